#!/usr/bin/env python
"""Micro-benchmark of yv4_conv_wgrad_h16 over the YOLOv4-L layer shapes (batch 64, bf16): microseconds per launch
(chained launches between HIP events), TFLOP/s, and the bytes of the fp32 partial-sum exchange per launch.
    python tools/wgrad_bench.py [--batch 64] [--chain 10] [--filter k3]"""
import argparse, ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mmdet_yolov4_amd as pkg
from mmdet_yolov4_amd._lib import ConvDesc
from conv_bench import SHAPES


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--chain', type=int, default=10)
    ap.add_argument('--filter', default='')
    ap.add_argument('--det', action='store_true', help='the deterministic form (workspace slabs + ordered reduction)')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    lib = pkg._lib.lib()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    tot_us = tot_fl = 0.0
    for (cin, cout, k, s, h, cnt) in SHAPES:
        tag = f'{cin}->{cout} k{k}s{s} @{h}'
        if a.filter and a.filter not in tag:
            continue
        cp = (cin + 7) // 8 * 8
        xcs = 16 if cin < 8 else cp      # the 16-bit stem: 8 weight channels against an image stored with 16 per pixel
        co = (cout + 7) // 8 * 8
        pad = k // 2
        ho = (h + 2 * pad - k) // s + 1
        x = torch.randn(a.batch * h * h * xcs, device=dev).bfloat16()
        dy = torch.randn(a.batch * ho * ho * co, device=dev).bfloat16()
        dw = torch.zeros(co * k * k * cp, device=dev)
        d = ConvDesc()
        d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = a.batch, h, h, cp, ho, ho, co
        d.KH = d.KW = k
        d.stride, d.pad = s, pad
        d.x_cstride, d.y_cstride = xcs, co
        need = int(lib.yv4_conv_wgrad_workspace(C.byref(d), 2)) if a.det else 0
        ws = torch.empty(max(need // 4, 4), device=dev)
        ts = []
        for r in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.chain):
                if a.det:
                    rc = lib.yv4_conv_wgrad_det(C.byref(d), 2, x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), need, stream)
                else:
                    rc = lib.yv4_conv_wgrad_h16(C.byref(d), 2, x.data_ptr(), dy.data_ptr(), dw.data_ptr(), stream)
            e1.record()
            torch.cuda.synchronize()
            assert rc == 0, lib.yv4_last_error()
            if r:
                ts.append(e0.elapsed_time(e1) * 1e3 / a.chain)
        us = sorted(ts)[len(ts) // 2]
        fl = 2.0 * a.batch * ho * ho * cout * k * k * cin
        tot_us += us * cnt
        tot_fl += fl * cnt
        print(f'{tag:28s} x{cnt:2d} {us:8.0f} us {fl / us / 1e6:7.1f} TF   dW {co * k * k * cp * 4 / 1e6:6.2f} MB', flush=True)
    print(f'weighted total {tot_us:.0f} us, {tot_fl / tot_us / 1e6:.1f} TFLOP/s')


if __name__ == '__main__':
    main()
