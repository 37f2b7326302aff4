#!/usr/bin/env python
"""Micro-benchmark of the train-mode BN kernels over the YOLOv4-L activation shapes (batch x map x channels).
Usage (GPU box):  python tools/bn_bench.py [--batch 64] [--dtype bf16]
Prints per shape: microseconds and achieved GB/s (algorithmic bytes) of stats / fwd / bwd (reduce+apply)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mmdet_yolov4_amd as pkg  # noqa: E402
from mmdet_yolov4_amd import train_ops as T  # noqa: E402

SHAPES = [(608, 32, 1), (304, 64, 2), (304, 32, 1), (152, 128, 2), (152, 64, 7), (76, 256, 2), (76, 128, 22),
          (38, 512, 3), (38, 256, 26), (19, 1024, 2), (19, 512, 30), (38, 128, 4), (76, 64, 0)]


def kernels(a):
    """apply (fwd), reduce (bwd), apply (bwd) launched back to back 20 times each between two events."""
    from mmdet_yolov4_amd import _lib
    from mmdet_yolov4_amd.ops import stream_ptr
    L = _lib.lib()
    dt = dict(f32=torch.float32, bf16=torch.bfloat16, f16=torch.float16)[a.dtype]
    code = T._DCODE[dt]
    es = 4 if a.dtype == 'f32' else 2
    dev = torch.device('cuda', 0)
    R = 20
    tot = [0.0, 0.0, 0.0]
    for hw, C_, cnt in SHAPES:
        M = a.batch * hw * hw
        x = torch.randn(M, C_, device=dev).to(dt)
        g = torch.randn(M, C_, device=dev).to(dt)
        y = torch.empty_like(x)
        mean, invstd = torch.zeros(C_, device=dev), torch.ones(C_, device=dev)
        ga, be = torch.ones(C_, device=dev), torch.zeros(C_, device=dev)
        dg, db = torch.empty(C_, device=dev), torch.empty(C_, device=dev)
        work = torch.zeros(2 * C_, dtype=torch.float64, device=dev)
        sp = stream_ptr()
        calls = [
            lambda: L.yv4_bn_act_fwd_h16(x.data_ptr(), code, C_, 0, mean.data_ptr(), invstd.data_ptr(), ga.data_ptr(),
                                         be.data_ptr(), None, C_, 0, y.data_ptr(), C_, 0, M, C_, 1, 0.0, sp),
            lambda: L.yv4_bn_act_bwd_sums(x.data_ptr(), code, C_, 0, g.data_ptr(), C_, 0, mean.data_ptr(), invstd.data_ptr(),
                                          ga.data_ptr(), be.data_ptr(), dg.data_ptr(), db.data_ptr(), work.data_ptr(), M, C_,
                                          1, 0.0, sp),
            lambda: L.yv4_bn_act_bwd_apply(x.data_ptr(), code, C_, 0, g.data_ptr(), C_, 0, mean.data_ptr(), invstd.data_ptr(),
                                           ga.data_ptr(), be.data_ptr(), y.data_ptr(), C_, 0, work.data_ptr(), M, M, None, C_,
                                           1, 0.0, sp)]
        us = []
        for fn in calls:
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(R):
                assert fn() == 0
            e1.record()
            torch.cuda.synchronize()
            us.append(e0.elapsed_time(e1) * 1e3 / R)
        nb = M * C_ * es
        print(f'{a.batch}x{hw}x{hw}x{C_:<5d} x{cnt:<3d} fwd {us[0]:7.1f} us {2 * nb / us[0] / 1e3:6.0f} GB/s | reduce {us[1]:7.1f} us '
              f'{2 * nb / us[1] / 1e3:6.0f} GB/s | apply {us[2]:7.1f} us {3 * nb / us[2] / 1e3:6.0f} GB/s')
        for k in range(3):
            tot[k] += us[k] * cnt
    print(f'weighted per step: fwd {tot[0] / 1e3:.2f} ms, reduce {tot[1] / 1e3:.2f} ms, apply {tot[2] / 1e3:.2f} ms')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--dtype', default='bf16')
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--kernels', action='store_true', help='time the three row passes alone (C ABI calls, 20 launches each)')
    a = ap.parse_args()
    if a.kernels:
        return kernels(a)
    dt = dict(f32=torch.float32, bf16=torch.bfloat16, f16=torch.float16)[a.dtype]
    es = 4 if a.dtype == 'f32' else 2
    dev = torch.device('cuda', 0)
    tot = [0.0, 0.0]
    for hw, C_, cnt in SHAPES:
        x = torch.randn(a.batch, C_, hw, hw, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
        bn = torch.nn.BatchNorm2d(C_).to(dev).train()
        g = torch.randn_like(x)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        tf, tb = [], []
        for r in range(a.reps + 1):
            xr = x.detach().requires_grad_(True)
            ev[0].record()
            y = T.bn_act(xr, bn, (1, 0.0))
            ev[1].record()
            y.backward(g)
            ev[2].record()
            torch.cuda.synchronize()
            if r:
                tf.append(ev[0].elapsed_time(ev[1]))
                tb.append(ev[1].elapsed_time(ev[2]))
        f, b = sorted(tf)[len(tf) // 2] * 1e3, sorted(tb)[len(tb) // 2] * 1e3
        nbytes = x.numel() * es
        print(f'{a.batch}x{hw}x{hw}x{C_:<5d} x{cnt:<3d} fwd(stats+apply) {f:8.1f} us {3 * nbytes / f / 1e3:7.0f} GB/s   '
              f'bwd(reduce+apply) {b:8.1f} us {5 * nbytes / b / 1e3:7.0f} GB/s')
        tot[0] += f * cnt
        tot[1] += b * cnt
    print(f'weighted per step: fwd {tot[0] / 1e3:.2f} ms, bwd {tot[1] / 1e3:.2f} ms')


if __name__ == '__main__':
    main()
