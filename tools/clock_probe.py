#!/usr/bin/env python
"""Which shader clock does the chip hold while the fp32 MFMA conv runs?  Loops one 3x3 layer (YOLOv4-L 256 -> 256 at
38 x 38, batch 32) for a few seconds while a thread samples `rocm-smi --showclocks`; prints the samples, the layer's
TFLOP/s and the matrix peak AT the sampled clock (256 CUs x 4 SIMDs x 64 FLOP/clk).
Usage (GPU box):  python tools/clock_probe.py [--dtype f32|bf16] [--seconds 6]"""
import argparse
import ctypes as C
import os
import re
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mmdet_yolov4_amd as pkg  # noqa: E402
from mmdet_yolov4_amd._lib import ConvDesc  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dtype', default='f32')
    ap.add_argument('--seconds', type=float, default=6.0)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    lib = pkg._lib.lib()
    h16 = a.dtype != 'f32'
    tdt = torch.float32 if not h16 else torch.bfloat16
    N, H, Cin, Cout = 32, 38, 256, 256
    x = torch.randn(N * H * H * Cin, device=dev).to(tdt)
    w = (torch.randn(Cout * 9 * Cin, device=dev) * 0.02).to(tdt)
    y = torch.empty(N * H * H * Cout, device=dev, dtype=tdt)
    sc, sh = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
    d = ConvDesc()
    d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = N, H, H, Cin, H, H, Cout
    d.KH = d.KW = 3
    d.stride, d.pad = 1, 1
    d.x_cstride, d.y_cstride = Cin, Cout
    d.act1 = 1
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def launch():
        if h16:
            return lib.yv4_conv_bn_act_fwd_h16(C.byref(d), 2, 2, x.data_ptr(), w.data_ptr(), sc.data_ptr(), sh.data_ptr(), None,
                                               None, None, y.data_ptr(), stream)
        return lib.yv4_conv_bn_act_fwd(C.byref(d), x.data_ptr(), w.data_ptr(), sc.data_ptr(), sh.data_ptr(), None, None, None,
                                       y.data_ptr(), stream)
    samples, stop = [], [False]

    def sampler():
        while not stop[0]:
            try:
                out = subprocess.run(['rocm-smi', '--showclocks'], capture_output=True, text=True, timeout=5).stdout
                m = re.search(r'sclk clock level:.*?\((\d+)Mhz\)', out)
                if m:
                    samples.append(int(m.group(1)))
            except Exception as e:      # noqa: BLE001
                samples.append(repr(e))
            time.sleep(0.3)
    assert launch() == 0
    torch.cuda.synchronize()
    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.perf_counter()
    n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.perf_counter() - t0 < a.seconds:
        for _ in range(50):
            launch()
        n += 50
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    stop[0] = True
    th.join()
    us = e0.elapsed_time(e1) * 1e3 / n
    tf = 2.0 * N * H * H * Cout * 9 * Cin / us / 1e6
    clocks = [s for s in samples if isinstance(s, int)]
    print('sclk samples (MHz):', samples)
    if clocks:
        mhz = sorted(clocks)[len(clocks) // 2]
        per_clk = 64 if not h16 else 1024
        peak = 256 * 4 * per_clk * mhz * 1e6 / 1e12
        print(f'{a.dtype}: {us:.1f} us per launch = {tf:.1f} TFLOP/s; median sclk {mhz} MHz -> matrix peak at that clock {peak:.1f} TFLOP/s '
              f'({tf / peak:.3f} of it)')
    else:
        print(f'{a.dtype}: {us:.1f} us per launch = {tf:.1f} TFLOP/s; no clock samples')


if __name__ == '__main__':
    main()
