"""The fused stem + stride-2 kernel (stem_down_h16.hip) alone, at the shapes of configs[3] (256 x 416 x 416, 16 -> 32, fp16)
and of the YOLOv4-L 16-bit plan (32 x 608 x 608, 32 -> 64, bf16): average launch time, algorithmic GB/s (fp32 NCHW image in,
16-bit NHWC map out), per-tile cycles per CU at 2.4 GHz.  With the measurement build the YV4_SD_* knobs apply.

    python tools/sd_bench.py [--reps 20]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmdet_yolov4_amd import _lib as L  # noqa: E402

DEV = 'cuda:0'
SHAPES = [('configs[3]', torch.float16, 256, 416, 16, 32), ('yolov4l', torch.bfloat16, 32, 608, 32, 64)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--act', type=int, default=1)
    a = ap.parse_args()
    for name, dt, N, S, C1, C2 in SHAPES:
        g = torch.Generator().manual_seed(0)
        x = torch.randn(N, 3, S, S, generator=g).to(DEV)
        w1 = torch.zeros(C1, 3, 3, 4)
        w1[..., :3] = torch.randn(C1, 3, 3, 3, generator=g) * (1.0 / 27) ** 0.5
        w1 = w1.to(DEV)
        w2 = (torch.randn(C2, 3, 3, C1, generator=g) * (1.0 / (9 * C1)) ** 0.5).to(dt).to(DEV)
        s1, t1 = (torch.rand(C1, generator=g) + 0.5).to(DEV), (torch.randn(C1, generator=g) * 0.1).to(DEV)
        s2, t2 = (torch.rand(C2, generator=g) + 0.5).to(DEV), (torch.randn(C2, generator=g) * 0.1).to(DEV)
        Ho = (S - 1) // 2 + 1
        y = torch.empty(N, Ho, Ho, C2, dtype=dt, device=DEV)
        code = 1 if dt == torch.float16 else 2
        st = torch.cuda.current_stream().cuda_stream

        def launch():
            L.check(L.lib().yv4_stem_down_fwd_h16(code, x.data_ptr(), N, S, S, w1.data_ptr(), s1.data_ptr(), t1.data_ptr(), C1,
                                                  a.act, 0.1, w2.data_ptr(), s2.data_ptr(), t2.data_ptr(), C2, a.act, 0.1,
                                                  y.data_ptr(), C2, 0, st), 'yv4_stem_down_fwd_h16')
        for _ in range(3):
            launch()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            launch()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.reps
        nbytes = x.numel() * 4 + y.numel() * 2
        tiles = N * ((Ho + 15) // 16) ** 2
        print(f'{name:10s} {N}x{S}x{S} {C1}->{C2}  {us:8.1f} us  {nbytes / us / 1e3:7.1f} GB/s  '
              f'{us * 2400 / (tiles / 256):8.0f} cycles per tile and CU', flush=True)


if __name__ == '__main__':
    main()
