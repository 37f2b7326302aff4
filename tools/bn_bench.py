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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--dtype', default='bf16')
    ap.add_argument('--reps', type=int, default=5)
    a = ap.parse_args()
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
