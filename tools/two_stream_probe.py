#!/usr/bin/env python
"""Does the inference step gain from running as TWO half-batch plans on two HIP streams?  Every conv kernel of a plan has a
compute phase (K loop: HBM nearly idle) and a memory phase (epilogue: the whole output, and the residual, in one burst while
the matrix pipe idles), and one persistent workgroup per CU keeps all CUs in the same phase.  Two independent half-batch
plans on two streams drift apart and fill each other's phases.

  python tools/two_stream_probe.py [--dtype bf16] [--batch 32] [--size 608] [--steps 20] [--parts 2]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import mmdet_yolov4_amd as pkg  # noqa: E402
from mmdet_yolov4_amd.calibrate import calibrate_bn  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dtype', default='bf16')
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--size', type=int, default=608)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--parts', type=int, default=2)
    ap.add_argument('--model', default='yolov4l')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    tdt = {'f32': torch.float32, 'f16': torch.float16, 'bf16': torch.bfloat16}[a.dtype]
    torch.manual_seed(0)
    det = pkg.build_detector(bench.model_cfg(a.model))
    det.init_weights()
    det.eval().to(dev)
    img = bench.synthetic_images(a.batch, a.size, 1000, dev)
    cal = det.compile(a.batch, a.size, a.size, device=dev, rescale=True)
    calibrate_bn(cal, img)
    bench.init_head(det, cal, img, 1500.0)
    del cal
    det._engines.clear()
    whole = det.compile(a.batch, a.size, a.size, device=dev, rescale=True, dtype=tdt)
    nb = a.batch // a.parts
    parts = []
    for _ in range(a.parts):
        det._engines.clear()
        parts.append(det.compile(nb, a.size, a.size, device=dev, rescale=True, dtype=tdt))
    streams = [torch.cuda.Stream() for _ in range(a.parts)]
    chunks = [img[i * nb:(i + 1) * nb].contiguous() for i in range(a.parts)]

    def run_whole():
        whole.run(img)

    def run_parts(sync_each_step):
        cur = torch.cuda.current_stream()
        for s in streams:
            s.wait_stream(cur)
        for s, p, c in zip(streams, parts, chunks):
            with torch.cuda.stream(s):
                p.run(c)
        if sync_each_step:
            for s in streams:
                cur.wait_stream(s)

    def timeit(fn, steps):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps

    for rep in range(2):
        tw = timeit(run_whole, a.steps)
        tp = timeit(lambda: run_parts(True), a.steps)
        tf = timeit(lambda: run_parts(False), a.steps)
        print(f'{a.model} {a.size} batch {a.batch} {a.dtype}: one plan {a.batch / tw:8.1f} img/s ({tw * 1e3:.3f} ms) | '
              f'{a.parts} x batch-{nb} plans on {a.parts} streams, joined per step {a.batch / tp:8.1f} img/s ({tp * 1e3:.3f} ms) | '
              f'free-running {a.batch / tf:8.1f} img/s', flush=True)
    # same detections?
    whole.run(img)
    run_parts(True)
    torch.cuda.synchronize()
    same = True
    for i, p in enumerate(parts):
        for n in range(nb):
            k = int(p.post['count'][n])
            g = i * nb + n
            same &= k == int(whole.post['count'][g]) and bool(torch.equal(p.post['dets'][n, :k], whole.post['dets'][g, :k]))
    print('detections of the parts == detections of the one plan (bit-exact):', same)


if __name__ == '__main__':
    main()
