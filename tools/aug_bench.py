#!/usr/bin/env python
"""Rate of the fused train-side input pipeline (csrc/augment.hip) at the recipe's sizes: sources resized to 640, mosaic,
pad 1920 / crop 1280 / random scale / centre crop, flip, colour jitter, GtBBoxesFilter, normalise -> (N, 3, out, out).
    python tools/aug_bench.py [--out 640] [--batch 64] [--steps 20]
Also times the oracle restatement (numpy, one core) on a few samples as the CPU yardstick."""
import argparse, json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mmdet_yolov4_amd as pkg
from mmdet_yolov4_amd.augment import FusedTrainPipeline


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', type=int, default=640)
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--pool', type=int, default=64, help='distinct source images resident on the device')
    ap.add_argument('--cpu-samples', type=int, default=3)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    rng = np.random.RandomState(0)
    pool = []
    for _ in range(a.pool):                                      # COCO-like sizes
        h, w = (480, 640) if rng.rand() < 0.7 else (640, 427)
        k = rng.randint(1, 15)
        xy = rng.rand(k, 2) * [w, h]
        wh = rng.rand(k, 2) * [w / 2, h / 2] + 4
        b = np.concatenate([xy, np.minimum(xy + wh, [w, h])], 1).astype(np.float32)
        pool.append((torch.from_numpy(rng.randint(0, 256, (h, w, 3)).astype(np.uint8)).to(dev), b,
                     rng.randint(0, 80, k).astype(np.int64)))
    scale = a.out / 640.0
    pipe = FusedTrainPipeline(img_scale=(a.out, a.out), pad_val=114, pad_to=int(1920 * scale), crop=int(1280 * scale),
                              out_size=a.out)
    g = np.random.default_rng(0)

    def batch():
        return [[pool[i] for i in g.integers(0, a.pool, 4)] for _ in range(a.batch)]
    for _ in range(3):
        out = pipe(batch(), rng=g)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = pipe(batch(), rng=g)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    # device time of the two launches alone (descriptors prepared once)
    samples = batch()
    params = [pipe.draw_params(g) for _ in samples]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    pipe(samples, params=params)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        pipe(samples, params=params)
    e1.record()
    torch.cuda.synchronize()
    res = dict(metric='images/sec (train-side input pipeline: resize + mosaic + crop/scale/flip + HSV jitter + box filter + '
                      'normalise)', value=round(a.batch * a.steps / el, 1), unit='images/sec', out_size=a.out, batch=a.batch,
               ms_per_batch_wall=round(el / a.steps * 1e3, 2), ms_per_batch_incl_host_descriptors=round(e0.elapsed_time(e1) / 5, 3),
               mean_boxes_per_image=round(float(np.mean([len(b) for b in out['gt_bboxes']])), 1))
    if a.cpu_samples:
        from oracle import augment_oracle as A                   # the CPU yardstick (test infrastructure)
        cs = [[(f[0].cpu().numpy(), f[1], f[2]) for f in four] for four in samples[:a.cpu_samples]]
        t0 = time.perf_counter()
        for four, prm in zip(cs, params):
            A.train_sample([f[0] for f in four], [f[1] for f in four], [f[2] for f in four], prm, scale=pipe.img_scale)
        res['cpu_baseline'] = dict(value=round(a.cpu_samples / (time.perf_counter() - t0), 2), unit='images/sec', cores=1,
                                   kind='port', sample=f'{a.cpu_samples} samples through oracle/augment_oracle.py (numpy)')
    print(json.dumps(res))


if __name__ == '__main__':
    main()
