#!/usr/bin/env python
"""Evaluation row: batched yv4_iou_coco / yv4_match_coco on a COCO-val-sized problem table
vs the per-problem CPU ops (the reference's compiled Cython from oracle/_ref when it travelled with the
snapshot -- kind "reference" -- else the oracle restatement -- kind "port").
Usage (GPU box):  python tools/eval_bench.py [--images 5000] [--classes 80] [--reps 10]
Prints ONE JSON line: problems/s for IoU + matching (10 thresholds x 4 breakdowns), inputs resident in HBM."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mmdet_yolov4_amd as pkg  # noqa: E402
from mmdet_yolov4_amd import eval_utils as EU  # noqa: E402


def table(rng, images, classes, mean_det, mean_gt):
    P = images * classes
    nd = rng.poisson(mean_det, P)
    ng = rng.poisson(mean_gt, P)
    live = (nd > 0) & (ng > 0)
    nd, ng = nd[live], ng[live]
    det_off, gt_off = EU._offsets(nd), EU._offsets(ng)

    def boxes(n):
        xy = rng.uniform(0, 600, (n, 2))
        return np.concatenate([xy, xy + rng.uniform(4, 200, (n, 2))], 1).astype(np.float32)
    gt, det = boxes(int(gt_off[-1])), boxes(int(det_off[-1]))
    crowd = rng.random(len(gt)) < 0.05
    return nd, ng, det_off, gt_off, det, gt, crowd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--images', type=int, default=5000)
    ap.add_argument('--classes', type=int, default=80)
    ap.add_argument('--mean-det', type=float, default=6.0)
    ap.add_argument('--mean-gt', type=float, default=1.5)
    ap.add_argument('--breakdowns', type=int, default=4)
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--cpu-seconds', type=float, default=10.0)
    a = ap.parse_args()
    rng = np.random.default_rng(0)
    dev = torch.device('cuda', 0)
    nd, ng, det_off, gt_off, det, gt, crowd = table(rng, a.images, a.classes, a.mean_det, a.mean_gt)
    P, B = len(nd), a.breakdowns
    thrs = np.array([0.5 + 0.05 * x for x in range(10)], np.float32)
    ign = rng.random((B, len(gt))) < 0.3
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    d_det, d_gt, d_crowd = t(det), t(gt), t(crowd)
    # matching problems = (problem, breakdown), sharing the IoU block of the problem
    q_det_off, q_gt_off = EU._offsets(np.repeat(nd, B)), EU._offsets(np.repeat(ng, B))
    q_ign = t(np.concatenate([ign[b, gt_off[k]:gt_off[k + 1]] for k in range(P) for b in range(B)]))
    q_crowd = t(np.concatenate([np.tile(crowd[gt_off[k]:gt_off[k + 1]], B) for k in range(P)]))
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    t_iou, t_match = [], []
    for r in range(a.reps + 2):
        ev[0].record()
        iou, iou_off = EU.iou_coco_batched(d_det, d_gt, d_crowd, det_off, gt_off)
        ev[1].record()
        EU.match_coco_batched(iou, q_det_off, q_gt_off, np.repeat(iou_off[:-1], B), thrs, q_ign, q_crowd)
        ev[2].record()
        torch.cuda.synchronize()
        if r >= 2:
            t_iou.append(ev[0].elapsed_time(ev[1]))
            t_match.append(ev[1].elapsed_time(ev[2]))
    ms_iou, ms_match = float(np.median(t_iou)), float(np.median(t_match))
    pairs = int(iou_off[-1])
    iou_bytes = 4 * pairs + 16 * (len(det) + len(gt)) + len(gt)
    # ---- CPU: per-problem ops on a bounded sample -----------------------------------------------------
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import build_ref, eval_oracle
    fns = build_ref.load_eval()
    kind = 'reference' if fns is not None else 'port'
    f_iou, f_match = fns if fns is not None else (eval_oracle.iou_coco, eval_oracle.match_coco)
    done, t0 = 0, time.perf_counter()
    while done < P and time.perf_counter() - t0 < a.cpu_seconds:
        k = done
        c = crowd[gt_off[k]:gt_off[k + 1]]
        m = f_iou(det[det_off[k]:det_off[k + 1]], gt[gt_off[k]:gt_off[k + 1]], c)
        for b in range(B):
            f_match(m, thrs, ign[b, gt_off[k]:gt_off[k + 1]], c)
        done += 1
    cpu_s = time.perf_counter() - t0
    out = dict(metric='evaluation problems (image x class) per second, IoU + matching', unit='problems/s',
               value=P / ((ms_iou + ms_match) * 1e-3), problems=P, pairs=pairs, thresholds=10, breakdowns=B,
               ms_iou=ms_iou, ms_match=ms_match, dtype='f32',
               data='synthetic (inputs resident in HBM; launch + table upload included)',
               roofline=dict(kernel='iou_coco_kernel', bound='hbm', achieved=iou_bytes / (ms_iou * 1e-3) / 1e9,
                             peak=8000.0, unit='GB/s', frac=iou_bytes / (ms_iou * 1e-3) / 1e9 / 8000.0, traffic=None),
               cpu_baseline=dict(value=done / cpu_s, unit='problems/s', cores=1, kind=kind,
                                 sample=f'first {done} of {P} problems, per-problem calls'))
    print(json.dumps(out))


if __name__ == '__main__':
    main()
