"""Teacher-forced gradient comparison along ONE fp32 trajectory (VERDICT round 4, item 1 iii): the measurement behind
tests/test_gpu_zz_trajectory.py.  Deterministic mode on; YOLOv4-L 608, fixed batch of 8, the recipe of
tests/test_gpu_zz_trajectory.py.  At the snapshot steps the fp32 weights are loaded into an fp32 / fp16 / bf16 model and
ONE forward + backward is run: per parameter group the relative distance, cosine, norm ratio and projection of the
16-bit gradient on the fp32 one -- and the same for an fp32 model whose weights carry one 16-bit rounding (the probe).

    python tools/teacher_forced.py [--steps 150] [--snaps 0,25,50,100,149]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import test_gpu_zz_trajectory as TZ  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=150)
    ap.add_argument('--snaps', default='0,25,50,100,149')
    ap.add_argument('--batch', type=int, default=8)
    a = ap.parse_args()
    snaps = [int(s) for s in a.snaps.split(',')]
    import mmdet_yolov4_amd as pkg
    pkg.set_deterministic(True)
    losses, states = TZ.fp32_trajectory(a.steps, a.batch, snaps)
    print(json.dumps(dict(kind='trajectory', every10=np.round(losses[::10], 4).tolist(),
                          blocks30=np.round(losses[:a.steps // 30 * 30].reshape(-1, 30).mean(1), 4).tolist())), flush=True)
    for s in snaps:
        rows = TZ.teacher_forced_stats(states[s], a.batch)
        for r in rows:
            r['step'] = s
            print(json.dumps(r), flush=True)


if __name__ == '__main__':
    main()
