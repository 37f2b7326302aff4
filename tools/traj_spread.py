"""Run-to-run spread of the 150-step recipe trajectory (tests/test_gpu_zz_trajectory.py's recipe: YOLOv4-L 608, one fixed
batch of 8, SGD-Nesterov lr 1e-3, clip 35, dynamic loss scale): every precision REPS times in ONE process on ONE box,
per step loss / skipped flag / loss scale.  Prints one JSON line per run and a summary; `--det` turns the library's
deterministic mode on first.

    python tools/traj_spread.py --reps 3 --steps 150 [--det] [--dtypes fp32,fp16,bf16]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import mmdet_yolov4_amd as pkg  # noqa: E402
from mmdet_yolov4_amd import hooks as H  # noqa: E402
from mmdet_yolov4_amd.optim import build_optimizer  # noqa: E402

DEV = 'cuda:0'
SIZE, LR, MOM, WD, CLIP = 608, 1e-3, 0.937, 5e-4, 35.0
DT = dict(fp32=torch.float32, fp16=torch.float16, bf16=torch.bfloat16)


def run(dtype, steps, batch):
    torch.manual_seed(0)
    det = pkg.build_detector(bench.model_cfg('yolov4l'))
    det.init_weights()
    det.train().to(DEV)
    if dtype != torch.float32:
        pkg.wrap_fp16_model(det, dtype)
    opt = build_optimizer(det, dict(type='SGD', lr=LR, momentum=MOM, weight_decay=WD, nesterov=True,
                                    paramwise_cfg=dict(bias_decay_mult=0., norm_decay_mult=0.)))
    runner = H.Runner(det, opt, max_epochs=1)
    runner.log_buffer = None
    hook = H.Fp16GradAccumulateOptimizerHook(accumulation=1, grad_clip=dict(max_norm=CLIP, norm_type=2),
                                             loss_scale='dynamic')
    runner.register_hook(hook, 'ABOVE_NORMAL')
    img = bench.synthetic_images(batch, SIZE, 1000, DEV)
    gtb, gtl = bench.synthetic_gts(batch, SIZE, 2000, DEV)
    data = dict(img=img, img_metas=[dict() for _ in range(batch)], gt_bboxes=gtb, gt_labels=gtl)
    runner.data_loader = H.BatchSource([data], batch)
    runner.call_hook('before_run')
    runner.call_hook('before_train_epoch')
    losses, skips, scales, norms = [], [], [], []
    for _ in range(steps):
        runner.call_hook('before_train_iter')
        runner.outputs = det.train_step(data, opt)
        runner.call_hook('after_train_iter')
        runner.iter += 1
        losses.append(float(runner.outputs['log_vars']['loss']))
        c = hook.ctrl.tolist()
        skips.append(int(c[2] != 0))
        norms.append(c[1])
        scales.append(float(hook.scale_state[0].item()))
    return np.array(losses), np.array(skips), np.array(scales), np.array(norms)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--steps', type=int, default=150)
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--dtypes', default='fp32,fp16,bf16')
    ap.add_argument('--det', action='store_true')
    a = ap.parse_args()
    if a.det:
        pkg.set_deterministic(True)
    out = {}
    for name in a.dtypes.split(','):
        rows = []
        for r in range(a.reps):
            lo, sk, sc, nm = run(DT[name], a.steps, a.batch)
            rec = dict(dtype=name, rep=r, det=bool(a.det), first10=float(lo[:10].mean()), last10=float(lo[-10:].mean()),
                       final=float(lo[-1]), skipped_steps=int(sk.sum()), skipped_at=np.nonzero(sk)[0].tolist(),
                       scale_first=float(sc[0]), scale_last=float(sc[-1]), scale_min=float(sc.min()),
                       grad_norm_max=float(np.nanmax(np.where(np.isfinite(nm), nm, np.nan))),
                       every10=np.round(lo[::10], 4).tolist(),
                       blocks30=np.round(lo[:a.steps // 30 * 30].reshape(-1, 30).mean(1), 4).tolist())
            print(json.dumps(rec), flush=True)
            rows.append((lo, rec))
        out[name] = rows
    print('---- summary (last-10 mean per run; spread = (max - min) / mean)')
    for name, rows in out.items():
        v = np.array([r[1]['last10'] for r in rows])
        same = all(np.array_equal(rows[0][0], r[0]) for r in rows[1:])
        print(f'{name}: last10 {np.round(v, 4).tolist()} spread {(v.max() - v.min()) / v.mean():.4f} '
              f'skips {[r[1]["skipped_steps"] for r in rows]} bit-identical-curves {same}')


if __name__ == '__main__':
    main()
