"""Training TRAJECTORIES of BASELINE.json configs[2] at its real depth and input size (YOLOv4-L 608x608), through the recipe
hooks -- the evidence that the 16-bit step trains, which gradient norms at initialisation cannot give (round 4).

Reference recipe: SGD-Nesterov with one group per parameter, weight decay on the conv weights only, gradient clip 35,
dynamic loss scale, fp32 master weights under fp16 autocast (mmdet/core/custom_hooks/accum_optim_hooks.py:9-60,
configs/yolov4/yolov4l_coco_mosaic.py:86-149); no warm-up hook and no EMA here (neither changes the trained weights'
trajectory on a fixed batch), lr 1e-3.

  * fp32 HIP vs the ORACLE (oracle/yolov4_oracle.forward_train + torch autograd + the hooks oracle's clip / SGD on the
    CPU, the arithmetic the golden fixtures pin to the reference) in float32 and float64: batch 2, the first
    ORACLE_STEPS optimizer steps, anchored on the float64 run (the trajectory is chaotic: see the test);
  * (round 4's second test -- fp16 / bf16 150-step curves within 5 % of fp32's -- compared samples of a chaotic family
    whose run-to-run spread on ONE box is 3.5-5.5 % (profiles/r05_traj_spread.md) and went red on the driver's box; it
    is replaced by tests/test_gpu_zz_trajectory.py: deterministic fp32 trajectory + teacher-forced 16-bit gradients.)"""
import os
import sys

import numpy as np
import pytest
import torch

import mmdet_yolov4_amd as pkg
from mmdet_yolov4_amd import hooks as H
from mmdet_yolov4_amd.optim import build_optimizer
from oracle import train_hooks_oracle as HO
from oracle import yolov4_oracle as O

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
SIZE, LR, MOM, WD, CLIP = 608, 1e-3, 0.937, 5e-4, 35.0
ORACLE_STEPS = 4


def _data(batch, dev):
    img = bench.synthetic_images(batch, SIZE, 1000, dev)
    gtb, gtl = bench.synthetic_gts(batch, SIZE, 2000, dev)
    return dict(img=img, img_metas=[dict() for _ in range(batch)], gt_bboxes=gtb, gt_labels=gtl)


def _hip_run(dtype, steps, batch):
    torch.manual_seed(0)
    det = pkg.build_detector(bench.model_cfg('yolov4l'))
    det.init_weights()
    det.train().to(DEV)
    sd0 = {k: v.detach().cpu().clone() for k, v in det.state_dict().items()}
    if dtype != torch.float32:
        pkg.wrap_fp16_model(det, dtype)
    opt = build_optimizer(det, dict(type='SGD', lr=LR, momentum=MOM, weight_decay=WD, nesterov=True,
                                    paramwise_cfg=dict(bias_decay_mult=0., norm_decay_mult=0.)))
    runner = H.Runner(det, opt, max_epochs=1)
    runner.log_buffer = None
    runner.register_hook(H.Fp16GradAccumulateOptimizerHook(accumulation=1, grad_clip=dict(max_norm=CLIP, norm_type=2),
                                                           loss_scale='dynamic'), 'ABOVE_NORMAL')
    data = _data(batch, DEV)
    runner.data_loader = H.BatchSource([data], batch)
    runner.call_hook('before_run')
    runner.call_hook('before_train_epoch')
    losses = []
    for _ in range(steps):
        runner.call_hook('before_train_iter')
        runner.outputs = det.train_step(data, opt)
        runner.call_hook('after_train_iter')
        runner.iter += 1
        losses.append(float(runner.outputs['log_vars']['loss']))       # the loss BEFORE this step's update
    return np.array(losses), sd0


def _oracle_run(sd0, steps, batch, dtype=torch.float32):
    """The same recipe on the CPU: oracle forward_train + autograd, clip_grad_norm, per-parameter SGD-Nesterov."""
    stages, reps = O.ARCH['v4l5p']
    data = _data(batch, 'cpu')
    torch.set_num_threads(max(1, bench.host_cpu_budget()))
    sd = {k: (v.clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
    names = [k for k, v in sd.items() if v.is_floating_point() and 'running_' not in k]
    bufs = {k: None for k in names}
    losses = []
    for _ in range(steps):
        for k in names:
            sd[k] = sd[k].detach().clone().requires_grad_(True)
        L = O.forward_train(data['img'].to(dtype), sd, stages, reps, [3, 4, 5], [b.to(dtype) for b in data['gt_bboxes']],
                            data['gt_labels'], neck='v4')
        total = O.total_loss(L)
        losses.append(float(total.detach()))
        total.backward()
        _, grads = HO.clip_grad_norm([sd[k].grad for k in names], CLIP)
        with torch.no_grad():
            for k, g in zip(names, grads):
                wd = WD if sd[k].dim() > 1 else 0.0            # bias_decay_mult = norm_decay_mult = 0
                newp, bufs[k] = HO.sgd_nesterov(sd[k].detach(), g, bufs[k], LR, MOM, wd, True)
                sd[k] = newp
    return np.array(losses)


def test_fp32_hip_trajectory_tracks_the_oracle():
    """A randomly initialised 110-layer network under batch-of-2 BatchNorm statistics is chaotic: two fp32 evaluations
    of the same recipe part ways within a handful of steps (measured: the fp32 CPU oracle is 1.3e-3 / 5.4e-3 away from
    its own float64 run after 2 / 3 updates).  So, as for the one-step gradients (test_gpu_fullsize_cfgs.py), the
    statement is relative to the truth: at every step the HIP fp32 trajectory's worst distance so far from the float64
    oracle's is at most 3 x the fp32 CPU oracle's worst so far (+ 1e-4).  Measured: HIP 4e-7, 2e-6, 6e-5, 1.0e-3 vs
    oracle-fp32 0, 9e-6, 1.3e-3, 5.4e-3."""
    hip, sd0 = _hip_run(torch.float32, ORACLE_STEPS, 2)
    o32 = _oracle_run(sd0, ORACLE_STEPS, 2, torch.float32)
    o64 = _oracle_run(sd0, ORACLE_STEPS, 2, torch.float64)
    e_hip = np.maximum.accumulate(np.abs(hip - o64) / np.abs(o64))
    e_cpu = np.maximum.accumulate(np.abs(o32 - o64) / np.abs(o64))
    print('fp32 HIP      :', np.round(hip, 4))
    print('oracle fp32   :', np.round(o32, 4))
    print('oracle fp64   :', np.round(o64, 4))
    print('HIP vs fp64   :', e_hip)
    print('oracle32 vs 64:', e_cpu)
    assert e_hip[0] <= 1e-5 and e_hip[1] <= 1e-4                 # the first step and the first UPDATE are the recipe's
    assert (e_hip <= 3 * e_cpu + 1e-4).all(), (e_hip, e_cpu)
