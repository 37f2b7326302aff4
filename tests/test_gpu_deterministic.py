"""Deterministic mode (``yv4_set_deterministic``, include/yv4.h "deterministic mode"): the reference's training step is
reproducible wherever torch's is -- its BatchNorm (torch.nn.BatchNorm2d through mmdet/models/backbones/darknetcsp.py:15-35)
sums in a fixed order -- while this library's default adds BatchNorm statistics, dbeta / dgamma, loss sums, positives' row
gradients, bias gradients, the SPP scatter and the gradient norm with atomics in arrival order.  With the mode on those
sums run on fixed-point integer words (order-independent), so:

  * every kernel touched gives the SAME BITS twice and the default path's values to rounding (kernel level);
  * a training step of the golden detector still matches the REFERENCE's losses and gradients (tests/golden/train_v4.npz);
  * YOLOv4-L 608 at batch 8 -- forward, fused loss, backward, clip, SGD, three optimizer steps -- twice from the same
    initialisation gives bit-identical losses, parameters, momentum buffers and running statistics, in fp32, bf16 and fp16.
"""
import os
import sys

import numpy as np
import pytest
import torch

import mmdet_yolov4_amd as pkg
from mmdet_yolov4_amd import hooks as H
from mmdet_yolov4_amd import train_ops as T
from mmdet_yolov4_amd.optim import build_optimizer
from conftest import arch_from, state_dict_from

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture
def det_mode():
    pkg.set_deterministic(True)
    assert pkg.deterministic()
    yield
    pkg.set_deterministic(False)
    assert not pkg.deterministic()


def _bn_case(dtype, C, N, H, W, seed, act=pkg._lib.ACT_MISH):
    g = torch.Generator(device='cpu').manual_seed(seed)
    x = (torch.randn(N, C, H, W, generator=g) * 3 + 0.5).to(DEV).to(dtype).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(N, C, H, W, generator=g).to(DEV).to(dtype).contiguous(memory_format=torch.channels_last)
    bn = torch.nn.BatchNorm2d(C).to(DEV).train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5, generator=None)
        bn.bias.normal_(0, 0.2)
    return x, dy, bn, (act, 0.1)


def _bn_run(x, dy, bn, act):
    bn.zero_grad()
    bn.running_mean.zero_()
    bn.running_var.fill_(1.0)
    xr = x.clone().requires_grad_(True)
    y = T.bn_act(xr, bn, act)
    y.backward(dy)
    return [t.detach().clone() for t in (y, xr.grad, bn.weight.grad, bn.bias.grad, bn.running_mean, bn.running_var)]


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize('shape', [(64, 4, 76, 76), (256, 3, 19, 19), (8, 2, 33, 17), (1024, 2, 10, 10)])
def test_bn_kernels_deterministic_and_equal_to_default(dtype, shape, det_mode):
    """BatchNorm statistics pass + backward reduction in fixed point: twice the same bits; against the default
    (double atomics) equal to the rounding of the fp32 results (1e-6 relative of each tensor's scale)."""
    C, N, Hh, Ww = shape
    x, dy, bn, act = _bn_case(dtype, C, N, Hh, Ww, seed=C + N)
    torch.manual_seed(1)
    a = _bn_run(x, dy, bn, act)
    b = _bn_run(x, dy, bn, act)
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    pkg.set_deterministic(False)
    ref = _bn_run(x, dy, bn, act)
    pkg.set_deterministic(True)
    tol = 2e-6 if dtype == torch.float32 else 1e-2      # 16-bit outputs: one rounding step of the stored type
    for name, u, v in zip(('y', 'dx', 'dgamma', 'dbeta', 'running_mean', 'running_var'), a, ref):
        scale = float(v.float().abs().max()) + 1e-12
        err = float((u.float() - v.float()).abs().max()) / scale
        assert err <= (2e-6 if name in ('dgamma', 'dbeta', 'running_mean', 'running_var') else tol), (name, err)


def test_bn_nonfinite_input_stays_loud(det_mode):
    """An infinite activation (an fp16 overflow) must poison the statistics in both modes: the fixed-point words carry a
    sticky non-finite bit that reads back as NaN."""
    x, dy, bn, act = _bn_case(torch.float32, 16, 2, 12, 12, seed=3)
    x[1, 5, 3, 4] = float('inf')
    out = _bn_run(x, dy, bn, act)
    assert not torch.isfinite(out[4][5]) and not torch.isfinite(out[0][:, 5]).all()
    assert torch.isfinite(out[4][[0, 1, 2, 3, 4, 6]]).all()          # the other channels' statistics are untouched
    x[1, 5, 3, 4] = 3e30        # finite, but its square leaves the fixed-point range: NaN variance, not a wrapped sum
    out = _bn_run(x, dy, bn, act)
    assert not torch.isfinite(out[5][5])


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_conv_epilogue_statistics_deterministic(dtype, det_mode):
    """The statistics a training conv leaves for its BatchNorm (replica pairs of fixed-point words): Conv (conv + BN +
    Mish) forward/backward twice = the same bits, and the default mode's values to rounding -- over the tile kernels the
    shapes select (1x1 weight-stationary, 3x3 wide, stride 2, few channels)."""
    torch.manual_seed(5)
    for cin, cout, k, s, hw, n in ((64, 64, 1, 1, 76, 4), (128, 128, 3, 1, 38, 8), (64, 128, 3, 2, 76, 4),
                                   (32, 64, 3, 2, 96, 2), (256, 512, 3, 1, 19, 8), (16, 32, 3, 1, 40, 2)):
        m = pkg.Conv(cin, cout, k, s).to(DEV).train()
        if dtype != torch.float32:
            pkg.wrap_fp16_model(m, dtype)
        x = torch.randn(n, cin, hw, hw, device=DEV)
        dy = torch.randn(n, cout, hw // s, hw // s, device=DEV)

        def run():
            m.zero_grad()
            m.bn.running_mean.zero_()
            m.bn.running_var.fill_(1.0)
            xr = x.clone().requires_grad_(True)
            y = m(xr)
            y.float().backward(dy)
            return [t.detach().float().clone() for t in (y, xr.grad, m.conv.weight.grad, m.bn.weight.grad, m.bn.bias.grad,
                                                         m.bn.running_mean, m.bn.running_var)]
        a, b = run(), run()
        for u, v in zip(a, b):
            assert torch.equal(u, v), (cin, cout, k, s)
        pkg.set_deterministic(False)
        ref = run()
        pkg.set_deterministic(True)
        for i, (u, v) in enumerate(zip(a, ref)):
            scale = float(v.abs().max()) + 1e-12
            err = float((u - v).abs().max()) / scale
            assert err <= (1e-4 if dtype == torch.float32 else 3e-2), (cin, cout, k, s, i, err)


def _golden_det(g):
    stages, reps, chans = arch_from(g)
    cfg = dict(type='SingleStageDetector',
               backbone=dict(type='DarknetCSP', scale=[stages, reps, chans], out_indices=[3, 4, 5]),
               neck=dict(type='YOLOV4Neck', in_channels=[32, 64, 64], out_channels=[32, 64, 128], csp_repetition=1),
               bbox_head=dict(type='YOLOCSPHead', num_classes=80, in_channels=[32, 64, 128]),
               train_cfg=None,
               test_cfg=dict(nms_pre=-1, score_thr=0.001, nms=dict(type='nms', iou_threshold=0.65), max_per_img=300))
    det = pkg.build_detector(cfg)
    det.load_state_dict(state_dict_from(g), strict=True)
    return det.to(DEV)


def test_det_train_step_matches_reference(golden, det_mode):
    """The reference's own losses and gradients (tests/golden/train_v4.npz) through the deterministic accumulators: the
    tolerances of test_gpu_train_parity.py::test_train_step_matches_reference, unchanged."""
    g = golden('train_v4')
    det = _golden_det(g).train()
    img = torch.from_numpy(g['img']).to(DEV)
    gtb = [torch.from_numpy(g['gt_bboxes0']).to(DEV), torch.from_numpy(g['gt_bboxes1']).to(DEV)]
    gtl = [torch.from_numpy(g['gt_labels0']).to(DEV), torch.from_numpy(g['gt_labels1']).to(DEV)]
    out = det.train_step(dict(img=img, img_metas=[dict(), dict()], gt_bboxes=gtb, gt_labels=gtl), None)
    np.testing.assert_allclose(out['log_vars']['loss'], float(g['loss_total']), rtol=1e-4)
    np.testing.assert_allclose(out['log_vars']['loss_cls'], float(g['loss/loss_cls'].sum()), rtol=1e-4)
    np.testing.assert_allclose(out['log_vars']['loss_conf'], float(g['loss/loss_conf'].sum()), rtol=1e-4)
    np.testing.assert_allclose(out['log_vars']['loss_bbox'], float(g['loss/loss_bbox'].sum()), rtol=1e-4)
    out['loss'].backward()
    params = dict(det.named_parameters())
    names = [str(n) for n in g['grad_names']]
    sums = g['grad_sums']
    for i, n in enumerate(names):
        gr = params[n].grad.double()
        got = np.array([float(gr.abs().sum()), float(gr.pow(2).sum().sqrt())])
        np.testing.assert_allclose(got, sums[i][1:], rtol=1e-2, atol=1e-5, err_msg=n)
    for k in g.files:
        if k.startswith('grad/'):
            ref = g[k]
            np.testing.assert_allclose(params[k[5:]].grad.cpu().numpy(), ref, rtol=1e-2,
                                       atol=2e-3 * float(np.abs(ref).max()) + 2e-5, err_msg=k)
        if k.startswith('after/'):
            got = dict(det.named_buffers())[k[6:]].cpu().numpy()
            np.testing.assert_allclose(got, g[k], rtol=1e-4, atol=1e-5, err_msg=k)


def _recipe_steps(dtype, steps, batch, model='yolov4l', size=608):
    """tests/test_gpu_zz_trajectory.py's recipe (SGD-Nesterov per-parameter groups, clip 35, dynamic loss scale) for a few
    optimizer steps; returns the losses and the whole flat state (parameters, momentum, buffers)."""
    torch.manual_seed(0)
    det = pkg.build_detector(bench.model_cfg(model))
    det.init_weights()
    det.train().to(DEV)
    if dtype != torch.float32:
        pkg.wrap_fp16_model(det, dtype)
    opt = build_optimizer(det, dict(type='SGD', lr=1e-3, momentum=0.937, weight_decay=5e-4, nesterov=True,
                                    paramwise_cfg=dict(bias_decay_mult=0., norm_decay_mult=0.)))
    runner = H.Runner(det, opt, max_epochs=1)
    runner.log_buffer = None
    hook = H.Fp16GradAccumulateOptimizerHook(accumulation=1, grad_clip=dict(max_norm=35.0, norm_type=2), loss_scale='dynamic')
    runner.register_hook(hook, 'ABOVE_NORMAL')
    img = bench.synthetic_images(batch, size, 1000, DEV)
    gtb, gtl = bench.synthetic_gts(batch, size, 2000, DEV)
    data = dict(img=img, img_metas=[dict() for _ in range(batch)], gt_bboxes=gtb, gt_labels=gtl)
    runner.data_loader = H.BatchSource([data], batch)
    runner.call_hook('before_run')
    runner.call_hook('before_train_epoch')
    losses, norms = [], []
    for _ in range(steps):
        runner.call_hook('before_train_iter')
        runner.outputs = det.train_step(data, opt)
        runner.call_hook('after_train_iter')
        runner.iter += 1
        losses.append(float(runner.outputs['log_vars']['loss']))
        norms.append(float(hook.ctrl[1]))
    state = {k: v.detach().clone() for k, v in det.state_dict().items()}
    mom = opt.momentum_buf.detach().clone() if hasattr(opt, 'momentum_buf') else None
    grads = hook.flat.grads.detach().clone()
    return np.array(losses), np.array(norms), state, mom, grads


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
def test_fullsize_train_steps_are_bit_identical(dtype, det_mode):
    """YOLOv4-L 608, batch 8, three optimizer steps of the recipe, twice: the same bits everywhere (VERDICT round 4,
    next-round item 1 ii).  The third step's loss has been through two updates of every parameter and two updates of
    every running statistic -- anything order-dependent left in forward, loss, backward, clip or update would show."""
    a = _recipe_steps(dtype, 3, 8)
    b = _recipe_steps(dtype, 3, 8)
    assert np.array_equal(a[0], b[0]), (a[0], b[0])
    assert np.array_equal(a[1], b[1]), (a[1], b[1])
    assert np.isfinite(a[0]).all()
    for k in a[2]:
        assert torch.equal(a[2][k], b[2][k]), k
    if a[3] is not None:
        assert torch.equal(a[3], b[3])
    assert torch.equal(a[4], b[4])


def test_default_mode_is_not_claimed_deterministic():
    """The default (atomics in arrival order) agrees with the deterministic mode to rounding on the first step -- the
    accumulators differ, the arithmetic does not."""
    pkg.set_deterministic(False)
    a = _recipe_steps(torch.float32, 1, 4)
    pkg.set_deterministic(True)
    try:
        b = _recipe_steps(torch.float32, 1, 4)
    finally:
        pkg.set_deterministic(False)
    assert abs(a[0][0] - b[0][0]) <= 1e-5 * abs(b[0][0]), (a[0], b[0])
    assert abs(a[1][0] - b[1][0]) <= 1e-3 * abs(b[1][0]), (a[1], b[1])


def test_side_stream_weight_gradients_are_the_same_gradients(det_mode, monkeypatch):
    """The weight gradients run on a side stream by default (train_ops._wgrad_side_stream: off backward's critical path,
    joined by an end-of-backward callback of the autograd engine).  In deterministic mode the whole state after two
    recipe steps must be bit-identical with and without it -- a missing wait (a weight gradient reading dY before the
    BatchNorm backward wrote it, the optimizer reading .grad before the side stream finished, the arena zeroed under a
    running kernel) would show as different bits or as NaN."""
    from mmdet_yolov4_amd import train_ops as T2
    monkeypatch.setattr(T2, '_WGRAD_STREAM', True)
    a = _recipe_steps(torch.bfloat16, 2, 8)
    assert T2._SIDE_STREAMS, 'the side stream was never used'
    monkeypatch.setattr(T2, '_WGRAD_STREAM', False)
    b = _recipe_steps(torch.bfloat16, 2, 8)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (a[0], b[0], a[1], b[1])
    for k in a[2]:
        assert torch.equal(a[2][k], b[2][k]), k
    assert torch.equal(a[3], b[3]) and torch.equal(a[4], b[4])


@pytest.mark.parametrize('det', [False, True])
def test_conv_stats_fold_in_both_modes(det):
    """yv4_conv_stats_fold (what SyncBN all-reduces): the totals of a conv epilogue's statistics replicas as doubles, in the
    default layout (64 replicas of doubles) and in the deterministic one (32 pairs of fixed-point words)."""
    import ctypes as C
    from mmdet_yolov4_amd import _lib
    from mmdet_yolov4_amd.ops import stream_ptr
    pkg.set_deterministic(det)
    try:
        torch.manual_seed(2)
        m = pkg.Conv(64, 128, 3, 1).to(DEV).train()
        pkg.wrap_fp16_model(m, torch.bfloat16)
        x = torch.randn(4, 64, 38, 38, device=DEV).bfloat16().contiguous(memory_format=torch.channels_last)
        stats = T.conv_stats_buffer(128, x.device)
        y = T.conv2d(x, m.conv.weight, 1, 1, dtype=torch.bfloat16, stats=stats)
        out = torch.empty(2 * 128, dtype=torch.float64, device=DEV)
        rc = _lib.lib().yv4_conv_stats_fold(stats.data_ptr(), 128, 0, out.data_ptr(), stream_ptr())
        assert rc == 0
        yf = y.float().permute(0, 2, 3, 1).reshape(-1, 128).double()
        ref = torch.cat([yf.sum(0), (yf * yf).sum(0)])
        err = float((out - ref.detach()).abs().max() / ref.detach().abs().max())
        assert err <= 1e-6, err
    finally:
        pkg.set_deterministic(False)
