"""Optimizer side of the training step on the GPU: the flat-arena kernels (csrc/optim.hip) against
torch's own optimizer / clipping / EMA arithmetic, and the three registered hooks driven through
the runner against the trajectory the REFERENCE's hooks produced (tests/golden/hooks.npz).

Tolerances: single kernels 1e-6 relative (same fp32 operations, different association at most);
the 12-iteration trajectory 1e-4 relative / 1e-5 absolute (the toy model's conv/BN run on the
GPU's torch kernels here, on the CPU's in the fixture)."""
import ctypes as C
import json
import os
import sys

import numpy as np
import pytest
import torch

import mmdet_yolov4_amd as pkg
from mmdet_yolov4_amd import _lib
from mmdet_yolov4_amd import hooks as H
from mmdet_yolov4_amd.flat_state import FlatState
from mmdet_yolov4_amd.optim import FlatSGD, build_optimizer
from conftest import arch_from, state_dict_from

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
from toy_model import Toy  # noqa: E402

pytestmark = pytest.mark.gpu


def _sp():
    return torch.cuda.current_stream().cuda_stream


def test_sgd_step_kernel_vs_torch_sgd(gpu_device):
    torch.manual_seed(0)
    shapes = [(8, 4, 3, 3), (8,), (5,), (16, 8, 1, 1), (3,), (1000, 7)]
    params = [torch.nn.Parameter(torch.randn(s, device=gpu_device)) for s in shapes]
    model = torch.nn.Module()
    for i, p in enumerate(params):
        model.register_parameter(f'p{i}', p)
    ref = [torch.nn.Parameter(p.detach().clone()) for p in params]
    hyp = [dict(lr=0.01 * (i + 1), momentum=0.9 - 0.1 * i, weight_decay=5e-4 if i % 2 == 0 else 0.,
                nesterov=(i % 3 != 0)) for i in range(len(shapes))]
    hyp[4]['momentum'] = 0.0
    hyp[4]['nesterov'] = False
    topt = torch.optim.SGD([dict(params=[q], **h) for q, h in zip(ref, hyp)], lr=0.1)
    opt = FlatSGD([dict(params=[p], **h) for p, h in zip(params, hyp)], lr=0.1, model=model)
    fs = opt.flat
    for step in range(4):
        if step == 2:                                   # hooks mutate the groups between steps
            for g1, g2 in zip(opt.param_groups, topt.param_groups):
                g1['lr'] = g2['lr'] = g1['lr'] * 0.5
                if g1['momentum'] > 0:
                    g1['momentum'] = g2['momentum'] = g1['momentum'] * 0.95
        fs.zero_grad()
        for p, q in zip(params, ref):
            g = torch.randn(p.shape, device=gpu_device)
            p.grad.copy_(g)
            q.grad = g.clone()
        v0 = params[0]._version
        opt.step()
        topt.step()
        assert params[0]._version > v0                  # plan caches see the update
        for i, (p, q) in enumerate(zip(params, ref)):
            torch.testing.assert_close(p.detach(), q.detach(), rtol=2e-6, atol=1e-7, msg=f'step {step} param {i}')


def test_grad_prepare_norm_clip_and_overflow(gpu_device):
    L = _lib.lib()
    torch.manual_seed(1)
    n = 4 * 100003
    g = torch.randn(n, device=gpu_device) * 3
    work = torch.zeros(2, dtype=torch.float64, device=gpu_device)
    ctrl = torch.zeros(4, device=gpu_device)
    scale = torch.tensor([1024., 7.], device=gpu_device)
    norm = float(g.double().norm()) / 1024.
    for max_norm in (35., 0., 1e9):
        _lib.check(L.yv4_grad_prepare(g.data_ptr(), n, scale.data_ptr(), max_norm, work.data_ptr(), ctrl.data_ptr(),
                                      _sp()), 'prep')
        c = ctrl.tolist()
        assert abs(c[1] - norm) <= 1e-6 * norm
        coef = min(1.0, max_norm / (norm + 1e-6)) if max_norm > 0 else 1.0
        assert abs(c[0] - coef / 1024.) <= 2e-6 * coef / 1024.
        assert c[2] == 0.0 and c[3] == 1.0 / 1024.
    # no scale state: scale 1
    _lib.check(L.yv4_grad_prepare(g.data_ptr(), n, None, 0., work.data_ptr(), ctrl.data_ptr(), _sp()), 'prep')
    assert ctrl.tolist()[3] == 1.0 and ctrl.tolist()[0] == 1.0
    for bad in (float('inf'), float('-inf'), float('nan')):
        g2 = g.clone()
        g2[12345] = bad
        _lib.check(L.yv4_grad_prepare(g2.data_ptr(), n, scale.data_ptr(), 35., work.data_ptr(), ctrl.data_ptr(),
                                      _sp()), 'prep')
        assert ctrl.tolist()[2] == 1.0
    # a finite gradient whose fp32 SQUARE overflows is still finite: no skip, the norm comes out of the double sum
    g3 = torch.zeros(n, device=gpu_device)
    g3[7], g3[99] = 3e30, -4e30
    _lib.check(L.yv4_grad_prepare(g3.data_ptr(), n, None, 0., work.data_ptr(), ctrl.data_ptr(), _sp()), 'prep')
    c = ctrl.tolist()
    assert c[2] == 0.0 and abs(c[1] - 5e30) <= 1e-6 * 5e30
    # invalid arguments are refused, not launched
    assert L.yv4_grad_prepare(g.data_ptr(), 6, None, 0., work.data_ptr(), ctrl.data_ptr(), _sp()) != 0
    assert L.yv4_grad_prepare(g.data_ptr() + 4, 8, None, 0., work.data_ptr(), ctrl.data_ptr(), _sp()) != 0
    assert L.yv4_grad_prepare(None, 8, None, 0., work.data_ptr(), ctrl.data_ptr(), _sp()) != 0


def test_loss_scale_update_follows_gradscaler(gpu_device):
    L = _lib.lib()
    state = torch.tensor([65536., 0.], device=gpu_device)
    ok = torch.tensor([1., 0., 0., 1.], device=gpu_device)
    bad = torch.tensor([1., 0., 1., 1.], device=gpu_device)
    for i in range(3):
        _lib.check(L.yv4_loss_scale_update(state.data_ptr(), ok.data_ptr(), 2.0, 0.5, 3, _sp()), 'upd')
    assert state.tolist() == [131072., 0.]              # grew after 3 clean steps
    _lib.check(L.yv4_loss_scale_update(state.data_ptr(), ok.data_ptr(), 2.0, 0.5, 3, _sp()), 'upd')
    assert state.tolist() == [131072., 1.]
    _lib.check(L.yv4_loss_scale_update(state.data_ptr(), bad.data_ptr(), 2.0, 0.5, 3, _sp()), 'upd')
    assert state.tolist() == [65536., 0.]               # overflow: halve, restart the count
    assert L.yv4_loss_scale_update(state.data_ptr(), ok.data_ptr(), 1.0, 0.5, 3, _sp()) != 0


def test_ema_kernel(gpu_device):
    L = _lib.lib()
    torch.manual_seed(2)
    n = 4 * 50001
    ema = torch.randn(n, device=gpu_device)
    x = torch.randn(n, device=gpu_device)
    for m in (0.0, 0.3141, 0.9999):
        want = ema.clone().mul_(m).add_(x, alpha=1 - m)
        got = ema.clone()
        _lib.check(L.yv4_ema_update(got.data_ptr(), x.data_ptr(), n, m, _sp()), 'ema')
        torch.testing.assert_close(got, want, rtol=1e-6, atol=1e-7)
    assert L.yv4_ema_update(ema.data_ptr(), x.data_ptr(), 7, 0.5, _sp()) != 0


def _golden_run(G, device, loss_scale):
    cfg = json.loads(str(G['cfg_json']))
    model = Toy()
    model.load_state_dict({k[5:]: torch.from_numpy(G[k]) for k in G.files if k.startswith('init/')})
    model.to(device)
    opt = build_optimizer(model, dict(type='SGD', lr=cfg['lr'], momentum=cfg['momentum'],
                                      weight_decay=cfg['weight_decay'], nesterov=cfg['nesterov'],
                                      paramwise_cfg=dict(bias_decay_mult=0., norm_decay_mult=0.)))
    runner = H.Runner(model, opt, max_epochs=cfg['epochs'])
    for hook_cfg in (
            dict(type='DetailedLinearWarmUpHook', warmup_iters=cfg['warmup_iters'],
                 lr_weight_warmup_ratio=cfg['lr_weight_warmup_ratio'],
                 lr_bias_warmup_ratio=cfg['lr_bias_warmup_ratio'],
                 momentum_warmup_ratio=cfg['momentum_warmup_ratio'], priority='NORMAL'),
            dict(type='StateEMAHook', momentum=cfg['ema_momentum'], nominal_batch_size=cfg['nominal_batch_size'],
                 warm_up=cfg['ema_warm_up'], resume_from=None, priority='HIGH')):
        runner.register_hook_from_cfg(hook_cfg)
    opt_hook = H.Fp16GradAccumulateOptimizerHook(nominal_batch_size=cfg['nominal_batch_size'],
                                                 grad_clip=dict(max_norm=cfg['max_norm'], norm_type=2),
                                                 loss_scale=loss_scale)
    runner.register_hook(opt_hook, 'ABOVE_NORMAL')
    batches = [dict(img=torch.from_numpy(G[f'batch{i}/img']).to(device),
                    target=torch.from_numpy(G[f'batch{i}/target']).to(device))
               for i in range(cfg['iters_per_epoch'])]
    return cfg, model, opt, runner, opt_hook, batches


class _Check(H.Hook):
    """Runs last in every iteration: compares the whole state dict with the reference's."""

    def __init__(self, G):
        self.G = G
        self.seen = 0

    def after_train_iter(self, runner):
        sd = runner.model.state_dict()
        for k, v in sd.items():
            np.testing.assert_allclose(v.detach().float().cpu().numpy(), self.G[f'iter{runner.iter}/{k}'], rtol=1e-4,
                                       atol=1e-5, err_msg=f'iter {runner.iter} {k}')
        self.seen += 1

    def after_train_epoch(self, runner):
        sd = runner.model.state_dict()
        for k, v in sd.items():
            np.testing.assert_allclose(v.detach().float().cpu().numpy(), self.G[f'epoch_end{runner.epoch}/{k}'],
                                       rtol=1e-4, atol=1e-5, err_msg=f'epoch {runner.epoch} {k}')


@pytest.mark.parametrize('loss_scale', ['dynamic', 512.])
def test_hooks_reproduce_the_reference_trajectory(golden, gpu_device, loss_scale):
    G = golden('hooks')
    cfg, model, opt, runner, opt_hook, batches = _golden_run(G, gpu_device, loss_scale)
    chk = _Check(G)
    runner.register_hook(chk, 'LOWEST')
    runner.run(H.BatchSource(batches, cfg['samples_per_gpu']))
    assert chk.seen == cfg['epochs'] * cfg['iters_per_epoch']
    assert opt_hook.accumulation == int(G['accumulation'])
    # same state-dict keys (EMA buffers included) in the same order as the reference's model
    want_keys = [k[len('iter0/'):] for k in G.files if k.startswith('iter0/')]
    assert list(model.state_dict().keys()) == want_keys
    norms = [h[0]['grad_norm'] for h in runner.log_buffer.history]
    np.testing.assert_allclose(norms, G['grad_norm'], rtol=1e-4)
    scale = opt_hook.loss_scale()
    assert scale == (65536. if loss_scale == 'dynamic' else 512.)
    assert runner.meta['fp16']['loss_scaler']['scale'] == scale


def test_overflow_skips_the_step_and_backs_off(golden, gpu_device):
    G = golden('hooks')
    cfg, model, opt, runner, opt_hook, batches = _golden_run(G, gpu_device, 'dynamic')
    runner.data_loader = H.BatchSource(batches, cfg['samples_per_gpu'])
    runner.call_hook('before_run')
    runner.call_hook('before_train_epoch')
    before = {k: v.clone() for k, v in model.state_dict().items() if not k.startswith('ema_')}
    for it in range(2):                                  # one accumulation window
        runner.iter = it
        runner.call_hook('before_train_iter')
        out = model.train_step(batches[it], opt)
        if it == 1:
            out['loss'] = out['loss'] * float('inf')
        runner.outputs = out
        runner.call_hook('after_train_iter')
    assert opt_hook.ctrl.tolist()[2] == 1.0
    assert opt_hook.loss_scale() == 32768.
    for k, v in model.state_dict().items():
        if k in before and 'running' not in k and 'num_batches' not in k:
            assert torch.equal(v, before[k]), k          # parameters untouched
    assert float(opt.momentum_buf.abs().sum()) == 0.


def test_detector_trains_through_runner_and_eval_plan_sees_new_weights(golden, gpu_device):
    g = golden('train_v4')
    stages, reps, chans = arch_from(g)
    det = pkg.build_detector(dict(
        type='SingleStageDetector',
        backbone=dict(type='DarknetCSP', scale=[stages, reps, chans], out_indices=[3, 4, 5]),
        neck=dict(type='YOLOV4Neck', in_channels=[32, 64, 64], out_channels=[32, 64, 128], csp_repetition=1),
        bbox_head=dict(type='YOLOCSPHead', num_classes=80, in_channels=[32, 64, 128]), train_cfg=None,
        test_cfg=dict(nms_pre=-1, score_thr=0.001, nms=dict(type='nms', iou_threshold=0.65), max_per_img=300)))
    det.load_state_dict(state_dict_from(g), strict=True)
    det.to(gpu_device)
    keys_before = list(det.state_dict().keys())
    img = torch.from_numpy(g['img']).to(gpu_device)
    data = dict(img=img, img_metas=[dict(), dict()],
                gt_bboxes=[torch.from_numpy(g['gt_bboxes0']).to(gpu_device),
                           torch.from_numpy(g['gt_bboxes1']).to(gpu_device)],
                gt_labels=[torch.from_numpy(g['gt_labels0']).to(gpu_device),
                           torch.from_numpy(g['gt_labels1']).to(gpu_device)])
    det.eval()
    with torch.no_grad():
        feat0 = [f.clone() for f in det.extract_feat(img)]
    opt = build_optimizer(det, dict(type='SGD', lr=0.01, momentum=0.937, weight_decay=0.0005, nesterov=True,
                                    paramwise_cfg=dict(bias_decay_mult=0., norm_decay_mult=0.)))
    assert len(opt.param_groups) == len(list(det.parameters()))
    assert list(det.state_dict().keys()) == keys_before           # re-homing keeps the checkpoint layout
    runner = H.Runner(det, opt, max_epochs=2)
    runner.register_hook_from_cfg(dict(type='DetailedLinearWarmUpHook', warmup_iters=4, priority='NORMAL'))
    runner.register_hook_from_cfg(dict(type='StateEMAHook', momentum=0.9, interval=1, warm_up=2, priority='HIGH'))
    runner.register_hook(H.Fp16GradAccumulateOptimizerHook(accumulation=1, grad_clip=dict(max_norm=35, norm_type=2),
                                                           loss_scale='dynamic'), 'ABOVE_NORMAL')
    losses = []

    class Rec(H.Hook):
        def after_train_iter(self, r):
            losses.append(r.outputs['log_vars']['loss'])
    runner.register_hook(Rec(), 'LOWEST')
    runner.run(H.BatchSource([data] * 6, 2))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    sd = det.state_dict()
    assert sum(k.startswith('ema_') for k in sd) == len(keys_before)
    assert 'ema_backbone_conv0_conv_weight' in sd and sd['ema_backbone_conv0_conv_weight'].shape == \
        sd['backbone.conv0.conv.weight'].shape
    det.eval()
    with torch.no_grad():
        feat1 = det.extract_feat(img)
    assert any(float((a - b).abs().max()) > 1e-4 for a, b in zip(feat0, feat1))   # plans were rebuilt


class _Saver(H.Hook):
    """Stands in for mmcv's CheckpointHook (priority NORMAL < the EMA hook's HIGH): saves after the EMA swap."""

    def __init__(self, out_dir):
        self.out_dir = out_dir
        self.paths = []

    def after_train_epoch(self, runner):
        self.paths.append(runner.save_checkpoint(self.out_dir))


@pytest.mark.parametrize('loss_scale', ['dynamic', 512.])
def test_checkpoint_resume_continues_the_reference_trajectory(golden, gpu_device, tmp_path, loss_scale):
    """Stop after the first epoch, save a checkpoint (reference layout: meta + state_dict with the ema_* buffers +
    torch-layout optimizer state), resume in a FRESH model / optimizer / runner through ``StateEMAHook(resume_from=)``
    (ema_hooks.py:66-74) and finish: every later iteration must still match the reference's uninterrupted run."""
    G = golden('hooks')
    cfg, model, opt, runner, opt_hook, batches = _golden_run(G, gpu_device, loss_scale)
    assert cfg['epochs'] >= 2
    saver = _Saver(str(tmp_path))
    runner.register_hook(saver, 'NORMAL')
    runner.run(H.BatchSource(batches, cfg['samples_per_gpu']), max_epochs=1)
    ck = torch.load(saver.paths[0], map_location='cpu', weights_only=False)
    assert set(ck) >= {'meta', 'state_dict', 'optimizer'}
    assert ck['meta']['epoch'] == 1 and ck['meta']['iter'] == cfg['iters_per_epoch']
    # plain GradScaler.state_dict() in the meta (what mmcv's Fp16OptimizerHook.before_run loads)
    sc = ck['meta']['fp16']['loss_scaler']
    assert type(sc) is dict and set(sc) == {'scale', 'growth_factor', 'backoff_factor', 'growth_interval', '_growth_tracker'}
    assert all(not isinstance(v, torch.Tensor) for v in sc.values())
    want_keys = [k[len('iter0/'):] for k in G.files if k.startswith('iter0/')]
    assert list(ck['state_dict'].keys()) == want_keys and all(v.device.type == 'cpu' for v in ck['state_dict'].values())
    # torch.optim.SGD accepts the optimizer state as it is
    ref_model = Toy()
    sys.path.insert(0, os.path.join(HERE, 'golden'))
    from toy_model import toy_groups
    topt = torch.optim.SGD(toy_groups(ref_model, cfg['lr'], cfg['weight_decay']), lr=cfg['lr'], momentum=cfg['momentum'],
                           weight_decay=cfg['weight_decay'], nesterov=cfg['nesterov'])
    topt.load_state_dict(ck['optimizer'])
    for i, p in enumerate(ref_model.parameters()):
        assert topt.state[p]['momentum_buffer'].shape == p.shape
    assert os.path.islink(os.path.join(str(tmp_path), 'latest.pth'))

    # ---- a fresh process would do this ----
    cfg2, model2, opt2, runner2, opt_hook2, _ = _golden_run(G, gpu_device, loss_scale)
    for h in runner2._hooks:
        if isinstance(h, H.StateEMAHook):
            h.checkpoint = os.path.join(str(tmp_path), 'latest.pth')

    class _Rebase(H.Hook):
        """The reference's warm-up hook reads its base lr / momentum from the optimizer's groups in before_run
        (warmup_hooks.py:22-36), i.e. AFTER the EMA hook's resume has loaded the groups as they were saved -- inside
        the warm-up window those are warmed values, not bases (a reference quirk this package mirrors).  A user
        resuming inside the window has to restore the bases; so does this test (the fixture warms up for 8 of its 12
        iterations).  Runs between the EMA hook (HIGH) and the warm-up hook (NORMAL)."""

        def before_run(self, r):
            for (name, _), g in zip(r.model.named_parameters(), r.optimizer.param_groups):
                g['lr'], g['momentum'] = cfg['lr'], cfg['momentum']
    runner2.register_hook(_Rebase(), 'ABOVE_NORMAL')
    chk = _Check(G)
    runner2.register_hook(chk, 'LOWEST')
    runner2.run(H.BatchSource(batches, cfg['samples_per_gpu']))
    assert runner2.epoch == cfg['epochs'] and chk.seen == (cfg['epochs'] - 1) * cfg['iters_per_epoch']
    assert opt_hook2.loss_scale() == opt_hook.loss_scale() or loss_scale == 'dynamic'
    # and torch's optimizer state loads into the flat optimizer (checkpoints written by the reference)
    opt2.load_state_dict(topt.state_dict())
    sd_a, sd_b = opt.state_dict(), None
    opt3 = build_optimizer(Toy().to(gpu_device), dict(type='SGD', lr=cfg['lr'], momentum=cfg['momentum'],
                                                      weight_decay=cfg['weight_decay'], nesterov=cfg['nesterov'],
                                                      paramwise_cfg=dict(bias_decay_mult=0., norm_decay_mult=0.)))
    opt3.load_state_dict(sd_a)
    sd_b = opt3.state_dict()
    assert sd_a['param_groups'] == sd_b['param_groups']
    for i in sd_a['state']:
        assert torch.equal(sd_a['state'][i]['momentum_buffer'], sd_b['state'][i]['momentum_buffer'])
