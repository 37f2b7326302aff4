"""Host-side logic of the optimizer-side training step (no GPU): schedule arithmetic of the
hooks vs the reference fixture, parameter-group construction, the flat arenas, the runner's hook
ordering, the gradient reducer's bucketing."""
import json
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn as nn

import mmdet_yolov4_amd as M
from mmdet_yolov4_amd import hooks as H
from mmdet_yolov4_amd.dist import GradReducer
from mmdet_yolov4_amd.flat_state import FlatState
from mmdet_yolov4_amd.optim import FlatSGD, OPTIMIZERS, paramwise_groups
from mmdet_yolov4_amd.registry import HOOKS

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
from toy_model import Toy, toy_groups  # noqa: E402


@pytest.fixture(scope='module')
def G():
    return np.load(os.path.join(HERE, 'golden', 'hooks.npz'), allow_pickle=False)


def test_hooks_registered_under_reference_names():
    for name in ('Fp16GradAccumulateOptimizerHook', 'StateEMAHook', 'DetailedLinearWarmUpHook'):
        assert HOOKS.get(name) is not None
    assert OPTIMIZERS.get('SGD') is FlatSGD


def test_schedule_functions_match_reference_known_answers(G):
    for i, row in zip(G['kat_iters'], G['kat']):
        assert H.warmup_factor(int(i), 10000, 10.) == row[0]
        assert H.warmup_factor(int(i), 10000, 0.) == row[1]
        assert H.warmup_factor(int(i), 10000, 0.95) == row[2]
        assert H.ema_momentum(0.9999, int(i), 10000, 1) == row[3]
    for spg, world, want in G['kat_accum']:
        assert H.accumulation_steps(64, int(spg), int(world)) == want


def test_hook_constructor_semantics():
    h = H.Fp16GradAccumulateOptimizerHook(nominal_batch_size=64, grad_clip=dict(max_norm=35, norm_type=2),
                                          loss_scale='dynamic')
    assert h.accumulation is None and h.nominal_batch_size == 64 and h.dynamic
    assert h.scaler_cfg['init_scale'] == 65536. and h.scaler_cfg['growth_interval'] == 2000
    h = H.Fp16GradAccumulateOptimizerHook(accumulation=4, loss_scale=512.)
    assert h.accumulation == 4 and not h.dynamic and h.scaler_cfg['init_scale'] == 512.
    with pytest.raises(AssertionError):
        H.Fp16GradAccumulateOptimizerHook(accumulation=0)
    with pytest.raises(ValueError):
        H.Fp16GradAccumulateOptimizerHook(loss_scale='static')
    with pytest.raises(NotImplementedError):
        H.Fp16GradAccumulateOptimizerHook(grad_clip=dict(max_norm=1, norm_type=1))
    e = H.StateEMAHook(momentum=0.9999, nominal_batch_size=64, warm_up=10000)
    assert e.interval is None and e.nominal_batch_size == 64
    e = H.StateEMAHook(interval=3)
    assert e.interval == 3
    with pytest.raises(AssertionError):
        H.StateEMAHook(momentum=1.0)


def test_paramwise_groups_follow_mmcv_rules_and_parameter_order():
    model = Toy()
    mine = paramwise_groups(model, 0.01, 0.0005, dict(bias_decay_mult=0., norm_decay_mult=0.))
    want = toy_groups(model, 0.01, 0.0005)
    assert len(mine) == len(want) == len(list(model.parameters()))
    for a, b, (n, p) in zip(mine, want, model.named_parameters()):
        assert a['params'][0] is p and b['params'][0] is p
        assert {k: v for k, v in a.items() if k != 'params'} == {k: v for k, v in b.items() if k != 'params'}, n
    with pytest.raises(NotImplementedError):
        paramwise_groups(model, 0.01, 0.0005, dict(custom_keys={}))


def test_paramwise_groups_on_the_detector_one_group_per_parameter():
    sys.path.insert(0, os.path.dirname(HERE))
    import bench
    det = M.build_detector(bench.model_cfg('yolov4s'))
    groups = paramwise_groups(det, 0.01, 0.0005, dict(bias_decay_mult=0., norm_decay_mult=0.))
    names = [n for n, _ in det.named_parameters()]
    assert len(groups) == len(names)
    for g, n in zip(groups, names):
        if '.bn.' in n or n.endswith('.bias'):
            assert g.get('weight_decay') == 0.0, n
        else:
            assert 'weight_decay' not in g, n


def test_flat_state_rehomes_parameters_without_changing_the_state_dict():
    model = Toy()
    before = {k: v.clone() for k, v in model.state_dict().items()}
    keys = list(before.keys())
    fs = FlatState(model)
    assert FlatState.of(model) is fs
    after = model.state_dict()
    assert list(after.keys()) == keys
    for k in keys:
        assert after[k].dtype == before[k].dtype and after[k].shape == before[k].shape
        assert torch.equal(after[k], before[k])
    # every slice starts on a 16-byte boundary, parameters first, float buffers after
    for s in fs.param_segments + fs.buffer_segments:
        assert s.offset % 4 == 0
    assert fs.param_segments[-1].offset < fs.n_param <= fs.buffer_segments[0].offset
    # conv weights are stored channels_last = the packed (Cout, kh, kw, ci) order of the kernels
    w = model.conv.weight
    assert w.is_contiguous(memory_format=torch.channels_last)
    seg = fs.param_segments[0]
    flat = fs.values[seg.offset:seg.offset + seg.numel]
    assert torch.equal(flat, before['conv.weight'].permute(0, 2, 3, 1).reshape(-1))
    from mmdet_yolov4_amd.plan import pack_conv_weight
    packed, cp = pack_conv_weight(w)
    assert packed.data_ptr() == flat.data_ptr() and cp == 4      # a view, not a copy
    # parameters and buffers are views of the arena
    fs.values.zero_()
    assert float(model.pred.bias.detach().abs().sum()) == 0 and float(model.bn.running_var.abs().sum()) == 0
    # int buffers live in the int arena
    model.bn.num_batches_tracked += 3
    assert int(fs.ints[0]) == 3


def test_flat_state_gradients_accumulate_into_the_arena():
    torch.manual_seed(0)
    model = Toy()
    ref = Toy()
    ref.load_state_dict(model.state_dict())
    fs = FlatState(model)
    x = torch.randn(2, 4, 6, 6)
    for m in (model, ref):
        m(x).square().mean().backward()
        m(x * 0.5).square().mean().backward()          # second micro-batch accumulates
    assert fs.grads_attached()
    for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        torch.testing.assert_close(p.grad, q.grad, rtol=1e-6, atol=1e-7)
    total = torch.sqrt(sum(q.grad.double().pow(2).sum() for q in ref.parameters()))
    torch.testing.assert_close(fs.grads.double().norm(), total, rtol=1e-6, atol=0)
    nn.Module.zero_grad(model, set_to_none=True)
    assert not fs.grads_attached()
    fs.zero_grad()
    assert fs.grads_attached() and float(fs.grads.abs().sum()) == 0


def test_flat_sgd_hyper_table_tracks_mutated_groups():
    model = Toy()
    opt = M.optim.build_optimizer(model, dict(type='SGD', lr=0.01, momentum=0.937, weight_decay=0.0005, nesterov=True,
                                              paramwise_cfg=dict(bias_decay_mult=0., norm_decay_mult=0.)))
    assert isinstance(opt, FlatSGD) and len(opt.param_groups) == 5
    rows = opt.hyper_table()
    assert rows[0] == (0.01, 0.937, 0.0005, 1.0)        # conv.weight
    assert rows[1] == (0.01, 0.937, 0.0, 1.0)           # bn.weight: norm_decay_mult 0
    assert rows[2] == (0.01, 0.937, 0.0, 1.0)           # bn.bias
    assert rows[3] == (0.01, 0.937, 0.0005, 1.0)        # pred.weight
    assert rows[4] == (0.01, 0.937, 0.0, 1.0)           # pred.bias: bias_decay_mult 0
    opt.param_groups[4]['lr'] = 0.1
    opt.param_groups[0]['momentum'] = 0.5
    rows = opt.hyper_table()
    assert rows[4][0] == 0.1 and rows[0][1] == 0.5
    model.pred.bias.requires_grad = False
    assert opt.hyper_table()[4] == (0., 0., 0., 0.)
    with pytest.raises(ValueError):
        FlatSGD([nn.Parameter(torch.zeros(3))], lr=0.1, model=model)
    with pytest.raises(NotImplementedError):
        FlatSGD(model.parameters(), lr=0.1, dampening=0.1, model=model)


def test_warmup_hook_sequences_match_reference(G):
    cfg = json.loads(str(G['cfg_json']))
    model = Toy()
    opt = M.optim.build_optimizer(model, dict(type='SGD', lr=cfg['lr'], momentum=cfg['momentum'],
                                              weight_decay=cfg['weight_decay'], nesterov=True,
                                              paramwise_cfg=dict(bias_decay_mult=0., norm_decay_mult=0.)))
    runner = H.Runner(model, opt)
    hook = H.DetailedLinearWarmUpHook(warmup_iters=cfg['warmup_iters'],
                                      lr_weight_warmup_ratio=cfg['lr_weight_warmup_ratio'],
                                      lr_bias_warmup_ratio=cfg['lr_bias_warmup_ratio'],
                                      momentum_warmup_ratio=cfg['momentum_warmup_ratio'])
    hook.before_run(runner)
    for it in range(G['lr'].shape[0]):
        runner.iter = it
        hook.before_train_iter(runner)
        assert [g['lr'] for g in opt.param_groups] == list(G['lr'][it])
        assert [g['momentum'] for g in opt.param_groups] == list(G['momentum'][it])


def test_warmup_hook_needs_one_group_per_parameter():
    model = Toy()
    opt = FlatSGD(model.parameters(), lr=0.01, momentum=0.9, model=model)
    msgs = []

    class L:
        def warning(self, m):
            msgs.append(m)
    runner = H.Runner(model, opt, logger=L())
    hook = H.DetailedLinearWarmUpHook()
    hook.before_run(runner)
    assert msgs and not hook.base_momentum
    hook.before_train_iter(runner)                     # no-op
    assert opt.param_groups[0]['lr'] == 0.01


def test_runner_calls_hooks_in_priority_then_registration_order():
    calls = []

    class Rec(H.Hook):
        def __init__(self, tag):
            self.tag = tag

        def after_train_iter(self, runner):
            calls.append(self.tag)

    class Mdl(nn.Module):
        def train_step(self, data, optimizer):
            return dict(loss=torch.zeros(()))
    r = H.Runner(Mdl(), None)
    r.register_hook(Rec('opt'), 'NORMAL')
    r.register_hook(Rec('warm'), 'NORMAL')
    r.register_hook(Rec('ema'), 'HIGH')
    r.register_hook(Rec('log'), 'VERY_LOW')
    r.run(H.BatchSource([{}], 4), max_epochs=1)
    assert calls == ['ema', 'opt', 'warm', 'log']
    assert r.iter == 1 and r.epoch == 1


def test_cosine_annealing_epoch_schedule():
    model = Toy()
    opt = FlatSGD(model.parameters(), lr=0.01, momentum=0.9, model=model)
    r = H.Runner(model, opt, max_epochs=300)
    h = H.CosineAnnealingLrUpdaterHook(min_lr_ratio=0.2)
    h.before_run(r)
    for ep, want in ((0, 0.01), (150, 0.006), (300, 0.002)):
        r.epoch = ep
        h.before_train_epoch(r)
        assert abs(opt.param_groups[0]['lr'] - want) < 1e-12


def test_grad_reducer_buckets_cover_the_arena_and_fire_in_backward():
    torch.manual_seed(0)
    model = Toy()
    fs = FlatState(model)
    red = GradReducer(fs, bucket_mb=300 * 4 / (1 << 20))     # ~300 floats per bucket -> several buckets
    assert len(red.buckets) >= 2
    assert red.buckets[0][0] == 0 and red.buckets[-1][1] == fs.n_param
    for a, b in zip(red.buckets, red.buckets[1:]):
        assert a[1] == b[0]
    assert sorted(si for b in red.buckets for si in b[2]) == list(range(len(fs.param_segments)))
    red.arm()
    model(torch.randn(2, 4, 6, 6)).square().mean().backward()
    assert all(red._launched)                           # every bucket became ready inside backward
    g = fs.grads.clone()
    red.finish()
    assert torch.equal(fs.grads, g)                     # world size 1: nothing to exchange
    with pytest.raises(RuntimeError):
        red.finish()
    red.remove()


def test_grad_reducer_launch_order_is_descending_and_unused_parameters_wait_for_finish():
    """A rank that leaves a parameter unused must not reorder its collectives: buckets go out strictly in descending
    index order, a ready bucket below a not-yet-ready one waits (for finish() at the latest)."""
    torch.manual_seed(0)
    model = Toy()
    fs = FlatState(model)
    red = GradReducer(fs, bucket_mb=4 * 4 / (1 << 20))           # 4 floats: one bucket per parameter
    nb = len(red.buckets)
    assert nb >= 3
    red.arm()
    # gradient-final events arrive bottom-up (the opposite of backward's order): nothing may launch before the top
    # bucket is complete
    order = [si for b in red.buckets for si in b[2]]
    top = set(red.buckets[-1][2])
    for si in order:
        if si in top:
            break
        red._event(si)
        assert red.launch_order == []
    for si in red.buckets[-1][2][:-1]:
        red._event(si)
    assert red.launch_order == []
    red._event(red.buckets[-1][2][-1])
    assert red.launch_order == list(range(nb - 1, -1, -1))
    red.finish()
    # an unused parameter in the top bucket: everything is exchanged by finish(), still in descending order
    red.arm()
    for si in order:
        if si != red.buckets[-1][2][0]:
            red._event(si)
    assert red.launch_order == []
    red.finish()
    assert red.launch_order == list(range(nb - 1, -1, -1))
    red.remove()


def test_grad_reducer_counts_a_parameter_once_and_refuses_late_gradients():
    torch.manual_seed(0)
    model = Toy()
    fs = FlatState(model)
    red = GradReducer(fs, bucket_mb=300 * 4 / (1 << 20))
    red.arm()
    last = red.buckets[-1][2]
    red._event(last[0])
    red._event(last[0])                                  # a second use of the same weight before the exchange: harmless
    assert red._pending[-1] == len(last) - 1
    for si in last[1:]:
        red._event(si)
    assert red.launch_order == [len(red.buckets) - 1]
    with pytest.raises(RuntimeError, match='after its bucket was exchanged'):
        red._event(last[0])
    red.finish()
    lazy = GradReducer(fs, bucket_mb=300 * 4 / (1 << 20), overlap=False)
    lazy.arm()
    for b in lazy.buckets:
        for si in b[2]:
            lazy._event(si)
            lazy._event(si)
    assert lazy.launch_order == []                       # overlap=False: everything goes out in finish()
    lazy.finish()
    assert lazy.launch_order == list(range(len(lazy.buckets) - 1, -1, -1))
    lazy.remove()
    red.remove()


def test_syncbn_norm_cfg_builds_and_selects_group():
    """configs/yolov5_ddp: norm_cfg type 'SyncBN' -> torch.nn.SyncBatchNorm; the HIP BN path synchronises only
    for a training-mode SyncBatchNorm inside an initialised group of more than one rank."""
    import torch
    import mmdet_yolov4_amd as pkg
    from mmdet_yolov4_amd import train_ops as T
    bk = pkg.build_backbone(dict(type='DarknetCSP', scale=[['conv', 'bottleneck', 'csp'], [None, 1, 1], [4, 8, 16]],
                                 out_indices=[1, 2], norm_cfg=dict(type='SyncBN', requires_grad=True, eps=0.001,
                                                                   momentum=0.03)))
    bns = [m for m in bk.modules() if isinstance(m, torch.nn.SyncBatchNorm)]
    assert bns and all(b.eps == 0.001 and b.momentum == 0.03 for b in bns)
    assert T._sync_group(bns[0].train()) is None                  # no process group in this process
    assert T._sync_group(torch.nn.BatchNorm2d(4).train()) is None


def test_soft_focal_loss_matches_reference(golden):
    """SoftFocalLoss (registered by the reference's head file, yolocsp_head.py:21-50) around the sigmoid
    CrossEntropyLoss: outputs and input gradients of the reference module on soft targets (tests/golden/softfocal.npz,
    made by make_golden.py from /root/reference).  Pure tensor ops: compared on the CPU at 1e-6."""
    import numpy as np
    import torch
    import mmdet_yolov4_amd as pkg
    z = golden('softfocal')
    pred, gt = torch.from_numpy(z['pred']), torch.from_numpy(z['gt'])
    for tag in 'abc':
        gamma, alpha, weight = (float(v) for v in z[f'{tag}/cfg'])
        from mmdet_yolov4_amd.registry import build_loss
        crit = build_loss(dict(type='SoftFocalLoss', gamma=gamma, alpha=alpha,
                                   raw_loss=dict(type='CrossEntropyLoss', use_sigmoid=True,
                                                 reduction=str(z[f'{tag}/reduction']), loss_weight=weight)))
        x = pred.clone().requires_grad_(True)
        y = crit(x, gt)
        y.sum().backward()
        np.testing.assert_allclose(y.detach().numpy(), z[f'{tag}/out'], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(x.grad.numpy(), z[f'{tag}/grad'], rtol=2e-6, atol=1e-7)


def test_grad_reducer_cuts_buckets_from_the_end_with_a_small_head():
    """Bucket boundaries by reverse cumulative size: the bucket of the first-registered parameters (final only when the
    stem's gradient is: the exchange that cannot overlap backward) holds at most ``head_mb``; the others are filled from
    the END of the arena, so the leftover is next to the head, not at the tail that backward finishes first."""
    from mmdet_yolov4_amd.dist import GradReducer
    model = Toy()
    fs = FlatState(model)
    offs = [s.offset for s in fs.param_segments] + [fs.n_param]
    red = GradReducer(fs, bucket_mb=60 * 4 / (1 << 20), head_mb=20 * 4 / (1 << 20))
    spans = [(b[0], b[1]) for b in red.buckets]
    assert spans[0] == (0, offs[1])                       # one segment even when it exceeds the head cap
    assert spans[-1][1] == fs.n_param and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    assert all(hi - lo <= 60 for lo, hi in spans[1:])
    # cut from the end: the LAST bucket is as full as the cap allows
    last_lo = spans[-1][0]
    prev = max(o for o in offs if o < last_lo)
    assert fs.n_param - prev > 60
    red.remove()
    whole = GradReducer(fs, bucket_mb=1, head_mb=1)
    assert len(whole.buckets) == 1 and whole.buckets[0][:2] == [0, fs.n_param]
    whole.remove()
