"""Host-side logic that needs no GPU: registry / config surface, state-dict layout,
plan compilation (shapes, fusion, algorithmic FLOPs), BN folding, anchors."""
import os
import textwrap

import numpy as np
import pytest
import torch

import mmdet_yolov4_amd as pkg
from conftest import arch_from, state_dict_from
from oracle import yolov4_oracle as O

V4L = dict(
    type='SingleStageDetector',
    backbone=dict(type='DarknetCSP', scale='v4l5p', out_indices=[3, 4, 5]),
    neck=dict(type='YOLOV4Neck', in_channels=[256, 512, 512], out_channels=[256, 512, 1024], csp_repetition=2),
    bbox_head=dict(type='YOLOCSPHead', num_classes=80, in_channels=[256, 512, 1024]),
    train_cfg=dict(),
    test_cfg=dict(min_bbox_size=0, nms_pre=-1, score_thr=0.001, nms=dict(type='nms', iou_threshold=0.65),
                  max_per_img=300))


def test_registry_names():
    for n in ('DarknetCSP', 'YOLOV4Neck', 'YOLOV5Neck', 'YOLOCSPHead', 'SingleStageDetector'):
        assert pkg.MODELS.get(n) is not None, n
    assert pkg.ANCHOR_GENERATORS.get('YOLOV4AnchorGenerator') is not None
    assert pkg.BBOX_CODERS.get('YOLOV4BBoxCoder') is not None
    assert pkg.ACTIVATION_LAYERS.get('Mish') is pkg.Mish
    assert isinstance(pkg.build_activation_layer(dict(type='Mish', inplace=True)), pkg.Mish)
    with pytest.raises(KeyError):
        pkg.build_backbone(dict(type='DarknetCSP', scale='nope'))
    with pytest.raises(NotImplementedError):
        pkg.YOLOV4BBoxCoder().encode(None, None, 8)


def test_v4l_parameter_and_key_counts():
    det = pkg.build_detector(V4L)
    assert sum(p.numel() for p in det.parameters()) == 52924765
    assert (len(det.backbone.state_dict()), len(det.neck.state_dict()), len(det.bbox_head.state_dict())) == (430, 222, 6)
    sd = det.state_dict()
    for k in ('backbone.conv0.conv.weight', 'backbone.conv0.bn.num_batches_tracked',
              'backbone.csp2.conv_csp.bottlenecks.0.conv1.conv.weight', 'backbone.sppv45.spp.conv7.bn.weight',
              'neck.pre_upsample_convs.0.conv.weight', 'neck.out_convs.2.bn.weight', 'bbox_head.convs_pred.2.bias'):
        assert k in sd, k
    # Q1: the SPP block's BNs keep mmcv's default eps, everything else the config's
    assert det.backbone.sppv45.spp.conv1.bn.eps == 1e-5 and det.backbone.sppv45.spp.bn.eps == 1e-5
    assert det.backbone.sppv45.conv_csp.bn.eps == 1e-3 and det.backbone.conv0.bn.momentum == 0.03
    # Q2: one bottleneck with the residual on
    assert det.backbone.bottleneck1.conv_bottleneck.shortcut is True


@pytest.mark.parametrize('name', ['tiny_v4', 'tiny_v5'])
def test_state_dict_layout_equals_reference(golden, name):
    g = golden(name)
    stages, reps, chans = arch_from(g)
    bb = pkg.DarknetCSP(scale=[stages, reps, chans], out_indices=[int(i) for i in g['meta_out_indices']])
    Neck = pkg.YOLOV4Neck if name == 'tiny_v4' else pkg.YOLOV5Neck
    neck = Neck(in_channels=[int(c) for c in g['meta_neck_in']], out_channels=[int(c) for c in g['meta_neck_out']],
                csp_repetition=int(g['meta_csp_rep']))
    head = pkg.YOLOCSPHead(num_classes=80, in_channels=[int(c) for c in g['meta_neck_out']])
    ours = [f'backbone.{k}' for k in bb.state_dict()] + [f'neck.{k}' for k in neck.state_dict()] + \
           [f'bbox_head.{k}' for k in head.state_dict()]
    assert ours == [str(k) for k in g['state_keys']]          # same names, same order
    sd = state_dict_from(g)
    for mod, pre in ((bb, 'backbone.'), (neck, 'neck.'), (head, 'bbox_head.')):
        mod.load_state_dict({k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}, strict=True)


@pytest.mark.parametrize('siblings', ['0', '1'])
def test_v4l_plan_fusion_and_flops(monkeypatch, siblings):
    monkeypatch.setenv('YV4_FUSE_SIBLINGS', siblings)
    det = pkg.build_detector(V4L).eval()
    plan = pkg.Plan('cpu')
    x = plan.add_input_nchw(1, 3, 608, 608)
    preds = det.emit(plan, x)
    det.bbox_head.emit_postprocess(plan, preds)
    kinds = {}
    for o in plan.ops:
        kinds[o.kind] = kinds.get(o.kind, 0) + 1
    # 115 convs; no BN / Mish / cat / add launches at all.  The eight pairs of 1x1 convs that read one
    # tensor (conv1 / conv2 of the five BottleneckCSP blocks, bottlenecks[0].conv1 / conv2 of three BottleneckCSP2 blocks,
    # and conv1 / conv2 of the SPPV4 block are one launch each: same FLOPs, 106 launches
    nconv = 115 if siblings == '0' else 106
    assert kinds == {'to_nhwc': 1, 'conv': nconv, 'spp': 1, 'resample': 4, 'reset': 1, 'decode': 1, 'nms': 1}
    assert abs(plan.total_flops() / 1e9 - 108.516) < 1e-3
    k3 = sum(o.flops for o in plan.ops if o.kind == 'conv' and o.info['k'] == 3) / 1e9
    assert abs(k3 - 87.513) < 1e-3
    assert [(v.H, v.W, v.C) for v in preds] == [(76, 76, 255), (38, 38, 255), (19, 19, 255)]
    assert plan.post['total_anchors'] == 22743


def test_16_bit_plan_fuses_the_first_two_layers_at_finalize(monkeypatch):
    """Host logic of Plan._fuse_stem_down (no launch): a 16-bit plan of YOLOv4-L replaces [repack, fp32 stem, stride-2
    conv] by ONE op that carries their FLOPs, keeps the replaced ops reachable, drops the two intermediate buffers
    from the allocation, and leaves the rest of the launch list alone; YV4_STEM_FUSE=0 and fp32 plans keep the three
    launches; calibrate_bn refuses a fused plan with a message that says what to do.  (Sibling 1x1 fusion off: this test
    counts launches and unallocated buffers of the stem fusion alone.)"""
    monkeypatch.setenv('YV4_FUSE_SIBLINGS', '0')
    det = pkg.build_detector(V4L).eval()

    def build(dtype):
        plan = pkg.Plan('cpu', dtype=dtype)
        x = plan.add_input_nchw(1, 3, 608, 608, dtype=torch.float32)
        plan.hint_single_consumer(x)
        preds = det.emit(plan, x)
        det.bbox_head.emit_postprocess(plan, preds)
        flops = plan.total_flops()
        plan.finalize()
        return plan, flops

    plan, flops = build(torch.bfloat16)
    kinds = [o.kind for o in plan.ops]
    assert kinds.count('to_nhwc') == 0 and kinds.count('conv') == 114
    fused = plan.ops[0]
    assert fused.name == 'stem_down' and fused.info['fused'] == 'stem_down'
    assert [o.kind for o in fused.info['parts']] == ['to_nhwc', 'conv', 'conv']
    assert abs(plan.total_flops() - flops) < 1.0                          # same algorithmic work
    assert (fused.info['Cin'], fused.info['Cout'], fused.info['stride'], fused.info['H']) == (3, 64, 2, 608)
    unused = [b for b in plan.bufs if getattr(b, 'unused', False)]
    assert len(unused) == 2 and all(b.tensor is None for b in unused)
    from mmdet_yolov4_amd.calibrate import calibrate_bn
    with pytest.raises(AssertionError, match='fp32 plan'):
        calibrate_bn(plan, torch.zeros(1, 3, 608, 608))
    monkeypatch.setenv('YV4_STEM_FUSE', '0')
    plain, _ = build(torch.bfloat16)
    assert [o.kind for o in plain.ops[:3]] == ['to_nhwc', 'conv', 'conv'] and not any(o.info.get('fused') for o in plain.ops)
    monkeypatch.delenv('YV4_STEM_FUSE')
    plan32 = pkg.Plan('cpu')
    x = plan32.add_input_nchw(1, 3, 608, 608)
    plan32.hint_single_consumer(x)
    det.emit(plan32, x)
    plan32.finalize()
    assert not any(o.info.get('fused') for o in plan32.ops)


def test_bn_fold_equals_batch_norm():
    bn = torch.nn.BatchNorm2d(16, eps=1e-3).eval()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(); bn.running_mean.normal_(); bn.running_var.uniform_(0.2, 2)
    from mmdet_yolov4_amd.plan import bn_affine, pack_conv_weight
    s, t = bn_affine(bn)
    x = torch.randn(2, 16, 5, 5)
    np.testing.assert_allclose((x * s[None, :, None, None] + t[None, :, None, None]).numpy(), bn(x).detach().numpy(),
                               rtol=1e-5, atol=1e-6)
    w = torch.randn(8, 3, 3, 3)
    wp, cp = pack_conv_weight(w)
    assert cp == 4 and wp.shape == (8, 36)
    assert torch.equal(wp.view(8, 3, 3, 4)[..., :3], w.permute(0, 2, 3, 1)) and float(wp.view(8, 3, 3, 4)[..., 3].abs().sum()) == 0


def test_anchor_generator_matches_oracle():
    gen = pkg.build_anchor_generator(dict(type='YOLOV4AnchorGenerator', base_sizes=O.DEFAULT_BASE_SIZES,
                                          strides=O.DEFAULT_STRIDES))
    assert gen.num_base_anchors == [3, 3, 3] and gen.num_levels == 3
    sizes = [(76, 76), (38, 38), (19, 19)]
    ours = gen.grid_anchors(sizes, device='cpu')
    ref = O.grid_anchors(sizes)
    assert [a.shape[0] for a in ours] == [17328, 4332, 1083]
    for a, b in zip(ours, ref):
        assert torch.equal(a, b)
    for a, b in zip(gen.base_anchors, O.base_anchors()):
        assert torch.equal(a, b)


def test_config_fromfile_base_and_delete(tmp_path):
    (tmp_path / 'base.py').write_text(textwrap.dedent('''
        model = dict(type='SingleStageDetector', backbone=dict(type='DarknetCSP', scale='v4s5p', out_indices=[3, 4, 5]),
                     neck=dict(type='YOLOV4Neck', in_channels=[128, 256, 256], out_channels=[128, 256, 512], csp_repetition=1),
                     bbox_head=dict(type='YOLOCSPHead', num_classes=80, in_channels=[128, 256, 512]),
                     train_cfg=dict(), test_cfg=dict(nms_pre=-1, score_thr=0.001, nms=dict(type='nms', iou_threshold=0.65), max_per_img=300))
        optimizer = dict(type='SGD', lr=0.01)
    '''))
    (tmp_path / 'child.py').write_text(textwrap.dedent('''
        _base_ = './base.py'
        model = dict(bbox_head=dict(num_classes=3))
        optimizer = dict(_delete_=True, type='Adam')
    '''))
    cfg = pkg.Config.fromfile(str(tmp_path / 'child.py'))
    assert cfg.model.bbox_head.num_classes == 3 and cfg.model.backbone.scale == 'v4s5p'
    assert dict(cfg.optimizer) == dict(type='Adam')
    det = pkg.build_detector(cfg.model)
    assert det.bbox_head.convs_pred[0].out_channels == 3 * 8
    assert det.bbox_head.test_cfg.score_thr == 0.001
    assert sum(p.numel() for p in det.backbone.parameters()) + sum(p.numel() for p in det.neck.parameters()) > 9e6


def test_head_loss_on_cpu_tensors_matches_oracle(golden):
    """The loss side is tensor ops (like the reference's): check it against the oracle on the
    reference's pred maps without a GPU."""
    g = golden('train_v4')
    head = pkg.YOLOCSPHead(num_classes=80, in_channels=[32, 64, 128])
    preds = [torch.from_numpy(g[f'pred{i}']) for i in range(3)]
    gtb = [torch.from_numpy(g['gt_bboxes0']), torch.from_numpy(g['gt_bboxes1'])]
    gtl = [torch.from_numpy(g['gt_labels0']), torch.from_numpy(g['gt_labels1'])]
    L = head.loss(preds, gtb, gtl, [dict(), dict()])
    for k in ('loss_cls', 'loss_conf', 'loss_bbox'):
        got = torch.stack([x.reshape(()) for x in L[k]]).numpy()
        np.testing.assert_allclose(got, g['loss/' + k], rtol=1e-6, atol=1e-7)
    assert float(L['num_gts']) == 2.5


def test_yolov3_registry_names_and_checkpoint_layout(golden):
    """YOLOv3 row (8f-3): the reference's registry names build, and the state-dict keys / order / shapes of
    Darknet + YOLOV3Neck + YOLOV3Head equal the reference's (fixture from its own modules)."""
    import mmdet_yolov4_amd as pkg
    from conftest import state_dict_from
    for reg, name in ((pkg.registry.BACKBONES, 'Darknet'), (pkg.registry.NECKS, 'YOLOV3Neck'),
                      (pkg.registry.HEADS, 'YOLOV3Head'), (pkg.registry.DETECTORS, 'YOLOV3'),
                      (pkg.registry.BBOX_CODERS, 'YOLOBBoxCoder')):
        assert reg.get(name) is not None, name
    g = golden('tiny_v3')

    class TinyDarknet(pkg.Darknet):
        arch_settings = {53: (tuple(int(v) for v in g['meta_layers']), tuple(tuple(int(c) for c in r) for r in g['meta_channels']))}
    backbone = TinyDarknet(depth=53, out_indices=(3, 4, 5))
    neck = pkg.YOLOV3Neck(num_scales=3, in_channels=[64, 64, 32], out_channels=[64, 32, 16])
    head = pkg.YOLOV3Head(num_classes=6, in_channels=[64, 32, 16], out_channels=[96, 64, 32])
    sd = state_dict_from(g)
    mine = {}
    for pre, m in (('backbone', backbone), ('neck', neck), ('bbox_head', head)):
        mine.update({f'{pre}.{k}': v for k, v in m.state_dict().items()})
    assert list(mine.keys()) == list(sd.keys())
    assert all(tuple(mine[k].shape) == tuple(sd[k].shape) for k in sd)
    # the full-size model has upstream's parameter count
    full = pkg.build_detector(dict(
        type='YOLOV3', backbone=dict(type='Darknet', depth=53, out_indices=(3, 4, 5)),
        neck=dict(type='YOLOV3Neck', num_scales=3, in_channels=[1024, 512, 256], out_channels=[512, 256, 128]),
        bbox_head=dict(type='YOLOV3Head', num_classes=80, in_channels=[512, 256, 128], out_channels=[1024, 512, 256])))
    assert sum(p.numel() for p in full.parameters()) == 61949149
    with pytest.raises(KeyError):
        pkg.Darknet(depth=19)
    # the coder's known answers (the reference's tests/test_utils/test_coder.py:8-24)
    coder = pkg.YOLOBBoxCoder()
    b, p_ = torch.from_numpy(g['coder/bboxes']), torch.from_numpy(g['coder/pred'])
    np.testing.assert_array_equal(coder.decode(b, p_, 32).numpy(), g['coder/decode_s32'])
    np.testing.assert_array_equal(coder.encode(b, torch.from_numpy(g['coder/encode_gt']), 32).numpy(), g['coder/encode_s32'])


def test_coco_test_annotation():
    """datasets/coco.py:357-409: xywh -> xyxy, crowd and unlisted categories flagged ignore (and kept), dtypes."""
    import numpy as np
    from mmdet_yolov4_amd.eval_utils import coco_test_annotation
    cat_ids, cat2label = [1, 3, 7], {1: 0, 3: 1, 7: 2}
    anns = [dict(bbox=[10, 20, 30, 40], category_id=3, area=1200.0),
            dict(bbox=[0.5, 1.5, 2, 3], category_id=7, area=6, iscrowd=1),
            dict(bbox=[5, 5, 1, 1], category_id=1, area=1, ignore=True)]
    a = coco_test_annotation(anns, cat_ids, cat2label)
    assert a['gt_bboxes'].dtype == np.float32 and a['gt_bboxes'].tolist() == [[10, 20, 40, 60], [0.5, 1.5, 2.5, 4.5], [5, 5, 6, 6]]
    assert a['gt_labels'].dtype == np.int64 and a['gt_labels'].tolist() == [1, 2, 0]
    assert a['gt_attrs']['ignore'].tolist() == [False, True, True] and a['gt_attrs']['iscrowd'].tolist() == [False, True, False]
    assert a['gt_attrs']['area'].dtype == np.float32 and a['gt_attrs']['ignore'].dtype == bool
    e = coco_test_annotation([], cat_ids, cat2label)
    assert e['gt_bboxes'].shape == (0, 4) and e['gt_labels'].shape == (0,) and e['gt_attrs']['iscrowd'].shape == (0,)
    import pytest
    with pytest.raises(KeyError):                          # the reference indexes cat2label before it could skip the box
        coco_test_annotation([dict(bbox=[0, 0, 1, 1], category_id=2, area=1)], cat_ids, cat2label)


def _layout_digest(module):
    import hashlib
    h = hashlib.sha256()
    for k, v in module.state_dict().items():
        h.update(f'{k}:{tuple(v.shape)}:{v.dtype}\n'.encode())
    return h.hexdigest()


def _n1():
    import json
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'configs_n1.json')) as f:
        return json.load(f)


@pytest.mark.parametrize('name', sorted(_n1()))
def test_reference_configs_build_unchanged(name):
    """N1 -- "configs/yolov4/* run unchanged": tests/golden/configs_n1.json holds, for each of the reference's 12
    YOLOv4 / YOLOv5 config files, the blocks this package's Config.fromfile parsed from the FILE in the build container
    and what the REFERENCE's own classes built from them (parameter counts, state-dict key order + shapes as a digest;
    tests/golden/make_golden_configs.py).  Rebuilding through this package's registries must give the same detector
    layout, one optimizer group per parameter (what DetailedLinearWarmUpHook requires, warmup_hooks.py:22-28), and the
    recipe's hooks under the reference's names."""
    from mmdet_yolov4_amd import hooks as H
    from mmdet_yolov4_amd.optim import paramwise_groups
    from mmdet_yolov4_amd.registry import HOOKS, build_from_cfg
    e = _n1()[name]
    det = pkg.build_detector(e['model'])
    for part, want in e['reference_parts'].items():
        m = getattr(det, part)
        assert sum(p.numel() for p in m.parameters()) == want['params'], part
        assert len(list(m.parameters())) == want['tensors'] and len(list(m.buffers())) == want['buffers'], part
        assert _layout_digest(m) == want['layout_sha256'], f'{part}: state-dict keys / shapes differ from the reference'
    syncbn = any(isinstance(m, torch.nn.SyncBatchNorm) for m in det.modules())
    assert syncbn == name.startswith('yolov5_ddp/')
    opt = dict(e['optimizer'])
    assert opt.pop('type') == 'SGD'
    groups = paramwise_groups(det, opt['lr'], opt['weight_decay'], opt.get('paramwise_cfg'))
    assert len(groups) == len(list(det.parameters()))
    for g, (n, p) in zip(groups, det.named_parameters()):
        assert g['params'][0] is p
        if n.endswith('.bias') or '.bn.' in n or n.endswith('bn.weight'):
            assert g.get('weight_decay') == 0.0, n          # bias_decay_mult = norm_decay_mult = 0
    # mmdet/apis/train.py:115-122: the fork's accumulate hook for configs/yolov4 + yolov5 (optimizer_config carries its
    # type), mmcv's Fp16OptimizerHook from optimizer_config + the top-level fp16 block for configs/yolov5_ddp
    hook = H.build_optimizer_hook(dict(optimizer_config=e['optimizer_config'], fp16=e['fp16']))
    assert isinstance(hook, H.Fp16GradAccumulateOptimizerHook) and hook.dynamic and hook.grad_clip['max_norm'] == 35
    if name.startswith('yolov5_ddp/'):
        assert type(hook) is H.Fp16OptimizerHook and hook.accumulation == 1
        assert hook.scaler_cfg == dict(init_scale=65536, growth_factor=2.0, backoff_factor=0.5, growth_interval=1000)
    else:
        assert type(hook) is H.Fp16GradAccumulateOptimizerHook and hook.nominal_batch_size == 64
    kinds = [build_from_cfg({k: v for k, v in h.items() if k != 'priority'}, HOOKS) for h in e['custom_hooks']]
    assert any(isinstance(k, H.StateEMAHook) for k in kinds) and any(isinstance(k, H.DetailedLinearWarmUpHook) for k in kinds)
    spg = e['samples_per_gpu']
    world = 8 if name.startswith('yolov5_ddp/') else 1
    assert H.accumulation_steps(64, spg, world) == -(-64 // (spg * world))
    tc = det.bbox_head.test_cfg
    assert tc.score_thr == 0.001 and tc.nms['iou_threshold'] == 0.65 and tc.max_per_img == 300


def test_deferred_log_vars_wait_at_the_first_value_access():
    """``deferred.DeferredLogVars``: keys / length / membership need no synchronisation; the first value access (and
    everything built on one: dict(), {**}, json, pickle, ==, repr) waits for the copy once and then behaves like the
    OrderedDict of python floats the reference returns (detectors/base.py:171-204)."""
    import json
    import pickle
    from collections import OrderedDict
    import torch
    from mmdet_yolov4_amd.deferred import DeferredLogVars

    class Event:
        waits = 0

        def synchronize(self):
            Event.waits += 1

    def make(finish=None):
        return DeferredLogVars(['loss_cls', 'loss'], torch.tensor([1.5, 4.0]), Event(), finish)

    lv = make()
    assert lv.pending and len(lv) == 2 and 'loss' in lv and list(lv.keys()) == ['loss_cls', 'loss'] and Event.waits == 0
    assert lv['loss'] == 4.0 and not lv.pending and Event.waits == 1
    assert lv['loss_cls'] == 1.5 and Event.waits == 1                       # resolved once
    want = OrderedDict(loss_cls=1.5, loss=4.0)
    for build in (dict, OrderedDict, lambda d: {**d}, lambda d: json.loads(json.dumps(d)), lambda d: pickle.loads(pickle.dumps(d)),
                  lambda d: dict(d.items()), lambda d: {k: d.get(k) for k in d}, lambda d: d.copy()):
        fresh = make()
        assert build(fresh) == want and not fresh.pending
    assert make() == want and make() == make() and 'loss_cls' in repr(make())
    assert [v for v in make().values()] == [1.5, 4.0]
    assert make(lambda c: [c[1], 1.0 / c[0]])['loss'] == 1.0 / 1.5          # a map from the copied floats to the values


def test_fused_losses_mutation_takes_the_general_path():
    """a detector or wrapper that adds an auxiliary loss to the fused head's loss mapping must see it in the
    optimised total and in log_vars (detectors/base.py:171-204 sums every key containing 'loss'): any mutation builds the
    mapping, which switches `_parse_losses` from the matrix's own total to the general sum."""
    import torch
    from mmdet_yolov4_amd.yolocsp_head import FusedLosses
    from mmdet_yolov4_amd.single_stage import SingleStageDetector
    w = torch.tensor([[1.0, 2.0, 3.0], [0.5, 0.25, 0.125]])
    fl = FusedLosses(w, torch.tensor(7.0), with_cls=True)
    assert not fl.built and float(fl.total) == 6.875
    total, log_vars = SingleStageDetector._parse_losses(None, fl)          # fast path: the matrix's own sums
    assert float(total) == 6.875 and not fl.built
    assert dict(log_vars)['loss'] == 6.875 and dict(log_vars)['loss_conf'] == 2.25
    for mutate in (lambda d: d.__setitem__('loss_aux', torch.tensor(10.0)), lambda d: d.update(loss_aux=torch.tensor(10.0)),
                   lambda d: d.setdefault('loss_aux', torch.tensor(10.0))):
        fl = FusedLosses(w, torch.tensor(7.0), with_cls=True)
        mutate(fl)
        assert fl.built and 'loss_aux' in fl and 'loss_cls' in fl
        total, log_vars = SingleStageDetector._parse_losses(None, fl)
        assert float(total) == 16.875 and dict(log_vars)['loss_aux'] == 10.0 and dict(log_vars)['loss'] == 16.875
    fl = FusedLosses(w, torch.tensor(7.0), with_cls=True)
    assert float(sum(fl.pop('loss_bbox')).sum()) == 3.125 and fl.built


def test_marker_region_stats_cuts_the_last_steps_of_a_two_stream_trace(tmp_path):
    """tools/summarize_prof.py: a kernel trace whose launch order is not periodic (weight gradients on a side stream) is cut at
    the optimizer's once-per-step kernel."""
    import importlib.util
    import os
    import sys
    rows = ['Kernel_Name,Start_Timestamp,End_Timestamp']
    t = 0
    for step in range(5):
        for k in range(3 + step % 2):                       # a varying number of launches per step
            rows.append(f'yv4::conv_kernel,{t},{t + 10}')
            t += 12
        rows.append(f'yv4::sgd_step_kernel,{t},{t + 5}')
        t += 7
    f = tmp_path / 'x_kernel_trace.csv'
    f.write_text('\n'.join(rows) + '\n')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    argv = sys.argv
    sys.argv = ['summarize_prof.py', '--r03', str(tmp_path / 'none'), str(tmp_path / 'out')]
    try:
        spec = importlib.util.spec_from_file_location('summarize_prof', os.path.join(root, 'tools', 'summarize_prof.py'))
        mod = importlib.util.module_from_spec(spec)
        try:
            spec.loader.exec_module(mod)
        except SystemExit:
            pass
    finally:
        sys.argv = argv
    out, info = mod.marker_region_stats(str(f), 2, 'sgd_step')
    calls = {r[0]: r[1] for r in out}
    assert calls['yv4::sgd_step_kernel'] == 2 and calls['yv4::conv_kernel'] == 4 + 3        # steps 3 and 4: 4 and 3 convs
    assert info[0] == (2 + 7) // 2
