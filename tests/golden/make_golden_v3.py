#!/usr/bin/env python
"""Generate tests/golden/tiny_v3.npz by importing the REFERENCE's YOLOv3 path
(mmdet/models/backbones/darknet.py, necks/yolo_neck.py, dense_heads/yolo_head.py,
core/bbox/coder/yolo_bbox_coder.py) with the mmcv shim of _ref_import.py.

``Darknet`` only knows depth 53 with fixed widths (40 M parameters); for a fixture its class attribute
``arch_settings`` is replaced AT RUN TIME by a narrow, shallow variant (the source is untouched, every
line of its forward runs as written).  Stored: checkpoint-layout state dict (fp16-representable values),
the image batch, every stage / neck / pred-map output, ``get_bboxes`` results of the reference's default
test_cfg shape (per-level ``nms_pre`` top-k, ``conf_thr``, ``score_thr``, NMS) with and without rescale,
and known answers of ``YOLOBBoxCoder.encode/decode`` on hand-written boxes
(the reference's own KAT: tests/test_utils/test_coder.py:8-24).

Run in the build container only; the GPU box never sees /root/reference.
"""
import importlib
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import _ref_import  # noqa: E402
from make_golden import quantize_fp16, randomize, sd_np  # noqa: E402
from oracle import build_ref  # noqa: E402

REF = '/root/reference'
ARCH = ((1, 1, 2, 2, 1), ((32, 16), (16, 32), (32, 32), (32, 64), (64, 64)))     # layers, (in, out) per stage


def import_v3(ref):
    imp = importlib.import_module
    core = sys.modules['mmdet.core']
    # images_to_levels (yolo_head.py:11-13 imports it from mmdet.core)
    au = imp('mmdet.core.anchor.utils')
    core.images_to_levels = au.images_to_levels
    # mmdet.core.export.get_k_for_topk (imported lazily inside yolo_head._get_bboxes): load the one file
    spec = importlib.util.spec_from_file_location('mmdet.core.export.onnx_helper',
                                                  os.path.join(REF, 'mmdet/core/export/onnx_helper.py'))
    oh = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(oh)
    exp = types.ModuleType('mmdet.core.export')
    exp.get_k_for_topk = oh.get_k_for_topk
    exp.add_dummy_nms_for_onnx = oh.add_dummy_nms_for_onnx
    sys.modules['mmdet.core.export'] = exp
    imp('mmdet.core.bbox.coder.yolo_bbox_coder')
    # training side: GridAssigner + PseudoSampler (package __init__s pull unrelated assigners: bypass them)
    for sub in ('core/bbox/assigners', 'core/bbox/samplers', 'utils'):
        name = 'mmdet.' + sub.replace('/', '.')
        if name not in sys.modules:
            _ref_import._pkg(name, os.path.join(REF, 'mmdet', sub))
    imp('mmdet.utils.util_mixins')
    imp('mmdet.core.bbox.assigners.grid_assigner')
    imp('mmdet.core.bbox.samplers.pseudo_sampler')
    # MSELoss is named by YOLOV3Head's defaults
    imp('mmdet.models.losses.mse_loss')
    return types.SimpleNamespace(darknet=imp('mmdet.models.backbones.darknet'),
                                 neck=imp('mmdet.models.necks.yolo_neck'),
                                 head=imp('mmdet.models.dense_heads.yolo_head'),
                                 coder=sys.modules['mmdet.core.bbox.coder.yolo_bbox_coder'])


def main():
    if not _ref_import.available():
        print('reference not present: nothing to do')
        return
    ref = _ref_import.install_shim(build_ref.load_ext())
    v3 = import_v3(ref)
    gen = torch.Generator().manual_seed(31)
    v3.darknet.Darknet.arch_settings = {53: ARCH}          # run-time narrowing, see the module docstring
    backbone = v3.darknet.Darknet(depth=53, out_indices=(3, 4, 5))
    # YOLOV3Neck applies conv_i (built for in_channels[i]) to the previous DetectionBlock's output, so
    # in_channels[i] must equal out_channels[i-1] (1024/512/256 -> 512/256/128 in configs/yolo/*)
    neck = v3.neck.YOLOV3Neck(num_scales=3, in_channels=[64, 64, 32], out_channels=[64, 32, 16])
    test_cfg = ref.ConfigDict(nms_pre=40, min_bbox_size=0, score_thr=0.05, conf_thr=0.005,
                              nms=dict(type='nms', iou_threshold=0.45), max_per_img=100)
    head = v3.head.YOLOV3Head(num_classes=6, in_channels=[64, 32, 16], out_channels=[96, 64, 32], train_cfg=None,
                              test_cfg=test_cfg)
    for m in (backbone, neck, head):
        randomize(m, gen)
        torch.nn.Module.eval(m)
    with torch.no_grad():
        for conv in head.convs_pred:
            conv.weight.normal_(0, 0.35, generator=gen)
            conv.bias.normal_(-0.5, 0.8, generator=gen)
            b = conv.bias.view(3, -1)
            b[:, 2:4] *= 0.3                                 # keep exp(t_w) in a sane range
    for m in (backbone, neck, head):
        quantize_fp16(m)
    N = 2
    img = torch.randint(0, 256, (N, 3, 64, 96), generator=gen).float() / 255.0     # img_norm_cfg of configs/yolo/*
    data = {'img': img.numpy()}
    for pre, m in (('backbone', backbone), ('neck', neck), ('bbox_head', head)):
        data.update(sd_np(pre, m))
    with torch.no_grad():
        x = img
        for i, name in enumerate(backbone.cr_blocks):
            x = getattr(backbone, name)(x)
            data[f'stage{i}'] = x.numpy()
        feats = backbone(img)                              # = stages 3, 4, 5 (out_indices)
        nouts = neck(feats)
        for i, f in enumerate(nouts):
            data[f'neck{i}'] = f.numpy()
        preds = head(nouts)[0]
        for i, f in enumerate(preds):
            data[f'pred{i}'] = f.numpy()
        sf = [np.array([1.25, 1.5, 1.25, 1.5], dtype=np.float32), np.array([0.75, 0.75, 0.75, 0.75], dtype=np.float32)]
        metas = [dict(scale_factor=s) for s in sf]
        data['scale_factors'] = np.stack(sf)
        for rescale, tag in ((True, ''), (False, '_norescale')):
            res = head.get_bboxes([p.clone() for p in preds], metas, rescale=rescale)
            for n, (d, l) in enumerate(res):
                data[f'dets{tag}{n}'] = d.numpy()
                data[f'labels{tag}{n}'] = l.numpy()
        # a second post-processing configuration: no top-k, no conf threshold
        cfg2 = ref.ConfigDict(nms_pre=-1, min_bbox_size=0, score_thr=0.3, conf_thr=-1,
                              nms=dict(type='nms', iou_threshold=0.6), max_per_img=30)
        res = head.get_bboxes([p.clone() for p in preds], metas, cfg=cfg2, rescale=True)
        for n, (d, l) in enumerate(res):
            data[f'cfg2/dets{n}'] = d.numpy()
            data[f'cfg2/labels{n}'] = l.numpy()
    # YOLOBBoxCoder known answers (the reference's own KAT + decode)
    coder = v3.coder.YOLOBBoxCoder()
    bboxes = torch.tensor([[-42., -29., 74., 61.], [-10., -29., 106., 61.], [22., -29., 138., 61.], [54., -29., 170., 61.]])
    pred = torch.tensor([[0.4709, 0.6152, 0.1690, -0.4056], [0.5399, 0.6653, 0.1162, -0.4162],
                         [0.4654, 0.6618, 0.1548, -0.4301], [0.4786, 0.6197, 0.1896, -0.4479]])
    data['coder/bboxes'] = bboxes.numpy()
    data['coder/pred'] = pred.numpy()
    data['coder/decode_s32'] = coder.decode(bboxes, pred, 32).numpy()
    gt = torch.tensor([[-40., -30., 70., 60.], [-8., -20., 100., 58.], [20., -25., 130., 70.], [50., -33., 175., 55.]])
    data['coder/encode_gt'] = gt.numpy()
    data['coder/encode_s32'] = coder.encode(bboxes, gt, 32).numpy()
    # ---- one training-mode forward + backward of the head's loss (yolo_head.py:393-560) ----------------
    train_head = v3.head.YOLOV3Head(
        num_classes=6, in_channels=[64, 32, 16], out_channels=[96, 64, 32],
        loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0, reduction='sum'),
        loss_conf=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0, reduction='sum'),
        loss_xy=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=2.0, reduction='sum'),
        loss_wh=dict(type='MSELoss', loss_weight=2.0, reduction='sum'),
        train_cfg=ref.ConfigDict(assigner=dict(type='GridAssigner', pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0)),
        test_cfg=test_cfg)
    train_head.load_state_dict(head.state_dict())
    for m in (backbone, neck, train_head):
        torch.nn.Module.train(m, True)       # batch-statistics BN everywhere (norm_eval is a Darknet.train() policy)
    gt_bboxes = [torch.tensor([[8.0, 10.0, 40.0, 44.0], [50.5, 5.0, 95.0, 30.0], [20.0, 30.0, 28.0, 62.0]]),
                 torch.tensor([[0.0, 0.0, 30.0, 20.0], [60.0, 20.0, 90.0, 63.5]])]
    gt_labels = [torch.tensor([3, 1, 5]), torch.tensor([0, 4])]
    for pm in (backbone, neck, train_head):
        pm.zero_grad()
    preds_t = train_head(neck(backbone(img)))[0]
    losses = train_head.loss(preds_t, gt_bboxes, gt_labels, [dict(), dict()])
    total = sum(sum(x.mean() for x in v) for k, v in losses.items() if 'loss' in k)
    total.backward()
    for k, v in losses.items():
        data['train/' + k] = torch.stack([x.reshape(()) for x in v]).detach().numpy()
    data['train/loss_total'] = total.detach().numpy()
    names, sums = [], []
    for pre, m in (('backbone', backbone), ('neck', neck), ('bbox_head', train_head)):
        for n_, prm in m.named_parameters():
            g_ = prm.grad
            assert g_ is not None, n_
            names.append(f'{pre}.{n_}')
            sums.append([float(g_.double().sum()), float(g_.double().abs().sum()), float(g_.double().pow(2).sum().sqrt())])
    data['train/grad_names'] = np.array(names)
    data['train/grad_sums'] = np.array(sums)
    data['train/grad/bbox_head.convs_pred.0.bias'] = train_head.convs_pred[0].bias.grad.numpy()
    data['train/grad/bbox_head.convs_pred.2.weight'] = train_head.convs_pred[2].weight.grad.numpy()
    data['train/grad/backbone.conv1.conv.weight'] = backbone.conv1.conv.weight.grad.numpy()
    for i, (b_, l_) in enumerate(zip(gt_bboxes, gt_labels)):
        data[f'train/gt_bboxes{i}'] = b_.numpy()
        data[f'train/gt_labels{i}'] = l_.numpy()
    # targets of the assigner for the same ground truth (per level: target map + negative map)
    sizes_ = [p_.shape[-2:] for p_ in preds_t]
    anchors_ = train_head.anchor_generator.grid_anchors(sizes_, 'cpu')
    flags_ = [train_head.anchor_generator.responsible_flags(sizes_, gb, 'cpu') for gb in gt_bboxes]
    tmaps, nmaps = train_head.get_targets([anchors_, anchors_], flags_, gt_bboxes, gt_labels)
    for i, (t_, n_) in enumerate(zip(tmaps, nmaps)):
        data[f'train/target_map{i}'] = t_.numpy()
        data[f'train/neg_map{i}'] = n_.numpy()
    data['meta_layers'] = np.array(ARCH[0])
    data['meta_channels'] = np.array(ARCH[1])
    out = os.path.join(HERE, 'tiny_v3.npz')
    np.savez_compressed(out, **data)
    print('v3', out, f'{os.path.getsize(out) / 1e6:.2f} MB', 'dets', [data[f'dets{n}'].shape for n in range(N)],
          'cfg2', [data[f'cfg2/dets{n}'].shape for n in range(N)], 'train loss', float(total),
          {k: v.tolist() for k, v in data.items() if k.startswith('train/loss_')})


if __name__ == '__main__':
    main()
