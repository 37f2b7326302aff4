"""YOLOv3 on the fused path: ``Darknet`` (53), ``YOLOV3Neck``, ``YOLOV3Head``, ``YOLOBBoxCoder``,
``YOLOV3`` -- registered under the reference's names, constructor arguments, attribute names (hence
checkpoint keys) and return structures.

Reference surface mirrored here:
  * ``mmdet/models/backbones/darknet.py:11-212``   ResBlock / Darknet (conv -> BN -> LeakyReLU(0.1))
  * ``mmdet/models/necks/yolo_neck.py:11-137``      DetectionBlock / YOLOV3Neck
  * ``mmdet/models/dense_heads/yolo_head.py:20-391`` YOLOV3Head: layers, forward, get_bboxes
  * ``mmdet/core/bbox/coder/yolo_bbox_coder.py``     YOLOBBoxCoder
  * ``mmdet/models/detectors/yolo.py``               YOLOV3 (a SingleStageDetector)
  * ``configs/yolo/yolov3_d53_*``                    the configs these classes are built from

Everything reuses the YOLOv4 machinery: ``Conv`` (mmcv ConvModule layout) with the LeakyReLU epilogue
of the fused conv kernel, the residual add in the conv epilogue, nearest-upsample + concat as channel-
offset stores, and decode + per-class NMS in two launches (``yv4_decode_filter_v3``: v3 box decode,
per-LEVEL top-k by objectness, ``conf_thr``, ``score_factors``).  Training of the v3 head (GridAssigner +
MSE / BCE losses, ``yolo_head.py:393-560``) runs on the HIP training kernels for the convs / BN and
torch tensor ops for the target assignment and the loss, like the YOLOv4 head's.
"""
import warnings

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.modules.batchnorm import _BatchNorm

from . import assigners as _assigners  # noqa: F401  (registers GridAssigner / PseudoSampler)
from . import ops
from . import train_ops as T
from .bricks import HipModule
from .darknetcsp import Conv
from .plan import Plan
from .registry import (BACKBONES, BBOX_CODERS, DETECTORS, HEADS, NECKS, ConfigDict, build_anchor_generator,
                       build_assigner, build_bbox_coder, build_loss, build_sampler)
from .single_stage import SingleStageDetector
from .yolocsp_head import collect_results, set_scale_factors

_NORM = dict(type='BN', requires_grad=True)
_ACT = dict(type='LeakyReLU', negative_slope=0.1)


class ResBlock(HipModule):
    """darknet.py:11-52: 1x1 (C -> C/2) -> 3x3 (C/2 -> C), input added after the second activation."""

    def __init__(self, in_channels, conv_cfg=None, norm_cfg=_NORM, act_cfg=_ACT, init_cfg=None):
        super().__init__(init_cfg)
        assert in_channels % 2 == 0
        cfg = dict(norm_cfg=norm_cfg, act_cfg=act_cfg)
        self.conv1 = Conv(in_channels, in_channels // 2, 1, **cfg)
        self.conv2 = Conv(in_channels // 2, in_channels, 3, padding=1, **cfg)

    def emit(self, plan, x, out=None):
        return self.conv2.emit(plan, self.conv1.emit(plan, x), out=out, residual=x)

    def fwd(self, x):
        return self.conv2.fwd(self.conv1.fwd(x), residual=x)

    def forward(self, x):
        return self._dispatch((x,), 'flat')


class _ConvResBlock(nn.Sequential):
    """darknet.py:182-212 ``make_conv_res_block``: Sequential('conv', 'res0', 'res1', ...)."""

    def emit(self, plan, x):
        for m in self:
            x = m.emit(plan, x)
        return x

    def fwd(self, x):
        for m in self:
            x = m.fwd(x)
        return x


@BACKBONES.register_module()
class Darknet(HipModule):
    """darknet.py:55-180."""

    arch_settings = {53: ((1, 2, 8, 8, 4), ((32, 64), (64, 128), (128, 256), (256, 512), (512, 1024)))}

    def __init__(self, depth=53, out_indices=(3, 4, 5), frozen_stages=-1, conv_cfg=None, norm_cfg=_NORM, act_cfg=_ACT,
                 norm_eval=True, pretrained=None, init_cfg=None):
        super().__init__(init_cfg)
        if depth not in self.arch_settings:
            raise KeyError(f'invalid depth {depth} for darknet')
        self.depth = depth
        self.out_indices = out_indices
        self.frozen_stages = frozen_stages
        self.layers, self.channels = self.arch_settings[depth]
        cfg = dict(norm_cfg=norm_cfg, act_cfg=act_cfg)
        self.conv1 = Conv(3, 32, 3, padding=1, **cfg)
        self.cr_blocks = ['conv1']
        for i, n_layers in enumerate(self.layers):
            layer_name = f'conv_res_block{i + 1}'
            in_c, out_c = self.channels[i]
            self.add_module(layer_name, self.make_conv_res_block(in_c, out_c, n_layers, **cfg))
            self.cr_blocks.append(layer_name)
        self.norm_eval = norm_eval
        assert not (init_cfg and pretrained), 'init_cfg and pretrained cannot be setting at the same time'
        if isinstance(pretrained, str):
            self.init_cfg = dict(type='Pretrained', checkpoint=pretrained)
        elif pretrained is None:
            if init_cfg is None:
                self.init_cfg = [dict(type='Kaiming', layer='Conv2d'),
                                 dict(type='Constant', val=1, layer=['_BatchNorm', 'GroupNorm'])]
        else:
            raise TypeError('pretrained must be a str or None')

    @staticmethod
    def make_conv_res_block(in_channels, out_channels, res_repeat, conv_cfg=None, norm_cfg=_NORM, act_cfg=_ACT):
        cfg = dict(norm_cfg=norm_cfg, act_cfg=act_cfg)
        model = _ConvResBlock()
        model.add_module('conv', Conv(in_channels, out_channels, 3, stride=2, padding=1, **cfg))
        for idx in range(res_repeat):
            model.add_module(f'res{idx}', ResBlock(out_channels, **cfg))
        return model

    def emit(self, plan, x):
        outs = []
        for i, name in enumerate(self.cr_blocks):
            x = getattr(self, name).emit(plan, x)
            if i in self.out_indices:
                outs.append(x)
        return tuple(outs)

    def fwd(self, x):
        outs = []
        for i, name in enumerate(self.cr_blocks):
            x = getattr(self, name).fwd(x)
            if i in self.out_indices:
                outs.append(x)
        return tuple(outs)

    def forward(self, x):
        return self._dispatch((x,), 'flat')

    def _freeze_stages(self):
        if self.frozen_stages >= 0:
            for i in range(self.frozen_stages):
                m = getattr(self, self.cr_blocks[i])
                m.eval()
                for param in m.parameters():
                    param.requires_grad = False

    def train(self, mode=True):
        super().train(mode)
        self._freeze_stages()
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, _BatchNorm):
                    m.eval()
        return self


class DetectionBlock(HipModule):
    """yolo_neck.py:11-61: 1x1 (n), 3x3 (2n), 1x1 (n), 3x3 (2n), 1x1 (n)."""

    def __init__(self, in_channels, out_channels, conv_cfg=None, norm_cfg=_NORM, act_cfg=_ACT, init_cfg=None):
        super().__init__(init_cfg)
        d = out_channels * 2
        cfg = dict(norm_cfg=norm_cfg, act_cfg=act_cfg)
        self.conv1 = Conv(in_channels, out_channels, 1, **cfg)
        self.conv2 = Conv(out_channels, d, 3, padding=1, **cfg)
        self.conv3 = Conv(d, out_channels, 1, **cfg)
        self.conv4 = Conv(out_channels, d, 3, padding=1, **cfg)
        self.conv5 = Conv(d, out_channels, 1, **cfg)

    def emit(self, plan, x, out=None):
        for c in (self.conv1, self.conv2, self.conv3, self.conv4):
            x = c.emit(plan, x)
        return self.conv5.emit(plan, x, out=out)

    def fwd(self, x):
        for c in (self.conv1, self.conv2, self.conv3, self.conv4, self.conv5):
            x = c.fwd(x)
        return x

    def forward(self, x):
        return self._dispatch((x,), 'flat')


@NECKS.register_module()
class YOLOV3Neck(HipModule):
    """yolo_neck.py:64-137: top-down; ``conv_i`` (built for ``in_channels[i]``) is applied to the previous
    DetectionBlock's output, upsampled x2 (nearest) and concatenated IN FRONT of the lateral feature."""

    def __init__(self, num_scales, in_channels, out_channels, conv_cfg=None, norm_cfg=_NORM, act_cfg=_ACT,
                 init_cfg=None):
        super().__init__(init_cfg)
        assert num_scales == len(in_channels) == len(out_channels)
        self.num_scales = num_scales
        self.in_channels = in_channels
        self.out_channels = out_channels
        cfg = dict(norm_cfg=norm_cfg, act_cfg=act_cfg)
        self.detect1 = DetectionBlock(in_channels[0], out_channels[0], **cfg)
        for i in range(1, self.num_scales):
            in_c, out_c = self.in_channels[i], self.out_channels[i]
            self.add_module(f'conv{i}', Conv(in_c, out_c, 1, **cfg))
            self.add_module(f'detect{i + 1}', DetectionBlock(in_c + out_c, out_c, **cfg))

    def emit(self, plan, feats):
        assert len(feats) == self.num_scales
        outs = []
        out = self.detect1.emit(plan, feats[-1])
        outs.append(out)
        for i, x in enumerate(reversed(feats[:-1])):
            tmp = getattr(self, f'conv{i + 1}').emit(plan, out)
            # F.interpolate(scale_factor=2) + torch.cat((tmp, x), 1): both land in one buffer
            cat = plan.new_buf(x.N, x.H, x.W, tmp.C + x.C, 'v3_up_cat')
            assert (x.H, x.W) == (2 * tmp.H, 2 * tmp.W), 'YOLOV3Neck: lateral map must be twice the upsampled one'
            plan.resample(tmp, cat.slice(0, tmp.C), name='upsample_nearest')
            plan.resample(x, cat.slice(tmp.C, x.C), name='concat_copy')
            out = getattr(self, f'detect{i + 2}').emit(plan, cat)
            outs.append(out)
        return tuple(outs)

    def fwd(self, feats):
        import torch.nn.functional as F
        outs = []
        out = self.detect1.fwd(feats[-1])
        outs.append(out)
        for i, x in enumerate(reversed(feats[:-1])):
            tmp = F.interpolate(getattr(self, f'conv{i + 1}').fwd(out), scale_factor=2)
            out = getattr(self, f'detect{i + 2}').fwd(torch.cat((tmp, x), 1))
            outs.append(out)
        return tuple(outs)

    def forward(self, feats):
        return self._dispatch((tuple(feats),), 'tuple')


@BBOX_CODERS.register_module()
class YOLOBBoxCoder:
    """core/bbox/coder/yolo_bbox_coder.py:8-89 (host-side API form; inference decodes inside
    ``yv4_decode_filter_v3``)."""

    def __init__(self, eps=1e-6):
        self.eps = eps

    def encode(self, bboxes, gt_bboxes, stride):
        assert bboxes.size(0) == gt_bboxes.size(0)
        assert bboxes.size(-1) == gt_bboxes.size(-1) == 4
        xg = (gt_bboxes[..., 0] + gt_bboxes[..., 2]) * 0.5
        yg = (gt_bboxes[..., 1] + gt_bboxes[..., 3]) * 0.5
        wg = gt_bboxes[..., 2] - gt_bboxes[..., 0]
        hg = gt_bboxes[..., 3] - gt_bboxes[..., 1]
        xc = (bboxes[..., 0] + bboxes[..., 2]) * 0.5
        yc = (bboxes[..., 1] + bboxes[..., 3]) * 0.5
        w = bboxes[..., 2] - bboxes[..., 0]
        h = bboxes[..., 3] - bboxes[..., 1]
        wt = torch.log((wg / w).clamp(min=self.eps))
        ht = torch.log((hg / h).clamp(min=self.eps))
        xt = ((xg - xc) / stride + 0.5).clamp(self.eps, 1 - self.eps)
        yt = ((yg - yc) / stride + 0.5).clamp(self.eps, 1 - self.eps)
        return torch.stack([xt, yt, wt, ht], dim=-1)

    def decode(self, bboxes, pred_bboxes, stride):
        assert pred_bboxes.size(0) == bboxes.size(0)
        assert pred_bboxes.size(-1) == bboxes.size(-1) == 4
        xc = (bboxes[..., 0] + bboxes[..., 2]) * 0.5
        yc = (bboxes[..., 1] + bboxes[..., 3]) * 0.5
        w = bboxes[..., 2] - bboxes[..., 0]
        h = bboxes[..., 3] - bboxes[..., 1]
        xp = (pred_bboxes[..., 0] - 0.5) * stride + xc
        yp = (pred_bboxes[..., 1] - 0.5) * stride + yc
        wp = torch.exp(pred_bboxes[..., 2]) * w
        hp = torch.exp(pred_bboxes[..., 3]) * h
        return torch.stack((xp - wp / 2, yp - hp / 2, xp + wp / 2, yp + hp / 2), dim=-1)


@HEADS.register_module()
class YOLOV3Head(HipModule):
    """yolo_head.py:20-391 (layers, forward, get_bboxes).  ``loss`` / training targets are not built."""

    def __init__(self, num_classes, in_channels, out_channels=(1024, 512, 256),
                 anchor_generator=dict(type='YOLOAnchorGenerator',
                                       base_sizes=[[(116, 90), (156, 198), (373, 326)],
                                                   [(30, 61), (62, 45), (59, 119)],
                                                   [(10, 13), (16, 30), (33, 23)]], strides=[32, 16, 8]),
                 bbox_coder=dict(type='YOLOBBoxCoder'), featmap_strides=[32, 16, 8], one_hot_smoother=0., conv_cfg=None,
                 norm_cfg=_NORM, act_cfg=_ACT,
                 loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
                 loss_conf=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
                 loss_xy=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
                 loss_wh=dict(type='MSELoss', loss_weight=1.0), train_cfg=None, test_cfg=None,
                 init_cfg=dict(type='Normal', std=0.01, override=dict(name='convs_pred'))):
        super().__init__(None)
        assert len(in_channels) == len(out_channels) == len(featmap_strides)
        self.num_classes = num_classes
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.featmap_strides = featmap_strides
        self.train_cfg = ConfigDict(train_cfg) if isinstance(train_cfg, dict) else train_cfg
        self.test_cfg = ConfigDict(test_cfg) if isinstance(test_cfg, dict) else test_cfg
        if self.train_cfg:                                        # yolo_head.py:91-97
            self.assigner = build_assigner(self.train_cfg.assigner)
            sampler_cfg = self.train_cfg.sampler if hasattr(self.train_cfg, 'sampler') else dict(type='PseudoSampler')
            self.sampler = build_sampler(sampler_cfg, context=self)
        self.one_hot_smoother = one_hot_smoother
        self.conv_cfg, self.norm_cfg, self.act_cfg = conv_cfg, norm_cfg, act_cfg
        self.bbox_coder = build_bbox_coder(bbox_coder)
        self.anchor_generator = build_anchor_generator(anchor_generator)
        self.loss_cls = build_loss(loss_cls)
        self.loss_conf = build_loss(loss_conf)
        self.loss_xy = build_loss(loss_xy)
        self.loss_wh = build_loss(loss_wh)
        self.num_anchors = self.anchor_generator.num_base_anchors[0]
        assert len(self.anchor_generator.num_base_anchors) == len(featmap_strides)
        self._init_layers()
        self.fp16_enabled = False
        self._post_cache = {}
        self._head_init = init_cfg

    @property
    def num_levels(self):
        return len(self.featmap_strides)

    @property
    def num_attrib(self):
        return 5 + self.num_classes

    def _init_layers(self):
        self.convs_bridge = nn.ModuleList()
        self.convs_pred = nn.ModuleList()
        for i in range(self.num_levels):
            self.convs_bridge.append(Conv(self.in_channels[i], self.out_channels[i], 3, padding=1,
                                          norm_cfg=self.norm_cfg, act_cfg=self.act_cfg))
            self.convs_pred.append(nn.Conv2d(self.out_channels[i], self.num_anchors * self.num_attrib, 1))

    def init_weights(self):
        """init_cfg Normal(std=0.01) with ``override=dict(name='convs_pred')`` (yolo_head.py:81-83)."""
        for m in self.convs_pred:
            nn.init.normal_(m.weight, 0, 0.01)
            nn.init.constant_(m.bias, 0)
        self._is_init = True

    # ---- plan contribution -----------------------------------------------------------------
    def emit(self, plan, feats):
        assert len(feats) == self.num_levels
        outs = []
        for i, x in enumerate(feats):
            x = self.convs_bridge[i].emit(plan, x)
            conv = self.convs_pred[i]
            outs.append(plan.conv(x, conv.weight, torch.ones(conv.out_channels), conv.bias.detach().float(), (0, 0.0),
                                  stride=1, pad=0, name=f'pred_conv{i}', out_f32=plan.h16))
        return tuple(outs)

    def emit_postprocess(self, plan, pred_views, cfg=None, rescale=True, want_cls=False):
        cfg = self.test_cfg if cfg is None else cfg
        nms_cfg = dict(cfg['nms'])
        if nms_cfg.get('type', 'nms') != 'nms':
            raise NotImplementedError('only nms type "nms" is built')
        return plan.postprocess(
            pred_views, self.featmap_strides, self.anchor_generator.base_anchors, self.num_classes,
            score_thr=cfg['score_thr'], iou_thr=nms_cfg.get('iou_threshold', nms_cfg.get('iou_thr')),
            max_per_img=cfg['max_per_img'], split_thr=nms_cfg.get('split_thr', ops.SPLIT_THR_DEFAULT),
            rescale=rescale, want_cls=want_cls, nms_pre=cfg.get('nms_pre', -1), v3=True,
            conf_thr=cfg.get('conf_thr', -1))

    # ---- reference API ------------------------------------------------------------------------
    def fwd(self, feats):
        """Training-mode forward: bridge ConvModule on the HIP training ops, biased 1x1 prediction conv with
        its output channels padded to a 16-byte multiple; dense fp32 pred maps (force_fp32, yolo_head.py:395)."""
        assert len(feats) == self.num_levels
        outs = []
        for i, x in enumerate(feats):
            x = self.convs_bridge[i].fwd(x)
            conv = self.convs_pred[i]
            co = conv.out_channels
            w = conv.weight
            dt = T.train_dtype(self, x)
            padc = (-co) % (4 if dt == torch.float32 else 8)
            if padc:
                w = F.pad(w, (0, 0, 0, 0, 0, 0, 0, padc))
            outs.append(T.conv2d(x, w, 1, 0, dtype=dt)[:, :co].float() + conv.bias.view(1, -1, 1, 1))
        return tuple(outs)

    def forward(self, feats):
        return self._dispatch((tuple(feats),), 'tuple'),

    def get_bboxes(self, pred_maps, img_metas, cfg=None, rescale=False, with_nms=True):
        assert len(pred_maps) == self.num_levels
        if not with_nms:
            raise NotImplementedError('YOLOV3Head.get_bboxes(with_nms=False) is not built')
        for t in pred_maps:
            ops._need_cuda(t, 'pred_map')
        cfg = self.test_cfg if cfg is None else cfg
        key = (tuple(tuple(p.shape) for p in pred_maps), bool(rescale), repr(dict(cfg)))
        plan = self._post_cache.get(key)
        if plan is None:
            self._post_cache.clear()
            plan = Plan(pred_maps[0].device)
            views = [plan.add_input_nchw(*p.shape, name=f'pred{i}', pad4=False) for i, p in enumerate(pred_maps)]
            self.emit_postprocess(plan, views, cfg, rescale=rescale)
            plan.finalize()
            self._post_cache[key] = plan
        set_scale_factors(plan.post, img_metas, rescale)
        plan.run(*[p.float() for p in pred_maps])
        return collect_results(plan.post, with_nms=True, head=self)

    # ---- training (yolo_head.py:393-560) ---------------------------------------------------------------
    def loss(self, pred_maps, gt_bboxes, gt_labels, img_metas, gt_bboxes_ignore=None):
        num_imgs = len(img_metas)
        pred_maps = [p.float() for p in pred_maps]
        device = pred_maps[0].device
        featmap_sizes = [pred_maps[i].shape[-2:] for i in range(self.num_levels)]
        multi_level_anchors = self.anchor_generator.grid_anchors(featmap_sizes, device)
        anchor_list = [multi_level_anchors for _ in range(num_imgs)]
        responsible_flag_list = [self.anchor_generator.responsible_flags(featmap_sizes, gt_bboxes[i], device)
                                 for i in range(num_imgs)]
        target_maps_list, neg_maps_list = self.get_targets(anchor_list, responsible_flag_list, gt_bboxes, gt_labels)
        res = [self.loss_single(p, t, n) for p, t, n in zip(pred_maps, target_maps_list, neg_maps_list)]
        losses_cls, losses_conf, losses_xy, losses_wh = (list(x) for x in zip(*res))
        return dict(loss_cls=losses_cls, loss_conf=losses_conf, loss_xy=losses_xy, loss_wh=losses_wh)

    def loss_single(self, pred_map, target_map, neg_map):
        num_imgs = len(pred_map)
        pred_map = pred_map.permute(0, 2, 3, 1).reshape(num_imgs, -1, self.num_attrib)
        neg_mask = neg_map.float()
        pos_mask = target_map[..., 4]
        pos_and_neg_mask = neg_mask + pos_mask
        pos_mask = pos_mask.unsqueeze(dim=-1)
        if torch.max(pos_and_neg_mask) > 1.:
            warnings.warn('There is overlap between pos and neg sample.')
            pos_and_neg_mask = pos_and_neg_mask.clamp(min=0., max=1.)
        loss_cls = self.loss_cls(pred_map[..., 5:], target_map[..., 5:], weight=pos_mask)
        loss_conf = self.loss_conf(pred_map[..., 4], target_map[..., 4], weight=pos_and_neg_mask)
        loss_xy = self.loss_xy(pred_map[..., :2], target_map[..., :2], weight=pos_mask)
        loss_wh = self.loss_wh(pred_map[..., 2:4], target_map[..., 2:4], weight=pos_mask)
        return loss_cls, loss_conf, loss_xy, loss_wh

    def get_targets(self, anchor_list, responsible_flag_list, gt_bboxes_list, gt_labels_list):
        num_imgs = len(anchor_list)
        num_level_anchors = [anchors.size(0) for anchors in anchor_list[0]]
        res = [self._get_targets_single(a, f, b, l)
               for a, f, b, l in zip(anchor_list, responsible_flag_list, gt_bboxes_list, gt_labels_list)]
        all_target_maps, all_neg_maps = (list(x) for x in zip(*res))
        assert num_imgs == len(all_target_maps) == len(all_neg_maps)
        return _images_to_levels(all_target_maps, num_level_anchors), _images_to_levels(all_neg_maps, num_level_anchors)

    def _get_targets_single(self, anchors, responsible_flags, gt_bboxes, gt_labels):
        anchor_strides = torch.cat([torch.tensor(self.featmap_strides[i], device=gt_bboxes.device).repeat(len(anchors[i]))
                                    for i in range(len(anchors))])
        concat_anchors = torch.cat(anchors)
        concat_responsible_flags = torch.cat(responsible_flags)
        assert len(anchor_strides) == len(concat_anchors) == len(concat_responsible_flags)
        assign_result = self.assigner.assign(concat_anchors, concat_responsible_flags, gt_bboxes)
        sampling_result = self.sampler.sample(assign_result, concat_anchors, gt_bboxes)
        target_map = concat_anchors.new_zeros(concat_anchors.size(0), self.num_attrib)
        pos = sampling_result.pos_inds
        target_map[pos, :4] = self.bbox_coder.encode(sampling_result.pos_bboxes, sampling_result.pos_gt_bboxes,
                                                     anchor_strides[pos])
        target_map[pos, 4] = 1
        one_hot = F.one_hot(gt_labels, num_classes=self.num_classes).float()
        if self.one_hot_smoother != 0:
            one_hot = one_hot * (1 - self.one_hot_smoother) + self.one_hot_smoother / self.num_classes
        target_map[pos, 5:] = one_hot[sampling_result.pos_assigned_gt_inds]
        neg_map = concat_anchors.new_zeros(concat_anchors.size(0), dtype=torch.uint8)
        neg_map[sampling_result.neg_inds] = 1
        return target_map, neg_map

    def forward_train(self, x, img_metas, gt_bboxes, gt_labels=None, gt_bboxes_ignore=None, proposal_cfg=None,
                      **kwargs):
        """base_dense_head.py:22-59."""
        outs = self(x)
        loss_inputs = outs + ((gt_bboxes, img_metas) if gt_labels is None else (gt_bboxes, gt_labels, img_metas))
        losses = self.loss(*loss_inputs, gt_bboxes_ignore=gt_bboxes_ignore)
        if proposal_cfg is None:
            return losses
        return losses, self.get_bboxes(*outs, img_metas, cfg=proposal_cfg)

    def aug_test(self, feats, img_metas, rescale=False):
        raise NotImplementedError('YOLOV3Head.aug_test (TTA) is not built')


def _images_to_levels(target, num_levels):
    """core/anchor/utils.py ``images_to_levels``: [per image: (A_total, ...)] -> [per level: (N, A_l, ...)]."""
    target = torch.stack(target, 0)
    out, start = [], 0
    for n in num_levels:
        out.append(target[:, start:start + n])
        start += n
    return out


@DETECTORS.register_module()
class YOLOV3(SingleStageDetector):
    """detectors/yolo.py: a SingleStageDetector under its own registry name."""

    def __init__(self, backbone, neck, bbox_head, train_cfg=None, test_cfg=None, pretrained=None, init_cfg=None):
        super().__init__(backbone, neck, bbox_head, train_cfg, test_cfg, pretrained, init_cfg)
