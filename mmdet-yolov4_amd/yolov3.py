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
MSE / BCE losses, ``yolo_head.py:393-604``) is not built.
"""
import torch
import torch.nn as nn
from torch.nn.modules.batchnorm import _BatchNorm

from . import ops
from .bricks import HipModule
from .darknetcsp import Conv
from .plan import Plan
from .registry import (BACKBONES, BBOX_CODERS, DETECTORS, HEADS, NECKS, ConfigDict, build_anchor_generator,
                       build_bbox_coder)
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
                 norm_cfg=_NORM, act_cfg=_ACT, loss_cls=None, loss_conf=None, loss_xy=None, loss_wh=None,
                 train_cfg=None, test_cfg=None,
                 init_cfg=dict(type='Normal', std=0.01, override=dict(name='convs_pred'))):
        super().__init__(None)
        assert len(in_channels) == len(out_channels) == len(featmap_strides)
        self.num_classes = num_classes
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.featmap_strides = featmap_strides
        self.train_cfg = ConfigDict(train_cfg) if isinstance(train_cfg, dict) else train_cfg
        self.test_cfg = ConfigDict(test_cfg) if isinstance(test_cfg, dict) else test_cfg
        self.one_hot_smoother = one_hot_smoother
        self.conv_cfg, self.norm_cfg, self.act_cfg = conv_cfg, norm_cfg, act_cfg
        self.bbox_coder = build_bbox_coder(bbox_coder)
        self.anchor_generator = build_anchor_generator(anchor_generator)
        self.loss_cfgs = dict(loss_cls=loss_cls, loss_conf=loss_conf, loss_xy=loss_xy, loss_wh=loss_wh)
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
        raise NotImplementedError('YOLOV3Head: the training graph (GridAssigner, MSE/BCE losses, '
                                  'yolo_head.py:393-604) is not built; call .eval()')

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

    def loss(self, *args, **kwargs):
        raise NotImplementedError('YOLOV3Head.loss (yolo_head.py:393-604) is not built')

    def forward_train(self, *args, **kwargs):
        raise NotImplementedError('YOLOV3Head training is not built')

    def aug_test(self, feats, img_metas, rescale=False):
        raise NotImplementedError('YOLOV3Head.aug_test (TTA) is not built')


@DETECTORS.register_module()
class YOLOV3(SingleStageDetector):
    """detectors/yolo.py: a SingleStageDetector under its own registry name."""

    def __init__(self, backbone, neck, bbox_head, train_cfg=None, test_cfg=None, pretrained=None, init_cfg=None):
        super().__init__(backbone, neck, bbox_head, train_cfg, test_cfg, pretrained, init_cfg)
