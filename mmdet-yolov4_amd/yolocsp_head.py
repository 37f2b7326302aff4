"""``YOLOCSPHead`` under the reference's registry name.

Mirror of ``mmdet/models/dense_heads/yolocsp_head.py:53-382`` for inference: three
biased 1x1 pred convs (``convs_pred.{0,1,2}.{weight,bias}``), ``forward`` returning the
1-tuple-of-tuple of NCHW pred maps (Q5), and ``get_bboxes`` returning, per image,
``(dets (n,5) fp32, labels (n,) int64)``.

On the HIP path ``get_bboxes`` is two launches over the batch -- ``yv4_decode_filter``
(sigmoid, xy/wh transform, anchor decode, conf*cls, rescale, threshold, compaction;
yolocsp_head.py:263-285,357-366 + bbox_nms.py:36-67) and ``yv4_nms_images``
(``batched_nms``, bbox_nms.py:84-88) -- with no host sync in between; the reference's
per-image python loop (:298-309), its (anchors x 81) score matrix and its 29 MB expanded
box tensor never exist.
"""
import ctypes as C
from collections import OrderedDict
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from . import losses as _losses  # noqa: F401  (registers GIoULoss / CrossEntropyLoss / SoftFocalLoss)
from . import ops
from ._lib import check
from . import train_ops as T
from .bricks import HipModule, normal_init
from .plan import Plan
from .registry import HEADS, LOSSES, ConfigDict, build_anchor_generator, build_bbox_coder, build_loss


class HeadTapFunction(torch.autograd.Function):
    """What the loss reads from one level's raw head-conv output, without ever materialising the
    reference's dense fp32 ``(N, H*W*A, 85)`` view of it (yolocsp_head.py:433-437: permute + reshape,
    ``pred_map[..., 4]``, ``pred_map[img_ind, anchor_ind]``): the objectness logit of every anchor
    box and the full attribute rows of the positive ones, bias added, in fp32.  The backward writes the
    two gradients straight into ONE zero-initialised tensor of the conv output's own type and layout --
    the reference's graph makes ~10 passes over 380 MB fp32 tensors per level for the same result."""

    @staticmethod
    def forward(ctx, raw, bias, img_ind, anchor_ind, A, attr):
        N, Cp, H, W = raw.shape
        nhwc = raw.permute(0, 2, 3, 1)                       # channels_last storage: a contiguous view
        assert nhwc.is_contiguous(), 'HeadTap: the conv output must be channels_last'
        conf = nhwc[..., 4:A * attr:attr].float() + bias[4::attr]
        cell = torch.div(anchor_ind, A, rounding_mode='floor')
        a = anchor_ind - cell * A
        rows = img_ind * (H * W) + cell
        cols = a[:, None] * attr + torch.arange(attr, device=raw.device)[None]
        pos = nhwc.reshape(N * H * W, Cp)[rows[:, None], cols].float() + bias[cols]
        ctx.save_for_backward(rows, cols)
        ctx.meta = (raw.shape, raw.dtype, A, attr, bias.shape[0])
        return conf.reshape(N, H * W * A), pos

    @staticmethod
    def backward(ctx, dconf, dpos):
        rows, cols = ctx.saved_tensors
        (N, Cp, H, W), dtype, A, attr, nb = ctx.meta
        draw = torch.zeros((N, Cp, H, W), dtype=dtype, device=dconf.device).contiguous(memory_format=torch.channels_last)
        d = draw.permute(0, 2, 3, 1)
        dbias = torch.zeros(nb, dtype=torch.float32, device=dconf.device)
        dc = dconf.reshape(N, H, W, A)
        d[..., 4:A * attr:attr] = dc.to(dtype)
        # (rows, A) -> A long contiguous rows: ATen's reduction over the OUTER dims of a 3-column matrix runs on
        # a handful of workgroups (5.8 ms for the stride-8 level at batch 64); the transposed copy is 13 MB
        dbias[4::attr] = dc.reshape(-1, A).t().contiguous().sum(1)
        if rows.numel():
            d.view(N * H * W, Cp).index_put_((rows[:, None], cols), dpos.to(dtype), accumulate=True)
            dbias.index_put_((cols.reshape(-1),), dpos.reshape(-1).float(), accumulate=True)
        return draw, dbias, None, None, None, None


class YoloLossFunction(torch.autograd.Function):
    """All levels' ``loss_single_no_assigner`` (+ ``responsible_indices`` and the target gathering) on
    ``yv4_yolo_loss_fwd`` / ``yv4_yolo_loss_bwd``: returns the (num_levels, 3) fp32 matrix
    [loss_cls, loss_conf (before the level balance weight), loss_bbox * loss_bbox_weight]; the backward
    writes each level's whole conv-output gradient and bias gradient in one pass.  Nothing is read back
    by the host in either direction (the reference path has two ``nonzero()`` syncs per level)."""

    @staticmethod
    def forward(ctx, head, gt, gt_label, gt_img, *maps):
        L = head.num_levels
        raws, biases = maps[:L], maps[L:]
        dev = raws[0].device
        A = head.num_anchors[0]
        C_ = 0 if head.class_agnostic else head.num_classes
        attr = 5 + C_
        N = raws[0].shape[0]
        G = int(gt.shape[0])
        d = _lib.LossDesc()
        d.num_levels, d.N, d.A, d.num_classes, d.G = L, N, A, C_, G
        d.dtype = T._DCODE[raws[0].dtype]
        keep = []
        TA = 0
        for l in range(L):
            raw = raws[l]
            _, Cp, H, W = raw.shape
            assert raw.permute(0, 2, 3, 1).is_contiguous(), 'fused loss: the conv output must be channels_last'
            assert head.num_anchors[l] == A and raw.dtype == raws[0].dtype
            b = biases[l].detach().float().contiguous()
            keep.append(b)
            lv = d.levels[l]
            lv.raw, lv.bias = raw.data_ptr(), b.data_ptr()
            lv.H, lv.W, lv.Cp, lv.stride = H, W, Cp, int(head.featmap_strides[l])
            ba = head.anchor_generator.base_anchors[l].float().cpu()
            for k in range(A):
                for c in range(4):
                    lv.base_anchors[k][c] = float(ba[k, c])
            TA += H * W * A
        S = 5 * A * G
        i32 = dict(dtype=torch.int32, device=dev)
        slot_anchor = torch.empty(max(L * S, 1), **i32)
        winner = torch.empty(N * TA, **i32)
        npos = torch.empty(L, **i32)
        conf_t = torch.empty(max(L * S, 1), dtype=torch.float32, device=dev)
        sums = torch.empty(2, L, 3, dtype=torch.float64, device=dev)[0]   # (second half: deterministic mode's lo words)
        gt = gt.detach().float().contiguous()
        gt_label = gt_label.long().contiguous()
        d.gt, d.gt_label, d.gt_img = gt.data_ptr(), gt_label.data_ptr(), gt_img.data_ptr()
        d.shape_thr, d.smooth, d.ratio = float(head.shape_match_thres), float(head.one_hot_smoother), \
            float(head.conf_iou_loss_ratio)
        d.eps = float(head.loss_bbox.eps)
        d.w_cls = float(head.loss_cls.loss_weight) if C_ else 0.
        d.w_conf, d.w_bbox = float(head.loss_conf.loss_weight), float(head.loss_bbox_weight)
        d.slot_anchor, d.winner, d.npos, d.conf_t, d.sums = (t.data_ptr() for t in (slot_anchor, winner, npos, conf_t,
                                                                                   sums))
        out = torch.empty(L, 3, dtype=torch.float32, device=dev)
        d.losses = out.data_ptr()
        check(_lib.lib().yv4_yolo_loss_fwd(C.byref(d), ops.stream_ptr()), 'yv4_yolo_loss_fwd')
        ctx.desc = d
        ctx.keep = (raws, keep, gt, gt_label, gt_img, slot_anchor, winner, npos, conf_t, sums)
        ctx.meta = (L, S, A, attr)
        # (the (L, 3) losses come out of the forward call itself: d.losses, set above)
        return out

    @staticmethod
    def backward(ctx, gout):
        d = ctx.desc
        raws = ctx.keep[0]
        L, S, A, attr = ctx.meta
        dev = raws[0].device
        gout = gout.float().contiguous()
        draws = [torch.empty_like(r) for r in raws]
        # scratch sized for the deterministic mode's fixed-point words (include/yv4.h: 2 x dbias, 4 x gpos)
        dbias = [torch.empty(2, A * attr, dtype=torch.float64, device=dev)[0] for _ in range(L)]
        gpos = torch.empty(max(L * S * attr, 1) * (4 if ops.deterministic() else 1), dtype=torch.float32, device=dev)
        for l in range(L):
            d.levels[l].draw, d.levels[l].dbias = draws[l].data_ptr(), dbias[l].data_ptr()
        d.gpos = gpos.data_ptr()
        check(_lib.lib().yv4_yolo_loss_bwd(C.byref(d), gout.data_ptr(), ops.stream_ptr()), 'yv4_yolo_loss_bwd')
        return (None, None, None, None) + tuple(draws) + tuple(b.float() for b in dbias)


class FusedLosses(OrderedDict):
    """The loss dict of ``YOLOCSPHead.loss`` (yolocsp_head.py:384-436: per key a list of per-level tensors) when the fused
    kernels produced it.  ``weighted`` is the (num_levels, 3) matrix [loss_cls | loss_conf * level balance | loss_bbox]
    and ``total`` its sum -- what ``_parse_losses`` (detectors/base.py:171-204) adds up one tensor at a time: a detector
    that knows this class takes the total and the column sums directly (a handful of launches instead of ~80 tiny ones
    through the forward and the backward of the aggregation, during which the device idles on a host-bound stream of
    5 us kernels).  The per-level lists of the reference's dict are built the first time anybody looks at the mapping."""

    def __init__(self, weighted, num_gts, with_cls):
        super().__init__()
        self.weighted, self.num_gts, self.with_cls = weighted, num_gts, with_cls
        self.total = weighted.sum()
        self.built = False
        self.mutated = False

    def _build(self):
        if not self.built:
            self.built = True
            w = self.weighted
            L = w.shape[0]
            if self.with_cls:
                OrderedDict.__setitem__(self, 'loss_cls', [w[l, 0].reshape(1) for l in range(L)])
            OrderedDict.__setitem__(self, 'loss_conf', [w[l, 1] for l in range(L)])
            OrderedDict.__setitem__(self, 'loss_bbox', [w[l, 2].reshape(1) for l in range(L)])
            OrderedDict.__setitem__(self, 'num_gts', self.num_gts)
        return self

    def __getitem__(self, k):
        return OrderedDict.__getitem__(self._build(), k)

    def __iter__(self):
        return OrderedDict.__iter__(self._build())

    def __len__(self):
        return OrderedDict.__len__(self._build())

    def __contains__(self, k):
        return OrderedDict.__contains__(self._build(), k)

    def keys(self):
        return OrderedDict.keys(self._build())

    def items(self):
        return OrderedDict.items(self._build())

    def values(self):
        return OrderedDict.values(self._build())

    def get(self, k, default=None):
        return OrderedDict.get(self._build(), k, default)

    def __repr__(self):
        return OrderedDict.__repr__(self._build())

    # Mutation (a detector or wrapper that adds an auxiliary loss before ``_parse_losses``): the mapping is built first, and
    # ``built`` then tells ``single_stage._parse_losses`` to sum every key the way detectors/base.py:171-204 does instead of
    # taking ``total`` -- an added loss must not silently drop out of the optimised sum and of ``log_vars``.
    def __setitem__(self, k, v):
        self._build()
        self.mutated = True
        OrderedDict.__setitem__(self, k, v)

    def __delitem__(self, k):
        self._build()
        self.mutated = True
        OrderedDict.__delitem__(self, k)

    def update(self, *a, **kw):
        self._build()
        self.mutated = True
        OrderedDict.update(self, *a, **kw)

    def setdefault(self, k, default=None):
        self._build()
        if not OrderedDict.__contains__(self, k):
            self.mutated = True
        return OrderedDict.setdefault(self, k, default)

    def pop(self, k, *default):
        self._build()
        self.mutated = True
        return OrderedDict.pop(self, k, *default)


def _upload(host, device):
    """Host tensor -> device through pinned memory, stream-ordered: a pageable source makes the copy wait for
    everything queued before it (the whole forward pass, when the loss is where it is issued), and with the host
    parked there the backward pass cannot be queued ahead.  torch's caching host allocator keeps the pinned block
    until the copy has executed."""
    if device.type != 'cuda':
        return host.to(device)
    return host.pin_memory().to(device, non_blocking=True)


class RawPredMap:
    """One level's head-conv output in training mode: the raw (bias-free, channel-padded, possibly
    16-bit) NHWC tensor plus what is needed to read it the way the loss does."""

    def __init__(self, raw, bias, A, attr):
        self.raw, self.bias, self.A, self.attr = raw, bias, A, attr
        self.shape = (raw.shape[0], A * attr, raw.shape[2], raw.shape[3])
        self.device = raw.device

    def __len__(self):
        return self.shape[0]

    def tap(self, img_ind, anchor_ind):
        return HeadTapFunction.apply(self.raw, self.bias, img_ind, anchor_ind, self.A, self.attr)

    def dense(self):
        """The reference's (N, A*attr, H, W) fp32 pred map (what ``forward`` returns)."""
        return self.raw[:, :self.A * self.attr].float() + self.bias.view(1, -1, 1, 1)


@HEADS.register_module()
class YOLOCSPHead(HipModule):

    def __init__(self, num_classes, in_channels,
                 anchor_generator=dict(type='YOLOV4AnchorGenerator',
                                       base_sizes=[[(12, 16), (19, 36), (40, 28)],
                                                   [(36, 75), (76, 55), (72, 146)],
                                                   [(142, 110), (192, 243), (459, 401)]],
                                       strides=[8, 16, 32]),
                 bbox_coder=dict(type='YOLOV4BBoxCoder'), featmap_strides=[8, 16, 32], one_hot_smoother=0.,
                 conv_cfg=None, norm_cfg=dict(type='BN', requires_grad=True, eps=0.001, momentum=0.03),
                 act_cfg=dict(type='Mish'),
                 loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=32.),
                 loss_conf=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=64.),
                 loss_bbox=dict(type='GIoULoss', loss_weight=3.2), class_agnostic=False, train_cfg=None,
                 test_cfg=None, init_cfg=None):
        super().__init__(init_cfg)
        assert len(in_channels) == len(featmap_strides)
        self.num_classes = num_classes
        self.in_channels = in_channels
        self.featmap_strides = featmap_strides
        self.train_cfg = ConfigDict(train_cfg) if isinstance(train_cfg, dict) else train_cfg
        self.test_cfg = ConfigDict(test_cfg) if isinstance(test_cfg, dict) else test_cfg
        self.assigner = None
        self.sampler = None
        self.shape_match_thres = 4.
        self.conf_iou_loss_ratio = 1.
        self.conf_level_balance_weight = [4.0, 1.0, 0.4, 0.1, 0.1]
        self.class_freq = None
        self.num_obj_avg = 8
        if self.train_cfg is not None:
            for key, attr in (('conf_iou_loss_ratio', 'conf_iou_loss_ratio'),
                              ('conf_level_balance_weight', 'conf_level_balance_weight'),
                              ('class_frequency', 'class_freq'), ('num_obj_per_image', 'num_obj_avg'),
                              ('shape_match_thres', 'shape_match_thres')):
                if hasattr(self.train_cfg, key):
                    setattr(self, attr, self.train_cfg[key])
        self.one_hot_smoother = one_hot_smoother
        self.conv_cfg, self.norm_cfg, self.act_cfg = conv_cfg, norm_cfg, act_cfg
        self.bbox_coder = build_bbox_coder(bbox_coder)
        self.anchor_generator = build_anchor_generator(anchor_generator)
        self.class_agnostic = class_agnostic
        if not self.class_agnostic:                              # yolocsp_head.py:155-156
            self.loss_cls = build_loss(loss_cls)
        self.loss_conf = build_loss(loss_conf)
        self.loss_bbox = build_loss(loss_bbox)
        self.loss_bbox_weight = self.loss_bbox.loss_weight      # yolocsp_head.py:160-162
        self.loss_bbox.loss_weight = 1.
        self.num_anchors = self.anchor_generator.num_base_anchors
        self._init_layers()
        self.fp16_enabled = False
        self._post_cache = {}

    @property
    def num_levels(self):
        return len(self.featmap_strides)

    @property
    def num_attrib(self):
        return 5 + self.num_classes if not self.class_agnostic else 5

    def _init_layers(self):
        self.convs_pred = nn.ModuleList()
        for i in range(self.num_levels):
            self.convs_pred.append(nn.Conv2d(self.in_channels[i], self.num_anchors[i] * self.num_attrib, 1))

    def init_weights(self):
        """yolocsp_head.py:187-201 with the values it intends (Q4: the reference's in-place
        edit of a leaf view raises on modern torch; done here under no_grad)."""
        for m in self.convs_pred:
            normal_init(m, std=0.01)
        with torch.no_grad():
            for m, stride in zip(self.convs_pred, self.featmap_strides):
                b = m.bias.view(-1, self.num_attrib)
                b[:, 4] += math.log(self.num_obj_avg / (640 / stride) ** 2)
                if not self.class_agnostic:
                    if self.class_freq is None:
                        b[:, 5:] += math.log(0.6 / (self.num_classes - 0.99))
                    else:
                        cf = torch.as_tensor(self.class_freq, dtype=b.dtype)
                        b[:, 5:] += torch.log(cf / cf.sum())

    # ---- plan contribution ----------------------------------------------------------------
    def emit(self, plan, feats):
        """1x1 pred convs -> dense NHWC pred maps (the layout yolocsp_head.py:264 permutes to)."""
        assert len(feats) == self.num_levels
        outs = []
        for i, x in enumerate(feats):
            conv = self.convs_pred[i]
            s = torch.ones(conv.out_channels)
            t = conv.bias.detach().float()
            outs.append(plan.conv(x, conv.weight, s, t, (0, 0.0), stride=1, pad=0, name=f'pred_conv{i}',
                                  out_f32=plan.h16))      # decode reads fp32 pred maps
        return tuple(outs)

    def emit_postprocess(self, plan, pred_views, cfg=None, rescale=True, want_cls=False):
        cfg = self.test_cfg if cfg is None else cfg
        nms_cfg = dict(cfg['nms'])
        if nms_cfg.get('type', 'nms') != 'nms':
            raise NotImplementedError('only nms type "nms" is built')
        nms_pre = cfg.get('nms_pre', -1)
        return plan.postprocess(
            pred_views, self.featmap_strides, self.anchor_generator.base_anchors, self.num_classes,
            score_thr=cfg['score_thr'], iou_thr=nms_cfg.get('iou_threshold', nms_cfg.get('iou_thr')),
            max_per_img=cfg['max_per_img'], split_thr=nms_cfg.get('split_thr', ops.SPLIT_THR_DEFAULT),
            rescale=rescale, want_cls=want_cls, nms_pre=nms_pre, class_agnostic=self.class_agnostic)

    # ---- reference API ----------------------------------------------------------------------
    def fwd_raw(self, feats):
        """Training-mode head: biased 1x1 convs through the HIP conv with the bias left out and the
        output channels padded 255 -> 256 (16-byte aligned NHWC gradient); the loss reads the result
        through ``RawPredMap.tap``."""
        assert len(feats) == self.num_levels
        outs = []
        for i, (conv, x) in enumerate(zip(self.convs_pred, feats)):
            co = conv.out_channels
            w = conv.weight
            dt = T.train_dtype(self, x)
            padc = (-co) % (4 if dt == torch.float32 else 8)
            if padc:
                w = F.pad(w, (0, 0, 0, 0, 0, 0, 0, padc))
            outs.append(RawPredMap(T.conv2d(x, w, 1, 0, dtype=dt), conv.bias, self.num_anchors[i], self.num_attrib))
        return tuple(outs)

    def fwd(self, feats):
        """Training-mode ``forward``: dense fp32 pred maps like the reference's (losses are computed in
        fp32, yolocsp_head.py:384 force_fp32)."""
        return tuple(r.dense() for r in self.fwd_raw(feats))

    def forward(self, feats):
        return self._dispatch((tuple(feats),), 'tuple'),

    def get_bboxes(self, pred_maps, img_metas, cfg=None, rescale=False, with_nms=True):
        """pred_maps: NCHW tensors as returned by ``forward``.  Returns
        ``[(dets(n,5), labels(n,) int64)] * num_images``."""
        assert len(pred_maps) == self.num_levels
        for t in pred_maps:
            ops._need_cuda(t, 'pred_map')
        cfg = self.test_cfg if cfg is None else cfg
        key = (tuple(tuple(p.shape) for p in pred_maps), bool(rescale), bool(with_nms), repr(dict(cfg)))
        plan = self._post_cache.get(key)
        if plan is None:
            self._post_cache.clear()
            plan = Plan(pred_maps[0].device)
            views = []
            for i, p in enumerate(pred_maps):
                N, Cc, H, W = p.shape
                views.append(plan.add_input_nchw(N, Cc, H, W, name=f'pred{i}', pad4=False))
            self.emit_postprocess(plan, views, cfg, rescale=rescale, want_cls=not with_nms)
            plan.finalize()
            self._post_cache[key] = plan
        set_scale_factors(plan.post, img_metas, rescale)
        plan.run(*[p.float() for p in pred_maps])
        return collect_results(plan.post, with_nms=with_nms, head=self)

    def aug_test(self, feats, img_metas, rescale=False):
        """yolocsp_head.py:577-593 delegates to ``BBoxTestMixin.aug_test_bboxes``
        (dense_test_mixins.py:38-100), which for this head cannot run in the reference either:
        ``get_bboxes(with_nms=False)`` returns ``((n,5) dets, (n,) class ids)`` (yolocsp_head.py:377-382)
        and the mixin hands the class-id vector to ``multiclass_nms`` as its ``(n, #class+1)`` score
        matrix (``multi_scores.size(1)`` on a 1-D tensor raises).  The ``with_nms=False`` branch itself
        is built (including its double objectness factor); the merge is not, because there
        is no reference behaviour to reproduce."""
        raise NotImplementedError('YOLOCSPHead.aug_test: the reference\'s TTA merge for this head is not '
                                  'executable (see docstring); use simple_test')

    # ---- training (yolocsp_head.py:384-575) -------------------------------------------------------
    def loss(self, pred_maps, gt_bboxes, gt_labels, img_metas, gt_bboxes_ignore=None):
        device = pred_maps[0].device
        num_gts = _upload(torch.tensor([g.size(0) for g in gt_bboxes], dtype=torch.float32).mean(), device)
        pred_maps = [p if isinstance(p, RawPredMap) else p.float() for p in pred_maps]
        featmap_sizes = [pred_maps[i].shape[-2:] for i in range(self.num_levels)]
        if self.assigner is not None:
            raise NotImplementedError
        if self._fused_loss_ok(pred_maps):
            return self._loss_fused(pred_maps, gt_bboxes, gt_labels, num_gts)
        resp = self.anchor_generator.responsible_indices(
            featmap_sizes, gt_bboxes, neighbor=2, shape_match_thres=self.shape_match_thres, device=device)
        pos, tb, tl = self.get_targets_no_assigner(resp, gt_bboxes, gt_labels)
        anchors = self.anchor_generator.grid_anchors(featmap_sizes, device)
        l_cls, l_conf, l_box = [], [], []
        for lvl in range(self.num_levels):
            c, f, b = self.loss_single_no_assigner(pred_maps[lvl], anchors[lvl], self.featmap_strides[lvl],
                                                   pos[lvl], tb[lvl], tl[lvl])
            l_cls.append(c)
            l_conf.append(f * self.conf_level_balance_weight[lvl])
            l_box.append(b)
        if not self.class_agnostic:
            return dict(loss_cls=l_cls, loss_conf=l_conf, loss_bbox=l_box, num_gts=num_gts)
        return dict(loss_conf=l_conf, loss_bbox=l_box, num_gts=num_gts)

    def _fused_loss_ok(self, pred_maps):
        """The fused kernels cover the configuration the recipes use: raw (training-mode) maps, sigmoid
        CrossEntropyLoss without class weights and GIoULoss, both with mean reduction.  Anything else takes the
        tensor-op path below (on the GPU as well).  YV4_FUSED_LOSS=0 forces that path (A/B, tests)."""
        if os.environ.get('YV4_FUSED_LOSS', '1') == '0':
            return False
        if not all(isinstance(p, RawPredMap) for p in pred_maps) or len(pred_maps) > 5:
            return False
        ok = type(self.loss_conf) is _losses.CrossEntropyLoss and type(self.loss_bbox) is _losses.GIoULoss
        ok = ok and self.loss_conf.class_weight is None and self.loss_conf.reduction == 'mean'
        ok = ok and self.loss_bbox.reduction == 'mean'
        if not self.class_agnostic:
            ok = ok and type(self.loss_cls) is _losses.CrossEntropyLoss and self.loss_cls.class_weight is None \
                and self.loss_cls.reduction == 'mean'
        return ok and len(set(self.num_anchors)) == 1 and self.num_anchors[0] <= 8

    def _loss_fused(self, pred_maps, gt_bboxes, gt_labels, num_gts):
        device = pred_maps[0].device
        sizes = [int(g.shape[0]) for g in gt_bboxes]
        gt = torch.cat(list(gt_bboxes), dim=0).reshape(-1, 4)
        labels = torch.cat(list(gt_labels), dim=0).reshape(-1)
        img = _upload(torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes)), device)
        out = YoloLossFunction.apply(self, gt, labels, img, *[p.raw for p in pred_maps], *[p.bias for p in pred_maps])
        key = (str(device), self.num_levels)
        wts = self._loss_wts.get(key) if hasattr(self, '_loss_wts') else None
        if wts is None:
            if not hasattr(self, '_loss_wts'):
                self._loss_wts = {}
            wts = _upload(torch.tensor([[1.0, float(self.conf_level_balance_weight[l]), 1.0] for l in range(self.num_levels)],
                                       dtype=torch.float32), device)
            self._loss_wts[key] = wts
        fl = FusedLosses(out * wts, num_gts, with_cls=not self.class_agnostic)
        if os.environ.get('YV4_LOSS_MATRIX', '1') == '0':      # A/B: the reference's one-tensor-at-a-time aggregation
            fl._build()
        return fl

    def loss_single_no_assigner(self, pred_map, anchors, stride, pos_indices, target_bboxes, target_labels):
        num_imgs = len(pred_map)
        img_ind, anchor_ind = pos_indices
        if isinstance(pred_map, RawPredMap):      # training fast path: no dense fp32 view of the map
            pred_conf, pos_all = pred_map.tap(img_ind, anchor_ind)
        else:
            pred_map = pred_map.permute(0, 2, 3, 1).reshape(num_imgs, -1, self.num_attrib)
            pred_conf = pred_map[..., 4]
            pos_all = pred_map[img_ind, anchor_ind] if anchor_ind.numel() else None
        target_conf = torch.zeros_like(pred_conf, requires_grad=False)
        loss_bbox = pred_conf.new_zeros((1,))
        loss_cls = pred_conf.new_zeros((1,))
        if anchor_ind.numel():
            pos = pos_all
            pb = pos[..., :4].sigmoid()
            xy = pb[..., :2] * 2. - 1.
            wh = (pb[..., 2:] * 2.) ** 2.
            box = self.bbox_coder.decode(anchors[anchor_ind], torch.cat((xy, wh), dim=-1), stride)
            giou_loss = self.loss_bbox(box, target_bboxes, reduction_override='none')
            loss_bbox = loss_bbox + _losses.reduce_loss(giou_loss, self.loss_bbox.reduction)
            if not self.class_agnostic:
                loss_cls = loss_cls + self.loss_cls(pos[..., 5:], target_labels)
            r = self.conf_iou_loss_ratio
            conf_t = (1 - r) + r * (1 - giou_loss).detach().clamp(0.0, 1.0)
            target_conf[img_ind, anchor_ind] = conf_t.type(target_conf.dtype)
        loss_conf = self.loss_conf(pred_conf, target_conf)
        return loss_cls, loss_conf, loss_bbox * self.loss_bbox_weight

    def get_targets_no_assigner(self, responsible_indices_list, gt_bboxes_list, gt_labels_list):
        gt_bboxes = torch.cat(gt_bboxes_list, dim=0)
        gt_labels = torch.cat(gt_labels_list, dim=0)
        pos, tb, tl = [], [], []
        for lvl in range(self.num_levels):
            img_ind, anchor_ind, gt_ind = responsible_indices_list[lvl]
            pos.append((img_ind, anchor_ind))
            tb.append(gt_bboxes[gt_ind])
            t = F.one_hot(gt_labels[gt_ind], num_classes=self.num_classes).float()
            if self.one_hot_smoother != 0:
                t = t * (1 - self.one_hot_smoother) + self.one_hot_smoother / self.num_classes
            tl.append(t)
        return pos, tb, tl

    def forward_train(self, x, img_metas, gt_bboxes, gt_labels=None, gt_bboxes_ignore=None, proposal_cfg=None,
                      **kwargs):
        """base_dense_head.py:22-59."""
        outs = (self.fwd_raw(x),) if (self.training and proposal_cfg is None) else self(x)
        if gt_labels is None:
            loss_inputs = outs + (gt_bboxes, img_metas)
        else:
            loss_inputs = outs + (gt_bboxes, gt_labels, img_metas)
        losses = self.loss(*loss_inputs, gt_bboxes_ignore=gt_bboxes_ignore)
        if proposal_cfg is None:
            return losses
        return losses, self.get_bboxes(*outs, img_metas, cfg=proposal_cfg)


def set_scale_factors(post, img_metas, rescale):
    if not rescale:
        return
    sf = torch.tensor([[float(v) for v in m['scale_factor']] for m in img_metas], dtype=torch.float32)
    assert sf.shape == (post['N'], 4), f'scale_factor must be 4 numbers per image, got {tuple(sf.shape)}'
    post['scale_factor'].copy_(sf, non_blocking=False)


def collect_results(post, with_nms=True, head=None):
    """One D2H round trip for the whole batch -> the reference's per-image list."""
    N = post['N']
    if not with_nms:
        # aug_test branch, yolocsp_head.py:377-382 (class score multiplied by conf twice, Q8)
        if post.get('nms_pre', -1) > 0:
            raise NotImplementedError('get_bboxes(with_nms=False) with nms_pre > 0 (aug_test only) is not built')
        out = []
        for n in range(N):
            if post.get('class_agnostic'):
                cls = post['conf'][n][:, None] * post['conf'][n][:, None]    # yolocsp_head.py:360,378
                score, cid = cls.max(dim=-1)
                out.append((torch.cat((post['boxes'][n], score[:, None]), dim=-1), cid))
                continue
            cls = post['cls'][n] * post['conf'][n][:, None]
            cls = cls * post['conf'][n][:, None]
            score, cid = cls.max(dim=-1)
            out.append((torch.cat((post['boxes'][n], score[:, None]), dim=-1), cid))
        return out
    counts = post['count'].cpu()
    if bool((counts < 0).any()):
        _run_split_path(post, counts)
        counts = post['count'].cpu()
    out = []
    for n in range(N):
        k = int(counts[n])
        if k == 0:
            # multiclass_nms empty case (Q7): boxes (0,4), labels int64 (0,)
            out.append((post['dets'].new_zeros((0, 4)), torch.zeros((0,), dtype=torch.int64,
                                                                    device=post['dets'].device)))
        else:
            out.append((post['dets'][n, :k].clone(), post['labels'][n, :k].to(torch.int64)))
    return out


def _run_split_path(post, counts):
    """Images whose candidate count reached mmcv's split_thr take the per-class path."""
    from ._lib import lib, check
    L = lib()
    maxc = post['max_coord'].cpu()
    stream = ops.stream_ptr()
    for n in range(post['N']):
        if int(counts[n]) >= 0:
            continue
        cnt = int(post['counts'][n].item())
        if cnt > post['key_cap']:
            raise RuntimeError('candidate key buffer overflow')
        work = torch.empty(max(L.yv4_nms_split_work(cnt), 16), dtype=torch.uint8, device=post['dets'].device)
        check(L.yv4_nms_split(post['keys'][n].data_ptr(), cnt, float(maxc[n]), post['boxes'][n].data_ptr(), None,
                              post['num_classes'], float(post['iou_thr']), post['max_per_img'], work.data_ptr(),
                              post['dets'][n].data_ptr(), post['labels'][n].data_ptr(),
                              post['index'][n].data_ptr(), post['count'][n:n + 1].data_ptr(), stream),
              'yv4_nms_split')
