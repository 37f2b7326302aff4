"""Import the reference's YOLOv4 modules from /root/reference in the BUILD CONTAINER only.

Used by make_golden.py to generate golden vectors.  Nothing here (and nothing from the
reference) is needed at test time: the GPU box has no /root/reference.

mmcv-full is not installed, so a minimal stand-in for the mmcv *composition* surface is
registered in sys.modules (Registry, ConvModule = conv->bn->act, BaseModule, no-op fp16
decorators ...).  All arithmetic stays torch's, exactly as with the real mmcv, EXCEPT
``mmcv.ops.nms.batched_nms``, whose arithmetic lives in mmcv-full: it is bound to the
restated definition in oracle/ (documented "parity unpinned").  ``mmdet.*`` packages are
registered as empty namespace shells whose ``__path__`` point into /root/reference, so the
real module files are executed but the heavyweight package ``__init__``s (pycocotools,
cv2 ...) are not.
"""
import importlib
import os
import sys
import types

import torch
import torch.nn as nn

REF = '/root/reference'


def available():
    return os.path.isdir(os.path.join(REF, 'mmdet'))


class _Registry:
    def __init__(self, name, parent=None, build_func=None):
        self.name = name
        self.parent = parent
        self._d = {}

    def get(self, k):
        if k in self._d:
            return self._d[k]
        return self.parent.get(k) if self.parent is not None else None

    def register_module(self, name=None, force=False, module=None):
        def reg(cls):
            self._d[name or cls.__name__] = cls
            return cls
        if module is not None:
            return reg(module)
        return reg

    def build(self, cfg, default_args=None):
        return _build_from_cfg(cfg, self, default_args)


def _build_from_cfg(cfg, registry, default_args=None):
    args = dict(cfg)
    if default_args:
        for k, v in default_args.items():
            args.setdefault(k, v)
    t = args.pop('type')
    cls = registry.get(t) if isinstance(t, str) else t
    if cls is None:
        raise KeyError(f'{t} not in {registry.name}')
    return cls(**args)


class _ConfigDict(dict):
    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError:
            raise AttributeError(k)
        return _ConfigDict(v) if isinstance(v, dict) and not isinstance(v, _ConfigDict) else v

    def __setattr__(self, k, v):
        self[k] = v


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    sys.modules[name] = m
    return m


def install_shim(mish_ext):
    """mish_ext: module exposing mish_forward / mish_backward (the reference's C++ kernels)."""
    ACT = _Registry('activation layer')
    for c in (nn.ReLU, nn.LeakyReLU, nn.Sigmoid, nn.Tanh):
        ACT.register_module(module=c)
    MODELS = _Registry('model')

    def build_activation_layer(cfg):
        return _build_from_cfg(cfg, ACT)

    def build_norm_layer(cfg, num_features, postfix=''):
        cfg_ = dict(cfg)
        t = cfg_.pop('type')
        assert t in ('BN', 'BN2d', 'SyncBN')
        requires_grad = cfg_.pop('requires_grad', True)
        cfg_.setdefault('eps', 1e-5)
        layer = (nn.SyncBatchNorm if t == 'SyncBN' else nn.BatchNorm2d)(num_features, **cfg_)
        for p in layer.parameters():
            p.requires_grad = requires_grad
        return 'bn' + str(postfix), layer

    class BaseModule(nn.Module):
        def __init__(self, init_cfg=None):
            super().__init__()
            self.init_cfg = init_cfg

        def init_weights(self):
            pass

    class ConvModule(nn.Module):
        """conv -> norm -> act with mmcv's defaults (bias='auto', inplace act)."""

        def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                     bias='auto', conv_cfg=None, norm_cfg=None, act_cfg=dict(type='ReLU'), inplace=True,
                     with_spectral_norm=False, padding_mode='zeros', order=('conv', 'norm', 'act')):
            super().__init__()
            self.with_norm = norm_cfg is not None
            self.with_activation = act_cfg is not None
            if bias == 'auto':
                bias = not self.with_norm
            self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=padding,
                                  dilation=dilation, groups=groups, bias=bias)
            if self.with_norm:
                self.norm_name, norm = build_norm_layer(norm_cfg, out_channels)
                self.add_module(self.norm_name, norm)
            if self.with_activation:
                act_cfg_ = dict(act_cfg)
                if act_cfg_['type'] not in ['Tanh', 'PReLU', 'Sigmoid', 'HSigmoid', 'Swish']:
                    act_cfg_.setdefault('inplace', inplace)
                self.activate = build_activation_layer(act_cfg_)
            nn.init.kaiming_normal_(self.conv.weight, a=0, mode='fan_out', nonlinearity='relu')

        @property
        def norm(self):
            return getattr(self, self.norm_name)

        def forward(self, x):
            x = self.conv(x)
            if self.with_norm:
                x = self.norm(x)
            if self.with_activation:
                x = self.activate(x)
            return x

    def _noop_deco(*a, **k):
        def deco(f):
            return f
        return deco

    def normal_init(module, mean=0, std=1, bias=0):
        nn.init.normal_(module.weight, mean, std)
        if getattr(module, 'bias', None) is not None:
            nn.init.constant_(module.bias, bias)

    from oracle import yolov4_oracle as O

    def batched_nms(boxes, scores, idxs, nms_cfg, class_agnostic=False):
        return O.batched_nms(boxes, scores, idxs, nms_cfg, class_agnostic)

    mmcv = _mod('mmcv', jit=_noop_deco, ConfigDict=_ConfigDict)
    mmcv.is_tuple_of = lambda seq, t: isinstance(seq, tuple) and all(isinstance(s, t) for s in seq)
    _mod('mmcv.utils', Registry=_Registry, build_from_cfg=_build_from_cfg, TORCH_VERSION=torch.__version__)
    _mod('mmcv.cnn', ConvModule=ConvModule, MODELS=MODELS, normal_init=normal_init)
    _mod('mmcv.cnn.bricks')
    _mod('mmcv.cnn.bricks.registry', ACTIVATION_LAYERS=ACT)
    _mod('mmcv.cnn.bricks.activation', build_activation_layer=build_activation_layer)
    _mod('mmcv.cnn.bricks.norm', build_norm_layer=build_norm_layer)
    _mod('mmcv.runner', BaseModule=BaseModule, auto_fp16=_noop_deco, force_fp32=_noop_deco)
    _mod('mmcv.runner.fp16_utils', auto_fp16=_noop_deco, force_fp32=_noop_deco)
    _mod('mmcv.ops')
    _mod('mmcv.ops.nms', batched_nms=batched_nms)

    m = os.path.join(REF, 'mmdet')
    _pkg('mmdet', m)
    for sub in ('ops', 'ops/mish_cuda', 'models', 'models/backbones', 'models/necks', 'models/dense_heads',
                'models/losses', 'core', 'core/anchor', 'core/bbox', 'core/bbox/coder', 'core/post_processing',
                'core/utils'):
        _pkg('mmdet.' + sub.replace('/', '.'), os.path.join(m, sub))
    sys.modules['mmdet.ops.mish_cuda.mish_cuda_ext'] = mish_ext

    imp = importlib.import_module
    mish = imp('mmdet.ops.mish_cuda.mish')          # registers Mish in ACTIVATION_LAYERS
    core = sys.modules['mmdet.core']
    ab = imp('mmdet.core.anchor.builder')
    bb = imp('mmdet.core.bbox.builder')
    imp('mmdet.core.anchor.anchor_generator')
    imp('mmdet.core.anchor.yolov4_anchor_generator')
    imp('mmdet.core.bbox.coder.yolov4_bbox_coder')
    iou = imp('mmdet.core.bbox.iou_calculators')
    tr = imp('mmdet.core.bbox.transforms')
    core.bbox_overlaps = iou.bbox_overlaps
    core.bbox2result = tr.bbox2result
    core.bbox_mapping_back = tr.bbox_mapping_back
    core.build_anchor_generator = ab.build_anchor_generator
    core.build_assigner = bb.build_assigner
    core.build_bbox_coder = bb.build_bbox_coder
    core.build_sampler = bb.build_sampler

    def multi_apply(func, *args, **kwargs):
        from functools import partial
        pfunc = partial(func, **kwargs) if kwargs else func
        return tuple(map(list, zip(*map(pfunc, *args))))
    core.multi_apply = multi_apply
    nmsmod = imp('mmdet.core.post_processing.bbox_nms')
    core.multiclass_nms = nmsmod.multiclass_nms

    lu = imp('mmdet.models.losses.utils')
    losses = sys.modules['mmdet.models.losses']
    losses.reduce_loss = lu.reduce_loss
    imp('mmdet.models.losses.cross_entropy_loss')
    imp('mmdet.models.losses.iou_loss')
    dk = imp('mmdet.models.backbones.darknetcsp')
    nk = imp('mmdet.models.necks.yolo_neck_csp')
    hd = imp('mmdet.models.dense_heads.yolocsp_head')
    return types.SimpleNamespace(mish=mish, darknetcsp=dk, neck=nk, head=hd, nms=nmsmod, ConfigDict=_ConfigDict,
                                 anchor=sys.modules['mmdet.core.anchor.yolov4_anchor_generator'],
                                 coder=sys.modules['mmdet.core.bbox.coder.yolov4_bbox_coder'])
