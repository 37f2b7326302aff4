"""Building bricks shared by the registered modules: the ``Mish`` activation plugin,
the mmcv-style layer builders and the base class that turns a module tree into a
launch plan.

Reference surface mirrored here:
  * ``Mish`` + ``ACTIVATION_LAYERS.register_module(module=Mish)`` --
    mmdet/ops/mish_cuda/mish.py:39-48
  * ``build_activation_layer`` / ``build_norm_layer`` / ``ConvModule`` attribute
    layout (``.conv``, ``.bn``, ``.activate``; ``bias='auto'``; ``inplace=True``
    injected unless the type is in mmcv's no-inplace list) -- mmcv 1.3.x, used at
    mmdet/models/backbones/darknetcsp.py:5-7,15-35,88-95
  * ``BaseModule`` (``init_cfg`` / recursive ``init_weights``) -- mmcv.runner
"""
import copy
import math

import torch
import torch.nn as nn

from . import ops
from .plan import Plan, act_id, bn_affine
from .registry import ACTIVATION_LAYERS, NORM_LAYERS, build_from_cfg


# ---- activations ---------------------------------------------------------------------
class Mish(nn.Module):
    """``x * tanh(softplus(x))`` through the HIP op; ``**kwargs`` (``inplace=True``) are
    accepted and ignored exactly like the reference (mish.py:41)."""

    def __init__(self, **kwargs):
        super().__init__()

    def forward(self, inp):
        return ops.MishFunction.apply(inp)


class Swish(nn.Module):
    """mmcv 1.3.x registers ``x * sigmoid(x)`` under 'Swish' (SURVEY 0.1)."""

    def forward(self, x):
        return x * torch.sigmoid(x)


ACTIVATION_LAYERS.register_module(module=Mish)
ACTIVATION_LAYERS.register_module(module=Swish)
ACTIVATION_LAYERS.register_module(module=nn.SiLU)
for _m in (nn.ReLU, nn.LeakyReLU, nn.Sigmoid, nn.Tanh, nn.Identity):
    ACTIVATION_LAYERS.register_module(module=_m)

NORM_LAYERS.register_module('BN', module=nn.BatchNorm2d)
NORM_LAYERS.register_module('BN2d', module=nn.BatchNorm2d)
NORM_LAYERS.register_module('SyncBN', module=nn.SyncBatchNorm)

_NO_INPLACE = ('Tanh', 'PReLU', 'Sigmoid', 'HSigmoid', 'Swish')


def build_activation_layer(cfg):
    return build_from_cfg(cfg, ACTIVATION_LAYERS)


def build_norm_layer(cfg, num_features, postfix=''):
    """mmcv semantics: returns ``(name, layer)``; pops ``requires_grad``; default
    ``eps=1e-5`` (this default is what gives the SPP block its different eps, Q1)."""
    if not isinstance(cfg, dict) or 'type' not in cfg:
        raise KeyError('the norm cfg must be a dict containing the key "type"')
    cfg_ = dict(cfg)
    layer_type = cfg_.pop('type')
    cls = NORM_LAYERS.get(layer_type)
    if cls is None:
        raise KeyError(f'Unrecognized norm type {layer_type}')
    requires_grad = cfg_.pop('requires_grad', True)
    cfg_.setdefault('eps', 1e-5)
    layer = cls(num_features, **cfg_)
    for p in layer.parameters():
        p.requires_grad = requires_grad
    return 'bn' + str(postfix), layer


# ---- init helpers (mmcv.cnn) -----------------------------------------------------------
def kaiming_init(module, a=0, mode='fan_out', nonlinearity='relu', bias=0, distribution='normal'):
    if distribution == 'uniform':
        nn.init.kaiming_uniform_(module.weight, a=a, mode=mode, nonlinearity=nonlinearity)
    else:
        nn.init.kaiming_normal_(module.weight, a=a, mode=mode, nonlinearity=nonlinearity)
    if getattr(module, 'bias', None) is not None:
        nn.init.constant_(module.bias, bias)


def xavier_init(module, gain=1, bias=0, distribution='normal'):
    if distribution == 'uniform':
        nn.init.xavier_uniform_(module.weight, gain=gain)
    else:
        nn.init.xavier_normal_(module.weight, gain=gain)
    if getattr(module, 'bias', None) is not None:
        nn.init.constant_(module.bias, bias)


def constant_init(module, val, bias=0):
    if getattr(module, 'weight', None) is not None:
        nn.init.constant_(module.weight, val)
    if getattr(module, 'bias', None) is not None:
        nn.init.constant_(module.bias, bias)


def normal_init(module, mean=0, std=1, bias=0):
    nn.init.normal_(module.weight, mean, std)
    if getattr(module, 'bias', None) is not None:
        nn.init.constant_(module.bias, bias)


def _layer_matches(m, layer):
    names = [layer] if isinstance(layer, str) else list(layer)
    mro = [c.__name__ for c in type(m).__mro__]
    return any(n in mro for n in names)


def apply_init_cfg(module, init_cfg):
    """The subset of mmcv's ``initialize`` the path's defaults use: Kaiming / Xavier /
    Constant / Normal selected by layer class name."""
    cfgs = init_cfg if isinstance(init_cfg, (list, tuple)) else [init_cfg]
    for cfg in cfgs:
        cfg = dict(cfg)
        t = cfg.pop('type')
        layer = cfg.pop('layer', None)
        if t == 'Pretrained':
            raise NotImplementedError('init_cfg type Pretrained: load the checkpoint with load_state_dict')
        for m in module.modules():
            if layer is not None and not _layer_matches(m, layer):
                continue
            if layer is None and not hasattr(m, 'weight'):
                continue
            if t == 'Kaiming':
                kaiming_init(m, **cfg)
            elif t == 'Xavier':
                xavier_init(m, **cfg)
            elif t == 'Constant':
                constant_init(m, **cfg)
            elif t == 'Normal':
                normal_init(m, **cfg)
            else:
                raise NotImplementedError(f'init_cfg type {t}')


# ---- plan-backed module base -------------------------------------------------------------
PLAN_CACHE_ENTRIES = 4


def plan_cache_get(cache, key):
    """LRU lookup: a hit becomes the most recently used entry."""
    plan = cache.get(key)
    if plan is not None:
        cache[key] = cache.pop(key)
    return plan


def plan_cache_put(cache, key, plan, version_index, cap=None):
    """Insert a freshly built plan.  Entries built for another parameter version can never be hit again (the version
    only grows) and are dropped; beyond ``cap`` geometries the least recently used one goes.  Alternating input shapes
    (keep-ratio resize, the short last batch of a dataset) therefore reuse their plans instead of rebuilding --
    re-packing ~115 conv weights, reallocating activations and re-capturing a hipGraph -- on every call."""
    cap = PLAN_CACHE_ENTRIES if cap is None else cap
    for k in [k for k in cache if k[version_index] != key[version_index]]:
        del cache[k]
    cache[key] = plan
    while len(cache) > cap:
        del cache[next(iter(cache))]


class HipModule(nn.Module):
    """Base of every registered module.  In eval mode ``forward`` compiles (once per input
    geometry and parameter version) the module's own launch plan via ``emit`` and replays it;
    in training mode it runs ``fwd``: an autograd graph over the HIP training ops
    (``train_ops.py``).  NCHW in, NCHW out, like the reference's modules."""

    def __init__(self, init_cfg=None):
        super().__init__()
        self.init_cfg = copy.deepcopy(init_cfg)
        self._is_init = False
        self._plan_cache = {}

    # mmcv.runner.BaseModule.init_weights: own init_cfg first, then children
    def init_weights(self):
        if not self._is_init:
            if self.init_cfg:
                apply_init_cfg(self, self.init_cfg)
            for m in self.children():
                if hasattr(m, 'init_weights'):
                    m.init_weights()
            self._is_init = True

    def zero_grad(self, set_to_none=True):
        """With a FlatState attached (training through the flat arenas) gradients are zeroed by one
        memset and stay views of the gradient arena; otherwise ``nn.Module.zero_grad``."""
        fs = getattr(self, '_flat_state', None)
        if fs is not None:
            fs.zero_grad()
        else:
            super().zero_grad(set_to_none=set_to_none)

    def emit(self, plan, *xs):  # pragma: no cover - abstract
        raise NotImplementedError

    def fwd(self, *xs):  # pragma: no cover - abstract
        """Training-mode forward on channels_last tensors through the autograd HIP ops."""
        raise NotImplementedError

    def _dispatch(self, args, structure):
        """forward() of every registered module: training mode -> autograd graph over the HIP
        training ops (batch-statistics BN); eval mode -> fused launch plan."""
        if self.training:
            from . import train_ops as _T
            _T._fwd_depth[0] += 1
            try:
                return self.fwd(*args)
            finally:
                _T._fwd_depth[0] -= 1
                if _T._fwd_depth[0] == 0:
                    _T.flush_batch_counters()
        return self._run_plan(args, structure)

    def _param_version(self):
        v = 0
        for t in list(self.parameters()) + list(self.buffers()):
            v += t._version
        return v

    def invalidate_plans(self):
        for m in self.modules():
            if isinstance(m, HipModule):
                m._plan_cache.clear()

    def _flatten_inputs(self, args):
        flat = []
        for a in args:
            if isinstance(a, (tuple, list)):
                flat.extend(a)
            else:
                flat.append(a)
        return flat

    def _run_plan(self, args, structure):
        flat = self._flatten_inputs(args)
        for t in flat:
            ops._need_cuda(t, 'input')
            if t.dtype != torch.float32:
                raise RuntimeError(f'{type(self).__name__}: the HIP path computes in fp32; got {t.dtype}')
        if torch.is_grad_enabled() and any(t.requires_grad for t in flat):
            raise NotImplementedError(
                f'{type(self).__name__}: the eval-mode (fused launch plan) path has no autograd; call .train() '
                'to differentiate through the HIP training ops')
        dtype = getattr(self, 'compute_dtype', torch.float32)
        key = (tuple(tuple(t.shape) for t in flat), str(flat[0].device), self._param_version(), dtype)
        plan = plan_cache_get(self._plan_cache, key)
        if plan is None:
            plan = Plan(flat[0].device, dtype)
            views = [plan.add_input_nchw(*t.shape, name=f'in{i}') for i, t in enumerate(flat)]
            outs = self.emit(plan, *(views if structure == 'flat' else [views]))
            outs_l = list(outs) if isinstance(outs, (tuple, list)) else [outs]
            for i, v in enumerate(outs_l):
                plan.add_output_nchw(v, name=f'out{i}')
            plan.single = not isinstance(outs, (tuple, list))
            plan.finalize()
            plan_cache_put(self._plan_cache, key, plan, 2)
        res = plan.run(*flat)
        return res[0] if plan.single else tuple(res)


def wrap_fp16_model(model, dtype=torch.float16):
    """mmcv ``wrap_fp16_model`` for the eval-mode fused path (what ``tools/test.py:175-177`` calls
    when the config has an ``fp16`` field): every plan-backed module computes its convs on 16-bit
    operands (fp32 accumulate, fp32 BN/activation epilogue) from then on; parameters stay fp32 master
    copies and are packed to ``dtype`` when a plan is built.  ``dtype`` may be ``torch.bfloat16``.
    Module boundaries keep the reference's fp32 NCHW tensors."""
    if dtype not in (torch.float16, torch.bfloat16, torch.float32):
        raise ValueError('wrap_fp16_model: dtype must be float16, bfloat16 or float32')
    for m in model.modules():
        if isinstance(m, HipModule):
            m.compute_dtype = dtype
            m._plan_cache.clear()
            if hasattr(m, '_engines'):
                m._engines.clear()
        if hasattr(m, 'fp16_enabled'):
            m.fp16_enabled = dtype != torch.float32
    return model


def conv_bn_stage(conv_module):
    """(scale, shift, act) of a ``Conv`` (conv->bn->act) as the kernel's first epilogue stage."""
    s, t = bn_affine(conv_module.bn)
    return s, t, act_id(conv_module.activate)
