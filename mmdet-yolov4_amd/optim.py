"""Optimizer of the training step: SGD (momentum, Nesterov, weight decay) over the flat arenas.

Reference surface mirrored here:
  * ``optimizer = dict(type='SGD', lr=0.01, momentum=0.937, weight_decay=0.0005, nesterov=True,
    paramwise_cfg=dict(bias_decay_mult=0., norm_decay_mult=0.))``
    (``configs/yolov4/yolov4l_coco_mosaic.py:108-115``), built by mmcv's
    ``DefaultOptimizerConstructor`` into ``torch.optim.SGD`` with **one param group per
    parameter**, in ``named_parameters()`` order -- which ``DetailedLinearWarmUpHook`` relies on
    (``core/custom_hooks/warmup_hooks.py:24-40``).
  * ``param_groups`` stays a list of plain dicts (``lr``, ``momentum``, ``weight_decay``,
    ``nesterov``, ``initial_lr`` ...) that hooks mutate between steps.

What differs: ``step()`` is one kernel (``yv4_sgd_step``) over the whole parameter arena with a
per-parameter hyper-parameter table uploaded from the (mutated) groups; the momentum buffers are
one allocation; ``zero_grad()`` is one memset.
"""
import torch
import torch.nn as nn

from . import _lib
from ._lib import check
from .flat_state import FlatState
from .ops import stream_ptr
from .registry import Registry, build_from_cfg

OPTIMIZERS = Registry('optimizer')

_NORM_TYPES = (nn.modules.batchnorm._BatchNorm, nn.GroupNorm, nn.LayerNorm, nn.modules.instancenorm._InstanceNorm)


def paramwise_groups(model, base_lr, base_wd, paramwise_cfg=None):
    """mmcv 1.3.x ``DefaultOptimizerConstructor.add_params`` for the keys the path's configs use
    (``bias_lr_mult``, ``bias_decay_mult``, ``norm_decay_mult``, ``dwconv_decay_mult``): one group
    per parameter, module pre-order == ``named_parameters()`` order."""
    cfg = dict(paramwise_cfg or {})
    unknown = set(cfg) - {'bias_lr_mult', 'bias_decay_mult', 'norm_decay_mult', 'dwconv_decay_mult'}
    if unknown:
        raise NotImplementedError(f'paramwise_cfg keys {sorted(unknown)} are not supported')
    bias_lr_mult = cfg.get('bias_lr_mult', 1.)
    bias_decay_mult = cfg.get('bias_decay_mult', 1.)
    norm_decay_mult = cfg.get('norm_decay_mult', 1.)
    dwconv_decay_mult = cfg.get('dwconv_decay_mult', 1.)
    groups, seen = [], set()

    def add(module):
        is_norm = isinstance(module, _NORM_TYPES)
        is_dw = isinstance(module, nn.Conv2d) and module.in_channels == module.groups and module.groups > 1
        for name, p in module.named_parameters(recurse=False):
            if id(p) in seen:
                continue
            seen.add(id(p))
            g = {'params': [p]}
            if p.requires_grad:
                if name == 'bias' and not is_norm:
                    g['lr'] = base_lr * bias_lr_mult
                if base_wd is not None:
                    if is_norm:
                        g['weight_decay'] = base_wd * norm_decay_mult
                    elif is_dw:
                        g['weight_decay'] = base_wd * dwconv_decay_mult
                    elif name == 'bias':
                        g['weight_decay'] = base_wd * bias_decay_mult
            groups.append(g)
        for child in module.children():
            add(child)

    add(model)
    return groups


@OPTIMIZERS.register_module(name='SGD')
class FlatSGD:
    """``torch.optim.SGD`` semantics (dampening 0) on a :class:`FlatState`.

    Parameters that took no part in a step: their slice of the gradient arena is zero (``zero_grad`` fills, it does
    not detach), so they still receive weight decay and their momentum buffer still decays -- what
    ``torch.optim.SGD`` does for a zero gradient TENSOR, i.e. the behaviour of the reference's stack (torch 1.x
    ``zero_grad()`` zero-fills; mmcv 1.3 ``OptimizerHook`` calls exactly that).  torch >= 2.0's default
    ``zero_grad(set_to_none=True)`` would skip such parameters instead; every parameter of the YOLOv4 / YOLOv5
    detectors is used in every step, so the two agree on the recipes.

    ``params``: parameters or group dicts, exactly what ``torch.optim.SGD`` accepts; every
    trainable parameter of the model must be in exactly one group."""

    def __init__(self, params, lr, momentum=0., dampening=0., weight_decay=0., nesterov=False, model=None,
                 flat=None):
        if dampening != 0.:
            raise NotImplementedError('FlatSGD: dampening must be 0 (the reference never sets it)')
        if nesterov and momentum <= 0:
            raise ValueError('Nesterov momentum requires a momentum and zero dampening')
        if flat is None:
            if model is None:
                raise ValueError('FlatSGD needs the model (or its FlatState) the parameters belong to')
            flat = FlatState.of(model)
        self.flat = flat
        self.defaults = dict(lr=lr, momentum=momentum, dampening=0., weight_decay=weight_decay, nesterov=nesterov)
        params = list(params)
        if not params:
            raise ValueError('optimizer got an empty parameter list')
        if not isinstance(params[0], dict):
            params = [{'params': params}]
        index = flat.param_index()
        self.param_groups = []
        self._seg_group = [None] * len(flat.param_segments)
        for gi, g in enumerate(params):
            g = dict(g)
            ps = g['params']
            g['params'] = [ps] if isinstance(ps, torch.Tensor) else list(ps)
            for k, v in self.defaults.items():
                g.setdefault(k, v)
            for p in g['params']:
                si = index.get(id(p))
                if si is None:
                    raise ValueError('FlatSGD: a parameter of the optimizer is not part of the model\'s FlatState')
                if self._seg_group[si] is not None:
                    raise ValueError('some parameters appear in more than one parameter group')
                self._seg_group[si] = gi
            self.param_groups.append(g)
        for si, gi in enumerate(self._seg_group):
            if gi is None and flat._params[si].requires_grad:
                raise ValueError(f'FlatSGD: trainable parameter {flat.param_segments[si].name} is in no group')
        dev = flat.device
        self.momentum_buf = torch.zeros_like(flat.grads)
        self._seg_off = torch.tensor(flat.segment_offsets(), dtype=torch.int64, device=dev)
        nseg = len(flat.param_segments)
        self._hyper_dev = torch.zeros((nseg, 4), dtype=torch.float32, device=dev)
        self._hyper_last = None
        self.state = {}     # torch.optim.Optimizer attribute some hooks poke at
        self._stepped = False

    # ---- torch.optim.Optimizer surface -------------------------------------------------------
    def zero_grad(self, set_to_none=False):
        self.flat.zero_grad()

    def hyper_table(self):
        """(nseg, 4) rows {lr, momentum, weight_decay, nesterov}; frozen parameters get lr 0."""
        rows = []
        for si, gi in enumerate(self._seg_group):
            if gi is None or not self.flat._params[si].requires_grad:
                rows.append((0., 0., 0., 0.))
            else:
                g = self.param_groups[gi]
                rows.append((float(g['lr']), float(g['momentum']), float(g['weight_decay']),
                             1. if g['nesterov'] else 0.))
        return rows

    def step(self, ctrl=None):
        """One ``yv4_sgd_step`` over the parameter arena.  ``ctrl``: the 4-float device vector of
        ``yv4_grad_prepare`` (gradient multiplier, skip flag) or ``None``."""
        if not self.flat.grads_attached():
            raise RuntimeError('FlatSGD.step: a parameter\'s .grad is not its slice of the gradient arena '
                               '(zero_grad(set_to_none=True) on the model? use the model\'s own zero_grad)')
        from .train_ops import join_side_streams
        join_side_streams()          # weight gradients launched on a side stream (a no-op when the backward's callback ran)
        rows = self.hyper_table()
        if rows != self._hyper_last:
            # a fresh pinned staging tensor per change: torch's caching host allocator keeps it alive
            # until the (stream-ordered) copy has executed, so the host may run steps ahead
            host = torch.tensor(rows, dtype=torch.float32)
            if self._hyper_dev.is_cuda:
                host = host.pin_memory()
            self._hyper_dev.copy_(host, non_blocking=True)
            self._hyper_last = rows
        f = self.flat
        check(_lib.lib().yv4_sgd_step(f.values.data_ptr(), f.grads.data_ptr(), self.momentum_buf.data_ptr(),
                                      f.n_param, self._seg_off.data_ptr(), self._hyper_dev.data_ptr(),
                                      len(f.param_segments), ctrl.data_ptr() if ctrl is not None else None,
                                      stream_ptr()), 'yv4_sgd_step')
        # the kernel wrote the parameters behind autograd's back: bump the version counters the
        # eval-mode plan caches key on (HipModule._param_version)
        f.bump_versions()
        self._stepped = True

    # ---- torch.optim.Optimizer.state_dict() layout, so checkpoints interchange with the reference's
    # torch.optim.SGD ({'state': {i: {'momentum_buffer': t}}, 'param_groups': [{..., 'params': [i, ...]}]},
    # i = running index over the groups' parameters).  Momentum buffers are slices of one arena here. ---------
    def _param_order(self):
        index = self.flat.param_index()
        return [index[id(p)] for g in self.param_groups for p in g['params']]

    def state_dict(self):
        from .flat_state import _view_as_param
        order = self._param_order()
        state = {}
        for i, si in enumerate(order):
            if self._stepped:
                state[i] = {'momentum_buffer': _view_as_param(self.momentum_buf, self.flat.param_segments[si]).clone()}
        groups, start = [], 0
        for g in self.param_groups:
            d = {k: v for k, v in g.items() if k != 'params'}
            d['params'] = list(range(start, start + len(g['params'])))
            start += len(g['params'])
            groups.append(d)
        return {'state': state, 'param_groups': groups}

    def load_state_dict(self, sd):
        from .flat_state import _view_as_param
        if 'momentum_buf' in sd:                       # round-1 layout of this package
            self.momentum_buf.copy_(sd['momentum_buf'])
            self._stepped = True
        else:
            order = self._param_order()
            if len(sd['param_groups']) != len(self.param_groups) or \
                    sum(len(g['params']) for g in sd['param_groups']) != len(order):
                raise ValueError("loaded state dict has a different number of parameter groups / parameters")
            self.momentum_buf.zero_()
            for i, st in sd.get('state', {}).items():
                buf = st.get('momentum_buffer')
                if buf is None:
                    continue
                seg = self.flat.param_segments[order[int(i)]]
                with torch.no_grad():
                    _view_as_param(self.momentum_buf, seg).copy_(buf.reshape(seg.shape))
                self._stepped = True
        for g, s in zip(self.param_groups, sd['param_groups']):
            g.update({k: v for k, v in s.items() if k != 'params'})
        self._hyper_last = None


def build_optimizer(model, cfg):
    """mmcv ``build_optimizer``: ``cfg`` = the config's ``optimizer`` dict (with optional
    ``paramwise_cfg``); returns the optimizer over ``model``'s FlatState."""
    cfg = dict(cfg)
    paramwise = cfg.pop('paramwise_cfg', None)
    if hasattr(model, 'module'):
        model = model.module
    lr = cfg.get('lr')
    wd = cfg.get('weight_decay')
    if paramwise:
        cfg['params'] = paramwise_groups(model, lr, wd, paramwise)
    else:
        cfg['params'] = list(model.parameters())
    cfg['model'] = model
    return build_from_cfg(cfg, OPTIMIZERS)
