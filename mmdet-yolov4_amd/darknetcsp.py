"""``DarknetCSP`` backbone and its blocks, registered under the reference's names.

Mirror of ``mmdet/models/backbones/darknetcsp.py`` (constructor arguments, attribute
names -> state-dict keys, stage tables, forward results); the arithmetic is not
torch's: every block contributes fused launches to a ``Plan`` through ``emit``.

Fusion map (reference line -> launch):
  Conv            conv->bn->Mish (darknetcsp.py:15-35)            one conv launch
  Bottleneck      x + conv2(conv1(x)) (:60-64)                    residual in conv2's epilogue
  BottleneckCSP   conv4(act(bn(cat(conv3(..), conv2(x))))) (:106-109)
                  bare conv3 / conv2 write the two halves of the cat buffer with their
                  half of the CSP-level BN + act as epilogue
  BottleneckCSP2  (:149-153) last bottleneck conv writes half 0 with a second
                  affine+act stage; bare conv2 writes half 1
  SPPV4 / SPPV5   max-pools complete the cat buffer in place (:176-181,220-229)
"""
import os
import warnings

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.modules.batchnorm import _BatchNorm

from .bricks import (HipModule, build_activation_layer, build_norm_layer, _NO_INPLACE)
from . import train_ops as T
from .plan import act_id, bn_affine

_CONV_STATS = os.environ.get('YV4_CONV_STATS', '1') != '0'    # BN statistics in the conv epilogue (A/B switch)
_CAT_SLOTS = os.environ.get('YV4_CAT_SLOTS', '1') != '0'      # CSP halves written straight into the concat buffer (A/B switch)
from .registry import BACKBONES


def _csp_act(csp_act_cfg):
    cfg = dict(csp_act_cfg)
    if cfg['type'] not in _NO_INPLACE:
        cfg.setdefault('inplace', True)
    return build_activation_layer(cfg)


def _act_from_cfg(act_cfg):
    if act_cfg is None:
        return None
    cfg = dict(act_cfg)
    if cfg['type'] not in _NO_INPLACE:
        cfg.setdefault('inplace', True)
    return build_activation_layer(cfg)


class Conv(HipModule):
    """Conv2d(bias=False) -> BN -> act with mmcv ``ConvModule``'s attribute layout
    (``conv``, ``bn``, ``activate``).  darknetcsp.py:15-35."""

    def __init__(self, in_channels, out_channels, kernel_size=1, stride=1, padding=None, groups=1,
                 norm_cfg=dict(type='BN'), act_cfg=dict(type='Mish'), **kwargs):
        super().__init__(None)
        if groups != 1:
            raise NotImplementedError('grouped convolution has no fused kernel (no YOLOv4/v5 config uses it)')
        padding = kernel_size // 2 if padding is None else padding
        self.with_norm = norm_cfg is not None
        self.with_activation = act_cfg is not None
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=padding,
                              groups=groups, bias=not self.with_norm)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = kernel_size, stride, padding
        if self.with_norm:
            self.norm_name, norm = build_norm_layer(norm_cfg, out_channels)
            self.add_module(self.norm_name, norm)
        self.activate = _act_from_cfg(act_cfg)
        # mmcv ConvModule.init_weights: kaiming (relu gain), BN weight 1 / bias 0
        nn.init.kaiming_normal_(self.conv.weight, a=0, mode='fan_out', nonlinearity='relu')
        if self.conv.bias is not None:
            nn.init.constant_(self.conv.bias, 0)

    @property
    def norm(self):
        return getattr(self, self.norm_name) if self.with_norm else None

    def stage1(self):
        """Epilogue stage of this block: (scale, shift, (act, slope))."""
        if self.with_norm:
            s, t = bn_affine(self.norm)
        else:
            s = torch.ones(self.out_channels)
            t = self.conv.bias.detach().float() if self.conv.bias is not None else torch.zeros(self.out_channels)
        return s, t, act_id(self.activate)

    def emit(self, plan, x, out=None, residual=None, post=None, name=None):
        s, t, a = self.stage1()
        s2, t2, a2, bn2 = post if post is not None else (None, None, (0, 0.0), None)
        return plan.conv(x, self.conv.weight, s, t, a, stride=self.stride, pad=self.padding,
                         residual=residual, s2=s2, t2=t2, act2=a2, out=out,
                         name=name or f'conv{self.kernel_size}x{self.kernel_size}',
                         bn1=(self.norm, 0, self.out_channels) if self.with_norm else None, bn2=bn2)

    def fwd(self, x, residual=None, sink=None, res_sink=None, cat=None):
        """``sink`` / ``res_sink``: a ``train_ops.GradSink`` this conv's data gradient consumes / the residual's
        gradient is parked in (both ends of a Bottleneck's shortcut, see ``Bottleneck.fwd``).  ``cat``: a
        ``train_ops.CatSlot`` -- the activation is written into its concat buffer, which is returned."""
        w = self.conv.weight
        dt = T.train_dtype(self, x)
        al = 4 if dt == torch.float32 else 8
        if x.shape[1] % al:          # the 3-channel image: zero-pad x and the weight to one 16-byte chunk
            if (dt != torch.float32 and x.dtype == torch.float32 and x.is_cuda and x.is_contiguous()
                    and not x.requires_grad and x.shape[1] < 16 and self.kernel_size == 3 and self.stride == 1
                    and self.padding == 1):
                # 16-bit stem: 16 channels per pixel, the narrowest input of the few-channel 3x3 kernel
                # (conv3x3_small_h16.hip; the generic tile's per-lane (tap, channel) decode of a 72-wide K takes
                # 1.6 ms for this HBM-bound layer at 64 x 608 x 608), converted from the fp32 NCHW batch in one pass
                # (the weight keeps its 3 channels: ConvFunction packs it to x's width and takes dW over 8)
                x = T.image_to_nhwc16(x, dt, 16)
            else:
                padc = al - x.shape[1] % al
                x = F.pad(x, (0, 0, 0, 0, 0, padc))
                w = F.pad(w, (0, 0, 0, 0, 0, padc))
        if self.with_norm:      # batch statistics in training mode, running statistics under norm_eval / frozen stages
            bn = self.norm
            stats = None
            if (bn.training or not bn.track_running_stats) and _CONV_STATS:
                # the conv's epilogue leaves the BN sums here; the buffer is this module's, kept zero by the
                # finalize kernel that reads it (no memset per step)
                stats = getattr(self, '_yv4_stats', None)
                if stats is None or stats.device != x.device or stats.numel() != T.stats_numel(w.shape[0]):
                    stats = self._yv4_stats = T.conv_stats_buffer(w.shape[0], x.device, persistent=True)
            try:
                y = T.conv2d(x, w, self.stride, self.padding, dtype=dt, stats=stats, sink=sink)
                return T.bn_act(y, bn, act_id(self.activate), residual, sums=stats, res_sink=res_sink, cat=cat)
            except Exception:
                self._yv4_stats = None       # a half-used statistics buffer is not clean: drop it
                raise
        assert cat is None, 'a concat slot needs the fused BN path'
        y = T.conv2d(x, w, self.stride, self.padding, dtype=dt)
        if self.conv.bias is not None:
            y = y + self.conv.bias.view(1, -1, 1, 1)
        if self.activate is not None:
            y = self.activate(y)
        return y if residual is None else y + residual

    def forward(self, x):
        return self._dispatch((x,), 'flat')


def _spp_cat(mod, x):
    """cat([x, maxpools(x)...]) in training mode: the fused HIP op for the (5, 9, 13) pyramid."""
    ks = tuple(int(mp.kernel_size) for mp in mod.maxpools)
    if ks == (5, 9, 13) and x.is_cuda and x.shape[1] % (4 if x.dtype == torch.float32 else 8) == 0:
        return T.spp_cat(x)
    return torch.cat([x] + [mp(x) for mp in mod.maxpools], 1)


def bare_conv_fwd(conv, x, cat=None, park=None, stats=None):
    return T.conv2d(x, conv.weight, conv.stride[0], conv.padding[0], cat=cat, park=park, stats=stats)   # follows x's dtype


def _cat_stats(mod, channels, device):
    """Persistent, kept-clean statistics buffers (``train_ops.conv_stats_buffer``) for the convs that fill the channel
    ranges of ``mod``'s concat buffer, or None when the joint BatchNorm does not take batch statistics from one rank
    (eval-mode BN inside a training graph, SyncBN: those keep the statistics pass over the buffer)."""
    bn = mod.bn
    if not _CONV_STATS or not (bn.training or not bn.track_running_stats) or T._sync_group(bn) is not None:
        return None
    st = getattr(mod, '_yv4_cat_stats', None)
    if st is None or st[0].device != device or [b.numel() for b in st] != [T.stats_numel(c) for c in channels]:
        st = mod._yv4_cat_stats = [T.conv_stats_buffer(c, device, persistent=True) for c in channels]
    return st


def _fanout_sink(first, x):
    """``x`` feeds ``first`` (a ``Conv`` with the fused BN path, stride 1) and a bare conv that writes the SECOND half of
    the concat buffer: the bare conv's backward necessarily runs before ``first``'s (the buffer's gradient passes
    through it on the way to the producer of the first half), so it parks its data gradient in this sink and
    ``first``'s data-gradient launch adds it -- autograd's add kernel over the whole activation goes away."""
    if first.with_norm and first.stride == 1 and T.train_dtype(first, x) == x.dtype:
        return T.grad_sink_for(x)
    return None


def _cat_ok(mod, *xs):
    """The producers of a CSP concat write straight into the concat buffer (``train_ops.CatSlot``) when every half has
    a channel count the kernels can address (a whole number of 16-byte chunks) and the fused BN path is on."""
    al = 4 if T.train_dtype(mod, xs[0]) == torch.float32 else 8
    return _CAT_SLOTS and mod.hidden % al == 0 and all(x.is_cuda for x in xs)


def emit_bare_conv(plan, conv, x, stage, out=None, name='conv1x1_bare'):
    """A bias-free ``nn.Conv2d`` whose only epilogue is its half of a CSP-level BN + act."""
    assert conv.bias is None and conv.groups == 1
    s, t, a, bn = stage
    return plan.conv(x, conv.weight, s, t, a, stride=conv.stride[0], pad=conv.padding[0], out=out, name=name,
                     bn1=bn)


def _fuse_siblings():
    return os.environ.get('YV4_FUSE_SIBLINGS', '1') != '0'


def emit_sibling_pair(plan, x, first, second_conv, second_stage, h, name, bufname):
    """Two 1x1 convs that read the same tensor -- a ``Conv`` (``first``: conv + BN + act) and a bare ``nn.Conv2d`` whose
    epilogue is its half of a CSP-level BN (``second_conv`` / ``second_stage``) -- as ONE launch: the input is read once,
    one launch boundary goes.  A new buffer holds [y1 | y2 | t] (3 h channels): the launch writes channels [h, 3h) =
    [second's output (the concat's half 1) | first's output (a temporary)], so whatever later produces half 0 never
    writes what it reads.  Returns (the concat view [y1 | y2], the view of ``first``'s output), or (None, None) when the
    pair does not fit
    (darknetcsp.py:67-153: conv1 / conv2 of BottleneckCSP, bottlenecks[0].conv1 / conv2 of BottleneckCSP2)."""
    s2, t2, a2, bn2 = second_stage
    s1, t1, a1 = first.stage1()
    if not (_fuse_siblings() and first.kernel_size == 1 and first.stride == 1 and first.padding == 0 and first.with_norm
            and second_conv.kernel_size == (1, 1) and second_conv.stride == (1, 1) and second_conv.padding == (0, 0)
            and second_conv.bias is None and second_conv.groups == 1 and tuple(a1) == tuple(a2)
            and first.out_channels == h and second_conv.out_channels == h and h % 8 == 0):
        return None, None
    buf = plan.new_buf(x.N, x.H, x.W, 3 * h, bufname)
    w = torch.cat([second_conv.weight.detach(), first.conv.weight.detach()], 0)
    s = torch.cat([s2.float().cpu(), s1.float().cpu()])
    t = torch.cat([t2.float().cpu(), t1.float().cpu()])
    plan.conv(x, w, s, t, a1, out=buf.slice(h, 2 * h), name=name, bn1=[bn2, (first.norm, 0, h)])
    return buf.slice(0, 2 * h), buf.slice(2 * h, h)


def csp_halves(bn, act, hidden):
    """Split the CSP-level BN over the two concat halves (inference only)."""
    s, t = bn_affine(bn)
    a = act_id(act)
    return ((s[:hidden].contiguous(), t[:hidden].contiguous(), a, (bn, 0, hidden)),
            (s[hidden:].contiguous(), t[hidden:].contiguous(), a, (bn, hidden, 2 * hidden)))


class Bottleneck(HipModule):
    """darknetcsp.py:38-64."""

    def __init__(self, in_channels, out_channels, shortcut=True, groups=1, expansion=0.5, init_cfg=None, **kwargs):
        super().__init__(init_cfg)
        hidden = int(out_channels * expansion)
        self.conv1 = Conv(in_channels, hidden, kernel_size=1, **kwargs)
        self.conv2 = Conv(hidden, out_channels, kernel_size=3, groups=groups, **kwargs)
        self.shortcut = shortcut and in_channels == out_channels

    def emit(self, plan, x, out=None, post=None):
        y = self.conv1.emit(plan, x)
        return self.conv2.emit(plan, y, out=out, residual=x if self.shortcut else None, post=post)

    def fwd(self, x, cat=None, sink=None):
        """``sink``: a ``GradSink`` another consumer of ``x`` parks its data gradient in (``_fanout_sink``); only
        without a shortcut -- with one, conv1's single joining input is the shortcut's gradient."""
        if not self.shortcut:
            return self.conv2.fwd(self.conv1.fwd(x, sink=sink), cat=cat)
        assert sink is None
        # out = x + f(x): d_out reaches x twice; the second path is added inside conv1's data-gradient launch
        # (train_ops.GradSink) instead of by autograd's add kernel -- when both convs run the fused BN path
        sink = T.grad_sink_for(x) if (self.conv1.with_norm and self.conv2.with_norm and self.conv1.stride == 1
                                      and T.train_dtype(self, x) == x.dtype) else None
        return self.conv2.fwd(self.conv1.fwd(x, sink=sink), residual=x, res_sink=sink, cat=cat)

    def forward(self, x):
        return self._dispatch((x,), 'flat')


def _emit_chain(plan, bottlenecks, x, out=None, post=None):
    """nn.Sequential of Bottlenecks; the LAST one may write into a concat half with a
    second epilogue stage.  With zero bottlenecks the input is copied into the half."""
    n = len(bottlenecks)
    if n == 0:
        if out is None:
            return x
        raise NotImplementedError('CSP2 block with repetition=0 feeding a concat half')
    for i, b in enumerate(bottlenecks):
        last = i == n - 1
        x = b.emit(plan, x, out=out if last else None, post=post if last else None)
    return x


class BottleneckCSP(HipModule):
    """darknetcsp.py:67-109."""

    def __init__(self, in_channels, out_channels, repetition=1, shortcut=True, groups=1, expansion=0.5,
                 csp_act_cfg=dict(type='Mish'), init_cfg=None, **kwargs):
        super().__init__(init_cfg)
        hidden = int(out_channels * expansion)
        self.hidden = hidden
        self.conv1 = Conv(in_channels, hidden, kernel_size=1, **kwargs)
        self.conv2 = nn.Conv2d(in_channels, hidden, 1, 1, bias=False)
        self.conv3 = nn.Conv2d(hidden, hidden, 1, 1, bias=False)
        self.conv4 = Conv(2 * hidden, out_channels, kernel_size=1, **kwargs)
        csp_norm_cfg = dict(kwargs.get('norm_cfg', dict(type='BN')))
        self.bn = build_norm_layer(csp_norm_cfg, 2 * hidden)[-1]
        self.csp_act = _csp_act(csp_act_cfg)
        self.bottlenecks = nn.Sequential(*[
            Bottleneck(hidden, hidden, shortcut, groups, expansion=1.0, **kwargs) for _ in range(repetition)])

    def emit(self, plan, x, out=None):
        h = self.hidden
        half0, half1 = csp_halves(self.bn, self.csp_act, h)
        cat, y = emit_sibling_pair(plan, x, self.conv1, self.conv2, half1, h, 'csp_conv1+2', 'csp_cat')
        if y is None:
            cat = plan.new_buf(x.N, x.H, x.W, 2 * h, 'csp_cat')
            y = self.conv1.emit(plan, x)
            emit_bare_conv(plan, self.conv2, x, half1, out=cat.slice(h, h), name='csp_conv2')
        y = _emit_chain(plan, self.bottlenecks, y)
        emit_bare_conv(plan, self.conv3, y, half0, out=cat.slice(0, h), name='csp_conv3')
        return self.conv4.emit(plan, cat, out=out)

    def fwd(self, x):
        h = self.hidden
        slot = _cat_ok(self, x)
        sink = _fanout_sink(self.conv1, x) if slot else None
        y = self.conv1.fwd(x, sink=sink)
        for b in self.bottlenecks:
            y = b.fwd(y)
        if slot:                     # both bare convs write their half of the concat buffer ...
            st = _cat_stats(self, (h, h), x.device)      # ... and leave the joint BatchNorm's sums of their channels
            z = bare_conv_fwd(self.conv3, y, cat=T.CatSlot(2 * h, 0), stats=st[0] if st else None)
            z = bare_conv_fwd(self.conv2, x, cat=T.CatSlot(2 * h, h, z), park=sink, stats=st[1] if st else None)
            try:
                return self.conv4.fwd(T.bn_act(z, self.bn, act_id(self.csp_act), sums=st))
            except Exception:
                self._yv4_cat_stats = None       # half-used statistics buffers are not clean: drop them
                raise
        z = torch.cat((bare_conv_fwd(self.conv3, y), bare_conv_fwd(self.conv2, x)), dim=1)
        return self.conv4.fwd(T.bn_act(z, self.bn, act_id(self.csp_act)))

    def forward(self, x):
        return self._dispatch((x,), 'flat')


class BottleneckCSP2(HipModule):
    """darknetcsp.py:112-153."""

    def __init__(self, in_channels, out_channels, repetition=1, shortcut=False, groups=1,
                 csp_act_cfg=dict(type='Mish'), init_cfg=None, **kwargs):
        super().__init__(init_cfg)
        hidden = int(out_channels)
        self.hidden = hidden
        self.conv1 = Conv(in_channels, hidden, kernel_size=1, **kwargs)
        self.conv2 = nn.Conv2d(hidden, hidden, 1, 1, bias=False)
        self.conv3 = Conv(2 * hidden, out_channels, kernel_size=1, **kwargs)
        csp_norm_cfg = dict(kwargs.get('norm_cfg', dict(type='BN')))
        self.bn = build_norm_layer(csp_norm_cfg, 2 * hidden)[-1]
        self.csp_act = _csp_act(csp_act_cfg)
        self.bottlenecks = nn.Sequential(*[
            Bottleneck(hidden, hidden, shortcut, groups, expansion=1.0, **kwargs) for _ in range(repetition)])

    def emit(self, plan, x, out=None):
        h = self.hidden
        half0, half1 = csp_halves(self.bn, self.csp_act, h)
        if len(self.bottlenecks) > 0 and not self.bottlenecks[0].shortcut:
            x1 = self.conv1.emit(plan, x)
            b0 = self.bottlenecks[0]
            cat, t0 = emit_sibling_pair(plan, x1, b0.conv1, self.conv2, half1, h, 'csp2_conv2+b0', 'csp2_cat')
            if t0 is not None:
                last = len(self.bottlenecks) == 1
                y = b0.conv2.emit(plan, t0, out=cat.slice(0, h) if last else None, post=half0 if last else None)
                if not last:
                    _emit_chain(plan, self.bottlenecks[1:], y, out=cat.slice(0, h), post=half0)
                return self.conv3.emit(plan, cat, out=out)
            cat = plan.new_buf(x.N, x.H, x.W, 2 * h, 'csp2_cat')
            _emit_chain(plan, self.bottlenecks, x1, out=cat.slice(0, h), post=half0)
            emit_bare_conv(plan, self.conv2, x1, half1, out=cat.slice(h, h), name='csp2_conv2')
            return self.conv3.emit(plan, cat, out=out)
        cat = plan.new_buf(x.N, x.H, x.W, 2 * h, 'csp2_cat')
        if len(self.bottlenecks) == 0:
            # y1 == x1: conv1 itself produces half 0 (two-stage epilogue) and a private copy for conv2
            x1 = self.conv1.emit(plan, x)
            self.conv1.emit(plan, x, out=cat.slice(0, h), post=half0)
        else:
            x1 = self.conv1.emit(plan, x)
            _emit_chain(plan, self.bottlenecks, x1, out=cat.slice(0, h), post=half0)
        emit_bare_conv(plan, self.conv2, x1, half1, out=cat.slice(h, h), name='csp2_conv2')
        return self.conv3.emit(plan, cat, out=out)

    def fwd(self, x):
        x1 = self.conv1.fwd(x)
        y1 = x1
        h = self.hidden
        n = len(self.bottlenecks)
        slot = n > 0 and _cat_ok(self, x1) and all(b.conv2.with_norm for b in self.bottlenecks)
        sink = _fanout_sink(self.bottlenecks[0].conv1, x1) if slot and not self.bottlenecks[0].shortcut else None
        for i, b in enumerate(self.bottlenecks):     # the last bottleneck's activation lands in the concat buffer
            y1 = b.fwd(y1, cat=T.CatSlot(2 * h, 0) if slot and i == n - 1 else None, sink=sink if i == 0 else None)
        if slot:
            z = bare_conv_fwd(self.conv2, x1, cat=T.CatSlot(2 * h, h, y1), park=sink)
        else:
            z = torch.cat((y1, bare_conv_fwd(self.conv2, x1)), dim=1)
        return self.conv3.fwd(T.bn_act(z, self.bn, act_id(self.csp_act)))

    def forward(self, x):
        return self._dispatch((x,), 'flat')


class SPPV5(HipModule):
    """darknetcsp.py:156-181."""

    def __init__(self, in_channels, out_channels, pooling_kernel_size=(5, 9, 13), init_cfg=None, **kwargs):
        super().__init__(init_cfg)
        if tuple(pooling_kernel_size) != (5, 9, 13):
            raise NotImplementedError('the SPP kernel is built for pooling sizes (5, 9, 13)')
        hidden = in_channels // 2
        self.hidden = hidden
        self.conv1 = Conv(in_channels, hidden, kernel_size=1, **kwargs)
        self.conv2 = Conv(hidden * (len(pooling_kernel_size) + 1), out_channels, kernel_size=1, **kwargs)
        self.maxpools = nn.ModuleList([nn.MaxPool2d(kernel_size=k, stride=1, padding=k // 2)
                                       for k in pooling_kernel_size])

    def emit(self, plan, x, out=None):
        h = self.hidden
        cat = plan.new_buf(x.N, x.H, x.W, 4 * h, 'spp_cat')
        self.conv1.emit(plan, x, out=cat.slice(0, h))
        plan.spp(cat, h)
        return self.conv2.emit(plan, cat, out=out)

    def fwd(self, x):
        x = self.conv1.fwd(x)
        return self.conv2.fwd(_spp_cat(self, x))

    def forward(self, x):
        return self._dispatch((x,), 'flat')


class SPPV4(HipModule):
    """darknetcsp.py:184-229."""

    def __init__(self, in_channels, out_channels, expansion=0.5, pooling_kernel_size=(5, 9, 13),
                 csp_act_cfg=dict(type='Mish'), init_cfg=None, **kwargs):
        super().__init__(init_cfg)
        if tuple(pooling_kernel_size) != (5, 9, 13):
            raise NotImplementedError('the SPP kernel is built for pooling sizes (5, 9, 13)')
        hidden = int(2 * out_channels * expansion)
        self.hidden = hidden
        self.conv1 = Conv(in_channels, hidden, kernel_size=1, **kwargs)
        self.conv2 = nn.Conv2d(in_channels, hidden, 1, 1, bias=False)
        self.conv3 = Conv(hidden, hidden, kernel_size=3, **kwargs)
        self.conv4 = Conv(hidden, hidden, kernel_size=1, **kwargs)
        self.maxpools = nn.ModuleList([nn.MaxPool2d(kernel_size=k, stride=1, padding=k // 2)
                                       for k in pooling_kernel_size])
        self.conv5 = Conv(4 * hidden, hidden, kernel_size=1, **kwargs)
        self.conv6 = Conv(hidden, hidden, kernel_size=3, **kwargs)
        csp_norm_cfg = dict(kwargs.get('norm_cfg', dict(type='BN')))
        self.bn = build_norm_layer(csp_norm_cfg, 2 * hidden)[-1]
        self.csp_act = _csp_act(csp_act_cfg)
        self.conv7 = Conv(2 * hidden, out_channels, kernel_size=1, **kwargs)

    def emit(self, plan, x, out=None):
        h = self.hidden
        sppcat = plan.new_buf(x.N, x.H, x.W, 4 * h, 'sppv4_poolcat')
        half0, half1 = csp_halves(self.bn, self.csp_act, h)
        cat, y = emit_sibling_pair(plan, x, self.conv1, self.conv2, half1, h, 'sppv4_conv1+2', 'sppv4_cat')
        if y is None:
            cat = plan.new_buf(x.N, x.H, x.W, 2 * h, 'sppv4_cat')
            y = self.conv1.emit(plan, x)
            emit_bare_conv(plan, self.conv2, x, half1, out=cat.slice(h, h), name='sppv4_conv2')
        y = self.conv3.emit(plan, y)
        self.conv4.emit(plan, y, out=sppcat.slice(0, h))
        plan.spp(sppcat, h)
        y = self.conv5.emit(plan, sppcat)
        self.conv6.emit(plan, y, out=cat.slice(0, h), post=half0)
        return self.conv7.emit(plan, cat, out=out)

    def fwd(self, x):
        slot = _cat_ok(self, x) and self.conv6.with_norm
        sink = _fanout_sink(self.conv1, x) if slot else None
        x1 = self.conv4.fwd(self.conv3.fwd(self.conv1.fwd(x, sink=sink)))
        h = self.hidden
        if slot:
            z = self.conv6.fwd(self.conv5.fwd(_spp_cat(self, x1)), cat=T.CatSlot(2 * h, 0))
            z = bare_conv_fwd(self.conv2, x, cat=T.CatSlot(2 * h, h, z), park=sink)
        else:
            y1 = self.conv6.fwd(self.conv5.fwd(_spp_cat(self, x1)))
            z = torch.cat((y1, bare_conv_fwd(self.conv2, x)), dim=1)
        return self.conv7.fwd(T.bn_act(z, self.bn, act_id(self.csp_act)))

    def forward(self, x):
        return self._dispatch((x,), 'flat')


class Focus(HipModule):
    """darknetcsp.py:232-259: the Focus slice realised as a (2k x 2k, stride 2) conv."""

    def __init__(self, in_channels, out_channels, kernel_size=1, stride=1, groups=1, init_cfg=None, **kwargs):
        super().__init__(init_cfg)
        padding = (kernel_size // 2) * 2
        self.conv = Conv(in_channels, out_channels, kernel_size=kernel_size * 2, stride=stride * 2,
                         padding=padding, groups=groups, **kwargs)

    def emit(self, plan, x, out=None):
        return self.conv.emit(plan, x, out=out)

    def fwd(self, x):
        return self.conv.fwd(x)

    def forward(self, x):
        return self._dispatch((x,), 'flat')


class CSPStage(HipModule):
    """darknetcsp.py:262-277."""

    def __init__(self, in_channels, out_channels, repetition, init_cfg=None, **kwargs):
        super().__init__(init_cfg)
        self.conv_downscale = Conv(in_channels, out_channels, kernel_size=3, stride=2, **kwargs)
        self.conv_csp = BottleneckCSP(out_channels, out_channels, repetition, **kwargs)

    def emit(self, plan, x, out=None):
        return self.conv_csp.emit(plan, self.conv_downscale.emit(plan, x), out=out)

    def fwd(self, x):
        return self.conv_csp.fwd(self.conv_downscale.fwd(x))

    def forward(self, x):
        return self._dispatch((x,), 'flat')


class SPPV5Stage(HipModule):
    """darknetcsp.py:280-297 (``SPPV5`` built WITHOUT the stage's cfg, Q1)."""

    def __init__(self, in_channels, out_channels, repetition, init_cfg=None, **kwargs):
        super().__init__(init_cfg)
        self.conv_downscale = Conv(in_channels, out_channels, kernel_size=3, stride=2, **kwargs)
        self.spp = SPPV5(out_channels, out_channels, pooling_kernel_size=(5, 9, 13))
        self.conv_csp = BottleneckCSP(out_channels, out_channels, repetition, **kwargs)

    def emit(self, plan, x, out=None):
        y = self.conv_downscale.emit(plan, x)
        return self.conv_csp.emit(plan, self.spp.emit(plan, y), out=out)

    def fwd(self, x):
        return self.conv_csp.fwd(self.spp.fwd(self.conv_downscale.fwd(x)))

    def forward(self, x):
        return self._dispatch((x,), 'flat')


class SPPV4Stage(HipModule):
    """darknetcsp.py:300-317 (``SPPV4`` built WITHOUT the stage's cfg: BN eps 1e-5, Q1)."""

    def __init__(self, in_channels, out_channels, repetition, init_cfg=None, **kwargs):
        super().__init__(init_cfg)
        self.conv_downscale = Conv(in_channels, out_channels * 2, kernel_size=3, stride=2, **kwargs)
        self.conv_csp = BottleneckCSP(out_channels * 2, out_channels * 2, repetition, **kwargs)
        self.spp = SPPV4(out_channels * 2, out_channels, pooling_kernel_size=(5, 9, 13))

    def emit(self, plan, x, out=None):
        y = self.conv_csp.emit(plan, self.conv_downscale.emit(plan, x))
        return self.spp.emit(plan, y, out=out)

    def fwd(self, x):
        return self.spp.fwd(self.conv_csp.fwd(self.conv_downscale.fwd(x)))

    def forward(self, x):
        return self._dispatch((x,), 'flat')


class BottleneckStage(HipModule):
    """darknetcsp.py:320-335.  ``repetition`` lands in ``Bottleneck``'s ``shortcut``
    parameter positionally (Q2): one bottleneck, residual on when repetition is truthy."""

    def __init__(self, in_channels, out_channels, repetition, init_cfg=None, **kwargs):
        super().__init__(init_cfg)
        self.conv_downscale = Conv(in_channels, out_channels, kernel_size=3, stride=2, **kwargs)
        self.conv_bottleneck = Bottleneck(out_channels, out_channels, repetition, **kwargs)

    def emit(self, plan, x, out=None):
        return self.conv_bottleneck.emit(plan, self.conv_downscale.emit(plan, x), out=out)

    def fwd(self, x):
        return self.conv_bottleneck.fwd(self.conv_downscale.fwd(x))

    def forward(self, x):
        return self._dispatch((x,), 'flat')


_STAGES = dict(bottleneck=BottleneckStage, csp=CSPStage, sppv4=SPPV4Stage, sppv5=SPPV5Stage)


@BACKBONES.register_module()
class DarknetCSP(HipModule):
    """CSP-Darknet for the v4 / v5 scales (darknetcsp.py:338-480): stage types,
    repetitions and channels per scale, three output maps at ``out_indices``."""

    arch_settings = {
        'v4s5p': [['conv', 'bottleneck', 'csp', 'csp', 'csp', 'sppv4'],
                  [None, 1, 1, 3, 3, 1], [16, 32, 64, 128, 256, 256]],
        'v4m5p': [['conv', 'bottleneck', 'csp', 'csp', 'csp', 'sppv4'],
                  [None, 1, 1, 5, 5, 3], [24, 48, 96, 192, 384, 384]],
        'v4l5p': [['conv', 'bottleneck', 'csp', 'csp', 'csp', 'sppv4'],
                  [None, 1, 2, 8, 8, 4], [32, 64, 128, 256, 512, 512]],
        'v4x5p': [['conv', 'bottleneck', 'csp', 'csp', 'csp', 'sppv4'],
                  [None, 1, 3, 11, 11, 5], [40, 80, 160, 320, 640, 640]],
        'v4l6p': [['conv', 'csp', 'csp', 'csp', 'csp', 'csp', 'sppv4'],
                  [None, 1, 3, 15, 15, 7, 7], [32, 64, 128, 256, 512, 1024, 512]],
        'v4x7p': [['conv', 'csp', 'csp', 'csp', 'csp', 'csp', 'csp', 'sppv4'],
                  [None, 1, 3, 15, 15, 7, 7, 7], [40, 80, 160, 320, 640, 1280, 1280, 640]],
        'v5s5p': [['focus', 'csp', 'csp', 'csp', 'sppv5'], [None, 1, 3, 3, 1], [32, 64, 128, 256, 512]],
        'v5m5p': [['focus', 'csp', 'csp', 'csp', 'sppv5'], [None, 2, 6, 6, 2], [48, 96, 192, 384, 768]],
        'v5l5p': [['focus', 'csp', 'csp', 'csp', 'sppv5'], [None, 3, 9, 9, 3], [64, 128, 256, 512, 1024]],
        'v5x5p': [['focus', 'csp', 'csp', 'csp', 'sppv5'], [None, 4, 12, 12, 4], [80, 160, 320, 640, 1280]],
    }

    def __init__(self, scale='x5p', out_indices=(3, 4, 5), frozen_stages=-1,
                 norm_cfg=dict(type='BN', requires_grad=True, eps=0.001, momentum=0.03),
                 act_cfg=dict(type='Mish'), csp_act_cfg=dict(type='Mish'), norm_eval=False,
                 pretrained=None, init_cfg=None):
        super().__init__(init_cfg)
        if isinstance(scale, str):
            if scale not in self.arch_settings:
                raise KeyError(f'invalid scale {scale} for DarknetCSP')
            stage, repetition, channels = self.arch_settings[scale]
        else:
            stage, repetition, channels = scale
        self.out_indices = out_indices
        self.frozen_stages = frozen_stages
        cfg = dict(norm_cfg=norm_cfg, act_cfg=act_cfg, csp_act_cfg=csp_act_cfg, init_cfg=init_cfg)
        self.layers = []
        cin = 3
        for i, (stg, rep, cout) in enumerate(zip(stage, repetition, channels)):
            layer_name = f'{stg}{i}'
            self.layers.append(layer_name)
            if stg == 'conv':
                self.add_module(layer_name, Conv(cin, cout, 3, **cfg))
            elif stg == 'focus':
                self.add_module(layer_name, Focus(cin, cout, 3, **cfg))
            elif stg in _STAGES:
                self.add_module(layer_name, _STAGES[stg](cin, cout, rep, **cfg))
            else:
                raise NotImplementedError
            cin = cout
        self.norm_eval = norm_eval
        self.fp16_enabled = False
        assert not (init_cfg and pretrained), 'init_cfg and pretrained cannot be setting at the same time'
        if isinstance(pretrained, str):
            warnings.warn('DeprecationWarning: pretrained is a deprecated, please use "init_cfg" instead')
            self.init_cfg = dict(type='Pretrained', checkpoint=pretrained)
        elif pretrained is None:
            if init_cfg is None:
                self.init_cfg = [dict(type='Kaiming', layer='Conv2d'),
                                 dict(type='Constant', val=1, layer=['_BatchNorm', 'GroupNorm'])]
        else:
            raise TypeError('pretrained must be a str or None')

    def emit(self, plan, x):
        outs = []
        for i, layer_name in enumerate(self.layers):
            x = getattr(self, layer_name).emit(plan, x)
            if i in self.out_indices:
                outs.append(x)
            elif i == 0:
                plan.hint_single_consumer(x)         # the stem's map is read by the next stage's first conv only
        return tuple(outs)

    def fwd(self, x):
        outs = []
        for i, layer_name in enumerate(self.layers):
            x = getattr(self, layer_name).fwd(x)
            if i in self.out_indices:
                outs.append(x)
        return tuple(outs)

    def forward(self, x):
        return self._dispatch((x,), 'flat')

    def _freeze_stages(self):
        if self.frozen_stages >= 0:
            for i in range(0, self.frozen_stages):
                m = getattr(self, self.layers[i])
                m.eval()
                for param in m.parameters():
                    param.requires_grad = False

    def train(self, mode=True):
        # The reference returns None here (Q3); returning self keeps `.eval()` chainable
        # and is what every caller that ignores the return value observes anyway.
        super().train(mode)
        self._freeze_stages()
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, _BatchNorm):
                    m.eval()
        return self
