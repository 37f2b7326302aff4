"""Autograd front ends of the training-side HIP kernels (``csrc/train.hip`` + the forward conv).

Training does not go through launch plans: modules build an ordinary autograd graph out of
the two Functions below, on ``torch.channels_last`` tensors (logical NCHW, physical NHWC --
exactly the layout the kernels use, so nothing is converted).  What they replace in the
reference: the autograd of mmcv ``ConvModule`` in train mode (cuDNN conv fwd / backward-data /
backward-filter, ATen batch_norm with batch statistics, ``MishCudaFunction``,
mmdet/models/backbones/darknetcsp.py:15-64, mmdet/ops/mish_cuda/mish.py:18-36).

  ConvFunction    y = conv(x, w)                       fwd: fused conv kernel with an identity
                                                       epilogue; bwd: dX = the same kernel on dY
                                                       (zero-dilated for stride 2) with flipped,
                                                       transposed weights, dW = yv4_conv_wgrad
  BNActFunction   y = act(BN_batchstats(x)) (+ res)    saves only x and the batch statistics; the
                                                       activation is recomputed in backward (the
                                                       reference keeps conv-out, BN-out and the Mish
                                                       input: SURVEY Q17)
"""
import ctypes as C
import os
import weakref

import torch

from . import _lib
from ._lib import ConvDesc, check
from .ops import _need_cuda, stream_ptr
from .plan import pack_conv_weight


def to_nhwc(x):
    """Logical NCHW tensor whose storage is dense NHWC."""
    return x.contiguous(memory_format=torch.channels_last)


def image_to_nhwc16(x, dtype, channels):
    """fp32 NCHW image batch -> dense 16-bit NHWC with ``channels`` per pixel (the real ones, then zeros) in ONE pass
    (``yv4_nchw_to_nhwc_h16``); returned as the logical (N, channels, H, W) channels_last tensor the conv ops take.
    ATen needs a zero fill, a padded copy, a cast and a layout copy for the same result (0.66 ms at 64 x 3 x 608 x 608)."""
    N, C_, H, W = x.shape
    out = torch.empty((N, channels, H, W), device=x.device, dtype=dtype, memory_format=torch.channels_last)
    check(_lib.lib().yv4_nchw_to_nhwc_h16(x.data_ptr(), out.data_ptr(), N, C_, H, W, channels, 0, channels - C_,
                                          _DCODE[dtype], stream_ptr()), 'yv4_nchw_to_nhwc_h16')
    return out


import os as _os0

_SLICE_GRADS = _os0.environ.get('YV4_SLICE_GRADS', '1') != '0'     # A/B switch


def nhwc_or_slice(t, dtype):
    """(tensor, pixel stride in elements): ``t`` itself when it already is ``dtype`` and either dense NHWC or a
    channel slice of a dense NHWC tensor (what ``torch.cat``'s backward hands to each branch: every kernel takes
    a pixel stride, so the slice needs no copy); else a dense NHWC copy."""
    if _SLICE_GRADS and t.dtype == dtype and t.dim() == 4:
        N, C_, H, W = t.shape
        sn, sc, sh, sw = t.stride()
        al = 4 if dtype == torch.float32 else 8
        if (sc == 1 or C_ == 1) and sw >= C_ and sh == W * sw and sn == H * W * sw and sw % al == 0 \
                and t.data_ptr() % 16 == 0:
            return t, sw
    t = to_nhwc(t.to(dtype))
    return t, t.shape[1]


def _is_nhwc(t):
    return t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last)


_IDENTITY = {}


def _identity_affine(device, C_):
    """(ones, zeros) of length C_, cached per device: the identity epilogue of a training conv."""
    key = (str(device), C_)
    if key not in _IDENTITY:
        _IDENTITY[key] = (torch.ones(C_, device=device), torch.zeros(C_, device=device))
    return _IDENTITY[key]


_DCODE = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}


def _conv_launch(x, w_packed, Cin_p, Cout, KH, KW, stride, pad, out, stats=None, x_cs=None, residual=None, res_cs=None,
                 y_cs=None, y_co=0):
    """Identity-epilogue conv of a channels_last tensor; fp32, or fp16 / bf16 operands (fp32 accumulate).
    ``stats``: a float64 buffer of ``STATS_REPLICAS * 2 * Cout`` entries that receives the BatchNorm sums of the
    output (``yv4_conv_fwd_stats``: accumulated in the conv kernel's epilogue)."""
    N, _, H, W = x.shape
    Ho, Wo = out.shape[2], out.shape[3]
    d = ConvDesc()
    d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = N, H, W, Cin_p, Ho, Wo, Cout
    d.KH, d.KW, d.stride, d.pad = KH, KW, stride, pad
    d.x_cstride, d.y_cstride = (x_cs if x_cs is not None else Cin_p), (y_cs if y_cs is not None else Cout)
    d.y_coff = y_co
    if residual is not None:        # out = conv + residual (a gradient that joins this one: see GradSink)
        assert stats is None and residual.dtype == x.dtype
        d.r_cstride, d.r_coff = (res_cs if res_cs is not None else Cout), 0
    rptr = residual.data_ptr() if residual is not None else None
    ones, zeros = _identity_affine(x.device, Cout)
    if stats is not None:
        assert stats.dtype == torch.float64 and stats.numel() >= _lib.STATS_REPLICAS * 2 * Cout
        clean = 1 if getattr(stats, '_yv4_kept_clean', False) else 0
        check(_lib.lib().yv4_conv_fwd_stats(C.byref(d), _DCODE[x.dtype], x.data_ptr(), w_packed.data_ptr(),
                                            ones.data_ptr(), zeros.data_ptr(), out.data_ptr(), stats.data_ptr(),
                                            clean, stream_ptr()), 'yv4_conv_fwd_stats')
        return d
    if x.dtype == torch.float32:
        check(_lib.lib().yv4_conv_bn_act_fwd(C.byref(d), x.data_ptr(), w_packed.data_ptr(), ones.data_ptr(),
                                             zeros.data_ptr(), None, None, rptr, out.data_ptr(), stream_ptr()),
              'yv4_conv_bn_act_fwd')
    else:
        code = _DCODE[x.dtype]
        check(_lib.lib().yv4_conv_bn_act_fwd_h16(C.byref(d), code, code, x.data_ptr(), w_packed.data_ptr(),
                                                 ones.data_ptr(), zeros.data_ptr(), None, None, rptr, out.data_ptr(),
                                                 stream_ptr()), 'yv4_conv_bn_act_fwd_h16')
    return d


class GradSink:
    """Joins two gradient paths of one tensor without autograd's add kernel.  For  out = x + f(x)  (a Bottleneck
    with shortcut, darknetcsp.py:60-64) the gradient of x is  d_out + f'(d_out): the BN + act + residual Function at
    the end of f parks d_out here instead of returning it for the residual, and the FIRST convolution of f -- whose
    backward runs last -- adds it in the epilogue of its data-gradient launch (``residual=`` of the conv kernels).
    Saves one read-read-write pass over the activation per bottleneck and step.  YV4_GRAD_SINK=0 switches it off."""
    __slots__ = ('value', 'cs')

    def __init__(self):
        self.value = None
        self.cs = None


_GRAD_SINK_ON = os.environ.get('YV4_GRAD_SINK', '1') != '0'


def grad_sink_for(x):
    """A sink for the residual path of ``x``, or None when the fusion does not apply (no gradient wanted)."""
    if _GRAD_SINK_ON and torch.is_grad_enabled() and x.requires_grad:
        return GradSink()
    return None


# ---- packed weight operands, replayed in one launch per optimizer step -----------------------------------------------
# A training step needs every conv weight twice as a packed 16-bit (or fp32) operand: (Cout, K) for the forward, the
# transposed / mirrored form for the data gradient (plus one per parity class of a stride-2 layer) -- 750 launches of
# ``yv4_pack_weight`` per YOLOv4-L step.  The operands only change when the weights do, so the requests of the first
# step are recorded in a table (``yv4_pack_desc``) and from then on ONE ``yv4_pack_weights_multi`` launch refreshes all
# of them the first time any operand is asked for after the weights' version counter moved.  The version is torch's
# (views of the flat arena share the arena's counter; ``FlatSGD.step`` bumps it for its raw-pointer kernel).
# YV4_PACK_CACHE=0 packs per call as before.
_PACK_CACHE_ON = os.environ.get('YV4_PACK_CACHE', '1') != '0'


class _PackCache:
    ROWS_TARGET = 16384         # output elements per workgroup

    def __init__(self, device):
        self.device = device
        self.entries = {}       # key -> dict(weight=weakref, desc fields, dst, version)
        self.table = None       # device copy of the descriptor table (rebuilt when entries were added)
        self.dirty_table = True
        self.total_blocks = 0

    @staticmethod
    def key(weight, dtype, mode):
        return (weight.data_ptr(), tuple(weight.shape), tuple(weight.stride()), dtype, mode)

    def lookup(self, weight, dtype, mode):
        e = self.entries.get(self.key(weight, dtype, mode))
        if e is None:
            return None
        if e['wref']() is not weight:       # the owner died and the allocator handed its address to another tensor of the
            del self.entries[self.key(weight, dtype, mode)]     # same shape: a miss (the table is rebuilt on the next add)
            self.dirty_table = True
            return None
        if e['version'] != weight._version or e['gen'] != _PACK_GENERATION[0]:
            self.refresh(weight)
        return e

    def add(self, weight, dtype, mode, fields, dst, cp):
        """Only PERSISTENT weights come here (``packed_weight``: leaves of the autograd graph, i.e. parameters); the
        entry holds a weak reference, so a model that is dropped takes its entries with it at the next refresh."""
        e = dict(fields=fields, dst=dst, cp=cp, version=weight._version, gen=_PACK_GENERATION[0], wref=weakref.ref(weight))
        self.entries[self.key(weight, dtype, mode)] = e
        self.dirty_table = True
        return e

    def _build(self):
        n = len(self.entries)
        tab = (_lib.PackDesc * n)()
        blk = 0
        for i, e in enumerate(self.entries.values()):
            d = tab[i]
            for k, v in e['fields'].items():
                setattr(d, k, v)
            d.dst = e['dst'].data_ptr()
            rows = (d.Cin if d.transpose else d.Cout) * d.KHo * d.KWo
            icp = ((d.Cout if d.transpose else d.Cin) + d.pad_to - 1) // d.pad_to * d.pad_to
            # whole rows r (all their taps) per workgroup; the data-gradient operand is read ACROSS r (the source is
            # contiguous along it), so its workgroups take groups of rows (pack_rows in csrc/train.hip)
            taps = d.KHo * d.KWo
            group = taps * (max(8, 64 // taps) if d.transpose else 1)
            d.rows_per_block = group * max(1, self.ROWS_TARGET // (icp * group))
            d.nblocks = (rows + d.rows_per_block - 1) // d.rows_per_block
            d.first_block = blk
            blk += d.nblocks
        self.total_blocks = blk
        raw = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8)
        self.table = raw.to(self.device)
        self.dirty_table = False

    def refresh(self, _weight):
        """Re-pack every recorded operand in one launch (the weights move together at an optimizer step)."""
        live = {}
        for k, e in self.entries.items():
            w = e['wref']()
            if w is not None and w.data_ptr() == k[0]:
                live[k] = e
        if len(live) != len(self.entries):
            self.entries = live
            self.dirty_table = True
        if not self.entries:
            return
        if self.dirty_table:
            self._build()
        check(_lib.lib().yv4_pack_weights_multi(self.table.data_ptr(), len(self.entries), self.total_blocks, stream_ptr()),
              'yv4_pack_weights_multi')
        for e in self.entries.values():
            e['version'] = e['wref']()._version
            e['gen'] = _PACK_GENERATION[0]


_PACK_CACHES = {}
# Staleness is detected through torch's version counters; an update that bypasses them (``p.data.copy_()``, a
# raw-pointer kernel) must call ``invalidate_packed_weights()`` -- FlatSGD.step, load_state_dict of a FlatState model and
# the EMA swap do (they also bump the versions) -- which makes the next request re-pack everything.
_PACK_GENERATION = [0]


def invalidate_packed_weights():
    _PACK_GENERATION[0] += 1


def clear_pack_cache():
    _PACK_CACHES.clear()


def packed_weight(weight, dtype, transpose_flip=False, taps=None, owner=None, pad_to=None):
    """The conv kernels' weight operand from an fp32 (Cout, Cin, KH, KW) parameter in one launch (``yv4_pack_weight``):
    rows x (KH'*KW'*Cp), K ordered (kh, kw, channel), channels zero-padded to a 16-byte chunk, cast to ``dtype``.
    ``transpose_flip``: the data gradient's operand (rows = Cin, channels = Cout, taps mirrored).  ``taps``:
    ((kh0, kh_step, KH'), (kw0, kw_step, KW')) selects source taps explicitly (rows = Cin, channels = Cout): the
    operand of one parity class of a stride-2 data gradient.  Returns (w, Cp).  The result is a cached buffer that the
    next refresh overwrites: use it on the current stream before the weights change again (the conv launches do).
    ``owner``: the parameter ``weight`` is a detached alias of (``conv2d`` hands ``ConvFunction`` the detached weight
    when dW goes straight into ``weight.grad``): the table records and weakly references the OWNER, so a fresh alias
    per step still hits its entry.  ``pad_to``: pad the channels to a multiple of this instead of one 16-byte chunk (the
    stem: 3 input channels against an activation stored with 16)."""
    Cout, Cin, KH, KW = weight.shape
    al = 4 if dtype == torch.float32 else 8
    if pad_to is not None:
        assert pad_to % al == 0
        al = pad_to
    transpose = bool(transpose_flip or taps is not None)
    rows, ic = (Cin, Cout) if transpose else (Cout, Cin)
    cp = (ic + al - 1) // al * al
    if taps is not None:
        (kh0, khs, KHo), (kw0, kws, KWo) = taps
    elif transpose_flip:
        kh0, khs, KHo, kw0, kws, KWo = KH - 1, -1, KH, KW - 1, -1, KW
    else:
        kh0, khs, KHo, kw0, kws, KWo = 0, 1, KH, 0, 1, KW
    w = weight.detach()
    # temporaries (the stem weight zero-padded to a 16-byte chunk in every step, darknetcsp.Conv.fwd: a fresh non-leaf
    # tensor each time) take the per-call launch below: cached, each step would add an entry that is never hit again
    ident = owner if owner is not None else weight
    persistent = ident.is_leaf and (ident.requires_grad or isinstance(ident, torch.nn.Parameter)) and \
        ident.data_ptr() == weight.data_ptr()
    cacheable = _PACK_CACHE_ON and persistent and w.dtype == torch.float32 and w.is_cuda
    cache = None
    mode = (bool(transpose_flip), taps, al)
    if cacheable:
        cache = _PACK_CACHES.get(w.device)
        if cache is None:
            cache = _PACK_CACHES[w.device] = _PackCache(w.device)
        e = cache.lookup(ident, dtype, mode)
        if e is not None:
            return e['dst'], e['cp']
    if w.dtype != torch.float32:
        w = w.float()
    out = torch.empty((rows, KHo * KWo * cp), device=w.device, dtype=dtype)
    st = w.stride()
    check(_lib.lib().yv4_pack_weight(w.data_ptr(), st[0], st[1], st[2], st[3], Cout, Cin, KH, KW, KHo, KWo, kh0, khs, kw0,
                                     kws, int(transpose), al, out.data_ptr(), _DCODE[dtype], stream_ptr()),
          'yv4_pack_weight')
    if cacheable:
        cache.add(ident, dtype, mode, dict(w=w.data_ptr(), s_co=st[0], s_ci=st[1], s_kh=st[2], s_kw=st[3], Cout=Cout, Cin=Cin,
                                       KHo=KHo, KWo=KWo, kh0=kh0, kh_step=khs, kw0=kw0, kw_step=kws,
                                       transpose=int(transpose), pad_to=al, dtype=_DCODE[dtype]), out, cp)
    return out, cp


def _dgrad_dilated(dy, weight, xshape, stride, pad, dtype, dy_cs=None, residual=None, res_cs=None, owner=None):
    """dX = correlate(dY zero-dilated by `stride`, W flipped in (kh,kw) and transposed in (co,ci)), pad k-1-p."""
    N, Cin, H, W = xshape
    Cout, _, KH, KW = weight.shape
    Ho, Wo = dy.shape[2], dy.shape[3]
    h16 = dtype != torch.float32
    L = _lib.lib()
    wtp, _ = packed_weight(weight, dtype, transpose_flip=True, owner=owner)     # rows = Cin, taps mirrored, cast: one launch
    src_cs = None
    if stride == 1:
        src, src_cs = dy, dy_cs
    elif stride == 2:
        if dy_cs is not None and dy_cs != Cout:
            dy = to_nhwc(dy)
        src = torch.empty((N, Cout, 2 * Ho, 2 * Wo), device=dy.device, dtype=dtype, memory_format=torch.channels_last)
        k = 2 if h16 else 1     # a 16-bit map with C % 8 == 0 is an fp32 map with C/2 channels
        check(L.yv4_dilate2_fwd(dy.data_ptr(), src.data_ptr(), N, Ho, Wo, Cout // k, Cout // k, 0, stream_ptr()),
              'yv4_dilate2_fwd')
    else:
        raise NotImplementedError('conv backward: stride must be 1 or 2')
    p2 = KH - 1 - pad
    Hs, Ws = src.shape[2], src.shape[3]
    Hx = Hs + 2 * p2 - KH + 1
    Wx = Ws + 2 * p2 - KW + 1
    dxf = torch.empty((N, Cin, Hx, Wx), device=dy.device, dtype=dtype, memory_format=torch.channels_last)
    if residual is not None:
        assert (Hx, Wx) == (H, W)
    _conv_launch(src, wtp, Cout, Cin, KH, KW, 1, p2, dxf, x_cs=src_cs, residual=residual, res_cs=res_cs)
    if (Hx, Wx) != (H, W):
        # stride 2 with odd input size: the dilated grid is one row/column larger or smaller
        dx = torch.empty((N, Cin, H, W), device=dy.device, dtype=dtype, memory_format=torch.channels_last).zero_()
        hh, ww = min(H, Hx), min(W, Wx)
        dx[:, :, :hh, :ww] = dxf[:, :, :hh, :ww]
        return dx
    return dxf


def _dgrad_s2_parity(dy, weight, xshape, dtype, dy_cs=None, owner=None):
    """Data gradient of a 3x3 / stride 2 / pad 1 convolution as four parity classes: with
    hi = 2*ho - 1 + kh, the input rows hi = 2i + a receive only the taps kh with (a + 1 - kh) even
    (a = 0: kh = 1 from dY row i;  a = 1: kh = 2 from row i and kh = 0 from row i + 1), likewise in x.
    Each class is a stride-1 correlation of dY with 1, 2, 2 or 4 taps whose result is scattered to
    dX[:, :, a::2, b::2] (``yv4_conv_scatter_fwd``): the forward FLOPs exactly, no dilated copy."""
    N, Cin, H, W = xshape
    Cout = weight.shape[0]
    Ho, Wo = dy.shape[2], dy.shape[3]
    h16 = dtype != torch.float32
    L = _lib.lib()
    dx = torch.empty((N, Cin, H, W), device=dy.device, dtype=dtype, memory_format=torch.channels_last)
    ones, zeros = _identity_affine(dy.device, Cin)
    taps = {0: (1, 1, 1), 1: (2, -2, 2)}          # (first source tap, step, count): [1] and [2, 0]
    wd = weight.detach()
    for a in (0, 1):
        Ha = (H - a + 1) // 2
        for b in (0, 1):
            Wb = (W - b + 1) // 2
            if Ha == 0 or Wb == 0:
                continue
            wp, _ = packed_weight(wd, dtype, taps=(taps[a], taps[b]), owner=owner if owner is not None else weight)
            d = ConvDesc()
            d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = N, Ho, Wo, Cout, Ha, Wb, Cin
            d.KH, d.KW, d.stride, d.pad = taps[a][2], taps[b][2], 1, 0
            d.x_cstride, d.y_cstride = (dy_cs if dy_cs is not None else Cout), Cin
            if h16:
                check(L.yv4_conv_scatter_fwd_h16(C.byref(d), _DCODE[dtype], dy.data_ptr(), wp.data_ptr(), ones.data_ptr(),
                                                 zeros.data_ptr(), dx.data_ptr(), H, W, 2, 2, a, b, stream_ptr()),
                      'yv4_conv_scatter_fwd_h16')
            else:
                check(L.yv4_conv_scatter_fwd(C.byref(d), dy.data_ptr(), wp.data_ptr(), ones.data_ptr(), zeros.data_ptr(),
                                             dx.data_ptr(), H, W, 2, 2, a, b, stream_ptr()), 'yv4_conv_scatter_fwd')
    return dx


_ROWPAIR_W = {}     # id(owner) -> (weakref(owner), version, generation, dtype, (W2 for a = 0, W2 for a = 1))


def _rowpair_weights(weight, dtype, owner):
    """Operands of ``_dgrad_s2_rowpair``: for row parity a, rows = (b, c) -- column parity and input channel --, K ordered
    (di, dj, co) over the dY taps (i + di, j + dj); zero where the (parity, tap) pair has no source tap.  Cached per
    parameter until its version (or the packed-weight generation) moves."""
    ident = owner if owner is not None else weight
    key = id(ident)
    hit = _ROWPAIR_W.get(key)
    if hit is not None and hit[0]() is ident and hit[1] == ident._version and hit[2] == _PACK_GENERATION[0] \
            and hit[3] == dtype:
        return hit[4]
    w = weight.detach().float()
    Co, Cx = w.shape[0], w.shape[1]
    out = []
    for khs in ((1,), (2, 0)):                       # a = 0: dY row i through kh = 1;  a = 1: row i (kh = 2), row i + 1 (kh = 0)
        t = w.new_zeros((2, Cx, len(khs), 2, Co))    # (b, c, di, dj, co)
        for di, kh in enumerate(khs):
            t[0, :, di, 0] = w[:, :, kh, 1].t()      # b = 0: column j through kw = 1
            t[1, :, di, 0] = w[:, :, kh, 2].t()      # b = 1: column j (kw = 2) ...
            t[1, :, di, 1] = w[:, :, kh, 0].t()      # ... and column j + 1 (kw = 0)
        out.append(t.reshape(2 * Cx, len(khs) * 2 * Co).to(dtype).contiguous())
    if len(_ROWPAIR_W) > 64:
        _ROWPAIR_W.clear()
    _ROWPAIR_W[key] = (weakref.ref(ident), ident._version, _PACK_GENERATION[0], dtype, tuple(out))
    return tuple(out)


def _dgrad_s2_rowpair(dy, weight, xshape, dtype, dy_cs=None, owner=None):
    """The same data gradient for FEW input channels (2 * Cin <= 64), even H and W: the two column parities of an input
    row pair are ONE output pixel of 2 * Cin channels -- dX viewed as (N, H, W / 2, 2 Cin) -- so each row parity a is a
    single stride-1 correlation of dY with a (1 + a) x 2 kernel into 2 Cin channels, scattered to the rows 2 i + a
    (``yv4_conv_scatter_fwd``, sh = 2, sw = 1).  Two launches instead of four, each with a full 64-column tile and whole
    128-byte lines per output pixel; a third more FLOPs (the zero blocks) on a layer that is bound by its bytes:
    conv1 of CSPDarknet (32 -> 64 at 608 -> 304, batch 64): 1.61 -> 0.8 ms."""
    N, Cin, H, W = xshape
    Cout = weight.shape[0]
    Ho, Wo = dy.shape[2], dy.shape[3]
    h16 = dtype != torch.float32
    L = _lib.lib()
    dx = torch.empty((N, Cin, H, W), device=dy.device, dtype=dtype, memory_format=torch.channels_last)
    ones, zeros = _identity_affine(dy.device, 2 * Cin)
    w2 = _rowpair_weights(weight, dtype, owner)
    for a in (0, 1):
        d = ConvDesc()
        d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = N, Ho, Wo, Cout, H // 2, W // 2, 2 * Cin
        d.KH, d.KW, d.stride, d.pad = 1 + a, 2, 1, 0
        d.x_cstride, d.y_cstride = (dy_cs if dy_cs is not None else Cout), 2 * Cin
        if h16:
            check(L.yv4_conv_scatter_fwd_h16(C.byref(d), _DCODE[dtype], dy.data_ptr(), w2[a].data_ptr(), ones.data_ptr(),
                                             zeros.data_ptr(), dx.data_ptr(), H, W // 2, 2, 1, a, 0, stream_ptr()),
                  'yv4_conv_scatter_fwd_h16')
        else:
            check(L.yv4_conv_scatter_fwd(C.byref(d), dy.data_ptr(), w2[a].data_ptr(), ones.data_ptr(), zeros.data_ptr(),
                                         dx.data_ptr(), H, W // 2, 2, 1, a, 0, stream_ptr()), 'yv4_conv_scatter_fwd')
    return dx


_ROWPAIR_ON = os.environ.get('YV4_DGRAD_ROWPAIR', '1') != '0'     # A/B switch


# ---- weight gradients straight into the gradient arena -----------------------------------------------------
# When a conv weight's ``.grad`` is a channels_last fp32 tensor that already exists at backward time (the flat
# gradient arena of ``flat_state.FlatState``, zeroed once per step), ``yv4_conv_wgrad*`` accumulates INTO it -- its
# (Cout, KH, KW, Cin) memory is exactly the kernel's dW layout and the kernels only ever atomicAdd -- and the
# Function returns no gradient for the weight.  That removes, per conv and step, the zero fill of a scratch dW and
# autograd's ``grad += dW`` (230 launches of the YOLOv4-L step).  autograd's post-accumulate hooks do not fire for
# such a weight (the Function is given the detached weight, so autograd never sees it), so whoever needs to know
# that a weight gradient is final registers a listener here
# (``dist.GradReducer`` does).  YV4_DIRECT_WGRAD=0 restores the autograd path.
import os as _os

_DIRECT_WGRAD = _os.environ.get('YV4_DIRECT_WGRAD', '1') != '0'
_direct_grad_listeners = []
_listeners_need_main_stream = []     # listeners that assume the gradient was written on the CURRENT stream


def add_direct_grad_listener(cb, side_stream_ok=False):
    """``cb(weight)`` is called after a conv's backward accumulated dW into ``weight.grad`` directly.  ``side_stream_ok``:
    the listener knows that the weight gradient may have been launched on the side stream and orders itself behind
    ``wgrad_side_stream(device)`` (the gradient exchange does: ``GradReducer._launch``); a listener that does not say so
    switches the side stream off while it is registered."""
    _direct_grad_listeners.append(cb)
    if not side_stream_ok:
        _listeners_need_main_stream.append(cb)
    return cb


def remove_direct_grad_listener(cb):
    if cb in _direct_grad_listeners:
        _direct_grad_listeners.remove(cb)
    if cb in _listeners_need_main_stream:
        _listeners_need_main_stream.remove(cb)


def _flat_f32(g, n):
    return g is not None and g.dtype == torch.float32 and g.numel() == n and g.is_contiguous()


class _ParamRef:
    """Carries a parameter through ``Function.apply`` without autograd seeing a tensor argument."""
    __slots__ = ('p',)

    def __init__(self, p):
        self.p = p


def _direct_grad_target(weight, cp):
    g = weight.grad
    if not _DIRECT_WGRAD or g is None or g.dtype != torch.float32 or g.shape != weight.shape or cp != weight.shape[1]:
        return None
    if not getattr(weight, '_yv4_grad_in_arena', False) or not g.permute(0, 2, 3, 1).is_contiguous():
        return None
    return g


# YV4_WGRAD_ATOMIC=1: the round-1 weight gradient (split-M partials added with float atomics: not run-to-run
# deterministic); default: partials in a workspace + ordered reduction (yv4_conv_wgrad_det)
_WGRAD_ATOMIC = os.environ.get('YV4_WGRAD_ATOMIC', '0') == '1'
_WGRAD_WS = {}


def _wgrad_workspace(nbytes, device, stream_key=None):
    """One growing fp32 scratch per (device, stream), shared by all layers (kernels on one stream run in order)."""
    if not nbytes:
        return None
    key = (device, stream_key)
    ws = _WGRAD_WS.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)
        _WGRAD_WS[key] = ws
    return ws


# The weight gradient of a layer is needed by nobody before the optimizer (or the gradient exchange), while the data gradient
# is on backward's critical path: with dW going straight into the gradient arena, ``yv4_conv_wgrad*`` is launched on a SIDE
# stream.  It then runs beside the BatchNorm backward passes of the layers in front of it -- matrix-pipe-bound work
# beside HBM-bound work -- instead of between them.  Ordering: the side stream waits for the current stream at every launch
# (dY, x and the zeroed arena are ready), the tensors are handed to the allocator with ``record_stream``, and the current
# stream waits for the side stream in a callback the autograd engine runs at the END of this backward pass (so every
# ``.backward()`` leaves finished gradients behind, whoever called it).  Gradient listeners: the multi-GPU exchange launches
# a bucket when its last weight gradient has been ISSUED, and orders the bucket's collective behind the side stream itself
# (``GradReducer._launch`` -> ``wgrad_side_stream``); any other listener switches the side stream off while registered.
# YV4_WGRAD_STREAM=0 keeps everything on one stream.
_WGRAD_STREAM = _os.environ.get('YV4_WGRAD_STREAM', '1') != '0'
_SIDE_STREAMS = {}
_side_join_pending = [False]     # a join callback is queued with the autograd engine for the backward pass in flight
_side_dirty = [False]            # a side stream holds weight gradients the current stream has not waited for


def wgrad_side_stream(device):
    """The side stream that holds weight gradients not yet joined into the current stream of ``device``, or None."""
    return _SIDE_STREAMS.get(torch.device(device)) if _side_dirty[0] else None


def _wgrad_side_stream(device):
    if not _WGRAD_STREAM or _listeners_need_main_stream:
        return None
    st = _SIDE_STREAMS.get(device)
    if st is None:
        st = torch.cuda.Stream(device=device)
        _SIDE_STREAMS[device] = st
    return st


def _join_side_streams():
    _side_join_pending[0] = False
    if _side_dirty[0]:
        _side_dirty[0] = False
        for dev, st in _SIDE_STREAMS.items():
            torch.cuda.current_stream(dev).wait_stream(st)


def join_side_streams():
    """Make the current stream wait for every weight gradient launched on a side stream.  Idempotent and cheap (one event
    wait per device when something is pending, nothing otherwise).  The autograd callback does this at the end of every
    backward pass -- but the engine DROPS queued callbacks when a backward raises (an OOM retry, ``pytest.raises``, a failed
    check in a later node), so nothing may rely on the callback alone: the optimizer step, the gradient hooks, the gradient
    exchange and the next forward pass all call this before they touch the gradient arena."""
    _join_side_streams()


class ConvFunction(torch.autograd.Function):
    """``dtype``: torch.float32, or torch.float16 / torch.bfloat16 -- then x, y and their gradients are
    that type (fp32 accumulation in every kernel) while ``weight`` and its gradient stay fp32 (the
    master copy the optimizer steps; autocast semantics of the reference's Fp16 hook)."""

    @staticmethod
    def forward(ctx, x, weight, stride, pad, dtype, stats=None, direct=None, sink=None, cat_buf=None, cat_total=0,
                cat_off=0, park=None):
        """``direct``: a ``_ParamRef`` to the parameter whose ``.grad`` receives dW in place (then ``weight`` is the
        detached parameter: autograd does not track it through this Function, see ``conv2d``).
        ``cat_total`` > 0: the output is channels [cat_off, cat_off + Cout) of a (N, cat_total, Ho, Wo) concat buffer --
        ``cat_buf`` if given (written in place and returned), else a fresh one whose other channels a later producer
        fills (see ``CatSlot``).  ``park``: a ``GradSink`` that receives this conv's data gradient instead of autograd
        (x also feeds a conv whose backward runs LATER and adds the parked gradient in its own launch)."""
        _need_cuda(x, 'x')
        if _side_join_pending[0] or _side_dirty[0]:
            # a forward pass while a join is still "pending": the backward that queued it never finished (its callback was
            # dropped with the exception) -- join now, so the one-shot flag cannot stay stuck for the rest of the process
            _join_side_streams()
        ctx.direct = direct
        ctx.park = park
        ctx.sink = sink if stride == 1 else None     # (the stride-2 parity form has no residual input)
        Cout, Cin, KH, KW = weight.shape
        al = 4 if dtype == torch.float32 else 8
        xc = x.shape[1]
        # ``x`` may carry MORE channels than the weight (zeros: the image stored with a whole number of 16-byte chunks
        # per pixel, ``image_to_nhwc16``): the forward operand is packed to x's width, the weight gradient is taken
        # over the weight's own (padded) channels with x's pixel stride; such an x has no gradient
        assert xc >= Cin and Cout % al == 0 and xc % al == 0 and (xc == Cin or not ctx.needs_input_grad[0]), \
            f'training conv needs channel counts that are multiples of {al} (got {xc}/{Cin}->{Cout})'
        ctx.x_dtype = x.dtype
        x = to_nhwc(x.to(dtype))
        N, _, H, W = x.shape
        Ho = (H + 2 * pad - KH) // stride + 1
        Wo = (W + 2 * pad - KW) // stride + 1
        wp, cp = packed_weight(weight, dtype, owner=direct.p if direct is not None else None,
                               pad_to=xc if xc != Cin else None)
        assert cp == xc
        ctx.cat = None
        if cat_total:
            assert cat_off % al == 0 and cat_off + Cout <= cat_total
            if cat_buf is None:
                y = torch.empty((N, cat_total, Ho, Wo), device=x.device, dtype=dtype, memory_format=torch.channels_last)
            else:
                assert tuple(cat_buf.shape) == (N, cat_total, Ho, Wo) and cat_buf.dtype == dtype and _is_nhwc(cat_buf)
                y = cat_buf
                ctx.mark_dirty(cat_buf)
            ctx.cat = (cat_off, cat_buf is not None)
            _conv_launch(x, wp, cp, Cout, KH, KW, stride, pad, y, stats, y_cs=cat_total, y_co=cat_off)
        else:
            y = torch.empty((N, Cout, Ho, Wo), device=x.device, dtype=dtype, memory_format=torch.channels_last)
            _conv_launch(x, wp, cp, Cout, KH, KW, stride, pad, y, stats)
        ctx.save_for_backward(x, weight)
        ctx.geom = (stride, pad, dtype, (Cin + al - 1) // al * al)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        stride, pad, dtype, cp = ctx.geom
        Cout, Cin, KH, KW = weight.shape
        N, _, H, W = x.shape
        dcat = None
        if ctx.cat is not None:          # dy is the gradient of the whole concat buffer: this conv's slice of it,
            off, passed = ctx.cat        # and the buffer's gradient handed on to the producer of the other channels
            dcat = dy if passed else None
            dy = dy[:, off:off + Cout]
        dy, dy_cs = nhwc_or_slice(dy, dtype)
        Ho, Wo = dy.shape[2], dy.shape[3]
        L = _lib.lib()
        h16 = dtype != torch.float32
        code = _DCODE[dtype]
        dx = dw = None
        if ctx.needs_input_grad[1] or ctx.direct is not None:
            target = _direct_grad_target(ctx.direct.p, cp) if ctx.direct is not None else None
            dwp = target if target is not None else torch.zeros((Cout, KH * KW * cp), device=x.device,
                                                                dtype=torch.float32)
            d = ConvDesc()
            d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = N, H, W, cp, Ho, Wo, Cout
            d.KH, d.KW, d.stride, d.pad = KH, KW, stride, pad
            d.x_cstride, d.y_cstride = x.shape[1], dy_cs
            if _WGRAD_ATOMIC:
                if h16:
                    check(L.yv4_conv_wgrad_h16(C.byref(d), code, x.data_ptr(), dy.data_ptr(), dwp.data_ptr(),
                                               stream_ptr()), 'yv4_conv_wgrad_h16')
                else:
                    check(L.yv4_conv_wgrad(C.byref(d), x.data_ptr(), dy.data_ptr(), dwp.data_ptr(), stream_ptr()),
                          'yv4_conv_wgrad')
            else:
                # deterministic form: partial sums of the reduction chunks in a workspace, added in chunk order
                need = int(L.yv4_conv_wgrad_workspace(C.byref(d), code))
                side = _wgrad_side_stream(x.device) if target is not None else None
                ws = _wgrad_workspace(need, x.device, 'side' if side is not None else None)
                if side is not None:
                    side.wait_stream(torch.cuda.current_stream(x.device))
                    with torch.cuda.stream(side):
                        check(L.yv4_conv_wgrad_det(C.byref(d), code, x.data_ptr(), dy.data_ptr(), dwp.data_ptr(),
                                                   ws.data_ptr() if need else None, need, stream_ptr()), 'yv4_conv_wgrad_det')
                    x.record_stream(side)
                    dy.record_stream(side)
                    if ws is not None:
                        ws.record_stream(side)       # (a grown workspace frees its predecessor while the side stream may still read it)
                    _side_dirty[0] = True
                    if not _side_join_pending[0]:
                        _side_join_pending[0] = True
                        torch.autograd.Variable._execution_engine.queue_callback(_join_side_streams)
                else:
                    check(L.yv4_conv_wgrad_det(C.byref(d), code, x.data_ptr(), dy.data_ptr(), dwp.data_ptr(),
                                               ws.data_ptr() if need else None, need, stream_ptr()), 'yv4_conv_wgrad_det')
            if target is None:
                dw = dwp.view(Cout, KH, KW, cp)[..., :Cin].permute(0, 3, 1, 2)
                if cp != Cin:
                    dw = dw.contiguous()
            if ctx.direct is not None:       # autograd does not track the weight here: hand the gradient over
                prm = ctx.direct.p
                if target is None:           # (.grad went away between forward and backward)
                    prm.grad = dw.clone() if prm.grad is None else prm.grad.add_(dw)
                dw = None
                for cb in _direct_grad_listeners:
                    cb(prm)
        if ctx.needs_input_grad[0]:
            joined = jcs = None
            if ctx.sink is not None and ctx.sink.value is not None:
                joined, jcs = ctx.sink.value, ctx.sink.cs
                ctx.sink.value = None
                if joined.dtype != dtype or tuple(joined.shape) != (N, Cin, H, W):
                    joined = to_nhwc(joined.to(dtype))
                    jcs = None
            if stride == 2 and (KH, KW, pad) == (3, 3, 1) and Cout % (8 if h16 else 32) == 0:
                own = ctx.direct.p if ctx.direct is not None else None
                if _ROWPAIR_ON and 2 * Cin <= 64 and H % 2 == 0 and W % 2 == 0 and Cin % (4 if h16 else 2) == 0:
                    dx = _dgrad_s2_rowpair(dy, weight, (N, Cin, H, W), dtype, dy_cs, owner=own)
                else:
                    dx = _dgrad_s2_parity(dy, weight, (N, Cin, H, W), dtype, dy_cs, owner=own)
            else:
                dx = _dgrad_dilated(dy, weight, (N, Cin, H, W), stride, pad, dtype, dy_cs, residual=joined, res_cs=jcs,
                                    owner=ctx.direct.p if ctx.direct is not None else None)
                joined = None
            if joined is not None:
                dx = dx + joined
            dx = dx.to(ctx.x_dtype)
        elif ctx.sink is not None and ctx.sink.value is not None:
            dx, ctx.sink.value = ctx.sink.value, None       # nobody wants this conv's share: hand the parked one on
        if ctx.park is not None and dx is not None:
            ctx.park.value, ctx.park.cs = dx, None          # joins the other consumer's data gradient (GradSink)
            dx = None
        return dx, dw, None, None, None, None, None, None, dcat, None, None, None


def train_dtype(module, x):
    """Operand type of a training-mode conv: the module's ``compute_dtype`` (``wrap_fp16_model``) if
    it is a 16-bit type, else the type ``x`` already has (16-bit activations stay 16-bit), else fp32."""
    dt = getattr(module, 'compute_dtype', None)
    if dt in (torch.float16, torch.bfloat16):
        return dt
    return x.dtype if x.dtype in (torch.float16, torch.bfloat16) else torch.float32


class CatSlot:
    """Where a producer writes inside a channel-concat buffer instead of returning a tensor of its own that
    ``torch.cat`` would copy: ``CatSlot(total, offset)`` for the FIRST producer (it allocates the (N, total, H, W)
    buffer and returns it), ``CatSlot(total, offset, buf)`` for every further one (``buf`` = what the previous producer
    returned; it is written in place and returned again).  The gradient of the buffer reaches every producer's
    backward whole; each takes its own channel slice (a strided view, no copy)."""
    __slots__ = ('total', 'off', 'buf')

    def __init__(self, total, off, buf=None):
        self.total, self.off, self.buf = int(total), int(off), buf

    def args(self):
        return self.buf, self.total, self.off


def conv2d(x, weight, stride=1, pad=0, dtype=None, stats=None, sink=None, cat=None, park=None):
    """``dtype`` None: follow ``x`` (a 16-bit activation keeps the path 16-bit, anything else is fp32).
    ``stats``: see ``_conv_launch`` / ``conv_stats_buffer``; pass the same buffer to ``bn_act(..., sums=)``.
    ``cat``: a ``CatSlot`` -- the result is the concat buffer with this conv's channels written."""
    if dtype is None:
        dtype = x.dtype if x.dtype in (torch.float16, torch.bfloat16) else torch.float32
    cargs = cat.args() if cat is not None else (None, 0, 0)
    if (_DIRECT_WGRAD and weight.requires_grad and weight.is_leaf and x.requires_grad and torch.is_grad_enabled()
            and _direct_grad_target(weight, weight.shape[1]) is not None
            and weight.shape[1] % (4 if dtype == torch.float32 else 8) == 0):
        # dW goes straight into weight.grad (see the note above ConvFunction): the Function sees the detached weight
        return ConvFunction.apply(x, weight.detach(), stride, pad, dtype, stats, _ParamRef(weight), sink, *cargs, park)
    return ConvFunction.apply(x, weight, stride, pad, dtype, stats, None, sink, *cargs, park)


def stats_numel(cout):
    return _lib.STATS_REPLICAS * 2 * cout


def conv_stats_buffer(cout, device, persistent=False):
    """Scratch for the BatchNorm sums a training conv leaves for the BN that follows it.  ``persistent``: a zeroed
    buffer its owner keeps across steps; it is marked so that the conv skips the memset and ``bn_act`` has the
    finalize kernel clear it again as it reads it (the owner must pair every ``conv2d(stats=)`` with
    ``bn_act(sums=)``)."""
    if not persistent:
        return torch.empty(_lib.STATS_REPLICAS * 2 * cout, dtype=torch.float64, device=device)
    buf = torch.zeros(_lib.STATS_REPLICAS * 2 * cout, dtype=torch.float64, device=device)
    buf._yv4_kept_clean = True
    return buf


class BNActFunction(torch.autograd.Function):
    """Train-mode BatchNorm2d + activation (+ residual add) as one forward and one backward."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, eps, momentum, act, slope, residual, training=True,
                sync_group=None, sums=None, direct=None, res_sink=None, cat_buf=None, cat_total=0, cat_off=0):
        """``sync_group``: None, or a (process group or 'world') to synchronise the batch statistics over
        (torch.nn.SyncBatchNorm semantics: statistics over all ranks' rows, local dgamma / dbeta).
        ``sums``: the replicated [sum | sum of squares] buffer the producing conv filled (``conv2d(stats=)``):
        the statistics pass over ``x`` is skipped."""
        _need_cuda(x, 'x')
        if x.dtype not in _DCODE:
            x = x.float()
        x = to_nhwc(x)
        code = _DCODE[x.dtype]
        N, Cc, H, W = x.shape
        assert Cc % 4 == 0, 'BatchNorm kernels need a channel count that is a multiple of 4'
        M = N * H * W
        dev = x.device
        L = _lib.lib()
        work = torch.empty(4 * Cc, dtype=torch.float64, device=dev)     # (4*C: room for the deterministic mode's lo words)
        mean = torch.empty(Cc, dtype=torch.float32, device=dev)
        invstd = torch.empty(Cc, dtype=torch.float32, device=dev)
        rows = None
        bwd_work = None
        if training and sync_group is not None:
            import torch.distributed as dist
            group = None if sync_group == 'world' else sync_group
            pre = sums
            sums = torch.empty(4 * Cc + 1, dtype=torch.float64, device=dev)[:2 * Cc + 1]   # [sum | sum of squares | rows]
            if pre is not None:
                check(L.yv4_conv_stats_fold(pre.data_ptr(), Cc, 1 if getattr(pre, '_yv4_kept_clean', False) else 0,
                                            sums.data_ptr(), stream_ptr()), 'yv4_conv_stats_fold')
            else:       # (works in 4*C doubles: the tensor above is a view of 4*C + 1)
                check(L.yv4_bn_partial_sums(x.data_ptr(), code, M, Cc, Cc, 0, sums.data_ptr(), stream_ptr()),
                      'yv4_bn_partial_sums')
            sums[2 * Cc:].fill_(float(M))
            dist.all_reduce(sums, group=group)
            rows = sums[2 * Cc:]
            check(L.yv4_bn_finalize(sums.data_ptr(), 1, 0, rows.data_ptr(), Cc, float(eps), float(momentum),
                                    mean.data_ptr(), invstd.data_ptr(),
                                    running_mean.data_ptr() if running_mean is not None else None,
                                    running_var.data_ptr() if running_var is not None else None, 0, None, stream_ptr()),
                  'yv4_bn_finalize')
            ctx.sync_group = group
        elif training and isinstance(sums, (list, tuple)):
            # x is a concat buffer whose channel ranges were produced by several convs, each leaving its own sums
            # (``CatSlot`` + ``conv2d(stats=)``): one finalize per range, on the ranges of mean / invstd / running stats
            # (the finalize kernel clears the 4*c words at the pointer it is handed: the ranges' pieces tile the 4*Cc words
            # of the backward's accumulator)
            bwd_work = torch.empty(4 * Cc, dtype=torch.float64, device=dev)
            off = 0
            for buf in sums:
                c = buf.numel() // (2 * _lib.STATS_REPLICAS)
                check(L.yv4_bn_finalize(buf.data_ptr(), _lib.STATS_REPLICAS, M, None, c, float(eps), float(momentum),
                                        mean.data_ptr() + 4 * off, invstd.data_ptr() + 4 * off,
                                        running_mean.data_ptr() + 4 * off if running_mean is not None else None,
                                        running_var.data_ptr() + 4 * off if running_var is not None else None,
                                        1 if getattr(buf, '_yv4_kept_clean', False) else 0,
                                        bwd_work.data_ptr() + 32 * off, stream_ptr()), 'yv4_bn_finalize')
                off += c
            assert off == Cc, 'the statistics buffers do not cover the concat buffer'
        elif training and sums is not None:
            # the finalize kernel also clears the backward's reduction buffer (and a persistent statistics buffer)
            bwd_work = torch.empty(4 * Cc, dtype=torch.float64, device=dev)
            check(L.yv4_bn_finalize(sums.data_ptr(), _lib.STATS_REPLICAS, M, None, Cc, float(eps), float(momentum),
                                    mean.data_ptr(), invstd.data_ptr(),
                                    running_mean.data_ptr() if running_mean is not None else None,
                                    running_var.data_ptr() if running_var is not None else None,
                                    1 if getattr(sums, '_yv4_kept_clean', False) else 0, bwd_work.data_ptr(),
                                    stream_ptr()), 'yv4_bn_finalize')
        elif training:
            check(L.yv4_bn_train_stats_h16(x.data_ptr(), code, M, Cc, Cc, 0, float(eps), float(momentum),
                                           work.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                           running_mean.data_ptr() if running_mean is not None else None,
                                           running_var.data_ptr() if running_var is not None else None, stream_ptr()),
                  'yv4_bn_train_stats')
        else:       # eval-mode BN inside a training graph: the running statistics are constants
            mean.copy_(running_mean.detach().float())
            torch.rsqrt(running_var.detach().float() + eps, out=invstd)
        res = to_nhwc(residual.to(x.dtype)) if residual is not None else None
        ctx.cat = None
        y_cs, y_co = Cc, 0
        if cat_total:                # the output is a channel slice of a concat buffer (``CatSlot``)
            assert cat_off % (4 if x.dtype == torch.float32 else 8) == 0 and cat_off + Cc <= cat_total
            if cat_buf is None:
                y = torch.empty((N, cat_total, H, W), device=dev, dtype=x.dtype, memory_format=torch.channels_last)
            else:
                assert tuple(cat_buf.shape) == (N, cat_total, H, W) and cat_buf.dtype == x.dtype and _is_nhwc(cat_buf)
                y = cat_buf
                ctx.mark_dirty(cat_buf)
            ctx.cat = (cat_off, cat_buf is not None)
            y_cs, y_co = cat_total, cat_off
        else:
            y = torch.empty_like(x, memory_format=torch.channels_last)
        g = gamma.detach().float().contiguous()
        b = beta.detach().float().contiguous()
        check(L.yv4_bn_act_fwd_h16(x.data_ptr(), code, Cc, 0, mean.data_ptr(), invstd.data_ptr(), g.data_ptr(),
                                   b.data_ptr(), res.data_ptr() if res is not None else None, Cc, 0, y.data_ptr(), y_cs,
                                   y_co, M, Cc, int(act), float(slope), stream_ptr()), 'yv4_bn_act_fwd')
        ctx.save_for_backward(x, mean, invstd, g, b)
        ctx.direct = direct      # (_ParamRef(weight), _ParamRef(bias)): dgamma / dbeta are added to their .grad in place
        ctx.bwd_work = bwd_work  # 2*C doubles already cleared by the finalize kernel, or None
        ctx.rows = rows
        ctx.act = (int(act), float(slope))
        ctx.training = bool(training)
        ctx.has_res = residual is not None
        ctx.res_sink = res_sink if residual is not None else None      # GradSink: the residual's gradient is parked
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, invstd, g, b = ctx.saved_tensors
        act, slope = ctx.act
        N, Cc, H, W = x.shape
        dcat = None
        if ctx.cat is not None:
            off, passed = ctx.cat
            dcat = dy if passed else None
            dy = dy[:, off:off + Cc]
        dy, dcs = nhwc_or_slice(dy, x.dtype)
        code = _DCODE[x.dtype]
        M = N * H * W
        dev = x.device
        dx = torch.empty_like(x, memory_format=torch.channels_last)
        dgamma = torch.empty(Cc, dtype=torch.float32, device=dev)
        dbeta = torch.empty(Cc, dtype=torch.float32, device=dev)
        work = torch.empty(4 * Cc, dtype=torch.float64, device=dev)
        L = _lib.lib()
        gw = gb = None
        if ctx.rows is not None:            # SyncBN: local sums -> all-reduce -> apply with the totals
            import torch.distributed as dist
            check(L.yv4_bn_act_bwd_sums(x.data_ptr(), code, Cc, 0, dy.data_ptr(), dcs, 0, mean.data_ptr(),
                                        invstd.data_ptr(), g.data_ptr(), b.data_ptr(), dgamma.data_ptr(),
                                        dbeta.data_ptr(), work.data_ptr(), M, Cc, act, slope, stream_ptr()),
                  'yv4_bn_act_bwd_sums')
            dist.all_reduce(work[:2 * Cc], group=ctx.sync_group)     # (the other half is the deterministic mode's scratch)
            check(L.yv4_bn_act_bwd_apply(x.data_ptr(), code, Cc, 0, dy.data_ptr(), dcs, 0, mean.data_ptr(),
                                         invstd.data_ptr(), g.data_ptr(), b.data_ptr(), dx.data_ptr(), Cc, 0,
                                         work.data_ptr(), M, 0, ctx.rows.data_ptr(), Cc, act, slope, stream_ptr()),
                  'yv4_bn_act_bwd_apply')
        else:
            if ctx.direct is not None:
                gw, gb = ctx.direct[0].p.grad, ctx.direct[1].p.grad
                if not (_flat_f32(gw, Cc) and _flat_f32(gb, Cc)):
                    gw = gb = None
            if gw is not None:       # dgamma / dbeta added to the parameters' gradients by the kernel itself
                flags = 0 if ctx.training else 1
                wk = work
                if ctx.bwd_work is not None:
                    wk, flags = ctx.bwd_work, flags | 2
                    ctx.bwd_work = None          # one use: a second backward through this node memsets again
                check(L.yv4_bn_act_bwd_accum(x.data_ptr(), code, Cc, 0, dy.data_ptr(), dcs, 0, mean.data_ptr(),
                                             invstd.data_ptr(), g.data_ptr(), b.data_ptr(), dx.data_ptr(), Cc, 0,
                                             gw.data_ptr(), gb.data_ptr(), wk.data_ptr(), M, Cc, act, slope,
                                             flags, stream_ptr()), 'yv4_bn_act_bwd_accum')
            else:
                fn = L.yv4_bn_act_bwd_h16 if ctx.training else L.yv4_bn_eval_act_bwd
                check(fn(x.data_ptr(), code, Cc, 0, dy.data_ptr(), dcs, 0, mean.data_ptr(),
                         invstd.data_ptr(), g.data_ptr(), b.data_ptr(), dx.data_ptr(), Cc, 0,
                         dgamma.data_ptr(), dbeta.data_ptr(), work.data_ptr(), M, Cc, act, slope,
                         stream_ptr()), 'yv4_bn_act_bwd')
        dres = dy if ctx.has_res else None
        if dres is not None and ctx.res_sink is not None:
            ctx.res_sink.value, ctx.res_sink.cs = dy, dcs       # added by the data-gradient launch that consumes the sink
            dres = None
        if ctx.direct is not None:   # autograd does not track gamma / beta through this Function
            if ctx.rows is not None or gw is None:
                for ref, gr in zip(ctx.direct, (dgamma, dbeta)):
                    ref.p.grad = gr.clone() if ref.p.grad is None else ref.p.grad.add_(gr)
            for ref in ctx.direct:
                for cb in _direct_grad_listeners:
                    cb(ref.p)
            dgamma = dbeta = None
        return dx, dgamma, dbeta, None, None, None, None, None, None, dres, None, None, None, None, None, dcat, None, None


def _sync_group(bn):
    """The group a ``torch.nn.SyncBatchNorm`` (norm_cfg type 'SyncBN', configs/yolov5_ddp) synchronises over:
    its ``process_group`` or the world; None for a plain BatchNorm2d or a single-process run (torch's
    SyncBatchNorm also falls back to local statistics when world_size == 1, batchnorm.py ``need_sync``)."""
    import torch.distributed as dist
    if not isinstance(bn, torch.nn.SyncBatchNorm) or not bn.training:
        return None
    if not (dist.is_available() and dist.is_initialized()):
        return None
    group = bn.process_group
    if dist.get_world_size(group) <= 1:
        return None
    return group if group is not None else 'world'


def bn_act(x, bn, act=(0, 0.0), residual=None, sums=None, res_sink=None, cat=None):
    """``bn``: a torch BatchNorm2d; in training mode it normalises with batch statistics and updates
    the running ones, in eval mode (``norm_eval`` / frozen stages inside a training graph) with the
    running statistics as constants.  act = (YV4_ACT_*, slope)."""
    mom = bn.momentum if bn.momentum is not None else 0.1
    use_batch = bn.training or not bn.track_running_stats
    gamma, beta, direct = bn.weight, bn.bias, None
    if (_DIRECT_WGRAD and x.requires_grad and torch.is_grad_enabled() and gamma.requires_grad and beta.requires_grad
            and getattr(gamma, '_yv4_grad_in_arena', False) and getattr(beta, '_yv4_grad_in_arena', False)
            and _flat_f32(gamma.grad, gamma.numel()) and _flat_f32(beta.grad, beta.numel())):
        # dgamma / dbeta go straight into the parameters' gradients (see the note above ConvFunction)
        direct = (_ParamRef(gamma), _ParamRef(beta))
        gamma, beta = gamma.detach(), beta.detach()
    out = BNActFunction.apply(x, gamma, beta, bn.running_mean if bn.track_running_stats else None,
                              bn.running_var if bn.track_running_stats else None, bn.eps, mom, act[0], act[1],
                              residual, use_batch, _sync_group(bn) if use_batch else None,
                              sums if use_batch else None, direct, res_sink,
                              *(cat.args() if cat is not None else (None, 0, 0)))
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        if _fwd_depth[0] > 0:        # inside a registered module's training forward: one multi-tensor add at its end
            _nbt_pending.append(bn.num_batches_tracked)
        else:
            bn.num_batches_tracked += 1
    return out


# ``num_batches_tracked += 1`` of every BatchNorm is a 5 us launch on a scalar (108 per YOLOv4-L forward): inside a
# registered module's training forward (``HipModule._dispatch`` keeps the depth) they are collected and applied by
# one ``torch._foreach_add_`` when the outermost forward returns.
_fwd_depth = [0]
_nbt_pending = []


def flush_batch_counters():
    if _nbt_pending:
        torch._foreach_add_(_nbt_pending, 1)
        _nbt_pending.clear()


class ResampleIntoFunction(torch.autograd.Function):
    """Nearest resample of ``x`` to (Hd, Wd) written into a channel range of a concat buffer (``CatSlot``) -- the
    ``F.interpolate(...)`` + ``torch.cat`` of necks/yolo_neck_csp.py:213-219,229 without the intermediate tensor; with
    Hd == Hs it is the plain copy of a saved tensor into its concat half.  Backward: the buffer's gradient slice, summed
    over the pixels that read each source pixel (``yv4_resample_nearest_bwd``; the slice itself for the copy)."""

    @staticmethod
    def forward(ctx, x, Hd, Wd, cat_buf, cat_total, cat_off):
        _need_cuda(x, 'x')
        if x.dtype not in _DCODE:
            x = x.float()
        x = to_nhwc(x)
        N, Cc, Hs, Ws = x.shape
        al = 4 if x.dtype == torch.float32 else 8
        assert Cc % al == 0 and cat_off % al == 0 and cat_total % al == 0 and cat_off + Cc <= cat_total
        if cat_buf is None:
            out = torch.empty((N, cat_total, Hd, Wd), device=x.device, dtype=x.dtype, memory_format=torch.channels_last)
        else:
            assert tuple(cat_buf.shape) == (N, cat_total, Hd, Wd) and cat_buf.dtype == x.dtype and _is_nhwc(cat_buf)
            out = cat_buf
            ctx.mark_dirty(cat_buf)
        k = 1 if x.dtype == torch.float32 else 2      # a 16-bit map with C % 8 == 0 is an fp32 map with C / 2 channels
        check(_lib.lib().yv4_resample_nearest_fwd(x.data_ptr(), out.data_ptr(), N, Hs, Ws, Hd, Wd, Cc // k, Cc // k, 0,
                                                  cat_total // k, cat_off // k, stream_ptr()), 'yv4_resample_nearest_fwd')
        ctx.geom = (N, Cc, Hs, Ws, Hd, Wd, cat_off, cat_buf is not None, x.dtype)
        return out

    @staticmethod
    def backward(ctx, dz):
        N, Cc, Hs, Ws, Hd, Wd, off, passed, dtype = ctx.geom
        dy, dcs = nhwc_or_slice(dz[:, off:off + Cc], dtype)
        if (Hs, Ws) == (Hd, Wd):
            dx = dy                                   # the copy's gradient is the slice itself (a strided view)
        else:
            dx = torch.empty((N, Cc, Hs, Ws), device=dz.device, dtype=dtype, memory_format=torch.channels_last)
            check(_lib.lib().yv4_resample_nearest_bwd(dy.data_ptr(), dx.data_ptr(), N, Hs, Ws, Hd, Wd, Cc, dcs, 0,
                                                      _DCODE[dtype], stream_ptr()), 'yv4_resample_nearest_bwd')
        return dx, None, None, (dz if passed else None), None, None


def resample_into(x, size, cat):
    """``x`` nearest-resampled to ``size`` (an integer multiple of its own, or the same) into ``cat`` (a ``CatSlot``)."""
    return ResampleIntoFunction.apply(x, int(size[0]), int(size[1]), *cat.args())


def resample_into_ok(x, size):
    """Integer scale factors (what the backward kernel sums over), 16-byte channel chunks."""
    Hs, Ws = x.shape[2], x.shape[3]
    al = 4 if x.dtype == torch.float32 else 8
    return x.is_cuda and x.dtype in _DCODE and x.shape[1] % al == 0 and size[0] % Hs == 0 and size[1] % Ws == 0 \
        and size[0] // Hs <= 8 and size[1] // Ws <= 8


class SPPCatFunction(torch.autograd.Function):
    """``torch.cat([x, mp5(x), mp9(x), mp13(x)], 1)`` (darknetcsp.py:176-181,203-206,222-226) as one
    forward (slice copy + ``yv4_spp_pool_fwd``) and one backward launch (``yv4_spp_pool_bwd``); the
    concat buffer itself is what is saved."""

    @staticmethod
    def forward(ctx, x):
        _need_cuda(x, 'x')
        if x.dtype not in _DCODE:
            x = x.float()
        x = to_nhwc(x)
        N, Cc, H, W = x.shape
        al = 4 if x.dtype == torch.float32 else 8
        assert Cc % al == 0, f'SPP kernels need a channel count that is a multiple of {al}'
        out = torch.empty((N, 4 * Cc, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        out[:, :Cc] = x
        L = _lib.lib()
        if x.dtype == torch.float32:
            check(L.yv4_spp_pool_fwd(out.data_ptr(), N, H, W, Cc, 4 * Cc, 0, stream_ptr()), 'yv4_spp_pool_fwd')
        else:
            check(L.yv4_spp_pool_fwd_h16(out.data_ptr(), N, H, W, Cc, 4 * Cc, 0, _DCODE[x.dtype], stream_ptr()),
                  'yv4_spp_pool_fwd_h16')
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, dcat):
        out, = ctx.saved_tensors
        N, C4_, H, W = out.shape
        Cc = C4_ // 4
        dcat = to_nhwc(dcat.to(out.dtype))
        dx = torch.zeros((N, Cc, H, W), dtype=torch.float32, device=out.device).contiguous(
            memory_format=torch.channels_last)
        check(_lib.lib().yv4_spp_pool_bwd(out.data_ptr(), 4 * Cc, 0, dcat.data_ptr(), 4 * Cc, 0, dx.data_ptr(), N, H, W,
                                          Cc, _DCODE[out.dtype], stream_ptr()), 'yv4_spp_pool_bwd')
        return dx.to(out.dtype)


def spp_cat(x):
    return SPPCatFunction.apply(x)
