"""Launch plan of the fused inference path.

A ``Plan`` is the flattened, fused form of a module tree: NHWC fp32 buffers in HBM,
channel-sliced views of them (a ``torch.cat`` becomes two producers writing into one
buffer at different channel offsets), and an ordered list of C-ABI launches
(``include/yv4.h``).  Modules contribute through their ``emit(plan, x, ...)`` methods;
``Plan.run()`` replays the list on the current HIP stream with no host sync and no
allocation, so it can be captured into a hipGraph (``Plan.capture()``).

What is fused away relative to the reference's eager graph:
BN and Mish (108 + 108 elementwise passes) -> conv epilogue; residual add -> conv
epilogue; every channel ``cat`` -> channel-offset stores; CSP-level cat->BN->Mish
(darknetcsp.py:106-109,149-153,220-229) -> a second affine+act stage in the producing
convs' epilogues; permute+reshape of the pred maps (yolocsp_head.py:264) -> NHWC is
already that layout.
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import ConvDesc, LevelDesc, check


class Buf:
    """One NHWC allocation: (N, H, W, C) with C elements per pixel (fp32, or fp16 / bf16 in a
    16-bit plan)."""

    def __init__(self, N, H, W, C_, name='', dtype=torch.float32):
        self.N, self.H, self.W, self.C = int(N), int(H), int(W), int(C_)
        self.name = name
        self.dtype = dtype
        self.tensor = None

    @property
    def numel(self):
        return self.N * self.H * self.W * self.C

    def ptr(self):
        return self.tensor.data_ptr()


class View:
    """Channels [coff, coff+C) of a Buf."""

    def __init__(self, buf, coff, C_):
        assert 0 <= coff and coff + C_ <= buf.C, (coff, C_, buf.C)
        self.buf, self.coff, self.C = buf, int(coff), int(C_)

    N = property(lambda s: s.buf.N)
    H = property(lambda s: s.buf.H)
    W = property(lambda s: s.buf.W)
    cstride = property(lambda s: s.buf.C)

    def slice(self, off, C_):
        return View(self.buf, self.coff + off, C_)

    @property
    def shape_nchw(self):
        return (self.N, self.C, self.H, self.W)

    def __repr__(self):
        return f'View({self.buf.name}[{self.N},{self.H},{self.W},{self.coff}:{self.coff + self.C}/{self.cstride}])'


class Op:
    """One launch.  ``kind`` in {'conv','spp','resample','to_nhwc','to_nchw','decode',
    'nms','reset'}; ``flops``/``bytes`` are the ALGORITHMIC work."""

    def __init__(self, kind, name, fn, flops=0.0, nbytes=0.0, info=None):
        self.kind, self.name, self.fn = kind, name, fn
        self.flops, self.bytes, self.info = float(flops), float(nbytes), info or {}


def act_id(act_module):
    """Map an activation nn.Module to (YV4_ACT_*, slope)."""
    if act_module is None:
        return _lib.ACT_NONE, 0.0
    n = type(act_module).__name__
    if n == 'Mish':
        return _lib.ACT_MISH, 0.0
    if n == 'LeakyReLU':
        return _lib.ACT_LEAKY, float(act_module.negative_slope)
    if n in ('Swish', 'SiLU'):
        return _lib.ACT_SWISH, 0.0
    if n == 'Identity':
        return _lib.ACT_NONE, 0.0
    raise NotImplementedError(f'activation {n} has no fused epilogue (built: Mish, LeakyReLU, Swish/SiLU, none)')


def bn_affine(bn):
    """Eval-mode BatchNorm as y = x*s + t, computed like ATen's CPU transform
    (alpha = weight * invstd, beta = bias - mean * alpha) in fp32."""
    w = bn.weight.detach().float() if bn.weight is not None else torch.ones_like(bn.running_mean)
    b = bn.bias.detach().float() if bn.bias is not None else torch.zeros_like(bn.running_mean)
    invstd = 1.0 / torch.sqrt(bn.running_var.detach().float() + bn.eps)
    s = w * invstd
    t = b - bn.running_mean.detach().float() * s
    return s.contiguous(), t.contiguous()


def pack_conv_weight(weight, cin_pad=None, align=4):
    """(Cout, Cin, KH, KW) -> (Cout, KH*KW*Cin_pad) with K ordered (kh, kw, ci), the order
    the kernel's NHWC gather walks; Cin is zero-padded to a multiple of ``align`` (4 floats or
    8 sixteen-bit elements = one 16-byte chunk)."""
    Cout, Cin, KH, KW = weight.shape
    cp = cin_pad if cin_pad is not None else (Cin + align - 1) // align * align
    w = weight.detach().float().permute(0, 2, 3, 1)  # Cout, KH, KW, Cin
    if cp != Cin:
        w = torch.nn.functional.pad(w, (0, cp - Cin))
    return w.reshape(Cout, KH * KW * cp).contiguous(), cp


_DCODE = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}      # YV4_F32 / YV4_F16 / YV4_BF16
_PLAN_NT = os.environ.get('YV4_PLAN_NT', '0') == '1'                 # non-temporal output stores in 16-bit plans (ABI 7): opt-in


class Plan:
    """``dtype``: torch.float32 (the parity dtype) or torch.float16 / torch.bfloat16: activations and
    weights are stored in that type, convs accumulate in fp32 (``yv4_conv_bn_act_fwd_h16``), folded BN
    scale/shift and the pred maps handed to decode stay fp32."""

    def __init__(self, device, dtype=torch.float32):
        self.device = torch.device(device)
        if dtype not in _DCODE:
            raise ValueError(f'Plan dtype must be float32, float16 or bfloat16 (got {dtype})')
        self.dtype = dtype
        self.h16 = dtype != torch.float32
        self.dcode = _DCODE[dtype]
        self.calign = 8 if self.h16 else 4       # channels per 16-byte chunk
        self.esize = 2 if self.h16 else 4
        self.bufs = []
        self.ops = []
        self.params = []      # device tensors kept alive (packed weights, scales, descs)
        self.inputs = []      # (Buf view, C_real) filled from NCHW tensors by run()
        self.finalized = False
        self.graph = None
        self._static_inputs = None

    # ---- buffers -----------------------------------------------------------------
    def new_buf(self, N, H, W, C_, name='', dtype=None):
        b = Buf(N, H, W, C_, name or f'b{len(self.bufs)}', dtype or self.dtype)
        self.bufs.append(b)
        return View(b, 0, C_)

    def _dev(self, t):
        t = t.to(self.device).contiguous()
        self.params.append(t)
        return t

    def add_input_nchw(self, N, C_, H, W, name='input', pad4=True, dtype=None):
        """Declare an NCHW fp32 input; returns its NHWC view in the plan's dtype (channels
        zero-padded to one 16-byte chunk -- 4 floats / 8 halves -- unless pad4=False, which keeps
        the buffer dense)."""
        dtype = dtype or self.dtype          # fp32 in a 16-bit plan: the image feeding the fp32 stem kernel
        h16 = dtype != torch.float32
        al = 8 if h16 else 4
        cp = (C_ + al - 1) // al * al if pad4 else C_
        v = self.new_buf(N, H, W, cp, name, dtype=dtype)
        slot = {'view': v, 'C': C_, 'src': None}
        self.inputs.append(slot)
        if h16:
            def fn(stream, slot=slot, v=v, C_=C_, cp=cp):
                check(_lib.lib().yv4_nchw_to_nhwc_h16(slot['src'].data_ptr(), v.buf.ptr(), v.N, C_, v.H, v.W, cp, 0,
                                                      cp - C_, self.dcode, stream), 'yv4_nchw_to_nhwc_h16')
        else:
            def fn(stream, slot=slot, v=v, C_=C_, cp=cp):
                src = slot['src']
                check(_lib.lib().yv4_nchw_to_nhwc(src.data_ptr(), v.buf.ptr(), v.N, C_, v.H, v.W, cp, 0,
                                                  cp - C_, stream), 'yv4_nchw_to_nhwc')
        self.ops.append(Op('to_nhwc', name, fn, nbytes=(4.0 + (2 if h16 else 4)) * N * C_ * H * W, info=dict(slot=slot)))
        return v

    # ---- ops ---------------------------------------------------------------------
    def conv(self, x, weight, s1, t1, act1=(0, 0.0), stride=1, pad=None, residual=None, s2=None,
             t2=None, act2=(0, 0.0), out=None, name='conv', tile=0, bn1=None, bn2=None, out_f32=False):
        """Fused conv launch.  weight: (Cout, Cin, KH, KW) torch tensor (any device).
        bn1 / bn2: optional (BatchNorm module, lo, hi) naming the BN channels that stage 1 /
        stage 2 of the epilogue were folded from (used by calibrate.py only).
        out_f32: in a 16-bit plan, store this conv's output in fp32 (pred maps feeding decode)."""
        Cout, Cin, KH, KW = weight.shape
        stem32 = self.h16 and x.buf.dtype == torch.float32      # fp32 image -> fp32 stem kernel -> 16-bit output
        if stem32:
            assert (Cin, KH, KW, stride) == (3, 3, 3, 1) and pad in (None, 1) and Cout <= 64 and residual is None \
                and s2 is None, f'{name}: an fp32 input in a 16-bit plan must feed the 3x3 stem'
        wp, cp = pack_conv_weight(weight, align=4 if stem32 else self.calign)
        if stem32:
            pass
        elif self.h16:
            wp = wp.to(self.dtype)
            assert x.buf.dtype == self.dtype, f'{name}: a 16-bit plan convolves {self.dtype} inputs, got {x.buf.dtype}'
            assert x.coff % 8 == 0 and x.cstride % 8 == 0, \
                f'{name}: 16-bit views need channel offsets / strides that are multiples of 8 ({x})'
        assert x.C == cp or (x.C == Cin and Cin % (4 if stem32 else self.calign) == 0), \
            f'{name}: input view has {x.C} channels, conv wants {Cin} (padded {cp})'
        if pad is None:
            pad = KH // 2
        Ho = (x.H + 2 * pad - KH) // stride + 1
        Wo = (x.W + 2 * pad - KW) // stride + 1
        if out is None:
            out = self.new_buf(x.N, Ho, Wo, Cout, name, dtype=torch.float32 if out_f32 else None)
        assert (out.N, out.H, out.W, out.C) == (x.N, Ho, Wo, Cout), f'{name}: output view mismatch {out} vs {(x.N, Ho, Wo, Cout)}'
        out_code = _DCODE[out.buf.dtype]
        if self.h16 and out_code != 0:
            assert out.coff % 8 == 0 and out.cstride % 8 == 0, f'{name}: 16-bit output view must be 8-channel aligned ({out})'
        if residual is not None:
            assert (residual.N, residual.H, residual.W, residual.C) == (x.N, Ho, Wo, Cout), f'{name}: residual mismatch'
        d = ConvDesc()
        d.N, d.H, d.W, d.Cin = x.N, x.H, x.W, cp
        d.Ho, d.Wo, d.Cout = Ho, Wo, Cout
        d.KH, d.KW, d.stride, d.pad = KH, KW, stride, pad
        d.x_cstride, d.x_coff = x.cstride, x.coff
        d.y_cstride, d.y_coff = out.cstride, out.coff
        d.r_cstride, d.r_coff = (residual.cstride, residual.coff) if residual is not None else (0, 0)
        d.act1, d.slope1 = act1
        d.act2, d.slope2 = act2 if s2 is not None else (0, 0.0)
        d.tile = tile
        # YV4_PLAN_NT=1: 16-bit inference plans write their outputs with non-temporal stores (ABI 7; the wide-tile kernels have
        # the form, the others ignore the flag; same bits).  Off by default: round 5 measured +0.6-1.3 % with it, round 6 on
        # other boxes -0.2 to -0.8 % (profiles/r06_ab_plan_nt.txt, r06_ab_wide_rd.txt) -- inside the box-to-box spread.
        if self.h16 and _PLAN_NT:
            d.flags = _lib.CONV_NT_OUT
        # everything the launch reads sits in one mutable record (calibrate.py swaps entries)
        L = dict(d=d, x=x.buf, y=out.buf, res=residual.buf if residual is not None else None,
                 w=self._dev(wp), s1=self._dev(s1.float()), t1=self._dev(t1.float()),
                 s2=self._dev(s2.float()) if s2 is not None else None,
                 t2=self._dev(t2.float()) if s2 is not None else None)
        self.params.append(d)

        if stem32:
            def fn(stream, L=L, out_code=out_code):
                check(_lib.lib().yv4_conv_stem_fwd(C.byref(L['d']), L['x'].ptr(), L['w'].data_ptr(), L['s1'].data_ptr(),
                                                   L['t1'].data_ptr(), L['y'].ptr(), out_code, stream),
                      'yv4_conv_stem_fwd')
        elif self.h16:
            # single-image plans split the K loop of the layers with too few tiles, as the fp32 plans below do
            ks = C.c_int(1)
            ws_bytes = 0
            if x.N == 1 and d.tile == _lib.TILE_AUTO and os.environ.get('YV4_SPLITK', '1') != '0':
                ws_bytes = int(_lib.lib().yv4_conv_h16_splitk_workspace(C.byref(d), C.byref(ks)))
            if ws_bytes:
                self._splitk_bytes = max(getattr(self, '_splitk_bytes', 0), ws_bytes)
                L['ksplit'] = ks.value

                def fn(stream, L=L, out_code=out_code):
                    ws = self._splitk_ws
                    check(_lib.lib().yv4_conv_bn_act_fwd_h16_splitk(
                        C.byref(L['d']), self.dcode, out_code, L['x'].ptr(), L['w'].data_ptr(), L['s1'].data_ptr(),
                        L['t1'].data_ptr(), L['s2'].data_ptr() if L['s2'] is not None else None,
                        L['t2'].data_ptr() if L['t2'] is not None else None,
                        L['res'].ptr() if L['res'] is not None else None, L['y'].ptr(), ws.data_ptr(), ws.numel() * 4,
                        stream), 'yv4_conv_bn_act_fwd_h16_splitk')
            else:
                def fn(stream, L=L, out_code=out_code):
                    check(_lib.lib().yv4_conv_bn_act_fwd_h16(
                        C.byref(L['d']), self.dcode, out_code, L['x'].ptr(), L['w'].data_ptr(), L['s1'].data_ptr(),
                        L['t1'].data_ptr(), L['s2'].data_ptr() if L['s2'] is not None else None,
                        L['t2'].data_ptr() if L['t2'] is not None else None,
                        L['res'].ptr() if L['res'] is not None else None, L['y'].ptr(), stream),
                        'yv4_conv_bn_act_fwd_h16')
        else:
            # single-image plans (the reference's benchmark protocol, tools/analysis_tools/benchmark.py:83-109): the
            # deep layers have too few output tiles for 256 CUs; split their K loop (yv4_conv_bn_act_fwd_splitk).
            # Batched plans never split, so their results stay bit-identical across batch sizes.
            ks = C.c_int(1)
            ws_bytes = 0
            if x.N == 1 and d.tile == _lib.TILE_AUTO and os.environ.get('YV4_SPLITK', '1') != '0':
                ws_bytes = int(_lib.lib().yv4_conv_splitk_workspace(C.byref(d), C.byref(ks)))
            if ws_bytes:
                self._splitk_bytes = max(getattr(self, '_splitk_bytes', 0), ws_bytes)
                L['ksplit'] = ks.value

                def fn(stream, L=L):
                    ws = self._splitk_ws
                    check(_lib.lib().yv4_conv_bn_act_fwd_splitk(
                        C.byref(L['d']), L['x'].ptr(), L['w'].data_ptr(), L['s1'].data_ptr(), L['t1'].data_ptr(),
                        L['s2'].data_ptr() if L['s2'] is not None else None,
                        L['t2'].data_ptr() if L['t2'] is not None else None,
                        L['res'].ptr() if L['res'] is not None else None, L['y'].ptr(), ws.data_ptr(),
                        ws.numel() * 4, stream), 'yv4_conv_bn_act_fwd_splitk')
            else:
                def fn(stream, L=L):
                    check(_lib.lib().yv4_conv_bn_act_fwd(
                        C.byref(L['d']), L['x'].ptr(), L['w'].data_ptr(), L['s1'].data_ptr(), L['t1'].data_ptr(),
                        L['s2'].data_ptr() if L['s2'] is not None else None,
                        L['t2'].data_ptr() if L['t2'] is not None else None,
                        L['res'].ptr() if L['res'] is not None else None, L['y'].ptr(), stream), 'yv4_conv_bn_act_fwd')
        M = x.N * Ho * Wo
        flops = 2.0 * M * Cout * KH * KW * Cin          # algorithmic: real Cin, not the padded one
        es = float(self.esize)
        ies = 4.0 if stem32 else es
        nbytes = ies * (x.N * x.H * x.W * Cin + Cout * KH * KW * Cin) + (4.0 if out_code == 0 else es) * M * Cout
        if residual is not None:
            nbytes += es * M * Cout
        self.ops.append(Op('conv', name, fn, flops, nbytes,
                           dict(Cin=Cin, Cout=Cout, k=KH, stride=stride, H=x.H, W=x.W, Ho=Ho, Wo=Wo,
                                N=x.N, desc=d, launch=L, out=out, bn1=bn1, bn2=bn2, stem32=stem32)))
        return out

    def spp(self, cat_view, C_, name='spp'):
        """cat_view: view of 4*C_ channels whose first C_ are filled; fills the other 3*C_."""
        assert cat_view.C == 4 * C_
        b = cat_view.buf

        if self.h16:
            def fn(stream, b=b, v=cat_view, C_=C_):
                check(_lib.lib().yv4_spp_pool_fwd_h16(b.ptr(), v.N, v.H, v.W, C_, v.cstride, v.coff, self.dcode,
                                                      stream), 'yv4_spp_pool_fwd_h16')
        else:
            def fn(stream, b=b, v=cat_view, C_=C_):
                check(_lib.lib().yv4_spp_pool_fwd(b.ptr(), v.N, v.H, v.W, C_, v.cstride, v.coff, stream),
                      'yv4_spp_pool_fwd')
        self.ops.append(Op('spp', name, fn, nbytes=self.esize * 4.0 * cat_view.N * cat_view.H * cat_view.W * C_))
        return cat_view

    def resample(self, src, dst, name='resample'):
        """Nearest resample src -> dst (same N and C; dst's H/W are the target size)."""
        assert src.N == dst.N and src.C == dst.C and src.buf.dtype == dst.buf.dtype
        # a 16-bit view with C % 8 == 0 is byte-identical to an fp32 view with C/2 channels
        k = 2 if src.buf.dtype != torch.float32 else 1
        if k == 2:
            assert all(v % 8 == 0 for v in (src.C, src.cstride, src.coff, dst.cstride, dst.coff)), \
                f'{name}: 16-bit resample needs 8-channel aligned views'

        def fn(stream, s=src, d=dst, k=k):
            check(_lib.lib().yv4_resample_nearest_fwd(s.buf.ptr(), d.buf.ptr(), s.N, s.H, s.W, d.H, d.W,
                                                      s.C // k, s.cstride // k, s.coff // k, d.cstride // k,
                                                      d.coff // k, stream), 'yv4_resample_nearest_fwd')
        self.ops.append(Op('resample', name, fn,
                           nbytes=(4.0 / k) * src.C * src.N * (src.H * src.W + dst.H * dst.W)))
        return dst

    def add_output_nchw(self, view, name='out'):
        """Materialise a view as a fresh NCHW tensor on every run; returns the slot index."""
        slot = {'view': view, 'dst': None}

        if view.buf.dtype != torch.float32:
            def fn(stream, slot=slot, v=view, code=_DCODE[view.buf.dtype]):
                check(_lib.lib().yv4_nhwc_to_nchw_h16(v.buf.ptr(), slot['dst'].data_ptr(), v.N, v.C, v.H, v.W,
                                                      v.cstride, v.coff, code, stream), 'yv4_nhwc_to_nchw_h16')
        else:
            def fn(stream, slot=slot, v=view):
                check(_lib.lib().yv4_nhwc_to_nchw(v.buf.ptr(), slot['dst'].data_ptr(), v.N, v.C, v.H, v.W,
                                                  v.cstride, v.coff, stream), 'yv4_nhwc_to_nchw')
        self.ops.append(Op('to_nchw', name, fn, nbytes=(4.0 + self.esize) * view.N * view.C * view.H * view.W))
        self.outputs = getattr(self, 'outputs', [])
        self.outputs.append(slot)
        return len(self.outputs) - 1

    def postprocess(self, pred_views, strides, base_anchors, num_classes, score_thr, iou_thr, max_per_img,
                    split_thr=10000, rescale=True, want_cls=False, nms_pre=-1, class_agnostic=False, v3=False,
                    conf_thr=-1.0):
        """decode+filter and per-image NMS over the head's NHWC pred maps.
        base_anchors: list (per level) of (A,4) float tensors.  Returns a dict of result
        tensors (allocated at finalize).  ``class_agnostic``: 5 attributes per box, the score is the
        objectness (one pseudo-class); ``nms_pre`` > 0: only the top-k boxes by objectness are
        candidates (yolocsp_head.py:349-360).  ``v3``: YOLOV3Head semantics (yolo_head.py:210-391): the
        v3 box decode, top-k PER LEVEL, objectness >= ``conf_thr``, class score > ``score_thr`` with the
        objectness multiplied in afterwards."""
        N = pred_views[0].N
        A = base_anchors[0].shape[0]
        kclasses = 0 if class_agnostic else num_classes          # what the kernels see
        num_classes = 1 if class_agnostic else num_classes       # columns of the score matrix
        attr = 5 + kclasses
        total = 0
        for v in pred_views:
            assert v.buf.dtype == torch.float32, 'decode reads fp32 pred maps (emit the head convs with out_f32)'
            assert v.buf.dtype == torch.float32, 'decode reads fp32 pred maps (emit the head convs with out_f32)'
            assert v.C == A * attr and v.coff == 0 and v.cstride == v.C, 'pred maps must be dense NHWC'
            total += v.H * v.W * A
        levels = (LevelDesc * len(pred_views))()
        res = dict(N=N, total_anchors=total, num_classes=num_classes, max_per_img=max_per_img,
                   key_cap=total * num_classes, want_cls=want_cls, rescale=rescale,
                   iou_thr=iou_thr, split_thr=split_thr, score_thr=score_thr, class_agnostic=class_agnostic)
        if v3:
            assert not class_agnostic, 'YOLOV3Head has no class-agnostic form'
            use_topk = nms_pre > 0 and any(v.H * v.W * A > nms_pre for v in pred_views)
        else:
            use_topk = 0 < nms_pre < total
        res['nms_pre'] = nms_pre if use_topk else -1
        res['v3'] = bool(v3)
        self.post = res
        self.params.append(levels)

        def alloc():
            dev = self.device
            res['boxes'] = torch.empty((N, total, 4), dtype=torch.float32, device=dev)
            res['conf'] = torch.empty((N, total), dtype=torch.float32, device=dev)
            res['cls'] = (torch.empty((N, total, num_classes), dtype=torch.float32, device=dev)
                          if want_cls and not class_agnostic else None)
            if use_topk:
                nbytes = (_lib.lib().yv4_conf_topk_levels_work(N, total, len(pred_views)) if v3
                          else _lib.lib().yv4_conf_topk_work(N, total))
                if nbytes == 0:
                    raise RuntimeError('yv4_conf_topk_work: batch * anchors too large for the top-k pre-selection')
                res['topk_work'] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                res['topk_keys'] = torch.zeros(N * (len(pred_views) if v3 else 1), dtype=torch.int64, device=dev)
            res['keys'] = torch.empty((N, res['key_cap']), dtype=torch.int64, device=dev)
            res['counts'] = torch.zeros(N, dtype=torch.int32, device=dev)
            res['max_coord'] = torch.zeros(N, dtype=torch.float32, device=dev)
            res['scale_factor'] = torch.ones((N, 4), dtype=torch.float32, device=dev)
            res['dets'] = torch.zeros((N, max_per_img, 5), dtype=torch.float32, device=dev)
            res['labels'] = torch.zeros((N, max_per_img), dtype=torch.int32, device=dev)
            res['index'] = torch.zeros((N, max_per_img), dtype=torch.int64, device=dev)
            res['count'] = torch.zeros(N, dtype=torch.int32, device=dev)
            for i, v in enumerate(pred_views):
                levels[i].pred = v.buf.ptr()
                levels[i].H, levels[i].W, levels[i].stride = v.H, v.W, int(strides[i])
                ba = base_anchors[i].float().cpu()
                for a in range(A):
                    for c in range(4):
                        levels[i].base_anchors[a][c] = float(ba[a, c])
        res['_alloc'] = alloc

        def reset(stream):
            check(_lib.lib().yv4_decode_reset(res['counts'].data_ptr(), res['max_coord'].data_ptr(), N, stream),
                  'yv4_decode_reset')

        def topk(stream):
            fn = _lib.lib().yv4_conf_topk_levels if v3 else _lib.lib().yv4_conf_topk
            check(fn(levels, len(pred_views), N, A, kclasses, int(nms_pre), res['topk_work'].data_ptr(),
                     res['topk_keys'].data_ptr(), stream), 'yv4_conf_topk')

        def decode_v3(stream):
            check(_lib.lib().yv4_decode_filter_v3(
                levels, len(pred_views), N, A, kclasses, float(score_thr), float(conf_thr),
                res['scale_factor'].data_ptr() if rescale else None, res['boxes'].data_ptr(),
                res['conf'].data_ptr(), res['cls'].data_ptr() if res['cls'] is not None else None,
                res['keys'].data_ptr(), res['key_cap'], res['counts'].data_ptr(),
                res['max_coord'].data_ptr(), res['topk_keys'].data_ptr() if use_topk else None, stream),
                'yv4_decode_filter_v3')

        def decode(stream):
            check(_lib.lib().yv4_decode_filter(
                levels, len(pred_views), N, A, kclasses, float(score_thr),
                res['scale_factor'].data_ptr() if rescale else None, res['boxes'].data_ptr(),
                res['conf'].data_ptr(), res['cls'].data_ptr() if res['cls'] is not None else None,
                res['keys'].data_ptr(), res['key_cap'], res['counts'].data_ptr(),
                res['max_coord'].data_ptr(), res['topk_keys'].data_ptr() if use_topk else None, stream),
                'yv4_decode_filter')

        def nms(stream):
            check(_lib.lib().yv4_nms_images(
                res['keys'].data_ptr(), res['key_cap'], res['counts'].data_ptr(), res['max_coord'].data_ptr(),
                res['boxes'].data_ptr(), total, None, 0, num_classes, N, float(iou_thr), max_per_img,
                int(split_thr), res['dets'].data_ptr(), res['labels'].data_ptr(), res['index'].data_ptr(),
                res['count'].data_ptr(), stream), 'yv4_nms_images')
        self.ops.append(Op('reset', 'decode_reset', reset))
        if use_topk:
            self.ops.append(Op('topk', 'conf_topk', topk))
        self.ops.append(Op('decode', 'decode_filter', decode_v3 if v3 else decode, nbytes=4.0 * N * total * attr))
        self.ops.append(Op('nms', 'nms_images', nms))
        return res

    # ---- lifecycle ---------------------------------------------------------------
    def hint_single_consumer(self, view):
        """The emitter's promise that exactly one later op reads this view's buffer (lets finalize() fuse the
        producer into that consumer and drop the buffer)."""
        view.buf.single_consumer = True
        return view

    def _fuse_stem_down(self):
        """16-bit plans: [NCHW fp32 image -> NHWC4 repack, fp32 3x3 stem, 3x3 / stride-2 conv] -> ONE launch of
        yv4_stem_down_fwd_h16 (stem_down_h16.hip) when the three ops follow each other, the widths are the kernel's
        (C1 in {16, 32}, C2 in {32, 64}) and the two intermediate buffers have no other reader.  The replaced ops
        stay reachable through ``info['parts']``; the launch reads their parameter records, so refreshed scale /
        shift / weight tensors are picked up.  YV4_STEM_FUSE=0 keeps the three launches."""
        if not self.h16 or os.environ.get('YV4_STEM_FUSE', '1') == '0':
            return
        for i in range(len(self.ops) - 2):
            o0, o1, o2 = self.ops[i:i + 3]
            if not (o0.kind == 'to_nhwc' and o1.kind == 'conv' and o2.kind == 'conv' and o1.info.get('stem32')):
                continue
            slot = o0.info.get('slot')
            L1, L2 = o1.info['launch'], o2.info['launch']
            d1, d2 = L1['d'], L2['d']
            v0, v1 = slot['view'], o1.info['out']
            ok = (L1['x'] is v0.buf and L2['x'] is v1.buf and getattr(v0.buf, 'single_consumer', False)
                  and getattr(v1.buf, 'single_consumer', False) and v1.coff == 0 and v1.cstride == v1.C
                  and (d2.KH, d2.KW, d2.stride, d2.pad) == (3, 3, 2, 1) and d1.Cout in (16, 32) and d2.Cin == d1.Cout
                  and d2.Cout in (32, 64) and L2['res'] is None and L2['s2'] is None
                  and o2.info['out'].buf.dtype == self.dtype and slot['C'] == 3)
            if not ok:
                continue
            out = o2.info['out']

            def fn(stream, slot=slot, L1=L1, L2=L2, d1=d1, d2=d2):
                check(_lib.lib().yv4_stem_down_fwd_h16(
                    self.dcode, slot['src'].data_ptr(), d1.N, d1.H, d1.W, L1['w'].data_ptr(), L1['s1'].data_ptr(),
                    L1['t1'].data_ptr(), d1.Cout, d1.act1, d1.slope1, L2['w'].data_ptr(), L2['s1'].data_ptr(),
                    L2['t1'].data_ptr(), d2.Cout, d2.act1, d2.slope1, L2['y'].ptr(), d2.y_cstride, d2.y_coff, stream),
                    'yv4_stem_down_fwd_h16')
            es = float(self.esize)
            nbytes = 4.0 * d1.N * 3 * d1.H * d1.W + es * d2.N * d2.Ho * d2.Wo * d2.Cout + 4.0 * d1.Cout * 27 + es * d2.Cout * 9 * d2.Cin
            info = dict(o2.info)
            info.update(Cin=3, H=d1.H, W=d1.W, stem32=True, fused='stem_down', parts=[o0, o1, o2], bn1=None, bn2=None)
            self.ops[i:i + 3] = [Op('conv', 'stem_down', fn, o1.flops + o2.flops, nbytes, info)]
            v0.buf.unused = v1.buf.unused = True          # never allocated
            return

    def finalize(self):
        self._fuse_stem_down()
        for b in self.bufs:
            if getattr(b, 'unused', False):
                continue
            b.tensor = torch.empty(b.numel, dtype=b.dtype, device=self.device)
        if getattr(self, '_splitk_bytes', 0):          # one workspace shared by the split-K layers (they run in turn)
            self._splitk_ws = torch.empty(self._splitk_bytes // 4, dtype=torch.float32, device=self.device)
        if getattr(self, 'post', None) is not None:
            self.post['_alloc']()
        self.finalized = True
        return self

    def autotune(self, candidates=None, reps=3):
        """Pick the conv tile per layer by timing the candidates on the plan's own buffers
        (device-to-device spread on MI355X is larger than the gap between tile shapes, so a static
        table is not robust).  Inputs must be set (run the plan once first).  Returns a dict
        layer-index -> chosen tile id."""
        assert self.finalized and self.graph is None and self.device.type == 'cuda'
        if self.h16:
            cands = candidates or (1, 2, 3)            # YV4_HTILE_128x128 / 128x64 / 64x64
        else:
            cands = candidates or (_lib.TILE_DMA_64x64, _lib.TILE_DMA_128x64, _lib.TILE_DMA_128x128)
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        chosen = {}
        for idx, op in enumerate(self.ops):
            if op.kind != 'conv':
                continue
            d = op.info['desc']
            if op.info.get('stem32'):
                continue
            auto = (_lib.lib().yv4_conv_h16_pick_tile if self.h16 else _lib.lib().yv4_conv_pick_tile)(C.byref(d))
            if auto not in cands:          # stem / generic path: nothing to choose from
                continue
            best, best_t = auto, None
            for t in cands:
                d.tile = t
                try:
                    op.fn(stream)          # warm-up + applicability check
                except _lib.Yv4Error:
                    continue
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    op.fn(stream)
                e1.record()
                torch.cuda.synchronize()
                dt = e0.elapsed_time(e1)
                if best_t is None or dt < best_t:
                    best, best_t = t, dt
            d.tile = best
            op.info['tile'] = best
            chosen[idx] = best
        return chosen

    def total_flops(self):
        return sum(o.flops for o in self.ops)

    def activation_bytes(self):
        return sum(b.numel * (4 if b.dtype == torch.float32 else 2) for b in self.bufs)

    def _launch_all(self, stream):
        for op in self.ops:
            op.fn(stream)

    def run(self, *inputs):
        """Replay the plan on the current stream.  inputs: NCHW fp32 CUDA tensors."""
        if not self.finalized:
            raise RuntimeError('Plan.finalize() has not been called')
        if self.device.type != 'cuda':
            raise RuntimeError('Plan.run needs a CUDA (ROCm) device: the HIP path has no CPU fallback')
        assert len(inputs) == len(self.inputs), f'plan takes {len(self.inputs)} inputs'
        for slot, t in zip(self.inputs, inputs):
            v = slot['view']
            if tuple(t.shape) != (v.N, slot['C'], v.H, v.W):
                raise ValueError(f'input shape {tuple(t.shape)} != planned {(v.N, slot["C"], v.H, v.W)}')
            if t.dtype != torch.float32 or not t.is_cuda:
                raise ValueError('inputs must be fp32 CUDA tensors')
            t = t.contiguous()
            if self.graph is not None:
                slot['src'].copy_(t)
            else:
                slot['src'] = t
        for slot in getattr(self, 'outputs', []):
            if self.graph is None or slot['dst'] is None:
                v = slot['view']
                slot['dst'] = torch.empty(v.shape_nchw, dtype=torch.float32, device=self.device)
        if self.graph is not None:
            self.graph.replay()
        else:
            self._launch_all(C.c_void_p(torch.cuda.current_stream().cuda_stream))
        return [s['dst'] for s in getattr(self, 'outputs', [])]

    def capture(self):
        """Capture the launch list into a hipGraph (torch.cuda.CUDAGraph is the plumbing).
        Inputs become static buffers that run() copies into."""
        assert self.finalized and self.graph is None
        for slot in self.inputs:
            v = slot['view']
            slot['src'] = torch.zeros((v.N, slot['C'], v.H, v.W), dtype=torch.float32, device=self.device)
        for slot in getattr(self, 'outputs', []):
            v = slot['view']
            slot['dst'] = torch.empty(v.shape_nchw, dtype=torch.float32, device=self.device)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):  # warm-up outside capture (function attributes, lazy init)
            self._launch_all(C.c_void_p(side.cuda_stream))
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._launch_all(C.c_void_p(torch.cuda.current_stream().cuda_stream))
        self.graph = g
        return self
