"""16-bit operand convolution (fp16 / bf16 in, fp32 accumulate) through the C ABI vs a float64
convolution of the SAME rounded operands + the same epilogue in float64.

Tolerance: the only differences are fp32 accumulation order (~1e-6 relative to the sum of
|products|) and the final rounding of the output to the 16-bit type (half an ulp: 2^-9 relative for
bf16, 2^-12 for fp16); fp32 outputs (`out_dtype = YV4_F32`) are held to 2e-5."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import mmdet_yolov4_amd as pkg
from mmdet_yolov4_amd import _lib as L
from oracle import yolov4_oracle as O

pytestmark = pytest.mark.gpu


HT = dict(t128x128=1, t128x64=2, t64x64=3)


def _h16_conv(dev, dtype, N, H, W, Cin, Cout, k, stride, pad, act, tile, residual=False, two_stage=False, x_off=0,
              y_off=0, out_f32=False, seed=0, raw=False, splitk=False, nt=False):
    g = torch.Generator().manual_seed(seed)
    cp = (Cin + 7) // 8 * 8
    x = torch.randn(N, H, W, Cin, generator=g).to(dtype)
    w = (torch.randn(Cout, k, k, Cin, generator=g) * (1.0 / (Cin * k * k)) ** 0.5).to(dtype)
    s1 = torch.rand(Cout, generator=g) + 0.5
    t1 = torch.randn(Cout, generator=g) * 0.1
    s2 = torch.rand(Cout, generator=g) + 0.5
    t2 = torch.randn(Cout, generator=g) * 0.1
    Ho = (H + 2 * pad - k) // stride + 1
    Wo = (W + 2 * pad - k) // stride + 1
    res = torch.randn(N, Ho, Wo, Cout, generator=g).to(dtype) if residual else None
    acts = {0: lambda v: v, 1: lambda v: v * torch.tanh(F.softplus(v)), 2: lambda v: F.leaky_relu(v, 0.1),
            3: lambda v: v * torch.sigmoid(v)}
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), None, stride, pad)
    ref = acts[act](ref * s1.double()[None, :, None, None] + t1.double()[None, :, None, None])
    if residual:
        ref = ref + res.double().permute(0, 3, 1, 2)
    if two_stage:
        ref = acts[act](ref * s2.double()[None, :, None, None] + t2.double()[None, :, None, None])
    ref = ref.permute(0, 2, 3, 1)                                   # NHWC

    xs = cp + x_off + (8 if x_off else 0)
    xbuf = torch.zeros(N, H, W, xs, dtype=dtype, device=dev)
    xbuf[..., x_off:x_off + Cin] = x.to(dev)
    wbuf = torch.zeros(Cout, k, k, cp, dtype=dtype, device=dev)
    wbuf[..., :Cin] = w.to(dev)
    odt = torch.float32 if out_f32 else dtype
    ys = Cout + y_off + (8 if y_off else 0)
    if not out_f32 and ys % 8:
        ys = (ys + 7) // 8 * 8
    ybuf = torch.full((N, Ho, Wo, ys), 7.0, dtype=odt, device=dev)
    rbuf = res.to(dev).contiguous() if residual else None
    d = L.ConvDesc()
    d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = N, H, W, cp, Ho, Wo, Cout
    d.KH, d.KW, d.stride, d.pad = k, k, stride, pad
    d.x_cstride, d.x_coff, d.y_cstride, d.y_coff = xs, x_off, ys, y_off
    d.r_cstride, d.r_coff = Cout, 0
    d.act1, d.act2, d.slope1, d.slope2 = act, act if two_stage else 0, 0.1, 0.1
    d.tile = tile
    d.flags = pkg._lib.CONV_NT_OUT if nt else 0
    dev_f = lambda t: t.to(dev).float().contiguous()
    s1d, t1d, s2d, t2d = dev_f(s1), dev_f(t1), dev_f(s2), dev_f(t2)
    dcode, ocode = 1 if dtype == torch.float16 else 2, 0 if out_f32 else (1 if dtype == torch.float16 else 2)
    if splitk:
        ks = C.c_int(0)
        nbytes = int(L.lib().yv4_conv_h16_splitk_workspace(C.byref(d), C.byref(ks)))
        assert (ks.value > 1) == (nbytes > 0)
        _h16_conv.last_ksplit = ks.value
        ws = torch.empty(max(nbytes, 16) // 4, dtype=torch.float32, device=dev)
        rc = L.lib().yv4_conv_bn_act_fwd_h16_splitk(C.byref(d), dcode, ocode, xbuf.data_ptr(), wbuf.data_ptr(),
                                                    s1d.data_ptr(), t1d.data_ptr(), s2d.data_ptr() if two_stage else None,
                                                    t2d.data_ptr() if two_stage else None,
                                                    rbuf.data_ptr() if residual else None, ybuf.data_ptr(), ws.data_ptr(),
                                                    ws.numel() * 4, torch.cuda.current_stream().cuda_stream)
    else:
        rc = L.lib().yv4_conv_bn_act_fwd_h16(C.byref(d), dcode, ocode, xbuf.data_ptr(), wbuf.data_ptr(),
                                             s1d.data_ptr(), t1d.data_ptr(), s2d.data_ptr() if two_stage else None,
                                             t2d.data_ptr() if two_stage else None,
                                             rbuf.data_ptr() if residual else None, ybuf.data_ptr(),
                                             torch.cuda.current_stream().cuda_stream)
    L.check(rc, 'yv4_conv_bn_act_fwd_h16')
    torch.cuda.synchronize()
    if raw:
        return ybuf[..., y_off:y_off + Cout].clone()
    got = ybuf[..., y_off:y_off + Cout].double().cpu()
    assert bool((ybuf[..., :y_off] == 7.0).all()) and bool((ybuf[..., y_off + Cout:] == 7.0).all())
    ulp = 2e-5 if out_f32 else (2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11)
    err = (got - ref).abs()
    bound = ulp * ref.abs() + 3e-5 + (0 if out_f32 else ulp * 1e-2)
    bad = err > bound
    assert not bool(bad.any()), (f'{int(bad.sum())} of {bad.numel()} outside tolerance; worst '
                                 f'{float((err / (ref.abs() + 1e-3)).max()):.3e} rel')


SHAPES = [
    # N, H, W, Cin, Cout, k, stride, pad
    (2, 19, 19, 64, 128, 3, 1, 1),     # uniform-tap path, ragged M (722 rows)
    (1, 16, 20, 64, 255, 1, 1, 0),     # head-like: Cout 255 not a tile multiple
    (2, 17, 23, 128, 64, 3, 2, 1),     # stride 2, odd sizes
    (1, 12, 12, 32, 40, 3, 1, 1),      # Cin % 64 != 0 -> per-lane tap decode, K tail
    (2, 32, 32, 3, 32, 3, 1, 1),       # stem: Cin 3 padded to 8, K = 72
    (1, 24, 24, 3, 16, 6, 2, 2),       # Focus conv k=6 s=2 p=2
    (1, 10, 10, 192, 96, 1, 1, 0),     # 1x1, K = 3 slices
]


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('tile', [0, 1, 2, 3])
@pytest.mark.parametrize('shape', SHAPES)
def test_h16_conv_shapes_and_tiles(gpu_device, dtype, tile, shape):
    _h16_conv(gpu_device, dtype, *shape, act=1, tile=tile)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('act', [0, 1, 2, 3])
def test_h16_conv_epilogues(gpu_device, dtype, act):
    _h16_conv(gpu_device, dtype, 2, 13, 13, 64, 72, 3, 1, 1, act, 0, residual=True, two_stage=True, x_off=8, y_off=16)
    _h16_conv(gpu_device, dtype, 1, 9, 11, 64, 255, 1, 1, 0, act, 0, out_f32=True)           # pred maps in fp32
    _h16_conv(gpu_device, dtype, 1, 9, 11, 40, 64, 3, 1, 1, act, 3, out_f32=True, y_off=4)


SPLITK_SHAPES = [
    # N, H, W, Cin, Cout, k, stride, pad : single-image layers of YOLOv4-L 608 (the batch-1 protocol) + awkward ones
    (1, 19, 19, 512, 512, 3, 1, 1),     # 48 tiles, 72 slices: 16 ways -> 15 splits of 5 slices, the last of 2
    (1, 19, 19, 1024, 512, 1, 1, 0),    # 1x1, 16 slices
    (1, 38, 38, 256, 256, 3, 1, 1),     # 92 tiles, 36 slices
    (1, 38, 38, 128, 256, 3, 2, 1),     # stride 2
    (1, 19, 19, 1024, 255, 1, 1, 0),    # head: Cout 255 (slab rows padded to 256)
    (1, 11, 13, 192, 72, 3, 1, 1),      # ragged M, Cout tail inside an 8-channel group
    (1, 76, 76, 128, 128, 3, 1, 1),     # 182 tiles: two ways
]


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('shape', SPLITK_SHAPES)
def test_h16_splitk_conv(gpu_device, dtype, shape):
    """yv4_conv_bn_act_fwd_h16_splitk (single-image plans, tools/analysis_tools/benchmark.py:83-109): within the same
    bound of the fp64 convolution as the unsplit tiles (fp32 partial slabs added in slab order, one rounding at the end),
    run-to-run bit-identical, residual + two-stage epilogue + views; fp32 output for the head shape."""
    head = shape[4] == 255
    kw = dict(out_f32=True) if head else dict(residual=True, two_stage=True, x_off=8, y_off=16)
    _h16_conv(gpu_device, dtype, *shape, act=1, tile=0, splitk=True, **kw)
    assert _h16_conv.last_ksplit > 1
    a = _h16_conv(gpu_device, dtype, *shape, act=1, tile=0, splitk=True, raw=True, **kw)
    b = _h16_conv(gpu_device, dtype, *shape, act=1, tile=0, splitk=True, raw=True, **kw)
    assert torch.equal(a, b)


def test_h16_splitk_forwards_when_not_split(gpu_device):
    """Enough tiles already, or Cin % 64 != 0: ksplit = 1, no workspace, the call is the plain entry."""
    for shape in [(8, 38, 38, 256, 256, 3, 1, 1), (1, 19, 19, 96, 64, 3, 1, 1)]:
        _h16_conv(gpu_device, torch.bfloat16, *shape, act=1, tile=0, splitk=True)
        assert _h16_conv.last_ksplit == 1


PP3_SHAPES = [
    # N, H, W, Cin, Cout  (3x3, stride 1, pad 1): the domain of conv3x3_pp_h16.hip, tile 4 (256 pixels x 128 channels)
    (2, 19, 19, 64, 128),      # two images in three M tiles: image borders inside a tile, ragged last tile
    (3, 7, 5, 128, 64),        # map narrower than a fragment row group: many left / right borders per tile; half-empty columns
    (1, 38, 38, 64, 192),      # Cout not a multiple of 128: half-empty last column tile
    (2, 16, 16, 192, 72),      # three channel chunks, Cout tail inside a 32-column group
    (1, 1, 300, 64, 64),       # one image row: every kh = 0 / 2 tap is padding
    (5, 3, 3, 64, 64),         # tiny images: almost every tap is masked somewhere
    (24, 38, 38, 64, 256),     # 136 x 2 = 272 tiles > 256 CUs: some workgroups walk two tiles, the ring runs on across them
    (70, 19, 19, 128, 384),    # 99 x 3 = 297 tiles, ragged last row tile, three column tiles, two chunks
]


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('tile', [4])
@pytest.mark.parametrize('shape', PP3_SHAPES)
def test_h16_pp3x3_kernel_shapes(gpu_device, dtype, tile, shape):
    N, H, W, Cin, Cout = shape
    _h16_conv(gpu_device, dtype, N, H, W, Cin, Cout, 3, 1, 1, act=1, tile=tile)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('tile', [4])
@pytest.mark.parametrize('act', [0, 1, 2, 3])
def test_h16_pp3x3_kernel_epilogues(gpu_device, dtype, tile, act):
    """Residual + two-stage epilogue + channel-offset views on both sides through the persistent 3x3 kernels."""
    _h16_conv(gpu_device, dtype, 2, 13, 13, 64, 72, 3, 1, 1, act, tile, residual=True, two_stage=True, x_off=8, y_off=16)
    _h16_conv(gpu_device, dtype, 1, 20, 9, 128, 128, 3, 1, 1, act, tile, residual=True, x_off=16)
    _h16_conv(gpu_device, dtype, 30, 38, 38, 64, 128, 3, 1, 1, act, tile, residual=True, two_stage=True, y_off=8)   # 170 tiles


W3_TILES = [5, 13, 21, 29, 37, 45, 53, 61]      # YV4_HTILE_W3x3 (shape chosen by the cost model) and YV4_HTILE_W3x3_SHAPE(0..6)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('tile', W3_TILES)
@pytest.mark.parametrize('shape', [s for s in PP3_SHAPES if s[4] % 16 == 0] + [(33, 19, 19, 128, 512), (2, 76, 76, 128, 128)])
def test_h16_wide3x3_kernel_shapes(gpu_device, dtype, tile, shape):
    """conv3x3_wide_h16.hip (16x16x32 MFMAs, wave tiles of 16 PT pixels x 64 channels, seven workgroup tile shapes): the
    ping-pong kernel's shape list -- image borders inside a tile, maps narrower than a fragment group, one-row images,
    ragged row / column tiles, workgroups that walk several tiles -- against float64."""
    N, H, W, Cin, Cout = shape
    _h16_conv(gpu_device, dtype, N, H, W, Cin, Cout, 3, 1, 1, act=1, tile=tile)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('tile', W3_TILES)
@pytest.mark.parametrize('act', [0, 1, 2, 3])
def test_h16_wide3x3_kernel_epilogues(gpu_device, dtype, tile, act):
    _h16_conv(gpu_device, dtype, 2, 13, 13, 64, 80, 3, 1, 1, act, tile, residual=True, two_stage=True, x_off=8, y_off=16)
    _h16_conv(gpu_device, dtype, 1, 20, 9, 128, 128, 3, 1, 1, act, tile, residual=True, x_off=16)
    _h16_conv(gpu_device, dtype, 30, 38, 38, 64, 128, 3, 1, 1, act, tile, residual=True, two_stage=True, y_off=8)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_wide3x3_matches_generic_bitwise(gpu_device, dtype):
    """The wide-tile kernel sums a 64-deep K slice with two 16x16x32 MFMAs where the other tiles use four 32x32x16 ones;
    same chunk-major K order, same epilogue expressions -- and, measured, the same BITS (the matrix pipe's accumulation
    does not depend on how the 64 products of a slice are grouped into instructions), so a plan may pick it by batch size
    like the ping-pong kernel (bench.py's batch-32 vs batch-2 output check stays bit-exact in 16 bits)."""
    for shape, kw in [((2, 19, 19, 128, 128), dict(residual=True, two_stage=True)), ((3, 38, 38, 64, 192), dict()),
                      ((9, 38, 38, 256, 256), dict(residual=True)), ((5, 19, 19, 512, 512), dict())]:
        N, H, W, Cin, Cout = shape
        outs = [_h16_conv(gpu_device, dtype, N, H, W, Cin, Cout, 3, 1, 1, act=1, tile=t, raw=True, **kw) for t in (2, 4, 5, 13, 21, 29, 37, 45, 53, 61)]
        for o in outs[1:]:
            assert torch.equal(outs[0], o)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_nontemporal_output_flag_gives_the_same_bits(gpu_device, dtype):
    """yv4_conv_desc.flags & YV4_CONV_NT_OUT (ABI 7, what 16-bit inference plans set): the wide-tile kernels store with
    non-temporal instructions, every other kernel ignores the flag -- the output is the same either way."""
    for shape, k, stride, pad, tiles, kw in [((9, 38, 38, 256, 256), 3, 1, 1, (5, 13, 21, 37, 53, 61), dict(residual=True)),
                                             ((3, 38, 38, 128, 256), 3, 2, 1, (8, 24), dict()),
                                             ((2, 19, 19, 256, 256), 1, 1, 0, (8, 2, 6), dict(two_stage=True))]:
        N, H, W, Cin, Cout = shape
        for t in tiles:
            a = _h16_conv(gpu_device, dtype, N, H, W, Cin, Cout, k, stride, pad, act=1, tile=t, raw=True, **kw)
            b = _h16_conv(gpu_device, dtype, N, H, W, Cin, Cout, k, stride, pad, act=1, tile=t, raw=True, nt=True, **kw)
            assert torch.equal(a, b), (shape, t)


WIDE_TILES = [8, 16, 24, 32, 40, 48]     # YV4_HTILE_WIDE (shape by the cost model) and YV4_HTILE_WIDE_SHAPE(0..4)
WIDE_SHAPES = [
    # N, H, W, Cin, Cout, k, stride, pad: the general wide-tile kernel's domain (Cin % 64 == 0, Cout % 16 == 0)
    (2, 17, 23, 128, 64, 3, 2, 1),     # stride 2, odd sizes, ragged M
    (3, 38, 38, 64, 128, 3, 2, 1),     # stride 2 downscale, image borders inside a tile
    (2, 19, 19, 512, 256, 1, 1, 0),    # deep 1x1
    (1, 16, 20, 64, 240, 1, 1, 0),     # Cout not a multiple of 64: ragged channel groups
    (2, 19, 19, 64, 128, 3, 1, 1),     # 3x3 stride 1 as a general gather
    (1, 12, 12, 64, 64, 5, 1, 2),      # 25 taps
    (33, 19, 19, 256, 512, 1, 1, 0),   # several tiles per workgroup column
    (40, 38, 38, 128, 256, 3, 2, 1),   # more tiles than CUs for the small shapes: the issue side crosses tiles
]


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('tile', WIDE_TILES)
@pytest.mark.parametrize('shape', WIDE_SHAPES)
def test_h16_wide_general_kernel_shapes(gpu_device, dtype, tile, shape):
    """conv_wide_h16.hip: the wide wave tiles as a general implicit GEMM (one (chunk, tap) pixel tile per K tile, padding
    and borders as out-of-range DMA offsets) against float64, every workgroup tile shape."""
    _h16_conv(gpu_device, dtype, *shape, act=1, tile=tile)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('tile', WIDE_TILES)
@pytest.mark.parametrize('act', [0, 1, 2, 3])
def test_h16_wide_general_kernel_epilogues(gpu_device, dtype, tile, act):
    _h16_conv(gpu_device, dtype, 2, 13, 13, 64, 80, 3, 2, 1, act, tile, residual=True, two_stage=True, x_off=8, y_off=16)
    _h16_conv(gpu_device, dtype, 1, 20, 9, 128, 128, 1, 1, 0, act, tile, residual=True, x_off=16)
    _h16_conv(gpu_device, dtype, 30, 38, 38, 64, 128, 3, 2, 1, act, tile, residual=True, two_stage=True, y_off=8)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_wide_general_matches_generic_bitwise(gpu_device, dtype):
    """Same K order (chunk-major, taps inside) and epilogue expressions as the generic tiles: the same bits."""
    for shape, kw in [((2, 19, 19, 128, 128, 3, 2, 1), dict(residual=True, two_stage=True)), ((3, 38, 38, 512, 256, 1, 1, 0), dict()),
                      ((9, 38, 38, 256, 256, 1, 1, 0), dict(residual=True)), ((2, 38, 38, 64, 192, 3, 1, 1), dict())]:
        outs = [_h16_conv(gpu_device, dtype, *shape, act=1, tile=t, raw=True, **kw) for t in (2, 8, 16, 24, 32, 40, 48)]
        for o in outs[1:]:
            assert torch.equal(outs[0], o)


def test_h16_wide_general_kernel_is_refused_outside_its_domain(gpu_device):
    for shape in [(1, 8, 8, 32, 64, 3, 1, 1), (1, 8, 8, 64, 72, 3, 1, 1), (1, 8, 8, 64, 32, 1, 1, 0)]:
        with pytest.raises(L.Yv4Error):
            _h16_conv(gpu_device, torch.bfloat16, *shape, act=1, tile=8)
    with pytest.raises(L.Yv4Error):
        _h16_conv(gpu_device, torch.bfloat16, 1, 9, 11, 64, 64, 3, 1, 1, act=1, tile=8, out_f32=True, y_off=4)


def test_h16_wide3x3_kernel_is_refused_outside_its_domain(gpu_device):
    for shape in [(1, 8, 8, 64, 64, 1, 1, 0), (1, 8, 8, 64, 64, 3, 2, 1), (1, 8, 8, 32, 64, 3, 1, 1), (1, 8, 8, 64, 72, 3, 1, 1)]:
        with pytest.raises(L.Yv4Error):
            _h16_conv(gpu_device, torch.bfloat16, *shape, act=1, tile=5)
    with pytest.raises(L.Yv4Error):
        _h16_conv(gpu_device, torch.bfloat16, 1, 9, 11, 64, 64, 3, 1, 1, act=1, tile=5, out_f32=True, y_off=4)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_pp3x3_matches_generic_bitwise(gpu_device, dtype):
    """Same K order (chunk-major, taps inside, four 16-deep MFMA steps per tap) and the same epilogue expressions on the
    same fp32 accumulators: the persistent 3x3 kernel and the generic tiles return the same BITS, so a plan may pick
    either by batch size (bench.py's batch-32 vs batch-2 output check, the plans' batch-composition independence)."""
    for shape, kw in [((2, 19, 19, 128, 128), dict(residual=True, two_stage=True)), ((3, 38, 38, 64, 192), dict()),
                      ((9, 38, 38, 256, 256), dict(residual=True))]:
        N, H, W, Cin, Cout = shape
        outs = [_h16_conv(gpu_device, dtype, N, H, W, Cin, Cout, 3, 1, 1, act=1, tile=t, raw=True, **kw) for t in (4, 1, 2, 3)]
        for o in outs[1:]:
            assert torch.equal(outs[0], o)


def test_h16_pp3x3_kernel_is_refused_outside_its_domain(gpu_device):
    for shape in [(1, 8, 8, 64, 64, 1, 1, 0), (1, 8, 8, 64, 64, 3, 2, 1), (1, 8, 8, 32, 64, 3, 1, 1)]:
        with pytest.raises(L.Yv4Error):
            _h16_conv(gpu_device, torch.bfloat16, *shape, act=1, tile=4)
    with pytest.raises(L.Yv4Error):                                      # fp32 output (pred maps) stays on the generic tiles
        _h16_conv(gpu_device, torch.bfloat16, 1, 9, 11, 64, 64, 3, 1, 1, act=1, tile=4, out_f32=True, y_off=4)


def test_h16_pp3x3_auto_choice(gpu_device):
    """The auto tile choice takes the persistent 3x3 kernel for the stride-1 3x3 layers with >= 128 input channels whose
    tiles fill the chip; a batch-2 plan's layers and the few-channel layers keep the generic tiles (same bits either way:
    test_pp3x3_matches_generic_bitwise)."""
    d = L.ConvDesc()
    d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = 32, 38, 38, 256, 38, 38, 256
    d.KH = d.KW = 3
    d.stride, d.pad = 1, 1
    d.x_cstride, d.y_cstride, d.r_cstride = 256, 256, 256
    assert L.lib().yv4_conv_h16_pick_tile(C.byref(d)) == 5           # 46 208 outputs per CU: the wide-tile kernel (round 4)
    d.N, d.H, d.W, d.Ho, d.Wo, d.Cin, d.Cout = 32, 19, 19, 19, 19, 512, 512
    d.x_cstride, d.y_cstride = 512, 512
    assert L.lib().yv4_conv_h16_pick_tile(C.byref(d)) == 4           # 184 tiles: one round at 72 %
    d.N = 2
    assert L.lib().yv4_conv_h16_pick_tile(C.byref(d)) in (1, 2, 3)   # 12 tiles
    d.N, d.H, d.W, d.Ho, d.Wo, d.Cin, d.Cout = 32, 152, 152, 152, 152, 64, 64
    d.x_cstride, d.y_cstride = 64, 64
    assert L.lib().yv4_conv_h16_pick_tile(C.byref(d)) != 4           # 64 input channels: the few-channel kernel's layer
    _h16_conv(gpu_device, torch.bfloat16, 32, 38, 38, 128, 128, 3, 1, 1, act=1, tile=0)     # auto -> persistent kernel
    _h16_conv(gpu_device, torch.bfloat16, 32, 38, 38, 256, 256, 3, 1, 1, act=1, tile=0)     # auto -> wide-tile kernel


WS_SHAPES = [
    # N, H, W, Cin, Cout  (1x1, stride 1): the domain of conv1x1_ws_h16.hip, tile 6
    (2, 19, 19, 128, 128),     # ragged last strip (722 pixels), one slab
    (1, 76, 76, 64, 64),       # more strips than one round of waves: Cin = one stage
    (3, 40, 33, 256, 128),     # four stages per strip, slab of 64 columns x 2
    (1, 31, 7, 256, 256),      # four column slabs reading the same strips
    (2, 64, 64, 64, 32),       # narrowest slab (one 32-column tile)
    (1, 5, 5, 128, 96),        # fewer pixels than one strip; Cout 96 = a full 64-column slab and a half-empty one
    (4, 152, 152, 64, 64),     # 2 888 strips: every wave walks its ring across several strips
    (2, 52, 52, 32, 16),       # the v4s widths: Cin 32 laid out as 64 (zero chunks), Cout 16 = half a column tile
    (1, 40, 40, 96, 48),       # the v4m widths: Cin 96 laid out as 128
    (1, 33, 33, 192, 96),      # Cin 192 laid out as 256
]


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('shape', WS_SHAPES)
def test_h16_conv1x1_ws_kernel_shapes(gpu_device, dtype, shape):
    N, H, W, Cin, Cout = shape
    _h16_conv(gpu_device, dtype, N, H, W, Cin, Cout, 1, 1, 0, act=1, tile=6)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('act', [0, 1, 2, 3])
def test_h16_conv1x1_ws_kernel_epilogues(gpu_device, dtype, act):
    """Two-stage epilogue + channel-offset views on both sides through the weight-stationary kernel."""
    _h16_conv(gpu_device, dtype, 2, 23, 17, 128, 64, 1, 1, 0, act, 6, two_stage=True, x_off=8, y_off=16)
    _h16_conv(gpu_device, dtype, 1, 50, 50, 64, 128, 1, 1, 0, act, 6, x_off=64, y_off=3 * 8)


def test_h16_conv1x1_ws_kernel_is_refused_outside_its_domain(gpu_device):
    for shape in [(1, 8, 8, 64, 64, 3, 1, 1), (1, 8, 8, 320, 64, 1, 1, 0), (1, 8, 8, 512, 64, 1, 1, 0),
                  (1, 8, 8, 64, 8, 1, 1, 0)]:
        with pytest.raises(L.Yv4Error):
            _h16_conv(gpu_device, torch.bfloat16, *shape, act=1, tile=6)
    with pytest.raises(L.Yv4Error):
        _h16_conv(gpu_device, torch.bfloat16, 1, 8, 8, 64, 64, 1, 1, 0, act=0, tile=6, residual=True, out_f32=True)   # residual: 16-bit outputs
    with pytest.raises(L.Yv4Error):
        _h16_conv(gpu_device, torch.bfloat16, 1, 8, 8, 64, 63, 1, 1, 0, act=1, tile=6)          # odd Cout needs fp32 output


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('shape', [(2, 19, 19, 128, 128), (1, 76, 76, 64, 64), (3, 40, 33, 256, 128), (2, 64, 64, 32, 64),
                                   (1, 5, 5, 128, 96), (2, 52, 52, 32, 16)])
def test_h16_conv1x1_ws_kernel_residual(gpu_device, dtype, shape):
    """The residual of the weight-stationary kernel (the data gradient of a Bottleneck's first conv adds the shortcut's
    gradient): read as channel-pair dwords in the store layout, exchanged between the lanes of a pair, added in fp32
    before the one rounding -- plain, behind an activation, and in front of a second epilogue stage."""
    N, H, W, Cin, Cout = shape
    _h16_conv(gpu_device, dtype, N, H, W, Cin, Cout, 1, 1, 0, act=0, tile=6, residual=True)
    _h16_conv(gpu_device, dtype, N, H, W, Cin, Cout, 1, 1, 0, act=1, tile=6, residual=True)
    _h16_conv(gpu_device, dtype, N, H, W, Cin, Cout, 1, 1, 0, act=2, tile=6, residual=True, two_stage=True)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_h16_conv1x1_ws_kernel_fp32_pred_maps(gpu_device, dtype):
    """The head convs: fp32 output, 255 channels (odd pitch), bias as the shift -- through the weight-stationary kernel
    with two and four column slabs."""
    _h16_conv(gpu_device, dtype, 2, 26, 26, 128, 255, 1, 1, 0, act=0, tile=6, out_f32=True)
    _h16_conv(gpu_device, dtype, 1, 19, 23, 256, 255, 1, 1, 0, act=0, tile=6, out_f32=True)
    _h16_conv(gpu_device, dtype, 1, 9, 11, 64, 18, 1, 1, 0, act=1, tile=6, out_f32=True, y_off=4)


S3_SHAPES = [
    # N, H, W, Cin, Cout  (3x3, stride 1, pad 1): the domain of conv3x3_small_h16.hip, tile 7
    (2, 32, 32, 16, 32),       # 2 x 2 full tiles per image, the v4s widths
    (1, 40, 56, 32, 64),       # ragged tiles on both axes, the v4l widths
    (3, 19, 23, 64, 64),       # odd sizes, four chunks per pixel pair ... Cin 64
    (1, 16, 300, 32, 32),      # one tile row: every top / bottom tap row is padding
    (5, 5, 5, 16, 16),         # images smaller than a tile, Cout 16 = half a column tile
    (2, 304, 304, 32, 64),     # 722 tiles: more than one round of the persistent grid, double-buffered prefetch
]


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('shape', S3_SHAPES)
def test_h16_conv3x3_small_kernel_shapes(gpu_device, dtype, shape):
    N, H, W, Cin, Cout = shape
    _h16_conv(gpu_device, dtype, N, H, W, Cin, Cout, 3, 1, 1, act=1, tile=7)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('act', [0, 1, 2, 3])
def test_h16_conv3x3_small_kernel_epilogues(gpu_device, dtype, act):
    """Residual + two-stage epilogue + channel-offset views on both sides through the few-channel 3x3 kernel."""
    _h16_conv(gpu_device, dtype, 2, 23, 37, 32, 64, 3, 1, 1, act, 7, residual=True, two_stage=True, x_off=8, y_off=16)
    _h16_conv(gpu_device, dtype, 1, 50, 18, 16, 32, 3, 1, 1, act, 7, residual=True, x_off=16)
    _h16_conv(gpu_device, dtype, 1, 20, 20, 64, 48, 3, 1, 1, act, 7, y_off=8)


def test_h16_conv3x3_small_kernel_is_refused_outside_its_domain(gpu_device):
    for shape in [(1, 8, 8, 32, 64, 1, 1, 0), (1, 8, 8, 32, 64, 3, 2, 1), (1, 8, 8, 128, 64, 3, 1, 1), (1, 8, 8, 32, 128, 3, 1, 1),
                  (1, 8, 8, 24, 32, 3, 1, 1)]:
        with pytest.raises(L.Yv4Error):
            _h16_conv(gpu_device, torch.bfloat16, *shape, act=1, tile=7)
    with pytest.raises(L.Yv4Error):
        _h16_conv(gpu_device, torch.bfloat16, 1, 8, 8, 32, 64, 3, 1, 1, act=1, tile=7, out_f32=True)


def test_h16_conv_big_k(gpu_device):
    _h16_conv(gpu_device, torch.bfloat16, 1, 8, 8, 512, 64, 3, 1, 1, 1, 1)       # K = 4608


def test_h16_conv_rejects_bad_arguments(gpu_device):
    d = L.ConvDesc()
    d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = 1, 8, 8, 12, 8, 8, 16
    d.KH = d.KW = 1
    d.stride, d.pad = 1, 0
    d.x_cstride, d.y_cstride = 12, 16
    t = torch.zeros(4096, device=gpu_device)
    f = L.lib().yv4_conv_bn_act_fwd_h16
    p = t.data_ptr()
    assert f(C.byref(d), 2, 2, p, p, p, p, None, None, None, p, None) != 0          # Cin % 8
    d.Cin = d.x_cstride = 16
    assert f(C.byref(d), 0, 0, p, p, p, p, None, None, None, p, None) != 0          # fp32 is the other entry
    assert f(C.byref(d), 2, 1, p, p, p, p, None, None, None, p, None) != 0          # bf16 in, fp16 out
    assert f(C.byref(d), 2, 2, p, p, p, p, p, None, None, p, None) != 0             # scale2 without shift2
    assert f(C.byref(d), 2, 2, p + 2, p, p, p, None, None, None, p, None) != 0      # misaligned x
    d.tile = 9
    assert f(C.byref(d), 2, 2, p, p, p, p, None, None, None, p, None) != 0
    assert b'tile' in L.lib().yv4_last_error()


# ---------------------------------------------------------------------------------------------
# whole detector in fp16 / bf16 (wrap_fp16_model) vs the oracle's reduced-precision emulation
# ---------------------------------------------------------------------------------------------
def _build_v5(golden, dev):
    from conftest import arch_from, state_dict_from
    g = golden('tiny_v5')
    stages, reps, chans = arch_from(g)
    det = pkg.build_detector(dict(
        type='SingleStageDetector',
        backbone=dict(type='DarknetCSP', scale=[stages, reps, chans], out_indices=[2, 3, 4]),
        neck=dict(type='YOLOV5Neck', in_channels=[32, 64, 128], out_channels=[32, 64, 128], csp_repetition=1),
        bbox_head=dict(type='YOLOCSPHead', num_classes=80, in_channels=[32, 64, 128]), train_cfg=None,
        test_cfg=dict(min_bbox_size=0, nms_pre=-1, score_thr=0.001, nms=dict(type='nms', iou_threshold=0.65),
                      max_per_img=300)))
    det.load_state_dict(state_dict_from(g), strict=True)
    return g, det.to(dev).eval(), (stages, reps)


@pytest.mark.parametrize('dtype,tol_fused,tol_ref', [(torch.float16, 1.2e-2, 2e-2), (torch.bfloat16, 8e-2, 2e-1)])
def test_detector_in_16_bit_vs_oracle_emulation(golden, gpu_device, dtype, tol_fused, tol_ref):
    """Pred maps of the fused 16-bit path vs (a) the oracle rounding where a single-rounding fused
    kernel rounds, (b) the oracle rounding where the reference's wrap_fp16_model rounds (conv out, BN
    out, activation out), (c) the fp32 golden.  Tolerances are absolute on logits of magnitude ~5:
    a few ulps of the 16-bit type accumulated over ~60 layers."""
    g, det, (stages, reps) = _build_v5(golden, gpu_device)
    img = torch.from_numpy(g['img']).to(gpu_device)
    sd = {k: v.detach().cpu() for k, v in det.state_dict().items()}
    with torch.no_grad():
        p32 = [p.cpu() for p in det.forward_dummy(img)[0]]
    pkg.wrap_fp16_model(det, dtype)
    assert det.bbox_head.fp16_enabled
    with torch.no_grad():
        p16 = [p.cpu() for p in det.forward_dummy(img)[0]]
    assert all(p.dtype == torch.float32 for p in p16)              # module boundaries stay fp32 NCHW
    with O.precision(dtype, 'fused'):
        of, _ = O.forward_pred_maps(img.cpu(), sd, stages, reps, [2, 3, 4], neck='v5')
    with O.precision(dtype, 'reference'):
        orf, _ = O.forward_pred_maps(img.cpu(), sd, stages, reps, [2, 3, 4], neck='v5')
    for i in range(3):
        scale = 1.0 + p32[i].abs()
        e_f = float(((p16[i] - of[i]).abs() / scale).max())
        e_r = float(((p16[i] - orf[i]).abs() / scale).max())
        e_32 = float(((p16[i] - p32[i]).abs() / scale).max())
        print(f'{dtype} level {i}: vs fused-emulation {e_f:.2e}, vs reference-emulation {e_r:.2e}, vs fp32 {e_32:.2e}')
        assert e_f <= tol_fused, (i, e_f)
        assert e_r <= tol_ref and e_32 <= tol_ref, (i, e_r, e_32)
        assert e_32 > 0                                            # it really ran in reduced precision
    # end to end: detections of the 16-bit plan are close to the fp32 ones
    metas = [dict(scale_factor=g['scale_factors'][i]) for i in range(2)]
    res = det.simple_test(img, metas, rescale=True)
    n16 = sum(len(r) for r in res[0])
    assert abs(n16 - len(g['dets0'])) <= max(6, len(g['dets0']) // 8)
    pkg.wrap_fp16_model(det, torch.float32)
    with torch.no_grad():
        back = [p.cpu() for p in det.forward_dummy(img)[0]]
    for a, b in zip(back, p32):
        assert torch.equal(a, b)


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('shape', [(16, 13, 11), (40, 19, 19), (64, 20, 20), (8, 30, 30)])
def test_h16_layout_and_spp_kernels(gpu_device, dtype, shape):
    """(the SPP of a map of <= 512 pixels is one LDS-tiled launch -- a partial, one-and-a-bit and two 32-channel slices
    here -- above that three chained launches)"""
    torch.manual_seed(0)
    C, H, W = shape
    plan = pkg.Plan(gpu_device, dtype)
    x = plan.add_input_nchw(2, C, H, W)
    cat = plan.new_buf(2, H, W, 4 * C, 'cat')
    plan.resample(x, cat.slice(0, C))
    plan.spp(cat, C)
    up = plan.new_buf(2, 2 * H, 2 * W, 4 * C, 'up')
    plan.resample(cat, up)
    plan.add_output_nchw(cat)
    plan.add_output_nchw(up)
    plan.finalize()
    inp = torch.randn(2, C, H, W, device=gpu_device)
    got_cat, got_up = plan.run(inp)
    xr = inp.to(dtype).float()
    ref = torch.cat([xr] + [F.max_pool2d(xr, k, 1, k // 2) for k in (5, 9, 13)], 1)
    assert torch.equal(got_cat, ref)
    assert torch.equal(got_up, F.interpolate(ref, size=(2 * H, 2 * W), mode='nearest'))


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
def test_h16_plan_with_fp32_stem(gpu_device, dtype):
    """A detector whose backbone starts with the 3x3 stem keeps the image fp32 in a 16-bit plan
    (yv4_conv_stem_fwd: fp32 arithmetic, 16-bit output)."""
    torch.manual_seed(0)
    det = pkg.build_detector(dict(
        type='SingleStageDetector',
        backbone=dict(type='DarknetCSP', scale=[['conv', 'bottleneck', 'csp', 'csp', 'csp', 'sppv4'],
                                                [None, 1, 1, 1, 1, 1], [8, 16, 32, 64, 64, 64]], out_indices=[3, 4, 5]),
        neck=dict(type='YOLOV4Neck', in_channels=[64, 64, 64], out_channels=[32, 64, 128], csp_repetition=1),
        bbox_head=dict(type='YOLOCSPHead', num_classes=80, in_channels=[32, 64, 128]), train_cfg=None,
        test_cfg=dict(nms_pre=-1, score_thr=0.001, nms=dict(type='nms', iou_threshold=0.65), max_per_img=300)))
    det.init_weights()
    det.to(gpu_device).eval()
    img = torch.randn(2, 3, 64, 96, device=gpu_device)
    p32 = det.compile(2, 64, 96, device=gpu_device, rescale=False)
    p32.run(img)
    ref = [v.buf.tensor.clone() for v in p32.pred_views]
    p16 = det.compile(2, 64, 96, device=gpu_device, rescale=False, dtype=dtype)
    convs = [o for o in p16.ops if o.kind == 'conv']
    assert convs[0].info['stem32'] and not any(o.info['stem32'] for o in convs[1:])
    assert p16.inputs[0]['view'].buf.dtype == torch.float32 and convs[0].info['out'].buf.dtype == dtype
    p16.run(img)
    tol = 2e-2 if dtype == torch.float16 else 2e-1
    for a, b in zip(ref, [v.buf.tensor for v in p16.pred_views]):
        assert float((a - b).abs().max()) <= tol * (1 + float(a.abs().max()))
    p16.autotune()
    p16.run(img)


# ---------------------------------------------------------------------------------------------
# 16-bit training kernels (bf16 / fp16 activations, fp32 master weights and statistics)
# ---------------------------------------------------------------------------------------------
from mmdet_yolov4_amd import train_ops as T  # noqa: E402


@pytest.mark.parametrize('dtype,tol', [(torch.bfloat16, 2e-2), (torch.float16, 3e-3)])
@pytest.mark.parametrize('shape', [(2, 16, 24, 13, 11, 3, 1, 1), (2, 64, 32, 12, 12, 3, 2, 1), (1, 32, 72, 9, 9, 1, 1, 0),
                                   (2, 8, 16, 13, 15, 3, 2, 1)])
def test_h16_conv_autograd(gpu_device, dtype, tol, shape):
    """ConvFunction on 16-bit operands vs torch autograd in fp64 on the same rounded x, w, dY."""
    N, Cin, Cout, H, W, k, s, p = shape
    torch.manual_seed(0)
    x = torch.randn(N, Cin, H, W, device=gpu_device).to(dtype)
    w = (torch.randn(Cout, Cin, k, k, device=gpu_device) * (Cin * k * k) ** -0.5)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    y = T.conv2d(xr, wr, s, p, dtype=dtype)
    assert y.dtype == dtype and y.is_contiguous(memory_format=torch.channels_last)
    gy = torch.randn_like(y.float()).to(dtype)
    y.backward(gy)
    x64 = x.double().requires_grad_(True)
    w64 = w.to(dtype).double().requires_grad_(True)
    y64 = F.conv2d(x64, w64, None, s, p)
    y64.backward(gy.double())

    def rel(a, b):
        return float((a.double() - b).abs().max() / (b.abs().max() + 1e-12))
    assert rel(y, y64.detach()) <= tol
    assert xr.grad.dtype == dtype and rel(xr.grad, x64.grad) <= tol
    assert wr.grad.dtype == torch.float32 and rel(wr.grad, w64.grad) <= max(tol / 4, 2e-3)   # dW accumulates in fp32


W3_SHAPES = [
    # N, Cin, Cout, H, W (3x3, stride 1, pad 1, Cin % 128 == 0): the domain of conv_wgrad3x3_h16_kernel
    (2, 128, 128, 19, 19),     # two images, ragged last slice (722 rows), image borders inside every slice
    (3, 256, 64, 7, 5),        # map narrower than a transposed block: many left / right borders; half-empty co tile
    (1, 128, 128, 1, 300),     # one image row: the kh = 0 / 2 tiles see only padding
    (5, 128, 192, 3, 3),       # tiny images; Cout tail inside a co tile
    (4, 384, 136, 16, 16),     # three ci tiles, two co tiles, 1 024 rows
    (16, 128, 128, 38, 38),    # 23 104 rows: several chunks per tile (the split reduction + ordered slab sum)
]


@pytest.mark.parametrize('dtype,tol', [(torch.bfloat16, 2e-2), (torch.float16, 3e-3)])
@pytest.mark.parametrize('shape', W3_SHAPES)
def test_h16_wgrad3x3_kernel(gpu_device, dtype, tol, shape):
    """The kw-shared weight-gradient kernel (train.hip conv_wgrad3x3_h16_kernel; auto-selected for 3x3 / stride 1 with
    Cin % 128 == 0) through ConvFunction, vs fp64 autograd on the same rounded operands; deterministic run to run."""
    N, Cin, Cout, H, W = shape
    torch.manual_seed(1)
    x = torch.randn(N, Cin, H, W, device=gpu_device).to(dtype)
    w = (torch.randn(Cout, Cin, 3, 3, device=gpu_device) * (Cin * 9) ** -0.5)
    gy = None
    grads = []
    for _ in range(2):
        xr = x.clone().requires_grad_(True)
        wr = w.clone().requires_grad_(True)
        y = T.conv2d(xr, wr, 1, 1, dtype=dtype)
        if gy is None:
            gy = torch.randn_like(y.float()).to(dtype)
        y.backward(gy)
        grads.append(wr.grad.clone())
    assert torch.equal(grads[0], grads[1])
    x64 = x.double()
    w64 = w.to(dtype).double().requires_grad_(True)
    F.conv2d(x64, w64, None, 1, 1).backward(gy.double())
    err = float((grads[0].double() - w64.grad).abs().max() / (w64.grad.abs().max() + 1e-12))
    assert err <= 2e-3, err                                             # dW accumulates in fp32
    # every tap of every (co, ci) block: a shifted-row or masking mistake shows as one wrong (kh, kw) plane
    per_tap = (grads[0].double() - w64.grad).abs().amax(dim=(0, 1)) / (w64.grad.abs().amax(dim=(0, 1)) + 1e-12)
    assert float(per_tap.max()) <= 4e-3, per_tap


FC_SHAPES = [
    # N, Cin, Cout, H, W (3x3, stride 1, pad 1; Cin 16 / 32 / 64 per pixel, Cout 32 / 64; at least 131 072 rows):
    # the domain of conv_wgrad_fc_h16_kernel
    (4, 32, 64, 192, 192),     # the first Bottleneck's widths
    (3, 16, 32, 224, 208),     # 32-byte pixel rows, two taps per 32-column block
    (2, 64, 64, 260, 256),     # 128-byte rows on both sides (swapped halves), six column blocks per kh
    (9, 32, 32, 121, 123),     # odd sizes: image and row borders anywhere in a slice, ragged last slice
    (2, 64, 32, 300, 230),
    (5, 16, 64, 170, 160),
]


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('shape', FC_SHAPES)
def test_h16_wgrad_few_channel_kernel(gpu_device, dtype, shape):
    """The few-channel 3x3 weight-gradient kernel (train.hip conv_wgrad_fc_h16_kernel: all of dW per workgroup, one
    source image per kh shared by its three kw taps) through ConvFunction, vs fp64 autograd on the same rounded
    operands; deterministic run to run; per-tap error (a shifted-row or masking mistake is one wrong (kh, kw) plane)."""
    N, Cin, Cout, H, W = shape
    torch.manual_seed(2)
    x = torch.randn(N, Cin, H, W, device=gpu_device).to(dtype)
    w = (torch.randn(Cout, Cin, 3, 3, device=gpu_device) * (Cin * 9) ** -0.5)
    gy = None
    grads = []
    for _ in range(2):
        xr = x.clone().requires_grad_(True)
        wr = w.clone().requires_grad_(True)
        y = T.conv2d(xr, wr, 1, 1, dtype=dtype)
        if gy is None:
            gy = torch.randn_like(y.float()).to(dtype)
        y.backward(gy)
        grads.append(wr.grad.clone())
    assert torch.equal(grads[0], grads[1])
    w64 = w.to(dtype).double().requires_grad_(True)
    F.conv2d(x.double(), w64, None, 1, 1).backward(gy.double())
    err = float((grads[0].double() - w64.grad).abs().max() / (w64.grad.abs().max() + 1e-12))
    assert err <= 2e-3, err
    per_tap = (grads[0].double() - w64.grad).abs().amax(dim=(0, 1)) / (w64.grad.abs().amax(dim=(0, 1)) + 1e-12)
    assert float(per_tap.max()) <= 4e-3, per_tap


def test_h16_wgrad_few_channel_kernel_stem(gpu_device):
    """The 16-bit stem: 3 weight channels against an image stored with 16 channels per pixel (``image_to_nhwc16``): the
    kernel loads 16, keeps the weight's (padded) 8."""
    torch.manual_seed(3)
    N, H, W = 4, 200, 180
    img = torch.randn(N, 3, H, W, device=gpu_device)
    w = (torch.randn(32, 3, 3, 3, device=gpu_device) * 27 ** -0.5).requires_grad_(True)
    x16 = T.image_to_nhwc16(img, torch.bfloat16, 16)
    y = T.conv2d(x16, w, 1, 1, dtype=torch.bfloat16)
    gy = torch.randn_like(y.float()).bfloat16()
    y.backward(gy)
    w64 = w.detach().bfloat16().double().requires_grad_(True)
    F.conv2d(img.bfloat16().double(), w64, None, 1, 1).backward(gy.double())
    err = float((w.grad.double() - w64.grad).abs().max() / (w64.grad.abs().max() + 1e-12))
    assert err <= 2e-3, err


@pytest.mark.parametrize('dtype,tol', [(torch.bfloat16, 3e-2), (torch.float16, 4e-3)])
@pytest.mark.parametrize('act', [0, 1, 2])
def test_h16_bn_act_autograd(gpu_device, dtype, tol, act):
    torch.manual_seed(1)
    N, C_, H, W = 3, 24, 9, 7
    x = (torch.randn(N, C_, H, W, device=gpu_device) * 1.5 + 0.3).to(dtype)
    res = torch.randn(N, C_, H, W, device=gpu_device).to(dtype)
    bn = torch.nn.BatchNorm2d(C_, eps=1e-3, momentum=0.03).to(gpu_device).train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_(0, 0.2)
    ref_bn = torch.nn.BatchNorm2d(C_, eps=1e-3, momentum=0.03).to(gpu_device).double().train()
    ref_bn.load_state_dict({k: v.double() if v.dtype.is_floating_point else v for k, v in bn.state_dict().items()})
    xr = x.clone().requires_grad_(True)
    rr = res.clone().requires_grad_(True)
    y = T.bn_act(xr, bn, (act, 0.1), rr)
    assert y.dtype == dtype
    gy = torch.randn(N, C_, H, W, device=gpu_device).to(dtype)
    y.backward(gy)
    x64 = x.double().requires_grad_(True)
    r64 = res.double().requires_grad_(True)
    z = ref_bn(x64)
    z = {0: lambda v: v, 1: lambda v: v * torch.tanh(F.softplus(v)), 2: lambda v: F.leaky_relu(v, 0.1)}[act](z) + r64
    z.backward(gy.double())

    def rel(a, b):
        return float((a.double() - b).abs().max() / (b.abs().max() + 1e-12))
    assert rel(y, z.detach()) <= tol
    assert rel(xr.grad, x64.grad) <= tol and rel(rr.grad, r64.grad) <= 1e-6
    assert rel(bn.weight.grad, ref_bn.weight.grad) <= 2e-3 and rel(bn.bias.grad, ref_bn.bias.grad) <= 2e-3
    assert rel(bn.running_mean, ref_bn.running_mean) <= 1e-5 and rel(bn.running_var, ref_bn.running_var) <= 1e-5


@pytest.mark.parametrize('dtype,tol', [(torch.bfloat16, 4e-2), (torch.float16, 6e-3)])
def test_detector_train_step_in_16_bit(golden, gpu_device, dtype, tol):
    """One training step with 16-bit activations vs the same step in fp32 (which is pinned against the
    reference by test_gpu_train_parity.py): losses within `tol` relative, every parameter gradient
    close in direction and norm, master weights and their gradients stay fp32."""
    g = golden('train_v4')
    torch.manual_seed(3)
    det = pkg.build_detector(dict(
        type='SingleStageDetector',
        backbone=dict(type='DarknetCSP', scale=[['conv', 'bottleneck', 'csp', 'csp', 'csp', 'sppv4'],
                                                [None, 1, 1, 2, 1, 1], [8, 16, 32, 64, 64, 64]], out_indices=[3, 4, 5]),
        neck=dict(type='YOLOV4Neck', in_channels=[64, 64, 64], out_channels=[32, 64, 128], csp_repetition=1),
        bbox_head=dict(type='YOLOCSPHead', num_classes=80, in_channels=[32, 64, 128]), train_cfg=None,
        test_cfg=dict(nms_pre=-1, score_thr=0.001, nms=dict(type='nms', iou_threshold=0.65), max_per_img=300)))
    det.init_weights()
    det.to(gpu_device).train()
    data = dict(img=torch.from_numpy(g['img']).to(gpu_device), img_metas=[dict(), dict()],
                gt_bboxes=[torch.from_numpy(g['gt_bboxes0']).to(gpu_device),
                           torch.from_numpy(g['gt_bboxes1']).to(gpu_device)],
                gt_labels=[torch.from_numpy(g['gt_labels0']).to(gpu_device),
                           torch.from_numpy(g['gt_labels1']).to(gpu_device)])
    sd0 = {k: v.clone() for k, v in det.state_dict().items()}
    out32 = det.train_step(data, None)
    out32['loss'].backward()
    g32 = {n: p.grad.clone() for n, p in det.named_parameters()}
    det.zero_grad()
    det.load_state_dict(sd0)                      # undo the BN running-stat update of the first step
    pkg.wrap_fp16_model(det, dtype)
    out16 = det.train_step(data, None)
    for k in ('loss', 'loss_cls', 'loss_conf', 'loss_bbox'):
        np.testing.assert_allclose(out16['log_vars'][k], out32['log_vars'][k], rtol=tol, err_msg=k)
    out16['loss'].backward()
    # Gradients deep in the network are not comparable element-wise between precisions: with batch
    # statistics over a handful of positions per channel the map input -> gradient is chaotic at the
    # 1e-2 perturbation level of bf16 (cosines of ~0.4 are observed in the backbone of this toy
    # model while every kernel passes its own 16-bit test above).  What must hold: fp32 master
    # gradients exist and are finite everywhere, and the layers next to the loss agree.
    for n, p in det.named_parameters():
        assert p.dtype == torch.float32 and p.grad is not None and p.grad.dtype == torch.float32, n
        assert bool(torch.isfinite(p.grad).all()), n
    for n, p in det.bbox_head.named_parameters():
        a, b = p.grad.double().flatten(), g32['bbox_head.' + n].double().flatten()
        cos = float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30))
        assert cos >= (0.8 if dtype == torch.bfloat16 else 0.99), (n, cos)
    # and SGD on 16-bit activations follows the fp32 run: same loss trajectory over a few steps
    def run(dt):
        det.load_state_dict(sd0)
        pkg.wrap_fp16_model(det, dt)
        opt = torch.optim.SGD(det.parameters(), lr=1e-3, momentum=0.9, nesterov=True)
        hist = []
        for _ in range(6):
            opt.zero_grad()
            o = det.train_step(data, None)
            o['loss'].backward()
            torch.nn.utils.clip_grad_norm_(det.parameters(), 35)
            opt.step()
            hist.append(o['log_vars']['loss'])
        return np.array(hist)
    h32, h16 = run(torch.float32), run(dtype)
    np.testing.assert_allclose(h16, h32, rtol=max(3 * tol, 0.04))      # six chaotic steps: a few percent


def test_wgrad_widening_fallback_matches(gpu_device):
    """YV4_WGRAD_WIDEN=1 routes 16-bit weight gradients through the widening fp32-MFMA kernel (the
    fallback of the ds_read_b64_tr_b16 form): both must agree with each other to fp32 accumulation noise."""
    import subprocess
    code = (
        "import torch, sys; sys.path.insert(0, %r); import mmdet_yolov4_amd; from mmdet_yolov4_amd import train_ops as T;"
        "torch.manual_seed(0); x = torch.randn(2, 64, 13, 11, device='cuda').bfloat16();"
        "w = (torch.randn(72, 64, 3, 3, device='cuda') * 0.04).requires_grad_(True);"
        "y = T.conv2d(x, w, 1, 1, dtype=torch.bfloat16); g = torch.randn_like(y.float()).bfloat16(); y.backward(g);"
        "print('DW', float(w.grad.double().sum()), float(w.grad.double().abs().sum()))"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for widen in ('0', '1'):
        env = dict(os.environ, YV4_WGRAD_WIDEN=widen)
        r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([float(v) for v in [l for l in r.stdout.splitlines() if l.startswith('DW')][0].split()[1:]])
    assert abs(outs[0][0] - outs[1][0]) <= 1e-3 * outs[0][1] and abs(outs[0][1] - outs[1][1]) <= 1e-4 * outs[0][1]
