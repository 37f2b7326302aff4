"""Parity of the HIP path (through the C-ABI) against the CPU oracle and the golden
vectors generated from the reference.  Run on the GPU box:  pytest -m gpu

Tolerances (stated per BASELINE.json north_star: boxes/scores within 1e-4 in fp32, NMS
selection indices bit-exact):
  * conv / feature maps: |err| <= 1e-4 * (1 + |ref|): fp32 MFMA accumulates in a different
    order than oneDNN, errors grow ~sqrt(K)*2^-24 per layer;
  * decode: boxes 1e-4 absolute, scores 1e-6;
  * NMS on IDENTICAL candidate inputs: bit-exact indices, boxes and scores.
"""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import mmdet_yolov4_amd as pkg
from conftest import arch_from, state_dict_from
from oracle import yolov4_oracle as O

pytestmark = pytest.mark.gpu
L = pkg._lib


def close(got, ref, tol=1e-4, what=''):
    got = got.detach().cpu().double() if isinstance(got, torch.Tensor) else torch.as_tensor(got).double()
    ref = ref.detach().cpu().double() if isinstance(ref, torch.Tensor) else torch.as_tensor(ref).double()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    err = ((got - ref).abs() / (1 + ref.abs())).max().item() if got.numel() else 0.0
    assert err <= tol, f'{what}: max scaled err {err:.3e} > {tol}'
    return err


# ---------------------------------------------------------------------------------------------
# Mish
# ---------------------------------------------------------------------------------------------
def test_mish_fp32_fp64_against_reference_vectors(golden, gpu_device):
    g = golden('mish')
    x = torch.from_numpy(g['x']).to(gpu_device)
    gr = torch.from_numpy(g['g']).to(gpu_device)
    y = pkg.mish_forward(x)
    gi = pkg.mish_backward(gr, x)
    # fp32 on device uses float libm; the reference's CPU kernel evaluates in double -> ~1 ulp
    np.testing.assert_allclose(y.cpu().numpy(), g['y'], rtol=3e-6, atol=1e-7)
    np.testing.assert_allclose(gi.cpu().numpy(), g['gin'], rtol=1e-4, atol=5e-6)
    y64 = pkg.mish_forward(x.double())
    gi64 = pkg.mish_backward(gr.double(), x.double())
    np.testing.assert_allclose(y64.cpu().numpy(), g['y64'], rtol=1e-14, atol=1e-300)
    np.testing.assert_allclose(gi64.cpu().numpy(), g['gin64'], rtol=1e-12, atol=1e-15)
    assert float(pkg.mish_forward(torch.tensor([20.0, 25.0], device=gpu_device))[1]) == 25.0


@pytest.mark.parametrize('dtype,rtol', [(torch.float16, 2e-3), (torch.bfloat16, 1.6e-2)])
def test_mish_half_types_and_autograd(gpu_device, dtype, rtol):
    torch.manual_seed(0)
    x = (torch.randn(1000 + 3, device=gpu_device) * 4).to(dtype)   # odd length: vector tail
    ref = O.mish(x.float().cpu())
    np.testing.assert_allclose(pkg.mish_forward(x).float().cpu().numpy(), ref.numpy(), rtol=rtol, atol=rtol)
    xr = (torch.randn(4, 5, 6, device=gpu_device)).requires_grad_(True)
    y = pkg.Mish(inplace=True)(xr.transpose(0, 2))                 # non-contiguous input is accepted
    y.sum().backward()
    ref_g = O.mish_bwd(torch.ones(6, 5, 4), xr.detach().cpu().transpose(0, 2))
    np.testing.assert_allclose(xr.grad.cpu().transpose(0, 2).numpy(), ref_g.numpy(), rtol=1e-4, atol=1e-5)
    with pytest.raises(RuntimeError):
        pkg.mish_forward(torch.zeros(4, 4, device=gpu_device).t())  # op itself wants contiguous


def test_mish_empty(gpu_device):
    assert pkg.mish_forward(torch.zeros(0, device=gpu_device)).numel() == 0


# ---------------------------------------------------------------------------------------------
# fused conv, op level
# ---------------------------------------------------------------------------------------------
def _conv_case(dev, N, H, W, Cin, Cout, k, stride, pad, act, tile, residual=False, two_stage=False, x_off=0,
               y_off=0, seed=0, raw=False, out_extra=3):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) * (1.0 / (Cin * k * k)) ** 0.5
    s1 = torch.rand(Cout, generator=g) + 0.5
    t1 = torch.randn(Cout, generator=g) * 0.1
    s2 = torch.rand(Cout, generator=g) + 0.5
    t2 = torch.randn(Cout, generator=g) * 0.1
    Ho = (H + 2 * pad - k) // stride + 1
    Wo = (W + 2 * pad - k) // stride + 1
    res = torch.randn(N, Cout, Ho, Wo, generator=g) if residual else None
    acts = {0: lambda v: v, 1: O.mish, 2: lambda v: F.leaky_relu(v, 0.1), 3: lambda v: v * torch.sigmoid(v)}
    ref = F.conv2d(x.double(), w.double(), None, stride, pad).float()
    ref = acts[act](ref * s1[None, :, None, None] + t1[None, :, None, None])
    if residual:
        ref = ref + res
    if two_stage:
        ref = acts[act](ref * s2[None, :, None, None] + t2[None, :, None, None])

    plan = pkg.Plan(dev)
    cp = (Cin + 3) // 4 * 4
    xin = plan.add_input_nchw(N, Cin, H, W)
    if x_off:   # read the input through a channel-offset view of a wider buffer
        wide = plan.new_buf(N, H, W, cp + x_off + 4, 'wide')
        plan.resample(xin, wide.slice(x_off, cp))
        xin = wide.slice(x_off, cp)
    out_buf = plan.new_buf(N, Ho, Wo, Cout + y_off + out_extra, 'out')     # (+ 3: rows that are not 16-byte aligned)
    out = out_buf.slice(y_off, Cout)
    rv = plan.add_input_nchw(N, Cout, Ho, Wo, name='res', pad4=False) if residual else None
    plan.conv(xin, w, s1, t1, (act, 0.1), stride=stride, pad=pad, residual=rv,
              s2=s2 if two_stage else None, t2=t2 if two_stage else None, act2=(act, 0.1), out=out, tile=tile)
    plan.add_output_nchw(out)
    plan.finalize()
    out_buf.buf.tensor.fill_(7.0)  # sentinel: channels outside the view must stay untouched
    args = [x.to(dev)] + ([res.to(dev)] if residual else [])
    got = plan.run(*args)[0]
    torch.cuda.synchronize()
    full = out_buf.buf.tensor.view(N, Ho, Wo, -1)
    assert bool((full[..., :y_off] == 7.0).all()) and bool((full[..., y_off + Cout:] == 7.0).all())
    if raw:
        return got.clone()
    return close(got, ref, 1e-4, f'conv {N}x{Cin}x{H}x{W}->{Cout} k{k}s{stride} tile{tile}')


W3F_TILES = [10, 26, 42, 58, 74, 90]      # YV4_TILE_W3x3 (shape by the cost model) and YV4_TILE_W3x3_SHAPE(0..4)
W3F_SHAPES = [
    # N, H, W, Cin, Cout  (3x3, stride 1, pad 1): conv3x3_wide_f32.hip
    (2, 19, 19, 64, 128),      # image borders inside a tile, ragged last tile
    (3, 7, 5, 128, 64),        # map narrower than a fragment row group
    (1, 38, 38, 32, 192),      # one 32-channel chunk; Cout not a multiple of 128
    (2, 16, 16, 96, 80),       # three chunks, Cout tail inside a 64-column wave slab
    (1, 1, 300, 64, 64),       # one image row
    (5, 3, 3, 64, 64),         # tiny images
    (24, 38, 38, 64, 256),     # more tiles than CUs for the small shapes: the issue side crosses tiles
]


@pytest.mark.parametrize('tile', W3F_TILES)
@pytest.mark.parametrize('shape', W3F_SHAPES)
def test_conv_wide3x3_f32_kernel(gpu_device, shape, tile):
    """The fp32 wide-tile 3x3 kernel (16x16x4 MFMAs, one accumulator set, five workgroup tile shapes) against the float64
    convolution at 1e-4, plus residual / two-stage epilogue / channel-offset views."""
    N, H, W, Cin, Cout = shape
    _conv_case(gpu_device, N, H, W, Cin, Cout, 3, 1, 1, act=1, tile=tile, out_extra=4)


@pytest.mark.parametrize('tile', W3F_TILES)
@pytest.mark.parametrize('act', [0, 1, 2, 3])
def test_conv_wide3x3_f32_kernel_epilogues(gpu_device, tile, act):
    _conv_case(gpu_device, 2, 13, 13, 64, 80, 3, 1, 1, act, tile, residual=True, two_stage=True, x_off=8, y_off=16, out_extra=4)
    _conv_case(gpu_device, 1, 20, 9, 128, 128, 3, 1, 1, act, tile, residual=True, x_off=16, out_extra=4)
    _conv_case(gpu_device, 30, 38, 38, 32, 128, 3, 1, 1, act, tile, residual=True, two_stage=True, y_off=8, out_extra=4)


def test_conv_wide3x3_f32_is_batch_invariant_per_tile_id(gpu_device):
    """A pinned tile id gives the same bits whatever the batch (what bench.py's batch-2 check plan relies on); the auto
    choice reports the pinned form."""
    g = torch.Generator().manual_seed(3)
    x = torch.randn(6, 64, 19, 19, generator=g)
    w = torch.randn(128, 64, 3, 3, generator=g) * (1.0 / 576) ** 0.5
    s1, t1 = torch.rand(128, generator=g) + 0.5, torch.randn(128, generator=g) * 0.1

    def run(xb, tile):
        plan = pkg.Plan(gpu_device)
        xin = plan.add_input_nchw(xb.shape[0], 64, 19, 19)
        out = plan.conv(xin, w, s1, t1, (1, 0.1), stride=1, pad=1, tile=tile)
        plan.add_output_nchw(out)
        plan.finalize()
        y = plan.run(xb.to(gpu_device))[0].clone()
        torch.cuda.synchronize()
        return y

    for tile in (26, 42, 74):
        big = run(x, tile)
        assert torch.equal(run(x[:2], tile), big[:2]) and torch.equal(run(x[4:5], tile), big[4:5])
    d = L.ConvDesc()
    d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = 32, 38, 38, 256, 38, 38, 256
    d.KH = d.KW = 3
    d.stride, d.pad = 1, 1
    d.x_cstride, d.y_cstride, d.r_cstride = 256, 256, 256
    import ctypes
    t = L.lib().yv4_conv_pick_tile(ctypes.byref(d))
    assert t in (26, 42, 58, 74, 90), t
    d.N = 2
    assert L.lib().yv4_conv_pick_tile(ctypes.byref(d)) in (5, 6, 7)        # small batches keep the 32x32x2 tiles


WGF_TILES = [11, 27, 43, 59, 75, 91]      # YV4_TILE_WIDE (shape by the cost model) and YV4_TILE_WIDE_SHAPE(0..4)
WGF_SHAPES = [
    # N, H, W, Cin, Cout, k, stride, pad: conv_wide_f32.hip
    (2, 19, 19, 64, 128, 3, 2, 1),     # stride 2, odd map: borders and the ragged last tile
    (3, 38, 38, 64, 64, 3, 2, 1),      # Cout below one workgroup tile
    (2, 17, 23, 32, 80, 1, 1, 0),      # 1x1, one K tile, Cout not a multiple of 64
    (1, 24, 24, 256, 256, 1, 1, 0),    # the deep 1x1 shape
    (2, 13, 13, 96, 144, 3, 1, 1),     # 3x3 stride 1 through the general kernel
    (1, 21, 21, 32, 64, 5, 2, 2),      # 25 taps, pad 2
    (40, 19, 19, 32, 320, 1, 1, 0),    # several rounds of the persistent grid, three column tiles of 128
]


@pytest.mark.parametrize('tile', WGF_TILES)
@pytest.mark.parametrize('shape', WGF_SHAPES)
def test_conv_wide_f32_kernel(gpu_device, shape, tile):
    """The general fp32 wide-tile kernel (the fp32 form of the 16-bit general wide kernel) against the float64 convolution
    at 1e-4."""
    N, H, W, Cin, Cout, k, stride, pad = shape
    _conv_case(gpu_device, N, H, W, Cin, Cout, k, stride, pad, act=1, tile=tile, out_extra=4)


@pytest.mark.parametrize('tile', WGF_TILES)
@pytest.mark.parametrize('act', [0, 1, 2, 3])
def test_conv_wide_f32_kernel_epilogues(gpu_device, tile, act):
    _conv_case(gpu_device, 2, 13, 13, 64, 80, 3, 2, 1, act, tile, residual=True, two_stage=True, x_off=8, y_off=16, out_extra=4)
    _conv_case(gpu_device, 1, 20, 9, 128, 128, 1, 1, 0, act, tile, residual=True, x_off=16, out_extra=4)


def test_conv_wide_f32_refuses_what_it_cannot_address(gpu_device):
    with pytest.raises(L.Yv4Error):
        _conv_case(gpu_device, 1, 12, 12, 24, 64, 3, 1, 1, 1, 11, out_extra=4)       # Cin % 32
    with pytest.raises(L.Yv4Error):
        _conv_case(gpu_device, 1, 12, 12, 32, 72, 3, 1, 1, 1, 27, out_extra=4)       # Cout % 16
    with pytest.raises(L.Yv4Error):
        _conv_case(gpu_device, 1, 12, 12, 32, 64, 3, 1, 1, 1, 43, out_extra=3)       # rows that are not 16-byte aligned


@pytest.mark.parametrize('tile', [L.TILE_128x128, L.TILE_128x64, L.TILE_64x128, L.TILE_64x64, L.TILE_DMA_64x64, L.TILE_DMA_128x64, L.TILE_DMA_128x128, L.TILE_STEM])
@pytest.mark.parametrize('shape', [
    # N, H, W, Cin, Cout, k, stride, pad
    (2, 19, 19, 64, 128, 3, 1, 1),     # uniform-tap path, ragged M (722 rows)
    (1, 16, 20, 32, 255, 1, 1, 0),     # head-like: Cout 255 not a tile multiple
    (2, 17, 23, 32, 64, 3, 2, 1),      # stride 2, odd sizes
    (1, 12, 12, 24, 40, 3, 1, 1),      # Cin % 32 != 0 -> per-chunk tap decode
    (2, 32, 32, 3, 32, 3, 1, 1),       # stem: Cin 3 padded to 4, K = 36
    (1, 24, 24, 3, 16, 6, 2, 2),       # Focus conv k=6 s=2 p=2
    (1, 9, 37, 3, 16, 3, 1, 1),        # stem, W not a multiple of 32, Cout 16 (v4s)
    (1, 8, 70, 3, 40, 3, 1, 1),        # stem, Cout 40 (v4x): two column tiles
])
def test_conv_shapes_and_tiles(gpu_device, shape, tile):
    stem_shape = shape[3] == 3 and shape[5:] == (3, 1, 1)
    if tile == L.TILE_STEM and not stem_shape:
        pytest.skip('stem kernel: 3x3 s1 p1 on the 3-channel image only')
    if L.TILE_DMA_64x64 <= tile <= L.TILE_DMA_128x128 and shape[3] % 32 != 0:
        pytest.skip('fast-path kernels need Cin % 32 == 0')
    _conv_case(gpu_device, *shape, act=1, tile=tile)


@pytest.mark.parametrize('act', [0, 1, 2, 3])
def test_conv_epilogues(gpu_device, act):
    _conv_case(gpu_device, 2, 13, 13, 32, 96, 3, 1, 1, act, 0, residual=True, two_stage=True, x_off=8, y_off=4)
    _conv_case(gpu_device, 1, 13, 13, 64, 64, 1, 1, 0, act, 0, residual=False, two_stage=True, y_off=8)


WS_SHAPES_F32 = [
    # N, H, W, Cin, Cout  (1x1, stride 1): the domain of conv1x1_ws_f32_kernel, tile 9
    (2, 19, 19, 128, 128),     # ragged last strip, one slab (64 KB of weights)
    (1, 76, 76, 64, 64),       # more strips than one round of waves
    (3, 40, 33, 256, 128),     # eight stages per strip, two slabs of 64 columns
    (1, 31, 7, 256, 256),      # four slabs reading the same strips
    (2, 64, 64, 64, 32),       # narrowest slab
    (1, 5, 5, 128, 96),        # fewer pixels than one strip, half-empty second slab
]


@pytest.mark.parametrize('shape', WS_SHAPES_F32)
def test_conv1x1_ws_kernel_f32(gpu_device, shape):
    """The persistent weight-stationary pointwise kernel: within 1e-4 of the fp64 convolution AND bit-identical to the
    LDS-DMA tiles (same summation order), which is what lets a plan pick either by batch size."""
    N, H, W, Cin, Cout = shape
    _conv_case(gpu_device, N, H, W, Cin, Cout, 1, 1, 0, act=1, tile=L.TILE_WS_1x1)
    outs = []
    for tile in (L.TILE_WS_1x1, L.TILE_DMA_64x64, L.TILE_DMA_128x64):
        outs.append(_conv_case(gpu_device, N, H, W, Cin, Cout, 1, 1, 0, act=1, tile=tile, two_stage=True, y_off=4, raw=True))
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def test_conv1x1_ws_kernel_f32_is_refused_outside_its_domain(gpu_device):
    for shape in [(1, 8, 8, 64, 64, 3, 1, 1), (1, 8, 8, 96, 64, 1, 1, 0), (1, 8, 8, 512, 64, 1, 1, 0), (1, 8, 8, 64, 16, 1, 1, 0)]:
        with pytest.raises(L.Yv4Error):
            _conv_case(gpu_device, *shape, act=1, tile=L.TILE_WS_1x1)
    with pytest.raises(L.Yv4Error):
        _conv_case(gpu_device, 1, 8, 8, 64, 64, 1, 1, 0, act=1, tile=L.TILE_WS_1x1, residual=True)


def test_removed_tile_id_is_refused(gpu_device):
    """Tile id 10 (the fp32 ping-pong 3x3 form of round 2, removed in round 3: DESIGN 9.12) is refused, not replaced."""
    with pytest.raises(L.Yv4Error):
        _conv_case(gpu_device, 1, 8, 8, 64, 64, 3, 1, 1, act=1, tile=10)


def test_conv_big_k_accuracy(gpu_device):
    # K = 9*512 = 4608: the deepest accumulation of YOLOv4-L
    err = _conv_case(gpu_device, 1, 19, 19, 512, 128, 3, 1, 1, 1, 0)
    assert err < 5e-5


def test_conv_rejects_bad_arguments(gpu_device):
    lib = L.lib()
    d = L.ConvDesc()
    d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout, d.KH, d.KW, d.stride, d.pad = 1, 8, 8, 6, 8, 8, 8, 3, 3, 1, 1
    d.x_cstride, d.y_cstride = 6, 8
    t = torch.zeros(4096, device=gpu_device)
    p = t.data_ptr()
    assert lib.yv4_conv_bn_act_fwd(ctypes.byref(d), p, p, p, p, None, None, None, p, None) == -1   # Cin % 4
    d.Cin = d.x_cstride = 8
    d.Ho = 7
    assert lib.yv4_conv_bn_act_fwd(ctypes.byref(d), p, p, p, p, None, None, None, p, None) == -1   # geometry
    d.Ho = 8
    assert lib.yv4_conv_bn_act_fwd(ctypes.byref(d), p, p, p, p, p, None, None, p, None) == -1      # s2 without t2
    assert lib.yv4_conv_bn_act_fwd(ctypes.byref(d), p, p, p, p, None, None, None, p, None) == 0
    torch.cuda.synchronize()


# ---------------------------------------------------------------------------------------------
# SPP / resample / layout
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('C', [24, 64, 8])
@pytest.mark.parametrize('hw', [(19, 19), (13, 13), (5, 7), (20, 20), (1, 1), (22, 23), (23, 23), (40, 12)])
def test_spp_pools(gpu_device, hw, C):
    """One LDS-tiled launch for maps of <= 512 pixels (whole, partial and several 16-channel slices per pixel), the three
    chained 5x5 launches above that; max is exact."""
    H, W = hw
    x = torch.randn(2, C, H, W)
    plan = pkg.Plan(gpu_device)
    xin = plan.add_input_nchw(2, C, H, W)
    cat = plan.new_buf(2, H, W, 4 * C + 4, 'cat')
    v = cat.slice(4, 4 * C)
    plan.resample(xin, v.slice(0, C))
    plan.spp(v, C)
    plan.add_output_nchw(v)
    plan.finalize()
    got = plan.run(x.to(gpu_device))[0].cpu()
    ref = torch.cat([x] + [F.max_pool2d(x, k, 1, k // 2) for k in (5, 9, 13)], 1)
    assert torch.equal(got, ref)      # max is exact


@pytest.mark.parametrize('src,dst', [((19, 19), (38, 38)), ((5, 7), (10, 14)), ((7, 5), (13, 11)), ((9, 9), (9, 9))])
def test_resample_nearest(gpu_device, src, dst):
    x = torch.randn(2, 16, *src)
    plan = pkg.Plan(gpu_device)
    xin = plan.add_input_nchw(2, 16, *src)
    out = plan.new_buf(2, dst[0], dst[1], 32, 'cat')
    plan.resample(xin, out.slice(16, 16))
    plan.add_output_nchw(out.slice(16, 16))
    plan.finalize()
    got = plan.run(x.to(gpu_device))[0].cpu()
    assert torch.equal(got, F.interpolate(x, size=dst, mode='nearest'))


def test_layout_roundtrip(gpu_device):
    x = torch.randn(3, 7, 9, 11)
    plan = pkg.Plan(gpu_device)
    v = plan.add_input_nchw(3, 7, 9, 11)
    assert v.C == 8
    plan.add_output_nchw(v)
    plan.finalize()
    got = plan.run(x.to(gpu_device))[0].cpu()
    assert torch.equal(got[:, :7], x) and float(got[:, 7].abs().sum()) == 0.0


# ---------------------------------------------------------------------------------------------
# modules and whole detector against the reference's golden vectors
# ---------------------------------------------------------------------------------------------
def _build_from_golden(g, name, dev):
    stages, reps, chans = arch_from(g)
    neck_out = [int(c) for c in g['meta_neck_out']]
    cfg = dict(type='SingleStageDetector',
               backbone=dict(type='DarknetCSP', scale=[stages, reps, chans],
                             out_indices=[int(i) for i in g['meta_out_indices']]),
               neck=dict(type='YOLOV4Neck' if name == 'tiny_v4' else 'YOLOV5Neck',
                         in_channels=[int(c) for c in g['meta_neck_in']], out_channels=neck_out,
                         csp_repetition=int(g['meta_csp_rep'])),
               bbox_head=dict(type='YOLOCSPHead', num_classes=80, in_channels=neck_out),
               train_cfg=None,
               test_cfg=dict(min_bbox_size=0, nms_pre=-1, score_thr=0.001, nms=dict(type='nms', iou_threshold=0.65),
                             max_per_img=300))
    det = pkg.build_detector(cfg)
    det.load_state_dict(state_dict_from(g), strict=True)
    return det.eval().to(dev)


def _match_dets(got, ref, box_tol=1e-3, score_tol=1e-5):
    """Detections of two runs whose inputs differ by rounding: same count, and every
    reference detection has a counterpart with the same label, score and box."""
    gd, gl = got
    rd, rl = ref
    assert gd.shape == rd.shape, (gd.shape, rd.shape)
    if rd.shape[0] == 0:
        return
    # identical order except where scores tie within rounding: compare as sorted sets
    used = np.zeros(len(gd), bool)
    for i in range(len(rd)):
        cand = np.where((gl == rl[i]) & ~used & (np.abs(gd[:, 4] - rd[i, 4]) <= score_tol))[0]
        ok = [j for j in cand if np.abs(gd[j, :4] - rd[i, :4]).max() <= box_tol * (1 + np.abs(rd[i, :4]).max())]
        assert ok, f'reference detection {i} (label {rl[i]}, score {rd[i, 4]}) has no counterpart'
        used[ok[0]] = True


@pytest.mark.parametrize('name', ['tiny_v4', 'tiny_v5'])
def test_sibling_1x1_convs_as_one_launch_give_the_same_maps(golden, gpu_device, name, monkeypatch):
    """conv1 / conv2 of a BottleneckCSP and bottlenecks[0].conv1 / conv2 of a BottleneckCSP2 read one tensor: the plans run
    each pair as ONE conv over the concatenated output channels (``darknetcsp.emit_sibling_pair``).  Same per-element
    arithmetic: the feature maps of the fused and of the plain plan are equal bit for bit on these shapes (every fp32
    tile kernel they select sums in the same order), and both match the reference's golden maps."""
    g = golden(name)
    det = _build_from_golden(g, name, gpu_device)
    img = torch.from_numpy(g['img']).to(gpu_device)

    def run(flag):
        monkeypatch.setenv('YV4_FUSE_SIBLINGS', flag)
        plan = pkg.Plan(gpu_device)
        x = plan.add_input_nchw(*img.shape)
        preds = det.emit(plan, x)
        for v in preds:
            plan.add_output_nchw(v)
        plan.finalize()
        outs = [o.clone() for o in plan.run(img)]
        torch.cuda.synchronize()
        return outs, sum(o.kind == 'conv' for o in plan.ops)

    plain, n_plain = run('0')
    fused, n_fused = run('1')
    assert n_fused < n_plain
    for i, (a, b) in enumerate(zip(plain, fused)):
        assert torch.equal(a, b), f'{name} pred{i}: {float((a - b).abs().max())}'
        close(b, g[f'pred{i}'], 1e-4, f'{name} pred{i} (fused plan)')


@pytest.mark.parametrize('name', ['tiny_v4', 'tiny_v5'])
def test_detector_against_reference_golden(golden, gpu_device, name):
    g = golden(name)
    det = _build_from_golden(g, name, gpu_device)
    img = torch.from_numpy(g['img']).to(gpu_device)
    # module-level API (NCHW in / NCHW out), stage by stage
    x = img
    for i, lname in enumerate(det.backbone.layers):
        x = getattr(det.backbone, lname)(x)
        close(x, g[f'stage{i}'], 1e-4, f'{name} stage{i}')
    feats = det.backbone(img)
    for i, f in enumerate(feats):
        close(f, g[f'feat{i}'], 1e-4, f'{name} feat{i}')
    nouts = det.neck(feats)
    for i, f in enumerate(nouts):
        close(f, g[f'neck{i}'], 1e-4, f'{name} neck{i}')
    outs = det.bbox_head(nouts)
    assert isinstance(outs, tuple) and len(outs) == 1 and len(outs[0]) == 3     # Q5
    for i, f in enumerate(outs[0]):
        close(f, g[f'pred{i}'], 1e-4, f'{name} pred{i}')
    # get_bboxes on the REFERENCE's pred maps: identical inputs -> boxes/scores 1e-4, same selection
    sf = g['scale_factors']
    metas = [dict(scale_factor=sf[i]) for i in range(2)]
    ref_preds = [torch.from_numpy(g[f'pred{i}']).to(gpu_device) for i in range(3)]
    for rescale, tag in ((True, 'dets'), (False, 'dets_norescale')):
        res = det.bbox_head.get_bboxes(ref_preds, metas, rescale=rescale)
        for n in range(2):
            d, l = res[n]
            assert d.dtype == torch.float32 and l.dtype == torch.int64
            rd, rl = g[f'{tag}{n}'], g[f'labels{"_norescale" if not rescale else ""}{n}']
            _match_dets((d.cpu().numpy(), l.cpu().numpy()), (rd, rl))
    # end to end (image -> detections): conv rounding shifts scores by ~1e-6
    res = det.simple_test(img, metas, rescale=True)
    assert len(res) == 2 and len(res[0]) == 80 and res[0][0].dtype == np.float32
    for n in range(2):
        rd, rl = g[f'dets{n}'], g[f'labels{n}']
        got = [(r, np.full(len(r), c)) for c, r in enumerate(res[n]) if len(r)]
        gd = np.concatenate([a for a, _ in got]); gl = np.concatenate([b for _, b in got])
        assert abs(len(gd) - len(rd)) <= 3
        if len(gd) == len(rd):
            _match_dets((gd, gl), (rd, rl), box_tol=2e-3, score_tol=2e-5)


# ---------------------------------------------------------------------------------------------
# decode + NMS kernels on identical inputs
# ---------------------------------------------------------------------------------------------
def _run_post(dev, preds_nchw, sf, score_thr=0.001, iou=0.65, max_out=300, rescale=True, want_cls=True):
    plan = pkg.Plan(dev)
    views = [plan.add_input_nchw(*p.shape, name=f'p{i}', pad4=False) for i, p in enumerate(preds_nchw)]
    post = plan.postprocess(views, O.DEFAULT_STRIDES, O.base_anchors(), 80, score_thr, iou, max_out,
                            rescale=rescale, want_cls=want_cls)
    plan.finalize()
    if rescale:
        post['scale_factor'].copy_(torch.as_tensor(sf))
    plan.run(*[p.to(dev) for p in preds_nchw])
    torch.cuda.synchronize()
    return post


@pytest.mark.parametrize('name', ['tiny_v4', 'tiny_v5'])
def test_decode_filter_against_oracle(golden, gpu_device, name):
    g = golden(name)
    preds = [torch.from_numpy(g[f'pred{i}']) for i in range(3)]
    sf = g['scale_factors']
    post = _run_post(gpu_device, preds, sf)
    boxes, conf, cls = O.decode_maps(preds, 80)
    ref_boxes = boxes / torch.as_tensor(sf)[:, None, :]
    assert (post['boxes'].cpu() - ref_boxes).abs().max().item() < 1e-4
    assert (post['conf'].cpu() - conf).abs().max().item() < 1e-6
    assert (post['cls'].cpu() - cls).abs().max().item() < 1e-6
    # candidate set: identical except scores within 1e-6 of the threshold
    counts = post['counts'].cpu().numpy()
    for n in range(2):
        score = (cls[n] * conf[n][:, None]).reshape(-1).numpy()
        keys = post['keys'][n, :counts[n]].cpu().numpy().astype(np.uint64)
        flat = (keys & np.uint64(0xFFFFFFFF)).astype(np.int64)
        assert len(np.unique(flat)) == len(flat)
        ref_set = set(np.nonzero(score > np.float32(0.001))[0].tolist())
        diff = ref_set.symmetric_difference(set(flat.tolist()))
        assert all(abs(score[i] - 0.001) < 1e-6 for i in diff), len(diff)
        assert abs(int(g['num_candidates'][n]) - int(counts[n])) <= len(diff)
        # max_coord = max over the candidate boxes
        cand_boxes = ref_boxes[n][np.unique(flat // 80)]
        assert abs(float(post['max_coord'][n]) - float(cand_boxes.max())) < 1e-3


@pytest.mark.parametrize('tag', ['small', 'mid', 'split', 'empty'])
def test_batched_nms_bit_exact_vs_oracle(golden, gpu_device, tag):
    g = golden('nms')
    b = torch.from_numpy(g[f'{tag}_boxes']); s = torch.from_numpy(g[f'{tag}_scores']); thr = float(g[f'{tag}_thr'])
    d, l, inds = pkg.multiclass_nms(b.to(gpu_device), s.to(gpu_device), thr, dict(type='nms', iou_threshold=0.65),
                                    300, return_inds=True)
    # same inputs as the reference run that produced the fixture: everything bit-exact
    np.testing.assert_array_equal(d.cpu().numpy(), g[f'{tag}_dets'])
    np.testing.assert_array_equal(l.cpu().numpy(), g[f'{tag}_labels'])
    assert l.dtype == torch.int64
    if tag == 'empty':
        assert tuple(d.shape) == (0, 4)                        # Q7
    else:
        ro = O.multiclass_nms(b, s, thr, dict(type='nms', iou_threshold=0.65), 300, return_inds=True)
        np.testing.assert_array_equal(inds.cpu().numpy(), ro[2].numpy())


@pytest.mark.parametrize('n', [1, 2, 63, 64, 65, 255, 256, 257, 1000, 5000, 9999, 10000, 10241, 30000, 70001])
def test_nms_sizes_and_ties_bit_exact(gpu_device, n):
    rng = np.random.RandomState(n)
    c = rng.rand(max(n // 20, 1), 2) * 200
    cxy = c[rng.randint(0, len(c), n)] + rng.randn(n, 2) * 4
    wh = np.abs(rng.randn(n, 2)) * 15 + 8
    b = np.concatenate([cxy - wh / 2, cxy + wh / 2], 1).astype(np.float32)
    s = rng.rand(n).astype(np.float32)
    s[::4] = np.float32(0.5)                                   # heavy ties -> index tie-break
    idx = rng.randint(0, 6, n).astype(np.int64)
    for agnostic in (False, True):
        d, keep = pkg.batched_nms(torch.from_numpy(b).to(gpu_device), torch.from_numpy(s).to(gpu_device),
                                  torch.from_numpy(idx), dict(type='nms', iou_threshold=0.5, class_agnostic=agnostic))
        rd, rkeep = O.batched_nms(torch.from_numpy(b), torch.from_numpy(s), torch.from_numpy(idx),
                                  dict(type='nms', iou_threshold=0.5, class_agnostic=agnostic))
        np.testing.assert_array_equal(keep.cpu().numpy(), rkeep.numpy())
        np.testing.assert_array_equal(d.cpu().numpy(), rd.numpy())


@pytest.mark.parametrize('n', [400003, 1048577])
def test_nms_split_sorts_at_scale(gpu_device, n):
    """The split path's own radix sorts (nms_split.hip: rs_* kernels) on hundreds of tiles: with iou_threshold = 1 nothing
    is suppressed, so the result must be ALL candidates in (score descending, index ascending) order -- a stable sort torch
    can state -- whatever the class-major regrouping in between did.  Scores on a grid of 64 values: ~n / 64 exact ties
    per value."""
    g = torch.Generator().manual_seed(n)
    xy = torch.rand(n, 2, generator=g) * 1000
    wh = torch.rand(n, 2, generator=g) * 30 + 1
    b = torch.cat([xy, xy + wh], 1)
    s = (torch.randint(1, 65, (n,), generator=g).float() / 64)
    idx = torch.randint(0, 80, (n,), generator=g)
    d, keep = pkg.batched_nms(b.to(gpu_device), s.to(gpu_device), idx, dict(type='nms', iou_threshold=1.0))
    order = torch.sort(s, descending=True, stable=True).indices
    assert keep.shape[0] == n
    assert torch.equal(keep.cpu(), order)
    assert torch.equal(d[:, 4].cpu(), s[order]) and torch.equal(d[:, :4].cpu(), b[order])


def test_nms_negative_coordinates_cross_class(gpu_device):
    """Boxes with negative coordinates make mmcv's class offset overlap adjacent classes; the
    kernel works on the offset boxes, so it reproduces whatever that implies."""
    rng = np.random.RandomState(5)
    n = 800
    b = (rng.rand(n, 4) * 60 - 40).astype(np.float32)
    b[:, 2:] = b[:, :2] + rng.rand(n, 2).astype(np.float32) * 50
    s = rng.rand(n).astype(np.float32)
    idx = rng.randint(0, 3, n).astype(np.int64)
    d, keep = pkg.batched_nms(torch.from_numpy(b).to(gpu_device), torch.from_numpy(s).to(gpu_device),
                              torch.from_numpy(idx), dict(type='nms', iou_threshold=0.3))
    rd, rkeep = O.batched_nms(torch.from_numpy(b), torch.from_numpy(s), torch.from_numpy(idx),
                              dict(type='nms', iou_threshold=0.3))
    np.testing.assert_array_equal(keep.cpu().numpy(), rkeep.numpy())


@pytest.mark.parametrize('name', ['tiny_v4', 'tiny_v5'])
def test_fused_postprocess_bit_exact_selection(golden, gpu_device, name):
    """decode+filter+NMS vs the oracle run on the DEVICE's own decoded boxes/scores: with the
    candidates' values identical, the kept flat indices must be identical too."""
    g = golden(name)
    preds = [torch.from_numpy(g[f'pred{i}']) for i in range(3)]
    sf = g['scale_factors']
    post = _run_post(gpu_device, preds, sf)
    cnt = post['count'].cpu().numpy()
    for n in range(2):
        boxes = post['boxes'][n].cpu()
        score = (post['cls'][n] * post['conf'][n][:, None]).cpu()
        mb = torch.cat([score, score.new_zeros(score.shape[0], 1)], 1)
        rd, rl, rinds = O.multiclass_nms(boxes, mb, 0.001, dict(type='nms', iou_threshold=0.65), 300, return_flat=True)
        k = int(cnt[n])
        assert k == rd.shape[0]
        np.testing.assert_array_equal(post['index'][n, :k].cpu().numpy(), rinds.numpy())
        np.testing.assert_array_equal(post['dets'][n, :k].cpu().numpy(), rd.numpy())
        np.testing.assert_array_equal(post['labels'][n, :k].cpu().numpy(), rl.numpy().astype(np.int32))


def test_fused_postprocess_split_path(golden, gpu_device):
    """>= 10000 candidates per image: mmcv's per-class branch (yv4_nms_split)."""
    from mmdet_yolov4_amd.yolocsp_head import collect_results
    g = golden('tiny_v4')
    preds = []
    for i in range(3):
        p = torch.from_numpy(g[f'pred{i}']).clone()
        v = p.view(2, 3, 85, *p.shape[-2:])
        v[:, :, 4:] += 3.5                      # raise objectness and class logits
        preds.append(p)
    sf = g['scale_factors']
    post = _run_post(gpu_device, preds, sf)
    counts = post['counts'].cpu().numpy()
    assert (counts >= 10000).all(), counts
    assert (post['count'].cpu().numpy() == -1).all()
    res = collect_results(post, with_nms=True)
    for n in range(2):
        boxes = post['boxes'][n].cpu()
        score = (post['cls'][n] * post['conf'][n][:, None]).cpu()
        mb = torch.cat([score, score.new_zeros(score.shape[0], 1)], 1)
        rd, rl, rflat = O.multiclass_nms(boxes, mb, 0.001, dict(type='nms', iou_threshold=0.65), 300, return_flat=True)
        d, l = res[n]
        np.testing.assert_array_equal(d.cpu().numpy(), rd.numpy())
        np.testing.assert_array_equal(l.cpu().numpy(), rl.numpy())
        np.testing.assert_array_equal(post['index'][n, :d.shape[0]].cpu().numpy(), rflat.numpy())


# ---------------------------------------------------------------------------------------------
# get_bboxes variants: nms_pre top-k pre-selection, class-agnostic head (yolocsp_head.py:349-360)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('tag,agnostic', [('aware', False), ('agnostic', True)])
@pytest.mark.parametrize('nms_pre', [-1, 60, 250])
def test_get_bboxes_nms_pre_and_class_agnostic(golden, gpu_device, tag, agnostic, nms_pre):
    g = golden('post_variants')
    ncls = int(g['num_classes'])
    head = pkg.build_head(dict(type='YOLOCSPHead', num_classes=ncls, in_channels=[8, 8, 8], class_agnostic=agnostic,
                               train_cfg=None,
                               test_cfg=dict(nms_pre=nms_pre, score_thr=0.05, nms=dict(type='nms', iou_threshold=0.5),
                                             max_per_img=50))).to(gpu_device)
    assert head.num_attrib == (5 if agnostic else 5 + ncls)
    assert head.convs_pred[0].out_channels == 3 * head.num_attrib
    assert hasattr(head, 'loss_cls') != agnostic                 # yolocsp_head.py:155-156
    preds = [torch.from_numpy(g[f'{tag}/pred{i}']).to(gpu_device) for i in range(3)]
    metas = [dict(scale_factor=g['scale_factors'][i]) for i in range(2)]
    res = head.get_bboxes(preds, metas, rescale=True)
    for n in range(2):
        d, l = res[n]
        rd, rl = g[f'{tag}/pre{nms_pre}/dets{n}'], g[f'{tag}/pre{nms_pre}/labels{n}']
        assert l.dtype == torch.int64
        # identical inputs: same selection in the same order, boxes/scores within 1e-4
        np.testing.assert_array_equal(l.cpu().numpy(), rl)
        np.testing.assert_allclose(d.cpu().numpy(), rd, rtol=1e-4, atol=1e-4)
    # bit-exact against the oracle restatement on the same inputs (indices, boxes and scores)
    ores = O.get_bboxes([p.cpu() for p in preds], g['scale_factors'], ncls, score_thr=0.05, iou_threshold=0.5,
                        max_per_img=50, rescale=True, nms_pre=nms_pre, class_agnostic=agnostic)
    for n in range(2):
        np.testing.assert_array_equal(res[n][1].cpu().numpy(), ores[n][1].numpy())
        np.testing.assert_allclose(res[n][0].cpu().numpy(), ores[n][0].numpy(), rtol=1e-5, atol=1e-5)


def test_conf_topk_threshold_keys(gpu_device):
    """yv4_conf_topk: the k-th (conf desc, index asc) key per image, ties included."""
    import ctypes as C
    from mmdet_yolov4_amd import _lib
    L = _lib.lib()
    torch.manual_seed(3)
    N, A, ncls = 3, 3, 4
    sizes = [(6, 5), (3, 3)]
    preds = [torch.randn(N, h, w, A * (5 + ncls), device=gpu_device) for h, w in sizes]
    preds[0].view(N, -1, 5 + ncls)[:, :40, 4] = 0.25            # a block of exact ties
    levels = (_lib.LevelDesc * 2)()
    for i, (p, (h, w)) in enumerate(zip(preds, sizes)):
        levels[i].pred = p.data_ptr()
        levels[i].H, levels[i].W, levels[i].stride = h, w, 8 * (i + 1)
    total = sum(h * w * A for h, w in sizes)
    conf = torch.cat([p.view(N, -1, 5 + ncls)[:, :, 4] for p in preds], 1).sigmoid().cpu().numpy()
    work = torch.empty(L.yv4_conf_topk_work(N, total), dtype=torch.uint8, device=gpu_device)
    out = torch.zeros(N, dtype=torch.int64, device=gpu_device)
    for k in (1, 17, 50, total - 1):
        _lib.check(L.yv4_conf_topk(levels, 2, N, A, ncls, k, work.data_ptr(), out.data_ptr(),
                                   torch.cuda.current_stream().cuda_stream), 'topk')
        got = out.cpu().numpy().astype(np.uint64)
        for n in range(N):
            order = np.lexsort((np.arange(total), -conf[n].astype(np.float64)))
            assert int(got[n] & np.uint64(0xffffffff)) == order[k - 1], (k, n)
    assert L.yv4_conf_topk(levels, 2, N, A, ncls, total, work.data_ptr(), out.data_ptr(), None) != 0
    assert L.yv4_conf_topk(levels, 2, N, A, ncls, 0, work.data_ptr(), out.data_ptr(), None) != 0


@pytest.mark.parametrize('per_level', [False, True])
def test_conf_topk_radix_select_at_full_size(gpu_device, per_level):
    """The radix select (csrc/conf_topk.hip: one workgroup per segment, eight 256-bin passes) at the real segment sizes:
    YOLOv4-L 608 (22 743 boxes per image) and YOLOv3's per-level form (1 083 / 4 332 / 17 328 boxes), quantised logits so
    that thousands of boxes tie exactly and the k-th key is decided by the index half -- against a full sort on the host."""
    import ctypes as C
    from mmdet_yolov4_amd import _lib
    L = _lib.lib()
    torch.manual_seed(11)
    N, A, ncls = 5, 3, 2
    sizes = [(76, 76), (38, 38), (19, 19)]
    preds = [torch.randn(N, h, w, A * (5 + ncls), device=gpu_device) for h, w in sizes]
    for p in preds:                                                      # 1/8-steps: ~50 distinct conf values
        c = p.view(N, -1, 5 + ncls)[:, :, 4]
        c.copy_((c * 8).round() / 8)
    preds[0].view(N, -1, 5 + ncls)[1, :, 4] = 0.5                      # image 1, level 0: ALL boxes tie
    levels = (_lib.LevelDesc * 3)()
    for i, (p, (h, w)) in enumerate(zip(preds, sizes)):
        levels[i].pred = p.data_ptr()
        levels[i].H, levels[i].W, levels[i].stride = h, w, 8 << i
    boxes = [h * w * A for h, w in sizes]
    total = sum(boxes)
    base = np.concatenate([[0], np.cumsum(boxes)])
    conf = torch.cat([p.view(N, -1, 5 + ncls)[:, :, 4] for p in preds], 1).sigmoid().cpu().numpy()
    s = torch.cuda.current_stream().cuda_stream
    for k in (1, 1000, 4332, 17000):
        if per_level:
            work = torch.empty(L.yv4_conf_topk_levels_work(N, total, 3), dtype=torch.uint8, device=gpu_device)
            out = torch.zeros(N * 3, dtype=torch.int64, device=gpu_device)
            _lib.check(L.yv4_conf_topk_levels(levels, 3, N, A, ncls, k, work.data_ptr(), out.data_ptr(), s), 'topk_levels')
            got = out.cpu().numpy().astype(np.uint64).reshape(N, 3)
            for n in range(N):
                for l in range(3):
                    if boxes[l] <= k:
                        assert got[n, l] == np.uint64(0xffffffffffffffff), (k, n, l)
                        continue
                    idx = np.arange(base[l], base[l + 1])
                    order = idx[np.lexsort((idx, -conf[n, idx].astype(np.float64)))]
                    assert int(got[n, l] & np.uint64(0xffffffff)) == order[k - 1], (k, n, l)
        else:
            work = torch.empty(L.yv4_conf_topk_work(N, total), dtype=torch.uint8, device=gpu_device)
            out = torch.zeros(N, dtype=torch.int64, device=gpu_device)
            _lib.check(L.yv4_conf_topk(levels, 3, N, A, ncls, k, work.data_ptr(), out.data_ptr(), s), 'topk')
            got = out.cpu().numpy().astype(np.uint64)
            for n in range(N):
                order = np.lexsort((np.arange(total), -conf[n].astype(np.float64)))
                assert int(got[n] & np.uint64(0xffffffff)) == order[k - 1], (k, n)
                # the key's score half decodes to the k-th conf (the kernel's own sigmoid: 1 ulp of torch's)
                u = ~np.uint32(int(got[n] >> np.uint64(32)))
                sc = (np.uint32(u & np.uint32(0x7fffffff)) if u & np.uint32(0x80000000) else ~u).view(np.float32)
                assert abs(float(sc) - float(conf[n, order[k - 1]])) <= 1e-6, (k, n)


@pytest.mark.parametrize('form,tag', [('div', 'div'), ('mul', 'mul')])
def test_nms_boundary_pairs_bit_exact(golden, gpu_device, form, tag):
    """IoU == threshold (fp32) and one-rounding-off pairs, under both of mmcv's predicates (tests/golden/
    nms_boundary.npz, see tests/test_oracle_golden.py::test_nms_boundary_fixture): the HIP kernels keep and drop
    exactly what the reference glue + restated mmcv nms did -- `>` not `>=`, IEEE division for 'div', one fp32
    product for 'mul', no contraction."""
    g = golden('nms_boundary')
    b, s = torch.from_numpy(g['boxes']), torch.from_numpy(g['scores'])
    assert pkg.get_nms_iou_form() == 'div'                      # the documented default
    pkg.set_nms_iou_form(form)
    try:
        d, l, inds = pkg.multiclass_nms(b.to(gpu_device), s.to(gpu_device), float(g['thr']),
                                        dict(type='nms', iou_threshold=float(g['iou_thr'])), -1, return_inds=True)
        # the per-class (n >= split_thr) branch must agree on the same pairs
        d2, l2, inds2 = pkg.multiclass_nms(b.to(gpu_device), s.to(gpu_device), float(g['thr']),
                                           dict(type='nms', iou_threshold=float(g['iou_thr']), split_thr=100), -1,
                                           return_inds=True)
    finally:
        pkg.set_nms_iou_form('div')
    np.testing.assert_array_equal(d.cpu().numpy(), g[f'{tag}_dets'])
    np.testing.assert_array_equal(l.cpu().numpy(), g[f'{tag}_labels'])
    np.testing.assert_array_equal(inds.cpu().numpy(), g[f'{tag}_inds'])
    np.testing.assert_array_equal(d2.cpu().numpy(), g[f'{tag}_dets'])
    np.testing.assert_array_equal(inds2.cpu().numpy(), g[f'{tag}_inds'])
    assert not np.array_equal(g['div_inds'], g['mul_inds'])     # the two mmcv kernels really disagree on this input
    with pytest.raises(Exception):
        pkg.set_nms_iou_form('floor')
