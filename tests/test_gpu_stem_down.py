"""The fused first two layers (stem_down_h16.hip, yv4_stem_down_fwd_h16) against an fp64 restatement of the two
reference modules: Conv(3 -> C1, 3x3, s1) + BN + act, rounded to the plan's 16-bit type as its stored output is, then
Conv(C1 -> C2, 3x3, s2) + BN + act (darknetcsp.py:15-35, 290-300, 357-366), and against the two unfused launches."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from mmdet_yolov4_amd import _lib as L

pytestmark = pytest.mark.gpu

ACTS = {0: lambda v: v, 1: lambda v: v * torch.tanh(F.softplus(v)), 2: lambda v: F.leaky_relu(v, 0.1),
        3: lambda v: v * torch.sigmoid(v)}


def _run(dev, dtype, N, H, W, C1, C2, act, y_off=0, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, 3, H, W, generator=g)
    w1 = torch.randn(C1, 3, 3, 3, generator=g) * (1.0 / 27) ** 0.5
    w2 = (torch.randn(C2, C1, 3, 3, generator=g) * (1.0 / (9 * C1)) ** 0.5).to(dtype)
    s1, t1 = torch.rand(C1, generator=g) + 0.5, torch.randn(C1, generator=g) * 0.1
    s2, t2 = torch.rand(C2, generator=g) + 0.5, torch.randn(C2, generator=g) * 0.1
    a = F.conv2d(x.double(), w1.double(), None, 1, 1)
    a = ACTS[act](a * s1.double()[None, :, None, None] + t1.double()[None, :, None, None])
    a16 = a.to(dtype).double()                              # the stem's stored output
    ref = F.conv2d(a16, w2.double(), None, 2, 1)
    ref = ACTS[act](ref * s2.double()[None, :, None, None] + t2.double()[None, :, None, None]).permute(0, 2, 3, 1)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1

    w1p = torch.zeros(C1, 3, 3, 4)
    w1p[..., :3] = w1.permute(0, 2, 3, 1)                   # (cout, kh, kw, ci padded to 4)
    w2p = w2.permute(0, 2, 3, 1).contiguous()               # (cout, kh, kw, ci)
    ys = C2 + y_off + (8 if y_off else 0)
    ybuf = torch.full((N, Ho, Wo, ys), 7.0, dtype=dtype, device=dev)
    d = lambda t: t.to(dev).contiguous()
    xd, w1d, w2d, s1d, t1d, s2d, t2d = d(x), d(w1p), d(w2p), d(s1), d(t1), d(s2), d(t2)
    code = 1 if dtype == torch.float16 else 2
    L.check(L.lib().yv4_stem_down_fwd_h16(code, xd.data_ptr(), N, H, W, w1d.data_ptr(), s1d.data_ptr(), t1d.data_ptr(), C1,
                                          act, 0.1, w2d.data_ptr(), s2d.data_ptr(), t2d.data_ptr(), C2, act, 0.1,
                                          ybuf.data_ptr(), ys, y_off, torch.cuda.current_stream().cuda_stream),
            'yv4_stem_down_fwd_h16')
    torch.cuda.synchronize()
    assert bool((ybuf[..., :y_off] == 7.0).all()) and bool((ybuf[..., y_off + C2:] == 7.0).all())
    got = ybuf[..., y_off:y_off + C2].double().cpu()
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    # the kernel's stem differs from the fp64 one by ~1e-5 relative before it is rounded, so a stored stem value now
    # and then rounds the other way: that moves an output by |w2| * ulp(a) <= ~0.2 * ulp * 4 (randn inputs)
    bound = 1.5 * ulp * ref.abs() + ulp + 3e-5
    err = (got - ref).abs()
    bad = err > bound
    assert not bool(bad.any()), (f'{int(bad.sum())} of {bad.numel()} outside tolerance; worst '
                                 f'{float((err / (ref.abs() + 1e-2)).max()):.3e} rel')
    return xd, w1d, w2d, s1d, t1d, s2d, t2d, ybuf[..., y_off:y_off + C2]


SHAPES = [
    # N, H, W, C1, C2
    (2, 64, 64, 32, 64),       # 2 x 2 full tiles per image
    (1, 96, 80, 16, 32),       # the v4s widths; 3 x 3 tiles, the last column half empty
    (3, 33, 47, 32, 64),       # odd sizes: ragged tiles, Ho = 17, Wo = 24
    (1, 20, 20, 16, 64),
    (1, 608, 608, 32, 32),     # 19 x 19 tiles: more tiles than workgroups (the persistent walk and its prefetch)
]


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('shape', SHAPES)
def test_stem_down_shapes(gpu_device, dtype, shape):
    _run(gpu_device, dtype, *shape, act=1)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('act', [0, 1, 2, 3])
def test_stem_down_epilogues(gpu_device, dtype, act):
    _run(gpu_device, dtype, 2, 40, 56, 32, 64, act, y_off=16)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_stem_down_equals_the_two_launches(gpu_device, dtype):
    """Same weights through yv4_conv_stem_fwd (fp32 arithmetic) + yv4_conv_bn_act_fwd_h16: the fused kernel's
    three-term split reproduces the fp32 stem up to rare one-ulp flips of the stored 16-bit stem output, so the final
    maps agree to ~1 ulp almost everywhere."""
    dev = gpu_device
    N, H, W, C1, C2 = 2, 72, 88, 32, 64
    xd, w1d, w2d, s1d, t1d, s2d, t2d, fused = _run(dev, dtype, N, H, W, C1, C2, act=1)
    code = 1 if dtype == torch.float16 else 2
    stream = torch.cuda.current_stream().cuda_stream
    x4 = torch.zeros(N, H, W, 4, device=dev)
    x4[..., :3] = xd.permute(0, 2, 3, 1)
    a = torch.empty(N, H, W, C1, dtype=dtype, device=dev)
    d = L.ConvDesc()
    d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = N, H, W, 4, H, W, C1
    d.KH = d.KW = 3
    d.stride, d.pad = 1, 1
    d.x_cstride, d.x_coff, d.y_cstride, d.y_coff = 4, 0, C1, 0
    d.act1, d.slope1 = 1, 0.0
    L.check(L.lib().yv4_conv_stem_fwd(C.byref(d), x4.data_ptr(), w1d.data_ptr(), s1d.data_ptr(), t1d.data_ptr(),
                                      a.data_ptr(), code, stream), 'yv4_conv_stem_fwd')
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty(N, Ho, Wo, C2, dtype=dtype, device=dev)
    d2 = L.ConvDesc()
    d2.N, d2.H, d2.W, d2.Cin, d2.Ho, d2.Wo, d2.Cout = N, H, W, C1, Ho, Wo, C2
    d2.KH = d2.KW = 3
    d2.stride, d2.pad = 2, 1
    d2.x_cstride, d2.x_coff, d2.y_cstride, d2.y_coff = C1, 0, C2, 0
    d2.act1, d2.slope1 = 1, 0.0
    L.check(L.lib().yv4_conv_bn_act_fwd_h16(C.byref(d2), code, code, a.data_ptr(), w2d.data_ptr(), s2d.data_ptr(),
                                            t2d.data_ptr(), None, None, None, y.data_ptr(), stream),
            'yv4_conv_bn_act_fwd_h16')
    torch.cuda.synchronize()
    diff = (y.double() - fused.double()).abs()
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    assert float((diff <= 1e-12).double().mean()) > 0.97, 'more than 3 % of the outputs differ from the two launches'
    assert bool((diff <= 2 * ulp * y.double().abs() + ulp).all())


def test_stem_down_rejects_bad_arguments(gpu_device):
    with pytest.raises(L.Yv4Error):
        _run(gpu_device, torch.bfloat16, 1, 32, 32, 24, 64, 1)
    with pytest.raises(L.Yv4Error):
        _run(gpu_device, torch.bfloat16, 1, 32, 32, 32, 128, 1)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('widths', [[16, 32, 64, 128, 128, 128], [32, 64, 64, 128, 128, 128]])
def test_plan_fuses_the_first_two_layers(gpu_device, dtype, widths, monkeypatch):
    """A 16-bit inference plan replaces [repack, stem, stride-2 conv] by the fused launch; its pred maps equal the
    three-launch plan's up to the rare one-ulp flips of the stored stem output."""
    import mmdet_yolov4_amd as pkg
    torch.manual_seed(0)
    scale = [['conv', 'bottleneck', 'csp', 'csp', 'csp', 'sppv4'], [None, 1, 1, 1, 1, 1], widths]
    cfg = dict(type='SingleStageDetector',
               backbone=dict(type='DarknetCSP', scale=scale, out_indices=[3, 4, 5]),
               neck=dict(type='YOLOV4Neck', in_channels=[widths[3], widths[4], widths[5]], out_channels=[64, 128, 256],
                         csp_repetition=1),
               bbox_head=dict(type='YOLOCSPHead', num_classes=80, in_channels=[64, 128, 256]),
               train_cfg=None,
               test_cfg=dict(min_bbox_size=0, nms_pre=-1, score_thr=0.001, nms=dict(type='nms', iou_threshold=0.65),
                             max_per_img=300))
    det = pkg.build_detector(cfg)
    with torch.no_grad():
        for m in det.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.1)
                m.running_var.uniform_(0.5, 1.5)
    det.eval().to(gpu_device)
    img = torch.rand(2, 3, 96, 128, device=gpu_device)
    plan = det.compile(2, 96, 128, device=gpu_device, dtype=dtype)
    fused = [op for op in plan.ops if op.info.get('fused') == 'stem_down']
    assert len(fused) == 1 and [o.kind for o in fused[0].info['parts']] == ['to_nhwc', 'conv', 'conv']
    assert not any(op.kind == 'to_nhwc' for op in plan.ops)
    plan.run(img)
    torch.cuda.synchronize()
    got = [v.buf.tensor.clone() for v in plan.pred_views]
    got_count = plan.post['count'].clone()
    det._engines.clear()
    monkeypatch.setenv('YV4_STEM_FUSE', '0')
    ref_plan = det.compile(2, 96, 128, device=gpu_device, dtype=dtype)
    assert not any(op.info.get('fused') for op in ref_plan.ops)
    ref_plan.run(img)
    torch.cuda.synchronize()
    tol = 6e-2 if dtype == torch.bfloat16 else 8e-3            # pred-map logits, a 30-conv 16-bit network downstream
    for a, v in zip(got, ref_plan.pred_views):
        b = v.buf.tensor
        assert float((a - b).abs().max()) <= tol * max(1.0, float(b.abs().max()))
        assert float((a - b).abs().mean()) <= tol * 0.05 * max(1.0, float(b.abs().mean()))
    assert int((got_count - ref_plan.post['count']).abs().max()) <= max(3, int(0.02 * int(got_count.max())))
