"""Training-side HIP ops (conv backward-data / backward-filter, train-mode BN+act) through their
autograd front ends, against torch CPU autograd in fp64.  Run with -m gpu."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import mmdet_yolov4_amd as pkg
from mmdet_yolov4_amd import train_ops as T
from oracle import yolov4_oracle as O

pytestmark = pytest.mark.gpu


def rel(got, ref):
    got = got.detach().cpu().double()
    ref = ref.detach().cpu().double()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    return float((got - ref).abs().max() / (ref.abs().max() + 1e-12))


@pytest.mark.parametrize('shape', [
    # N, Cin, H, W, Cout, k, stride, pad
    (2, 32, 13, 17, 64, 3, 1, 1),
    (2, 64, 9, 9, 32, 1, 1, 0),
    (2, 32, 16, 20, 64, 3, 2, 1),      # stride 2, even size (Cin <= 32: the row-pair data gradient, two launches)
    (2, 16, 12, 8, 32, 3, 2, 1),       # ... 32 output columns per launch
    (1, 8, 10, 14, 64, 3, 2, 1),       # ... 16
    (2, 64, 12, 12, 64, 3, 2, 1),      # stride 2, Cin > 32: four parity classes
    (1, 32, 15, 19, 32, 3, 2, 1),      # stride 2, odd size (dilated grid cropped)
    (2, 24, 10, 10, 40, 3, 1, 1),      # Cin, Cout not multiples of 32/64
    (2, 4, 12, 12, 16, 3, 1, 1),       # stem-like (padded image), weight gradient only
    (3, 128, 19, 19, 128, 3, 1, 1),    # several reduction chunks
])
def test_conv_forward_backward(gpu_device, shape):
    N, Cin, H, W, Cout, k, s, p = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    need_dx = Cin != 4
    xr = x.double().requires_grad_(need_dx)
    wr = w.double().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, s, p)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy.double())

    xd = x.to(gpu_device).requires_grad_(need_dx)
    wd = w.to(gpu_device).requires_grad_(True)
    y = T.conv2d(xd, wd, s, p)
    assert y.is_contiguous(memory_format=torch.channels_last)
    y.backward(gy.to(gpu_device))
    assert rel(y, yr) < 2e-5
    assert rel(wd.grad, wr.grad) < 5e-5, 'dW'
    if need_dx:
        assert rel(xd.grad, xr.grad) < 5e-5, 'dX'


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize('shape', [(2, 32, 64, 40, 24), (3, 16, 32, 16, 36), (1, 8, 64, 22, 22)])
def test_stride2_data_gradient_row_pair_form(gpu_device, monkeypatch, dtype, shape):
    """``_dgrad_s2_rowpair`` (few input channels: the column parities of a row pair as ONE output pixel of 2 Cin channels,
    two scattered stride-1 correlations) against the four parity classes and against float64 autograd, with the
    weight operands rebuilt after an in-place weight update."""
    N, Cin, Cout, H, W = shape
    g = torch.Generator().manual_seed(H * W + Cin)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    gy = torch.randn(N, Cout, H // 2, W // 2, generator=g)
    wd = torch.nn.Parameter(w.to(gpu_device))

    def run(rowpair):
        monkeypatch.setattr(T, '_ROWPAIR_ON', rowpair)
        xd = x.to(gpu_device).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        y = T.conv2d(xd, wd, 2, 1, dtype=dtype)
        y.backward(gy.to(gpu_device).to(dtype))
        return xd.grad.float()

    tol = 5e-5 if dtype == torch.float32 else (2e-2 if dtype == torch.bfloat16 else 3e-3)
    for step in range(2):
        xr = x.double().requires_grad_(True)
        F.conv2d(xr, wd.detach().cpu().double(), None, 2, 1).backward(gy.double())
        a, b = run(True), run(False)
        assert rel(a, xr.grad) < tol and rel(b, xr.grad) < tol
        assert rel(a, b) < tol
        with torch.no_grad():
            wd.mul_(-0.7).add_(0.05)        # version counter moves: the cached operands must follow


@pytest.mark.parametrize('act', [0, 1, 2, 3])
@pytest.mark.parametrize('with_res', [False, True])
def test_bn_act_forward_backward(gpu_device, act, with_res):
    g = torch.Generator().manual_seed(act * 2 + int(with_res))
    N, Cc, H, W = 3, 24, 7, 9
    x = torch.randn(N, Cc, H, W, generator=g) * 1.5 + 0.3
    res = torch.randn(N, Cc, H, W, generator=g) if with_res else None
    gy = torch.randn(N, Cc, H, W, generator=g)
    acts = {0: lambda v: v, 1: O.mish, 2: lambda v: F.leaky_relu(v, 0.1), 3: lambda v: v * torch.sigmoid(v)}

    bn_ref = torch.nn.BatchNorm2d(Cc, eps=1e-3, momentum=0.03).double()
    bn = torch.nn.BatchNorm2d(Cc, eps=1e-3, momentum=0.03).to(gpu_device)
    with torch.no_grad():
        wv = torch.rand(Cc, generator=g) + 0.5
        bv = torch.randn(Cc, generator=g) * 0.2
        for m in (bn_ref, bn):
            m.weight.copy_(wv); m.bias.copy_(bv)
    xr = x.double().requires_grad_(True)
    rr = res.double().requires_grad_(True) if with_res else None
    yr = acts[act](bn_ref(xr))
    if with_res:
        yr = yr + rr
    yr.backward(gy.double())

    xd = x.to(gpu_device).requires_grad_(True)
    rd = res.to(gpu_device).requires_grad_(True) if with_res else None
    y = T.bn_act(xd, bn, (act, 0.1), rd)
    y.backward(gy.to(gpu_device))
    assert rel(y, yr) < 1e-5
    assert rel(xd.grad, xr.grad) < 1e-4, 'dx'
    assert rel(bn.weight.grad, bn_ref.weight.grad) < 1e-4, 'dgamma'
    assert rel(bn.bias.grad, bn_ref.bias.grad) < 1e-4, 'dbeta'
    if with_res:
        assert rel(rd.grad, rr.grad) < 1e-6
    assert rel(bn.running_mean, bn_ref.running_mean) < 1e-5 and rel(bn.running_var, bn_ref.running_var) < 1e-5
    assert int(bn.num_batches_tracked) == 1


def test_wgrad_rejects_unaligned(gpu_device):
    import ctypes
    lib = pkg._lib.lib()
    d = pkg._lib.ConvDesc()
    d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout, d.KH, d.KW, d.stride, d.pad = 1, 8, 8, 8, 8, 8, 255, 1, 1, 1, 0
    d.x_cstride, d.y_cstride = 8, 255
    t = torch.zeros(1 << 16, device=gpu_device)
    assert lib.yv4_conv_wgrad(ctypes.byref(d), t.data_ptr(), t.data_ptr(), t.data_ptr(), None) == -1


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize('hw', [(19, 19), (7, 5), (13, 20)])
def test_spp_cat_forward_and_backward_match_max_pool_autograd(gpu_device, dtype, hw):
    """cat([x, mp5, mp9, mp13]) as one HIP forward + one HIP backward vs torch's max_pool2d autograd."""
    from mmdet_yolov4_amd import train_ops as T
    torch.manual_seed(0)
    H, W = hw
    x = torch.randn(2, 16, H, W, device=gpu_device).to(dtype)
    xr = x.clone().requires_grad_(True)
    out = T.spp_cat(xr)
    g = torch.randn(2, 64, H, W, device=gpu_device).to(dtype)
    out.backward(g)
    x2 = x.float().requires_grad_(True)
    ref = torch.cat([x2] + [F.max_pool2d(x2, k, 1, k // 2) for k in (5, 9, 13)], 1)
    ref.backward(g.float())
    assert out.dtype == dtype and torch.equal(out.float(), ref.detach())
    # fp32 is exact up to the order of the atomic adds; 16-bit rounds the accumulated gradient once.
    # (exact ties between window elements -- common in 16 bits -- may send a gradient to a different,
    # equal-valued element than ATen does: compare after summing each window's mass, i.e. per map)
    tol = 1e-5 if dtype == torch.float32 else (2e-2 if dtype == torch.bfloat16 else 3e-3)
    a, b = xr.grad.float(), x2.grad
    if dtype == torch.float32:
        torch.testing.assert_close(a, b, rtol=tol, atol=tol)
    else:
        torch.testing.assert_close(a.sum((2, 3)), b.sum((2, 3)), rtol=tol, atol=tol * 10)
        assert float((a - b).abs().max()) <= 0.05 * float(b.abs().max()) or (a != b).float().mean() < 0.02


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_eval_mode_bn_inside_a_training_graph(gpu_device, dtype):
    """norm_eval / frozen stages (darknetcsp.py:466-480): BN uses its running statistics as constants,
    they are not updated, and the backward has no mean / variance terms."""
    torch.manual_seed(4)
    N, C_, H, W = 2, 16, 7, 9
    bn = torch.nn.BatchNorm2d(C_, eps=1e-3, momentum=0.03).to(gpu_device)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.2)
        bn.running_mean.normal_(0, 0.5); bn.running_var.uniform_(0.5, 2.0)
    bn.eval()
    rm, rv, nbt = bn.running_mean.clone(), bn.running_var.clone(), int(bn.num_batches_tracked)
    x = torch.randn(N, C_, H, W, device=gpu_device).to(dtype)
    xr = x.clone().requires_grad_(True)
    y = T.bn_act(xr, bn, (1, 0.0))
    g = torch.randn(N, C_, H, W, device=gpu_device).to(dtype)
    y.backward(g)
    assert torch.equal(bn.running_mean, rm) and torch.equal(bn.running_var, rv) and int(bn.num_batches_tracked) == nbt
    x64 = x.double().requires_grad_(True)
    w64 = bn.weight.detach().double().requires_grad_(True)
    b64 = bn.bias.detach().double().requires_grad_(True)
    z = F.batch_norm(x64, rm.double(), rv.double(), w64, b64, False, 0.0, 1e-3)
    ref = z * torch.tanh(F.softplus(z))
    ref.backward(g.double())
    tol = 1e-5 if dtype == torch.float32 else 2e-2

    def rel(a, b):
        return float((a.detach().double() - b).abs().max() / (b.abs().max() + 1e-12))
    assert rel(y, ref.detach()) <= tol and rel(xr.grad, x64.grad) <= tol
    assert rel(bn.weight.grad, w64.grad) <= max(tol / 10, 1e-5) and rel(bn.bias.grad, b64.grad) <= max(tol / 10, 1e-5)


def test_detector_with_frozen_stage_and_norm_eval_trains(gpu_device):
    torch.manual_seed(5)
    det = pkg.build_detector(dict(
        type='SingleStageDetector',
        backbone=dict(type='DarknetCSP', scale=[['conv', 'bottleneck', 'csp', 'csp', 'csp', 'sppv4'],
                                                [None, 1, 1, 1, 1, 1], [8, 16, 32, 64, 64, 64]], out_indices=[3, 4, 5],
                      frozen_stages=1, norm_eval=True),
        neck=dict(type='YOLOV4Neck', in_channels=[64, 64, 64], out_channels=[32, 64, 128], csp_repetition=1),
        bbox_head=dict(type='YOLOCSPHead', num_classes=80, in_channels=[32, 64, 128]), train_cfg=None,
        test_cfg=dict(nms_pre=-1, score_thr=0.001, nms=dict(type='nms', iou_threshold=0.65), max_per_img=300)))
    det.init_weights()
    det.to(gpu_device).train()
    bns = [m for m in det.backbone.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    assert bns and not any(m.training for m in bns)                  # norm_eval
    rm = [m.running_mean.clone() for m in bns]
    data = dict(img=torch.randn(2, 3, 64, 96, device=gpu_device), img_metas=[dict(), dict()],
                gt_bboxes=[torch.tensor([[8., 10., 40., 44.]], device=gpu_device),
                           torch.tensor([[20., 5., 90., 60.]], device=gpu_device)],
                gt_labels=[torch.tensor([3], device=gpu_device), torch.tensor([7], device=gpu_device)])
    out = det.train_step(data, None)
    out['loss'].backward()
    assert np.isfinite(out['log_vars']['loss'])
    assert all(torch.equal(a, m.running_mean) for a, m in zip(rm, bns))   # statistics untouched
    frozen = [p for n, p in det.backbone.named_parameters() if not p.requires_grad]
    assert frozen and all(p.grad is None for p in frozen)
    live = [p for p in det.parameters() if p.requires_grad]
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in live)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize('shape', [(3, 32, 64, 19, 3, 1), (2, 64, 40, 23, 1, 1), (2, 16, 32, 30, 3, 2), (1, 128, 256, 38, 3, 1),
                                   (4, 32, 64, 250, 3, 1),      # 1 024 tiles: the few-channel 3x3 kernel (16-bit)
                                   (2, 64, 64, 260, 1, 1),      # 4 225 strips: the weight-stationary 1x1 kernels
                                   (8, 128, 256, 152, 3, 2),    # the automatic choice is the general wide-tile kernel (fp32 and 16-bit)
                                   (16, 128, 128, 76, 3, 1)])   # ... and the wide-tile 3x3 kernel
def test_conv_epilogue_leaves_the_bn_sums(dtype, shape):
    """yv4_conv_fwd_stats: the conv kernel's epilogue accumulates [sum | sum of squares] of the STORED outputs
    (rounded to the output type) over YV4_STATS_REPLICAS copies; kernels without that epilogue (Cin % 32 != 0 in
    fp32) fall back to the reduction kernel behind the same entry point.  Checked against float64 sums of the
    output tensor (1e-6 relative) and through bn_act with / without the precomputed sums (identical up to 2e-6)."""
    from mmdet_yolov4_amd import train_ops as T
    from mmdet_yolov4_amd import _lib
    N, Cin, Cout, hw, k, stride = shape
    dev = torch.device('cuda', 0)
    g = torch.Generator().manual_seed(Cin + Cout)
    x = torch.randn(N, Cin, hw, hw, generator=g).to(dev).to(dtype).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5).to(dev)
    stats = T.conv_stats_buffer(Cout, dev)
    stats.fill_(float('nan'))                                  # the entry point clears it
    y = T.conv2d(x, w, stride, k // 2, dtype=dtype, stats=stats)
    y_plain = T.conv2d(x, w, stride, k // 2, dtype=dtype)
    assert torch.equal(y, y_plain)
    tot = stats.view(_lib.STATS_REPLICAS, 2, Cout).sum(0)
    yd = y.double()
    want = torch.stack([yd.sum((0, 2, 3)), (yd * yd).sum((0, 2, 3))])
    assert float((tot - want).abs().max() / want.abs().max()) < 1e-6
    bn_a, bn_b = torch.nn.BatchNorm2d(Cout).to(dev).train(), torch.nn.BatchNorm2d(Cout).to(dev).train()
    out_a = T.bn_act(y, bn_a, (1, 0.0), sums=stats)
    out_b = T.bn_act(y, bn_b, (1, 0.0))
    tol = 2e-6 if dtype == torch.float32 else 1e-2
    out_a, out_b = out_a.detach().float(), out_b.detach().float()
    assert float((out_a - out_b).abs().max()) <= tol * float(out_b.abs().max())
    assert float((bn_a.running_var - bn_b.running_var).abs().max()) < 1e-6
    assert float((bn_a.running_mean - bn_b.running_mean).abs().max()) < 1e-6


@pytest.mark.parametrize('shape', [(3, 80, 19, 19), (1, 40, 40, 36), (2, 36, 9, 30)])
def test_spp_backward_lds_and_atomic_forms(gpu_device, shape):
    """The SPP backward keeps one (image, 32-channel group) accumulator in LDS when H*W*128 B fits 64 KB (every map an
    SPP block sees in the recipes) and falls back to global float atomics otherwise: both against ATen's autograd,
    fp32, channel counts that are not multiples of the group."""
    from mmdet_yolov4_amd import train_ops as T
    N, C_, H, W = shape
    torch.manual_seed(1)
    x = torch.randn(N, C_, H, W, device=gpu_device)
    xr = x.clone().requires_grad_(True)
    out = T.spp_cat(xr)
    g = torch.randn(N, 4 * C_, H, W, device=gpu_device)
    out.backward(g)
    x2 = x.clone().requires_grad_(True)
    ref = torch.cat([x2] + [F.max_pool2d(x2, k, 1, k // 2) for k in (5, 9, 13)], 1)
    ref.backward(g)
    assert torch.equal(out, ref.detach())
    torch.testing.assert_close(xr.grad, x2.grad, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('block', ['csp', 'csp2', 'csp2_shortcut', 'sppv4'])
def test_csp_halves_written_into_the_concat_buffer(gpu_device, monkeypatch, block, dtype):
    """``train_ops.CatSlot``: the producers of a CSP concat (two bare convs; a bottleneck's BN + act + shortcut and a bare
    conv; SPPV4's conv6 and conv2) write their channel halves of the concat buffer and the buffer's gradient reaches each
    of them whole -- against the same module run through ``torch.cat`` (YV4_CAT_SLOTS off): identical outputs,
    gradients of the input and of every parameter equal up to the BN reductions' atomic order."""
    from mmdet_yolov4_amd import darknetcsp as D
    from mmdet_yolov4_amd import train_ops as T
    torch.manual_seed(5)
    kw = dict(norm_cfg=dict(type='BN'), act_cfg=dict(type='Mish'))
    if block == 'csp':
        mod = D.BottleneckCSP(32, 64, repetition=2, **kw)
    elif block == 'csp2':
        mod = D.BottleneckCSP2(32, 32, repetition=2, shortcut=False, **kw)
    elif block == 'csp2_shortcut':
        mod = D.BottleneckCSP2(48, 24, repetition=1, shortcut=True, **kw)
    else:
        mod = D.SPPV4(64, 32, **kw)
    mod = mod.to(gpu_device).train()
    if dtype != torch.float32:
        pkg.wrap_fp16_model(mod, dtype)
    cin = 48 if block == 'csp2_shortcut' else (64 if block == 'sppv4' else 32)
    x0 = torch.randn(3, cin, 19, 21, device=gpu_device).to(dtype).contiguous(memory_format=torch.channels_last)

    def run(slots):
        monkeypatch.setattr(D, '_CAT_SLOTS', slots)
        mod.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_(True)
        out = mod(x)
        T.flush_batch_counters()
        gout = torch.linspace(-1, 1, out.numel(), device=gpu_device).view_as(out).to(out.dtype)
        out.backward(gout)
        return out.detach().float(), x.grad.float(), [p.grad.detach().float().clone() for p in mod.parameters()]

    launches = []
    real_cat = torch.cat
    monkeypatch.setattr(torch, 'cat', lambda *a, **k: (launches.append(1), real_cat(*a, **k))[1])
    out_a, dx_a, gp_a = run(True)
    n_slots = len(launches)
    out_b, dx_b, gp_b = run(False)
    assert len(launches) - n_slots == n_slots + 1          # the CSP concat itself is the one torch.cat that went away
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    if block == 'csp':       # the joint BN's sums come from the two conv epilogues: same values, another summation order
        assert float((out_a - out_b).abs().max()) <= tol * float(out_b.abs().max())
    else:
        assert torch.equal(out_a, out_b)
    assert float((dx_a - dx_b).abs().max()) <= tol * float(dx_b.abs().max())
    for a, b in zip(gp_a, gp_b):
        assert float((a - b).abs().max()) <= tol * max(float(b.abs().max()), 1e-6)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('shape', [(2, 16, 7, 9, 2), (3, 24, 5, 5, 1), (1, 8, 6, 4, 4)])
def test_resample_into_concat_slot(gpu_device, dtype, shape):
    """``train_ops.resample_into``: nearest upsample by an integer factor (or the plain copy) straight into a channel
    range of a concat buffer, and the backward that sums the buffer's gradient slice over the pixels that read each
    source pixel -- against F.interpolate + torch.cat autograd (the forward bit for bit; fp32 sums on both sides)."""
    N, C_, H, W, f = shape
    torch.manual_seed(7)
    a = torch.randn(N, C_, H * f, W * f, device=gpu_device).to(dtype)
    b = torch.randn(N, C_, H, W, device=gpu_device).to(dtype)
    ar, br = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    z = T.resample_into(ar, (H * f, W * f), T.CatSlot(2 * C_, 0))                 # copy into the first half
    z = T.resample_into(br, (H * f, W * f), T.CatSlot(2 * C_, C_, z))             # upsample into the second
    g = torch.randn(N, 2 * C_, H * f, W * f, device=gpu_device).to(dtype).contiguous(memory_format=torch.channels_last)
    z.backward(g)
    a2, b2 = a.float().requires_grad_(True), b.float().requires_grad_(True)
    ref = torch.cat((a2, F.interpolate(b2, size=(H * f, W * f), mode='nearest')), 1)
    ref.backward(g.float())
    assert torch.equal(z.detach().float(), ref.detach())
    assert torch.equal(ar.grad.float(), a2.grad)
    tol = 1e-6 if dtype == torch.float32 else 1e-2
    assert rel(br.grad.float(), b2.grad) <= tol


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('neck', ['v4', 'v5'])
def test_neck_halves_written_into_the_concat_buffers(gpu_device, monkeypatch, neck, dtype):
    """The necks' concatenations (lateral conv / backbone map with the upsampled map; stride-2 conv with the saved
    top-down map) through ``CatSlot`` producers against the same neck through ``torch.cat`` (YV4_CAT_SLOTS off)."""
    from mmdet_yolov4_amd import darknetcsp as D
    from mmdet_yolov4_amd import yolo_neck_csp as NK
    torch.manual_seed(11)
    if neck == 'v4':
        mod = NK.YOLOV4Neck(in_channels=[32, 64, 64], out_channels=[32, 64, 128], csp_repetition=1)
        chans = [32, 64, 64]
    else:
        mod = NK.YOLOV5Neck(in_channels=[32, 64, 128], out_channels=[32, 64, 128], csp_repetition=1)
        chans = [32, 64, 128]
    mod = mod.to(gpu_device).train()
    if dtype != torch.float32:
        pkg.wrap_fp16_model(mod, dtype)
    xs0 = [torch.randn(2, c, 24 >> i, 16 >> i, device=gpu_device).to(dtype).contiguous(memory_format=torch.channels_last)
           for i, c in enumerate(chans)]

    def run(slots):
        monkeypatch.setattr(D, '_CAT_SLOTS', slots)
        mod.zero_grad(set_to_none=True)
        xs = [x.clone().requires_grad_(True) for x in xs0]
        outs = mod(xs)
        T.flush_batch_counters()
        loss = sum((o.float() * torch.linspace(-1, 1, o.numel(), device=gpu_device).view_as(o)).sum() for o in outs)
        loss.backward()
        return [o.detach().float() for o in outs], [x.grad.float() for x in xs], \
            [p.grad.detach().float().clone() for p in mod.parameters()]

    cats = []
    real_cat = torch.cat
    monkeypatch.setattr(torch, 'cat', lambda *a, **k: (cats.append(1), real_cat(*a, **k))[1])
    out_a, dx_a, gp_a = run(True)
    with_slots = len(cats)
    out_b, dx_b, gp_b = run(False)
    assert with_slots == 0 and len(cats) >= 4          # two top-down and two bottom-up concatenations (+ the CSP ones)
    # fp32 is the check of the plumbing.  In bf16 the forward is bit for bit the same, but the slot path also joins the
    # CSP blocks' fan-out gradients inside a data-gradient launch (one rounding) where autograd adds two rounded
    # tensors: rounding noise that a BatchNorm weight gradient three blocks upstream (a sum with cancellation) shows
    # as a few per cent of its largest entry
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    for a, b in zip(out_a, out_b):
        assert float((a - b).abs().max()) <= tol * float(b.abs().max())
    gtol = 1e-5 if dtype == torch.float32 else 1e-1
    for a, b in zip(dx_a + gp_a, dx_b + gp_b):
        assert float((a - b).abs().max()) <= gtol * max(float(b.abs().max()), 1e-6)


def test_direct_gradient_accumulation_matches_autograd(gpu_device):
    """Conv dW and BatchNorm dgamma / dbeta written straight into the flat gradient arena (train_ops' direct path:
    detached weights, kernels that accumulate) over two micro-batches == autograd's own accumulation on a copy of
    the model that has no arena (fp32, 2e-5 of each gradient's largest entry)."""
    import copy
    from mmdet_yolov4_amd.flat_state import FlatState
    from mmdet_yolov4_amd import train_ops as T
    torch.manual_seed(3)
    net = torch.nn.Sequential(pkg.Conv(8, 16, 3, stride=2), pkg.Conv(16, 16, 1), pkg.Conv(16, 32, 3)).to(gpu_device).train()
    ref = copy.deepcopy(net)
    fs = FlatState(net)
    fs.zero_grad()
    assert all(getattr(p, '_yv4_grad_in_arena', False) for p in net.parameters())
    fired = []
    cb = T.add_direct_grad_listener(lambda p: fired.append(id(p)))
    try:
        for seed in (0, 1):
            x = torch.randn(2, 8, 20, 24, device=gpu_device, generator=torch.Generator(gpu_device).manual_seed(seed))
            for m in (net, ref):
                xi = x.clone().requires_grad_(True)
                m(xi).square().mean().backward()
    finally:
        T.remove_direct_grad_listener(cb)
    # the second and third convs see an input that requires grad: their weights and all three BatchNorms go direct
    assert len(fired) >= 2 * (2 + 4)
    assert fs.grads_attached()
    for (n, p), q in zip(net.named_parameters(), ref.parameters()):
        assert p.grad.data_ptr() == fs.grads.data_ptr() + 4 * fs.param_segments[[id(t) for t in fs._params].index(id(p))].offset
        err = float((p.grad - q.grad).abs().max() / (q.grad.abs().max() + 1e-12))
        assert err < 2e-5, (n, err)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('layout', ['contiguous', 'channels_last'])
def test_pack_weight_forms(gpu_device, dtype, layout):
    """yv4_pack_weight against tensor ops: forward form, data-gradient form (transposed, taps mirrored) and the tap
    subsets of the stride-2 parity classes; contiguous and channels_last sources; channel padding."""
    from mmdet_yolov4_amd import train_ops as T
    torch.manual_seed(5)
    for (Cout, Cin) in ((24, 16), (16, 40)):
        w = torch.randn(Cout, Cin, 3, 3, device=gpu_device)
        if layout == 'channels_last':
            w = w.contiguous(memory_format=torch.channels_last)
        al = 4 if dtype == torch.float32 else 8

        def pack_ref(t):          # (R, IC, KH, KW) -> (R, KH*KW*ICp) with K = (kh, kw, ic), zero-padded ic
            R, IC, KH, KW = t.shape
            icp = (IC + al - 1) // al * al
            out = torch.zeros(R, KH, KW, icp, device=t.device)
            out[..., :IC] = t.permute(0, 2, 3, 1)
            return out.reshape(R, -1).to(dtype)

        got, cp = T.packed_weight(w, dtype)
        assert cp == (Cin + al - 1) // al * al and torch.equal(got, pack_ref(w))
        got, cp = T.packed_weight(w, dtype, transpose_flip=True)
        assert cp == (Cout + al - 1) // al * al and torch.equal(got, pack_ref(w.flip(2, 3).transpose(0, 1)))
        for ta, la in (((1, 1, 1), [1]), ((2, -2, 2), [2, 0])):
            for tb, lb in (((1, 1, 1), [1]), ((2, -2, 2), [2, 0])):
                got, _ = T.packed_weight(w, dtype, taps=(ta, tb))
                assert torch.equal(got, pack_ref(w[:, :, la][:, :, :, lb].permute(1, 0, 2, 3)))


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_wgrad_deterministic_form(gpu_device, dtype):
    """yv4_conv_wgrad_det: the chunks of the N*Ho*Wo reduction go to workspace slabs that are added to dw in chunk
    order.  Same accumulate-into-dw contract and the same value (fp32 reassociation apart) as the atomic form, and
    run-to-run bit-identical where the atomic form is not guaranteed to be."""
    import ctypes
    from mmdet_yolov4_amd import _lib
    from mmdet_yolov4_amd._lib import ConvDesc
    lib = _lib.lib()
    torch.manual_seed(0)
    N, H, W, Cin, Cout, k = 6, 38, 38, 64, 72, 3
    code = 0 if dtype == torch.float32 else 2
    x = torch.randn(N, H, W, Cin, device=gpu_device).to(dtype)
    dy = torch.randn(N, H, W, Cout, device=gpu_device).to(dtype)
    d = ConvDesc()
    d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = N, H, W, Cin, H, W, Cout
    d.KH = d.KW = k
    d.stride, d.pad = 1, 1
    d.x_cstride, d.y_cstride = Cin, Cout
    need = int(lib.yv4_conv_wgrad_workspace(ctypes.byref(d), code))
    assert need > 0 and need % (Cout * k * k * Cin * 4) == 0                       # several chunks
    ws = torch.empty(need // 4, device=gpu_device)
    s = torch.cuda.current_stream().cuda_stream
    base = torch.randn(Cout, k * k * Cin, device=gpu_device)                       # dw is accumulated into
    outs = []
    for _ in range(3):
        dw = base.clone()
        _lib.check(lib.yv4_conv_wgrad_det(ctypes.byref(d), code, x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(),
                                          need, s), 'wgrad_det')
        outs.append(dw)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    dwa = base.clone()
    if dtype == torch.float32:
        _lib.check(lib.yv4_conv_wgrad(ctypes.byref(d), x.data_ptr(), dy.data_ptr(), dwa.data_ptr(), s), 'wgrad')
    else:
        _lib.check(lib.yv4_conv_wgrad_h16(ctypes.byref(d), code, x.data_ptr(), dy.data_ptr(), dwa.data_ptr(), s), 'wgrad')
    ref = torch.nn.grad.conv2d_weight(x.double().permute(0, 3, 1, 2), (Cout, Cin, k, k), dy.double().permute(0, 3, 1, 2),
                                      stride=1, padding=1).permute(0, 2, 3, 1).reshape(Cout, -1) + base.double()
    scale = float(ref.abs().max())
    assert float((outs[0].double() - ref).abs().max()) <= 2e-5 * scale
    assert float((outs[0] - dwa).abs().max()) <= 2e-5 * scale
    # too small a workspace is refused, none at all falls back to the atomic form
    assert lib.yv4_conv_wgrad_det(ctypes.byref(d), code, x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), 64, s) != 0
    dwn = base.clone()
    _lib.check(lib.yv4_conv_wgrad_det(ctypes.byref(d), code, x.data_ptr(), dy.data_ptr(), dwn.data_ptr(), None, 0, s), 'x')
    assert float((dwn - dwa).abs().max()) <= 2e-5 * scale


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16, torch.float32])
def test_pack_table_launch_matches_the_per_call_pack(gpu_device, dtype):
    """``yv4_pack_weights_multi`` (LDS-staged boxes of the source, one launch for the whole table) writes bit for bit
    what ``yv4_pack_weight`` writes per call: forward operand, data-gradient operand, the four parity classes of a
    stride-2 layer; 1x1 and 3x3; channel counts that are not multiples of the staging chunk; a channels_last source."""
    from mmdet_yolov4_amd import train_ops as T
    torch.manual_seed(3)
    shapes = [(255, 512, 1), (512, 256, 3), (64, 32, 3), (1024, 512, 3), (96, 1000, 1), (40, 24, 3), (8, 1032, 3)]
    ws = [torch.nn.Parameter(torch.randn(co, ci, k, k, device=gpu_device)) for co, ci, k in shapes]
    ws.append(torch.nn.Parameter(torch.randn(48, 72, 3, 3, device=gpu_device).contiguous(memory_format=torch.channels_last)))
    parity = [((1, 1, 1), (1, 1, 1)), ((1, 1, 1), (2, -2, 2)), ((2, -2, 2), (1, 1, 1)), ((2, -2, 2), (2, -2, 2))]

    def modes(w):
        m = [dict(), dict(transpose_flip=True)]
        if w.shape[2] == 3:
            m += [dict(taps=t) for t in parity]
        return m

    T.clear_pack_cache()
    for w in ws:                                    # record every request (each packed per call the first time)
        for m in modes(w):
            T.packed_weight(w, dtype, **m)
    with torch.no_grad():
        for w in ws:
            w.mul_(1.5).add_(0.125)                 # version counters move: the next request replays the table
    table = [[T.packed_weight(w, dtype, **m)[0].clone() for m in modes(w)] for w in ws]
    assert len(T._PACK_CACHES[gpu_device].entries) == sum(len(modes(w)) for w in ws)
    for w, got in zip(ws, table):
        for m, g in zip(modes(w), got):
            T.clear_pack_cache()
            want = T.packed_weight(w, dtype, **m)[0]
            assert torch.equal(g, want), (tuple(w.shape), m)
    T.clear_pack_cache()


def test_packed_weight_cache_follows_the_weights(gpu_device):
    """The recorded operands are re-packed in one launch when the weight's version counter moves (an in-place update,
    an optimizer step), and an operand asked for after that is the fresh one -- for the forward, the data-gradient and
    a stride-2 parity-class form."""
    from mmdet_yolov4_amd import train_ops as T
    T.clear_pack_cache()
    torch.manual_seed(0)
    ws = [torch.nn.Parameter(torch.randn(16, 8, 3, 3, device=gpu_device)),
          torch.nn.Parameter(torch.randn(24, 16, 3, 3, device=gpu_device))]
    modes = [dict(), dict(transpose_flip=True), dict(taps=((2, -2, 2), (1, 1, 1)))]

    def pack_all(cached):
        out = []
        for w in ws:
            for m in modes:
                if not cached:
                    T.clear_pack_cache()
                out.append(T.packed_weight(w, torch.bfloat16, **m)[0].clone())
        return out

    first = pack_all(True)
    again = pack_all(True)                      # served from the table, no change
    for a, b in zip(first, again):
        assert torch.equal(a, b)
    cache = T._PACK_CACHES[gpu_device]
    assert len(cache.entries) == 6
    with torch.no_grad():
        for w in ws:
            w.mul_(-0.5).add_(0.25)             # in-place: the version counters move
    fresh = pack_all(True)
    ref = pack_all(False)                       # per-call packing of the new values
    for a, b, c in zip(fresh, ref, first):
        assert torch.equal(a, b) and not torch.equal(a, c)
    # an update that bypasses the version counters (p.data.copy_, a raw-pointer kernel) + invalidate_packed_weights()
    pack_all(True)
    for w in ws:
        w.data.mul_(2.0)
    assert all(torch.equal(a, b) for a, b in zip(pack_all(True), fresh))        # stale by contract ...
    T.invalidate_packed_weights()
    for a, b in zip(pack_all(True), pack_all(False)):                           # ... fresh after the call
        assert torch.equal(a, b)
    # temporaries are not recorded (the stem weight is re-padded in every step -- a fresh non-leaf tensor
    # whose entry would never be hit again and whose storage the table would pin)
    pack_all(True)                                                              # (pack_all(False) cleared the table)
    cache = T._PACK_CACHES[gpu_device]
    n0 = len(cache.entries)
    assert n0 == 6
    for _ in range(3):
        tmp = torch.nn.functional.pad(ws[0], (0, 0, 0, 0, 0, 5))
        got, cp = T.packed_weight(tmp, torch.bfloat16)
        assert cp == 16 and got.shape == (16, 9 * 16)
        got2, _ = T.packed_weight(ws[0].detach() * 1.0, torch.bfloat16)
    assert len(cache.entries) == n0
    # a dropped parameter takes its entries with it at the next refresh (weak references)
    w = tmp = None
    del ws[1]
    import gc
    gc.collect()
    with torch.no_grad():
        ws[0].add_(1.0)
    T.packed_weight(ws[0], torch.bfloat16)
    assert len(cache.entries) == 3
    # an address the allocator hands to a NEW parameter of the same shape is a miss, not a stale hit
    shape, old_ptr = tuple(ws[0].shape), ws[0].data_ptr()
    ws.clear()
    gc.collect()
    fresh_w = torch.nn.Parameter(torch.randn(shape, device=gpu_device))
    got, _ = T.packed_weight(fresh_w, torch.bfloat16)
    T.clear_pack_cache()
    assert torch.equal(got, T.packed_weight(fresh_w, torch.bfloat16)[0]), (fresh_w.data_ptr() == old_ptr)
    T.clear_pack_cache()


def _ulp16(ref, dtype):
    """One unit in the last place of the 16-bit type at |ref| (fp32 tensor), with the type's smallest normal as the floor."""
    mant, emin = (7, -126) if dtype == torch.bfloat16 else (10, -14)
    e = torch.floor(torch.log2(ref.abs().clamp(min=2.0 ** emin)))
    return torch.pow(2.0, e - mant)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('shape', [(128, 3, 37, 41), (32, 2, 97, 53), (256, 5, 19, 19), (36, 1, 23, 9), (512, 2, 10, 10)])
@pytest.mark.parametrize('with_res', [False, True])
def test_bn16_passes_within_one_ulp_of_fp32_evaluation(gpu_device, dtype, shape, with_res):
    """The pipelined 16-bit BatchNorm + Mish passes (train.hip bn16_*: folded affine constants, exponent clamp instead of
    the asymptote select; semantics mmdet/ops/mish_cuda/src/mish.h:16-50 + torch.nn.BatchNorm2d of darknetcsp.py:15-35)
    against the SAME 16-bit values pushed through the fp32 kernels: every 16-bit output within one ulp of the fp32-evaluated
    value rounded to the type, dgamma / dbeta / running statistics to fp32 rounding.  Shapes
    with row counts that leave every tail length of the software pipeline (and C % 8 != 0: the 4-channel form)."""
    C, N, H, W = shape
    g = torch.Generator(device='cpu').manual_seed(C * 7 + N)
    x16 = (torch.randn(N, C, H, W, generator=g) * 2.5 + 0.7).to(gpu_device).to(dtype).contiguous(memory_format=torch.channels_last)
    dy16 = torch.randn(N, C, H, W, generator=g).to(gpu_device).to(dtype).contiguous(memory_format=torch.channels_last)
    res16 = torch.randn(N, C, H, W, generator=g).to(gpu_device).to(dtype).contiguous(memory_format=torch.channels_last) \
        if with_res else None
    bn = torch.nn.BatchNorm2d(C).to(gpu_device).train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_(0, 0.3)

    def run(x, dy, res):
        bn.zero_grad()
        bn.running_mean.zero_()
        bn.running_var.fill_(1.0)
        xr = x.clone().requires_grad_(True)
        rr = res.clone().requires_grad_(True) if res is not None else None
        y = T.bn_act(xr, bn, (pkg._lib.ACT_MISH, 0.0), residual=rr)
        y.backward(dy)
        return (y.detach().float(), xr.grad.float(), bn.weight.grad.clone(), bn.bias.grad.clone(),
                bn.running_mean.clone(), bn.running_var.clone())
    got = run(x16, dy16, res16)
    ref = run(x16.float(), dy16.float(), res16.float() if with_res else None)
    for name, a, b in zip(('y', 'dx'), got[:2], ref[:2]):
        rounded = b.to(dtype).float()                    # the fp32-evaluated value, rounded once to the type
        err = (a - rounded).abs()
        # one ulp of the type; where an output is itself a difference of O(1) terms (dx = k1 (g - dbm - xhat dgm), y near
        # Mish's zero) the re-associated fp32 affine map moves it by fp32 rounding of those terms: floor 1e-6 of the range
        tol = torch.maximum(_ulp16(b, dtype), torch.full_like(b, 1e-6 * float(b.abs().max())))
        bad = err > tol
        assert not bad.any(), (name, float(err.max()), int(bad.sum()), float((err / tol).max()))
        assert float((a == rounded).float().mean()) > 0.9, name      # and the large majority land on the same value
    for name, a, b in zip(('dgamma', 'dbeta', 'running_mean', 'running_var'), got[2:], ref[2:]):
        scale = float(b.abs().max()) + 1e-12
        assert float((a - b).abs().max()) <= 2e-5 * scale, (name, float((a - b).abs().max()), scale)
