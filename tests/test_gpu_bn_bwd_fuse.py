"""The reduction pass of the train-mode BatchNorm backward fused into the data-gradient launch that produces the
gradient (``yv4_bnred`` / ``train_ops.BnLink``, round 4).

Reference semantics: ATen's batch_norm_backward behind mmcv ``ConvModule`` + ``MishCudaFunction.backward``
(mmdet/models/backbones/darknetcsp.py:15-35, mmdet/ops/mish_cuda/mish.py:18-36): dbeta = sum dz act'(z^),
dgamma = sum dz act'(z^) x^, then dy from both.  What is checked here:
  * every 16-bit tile kernel (generic 128x64 / 64x64, scattered stride-2 classes incl. the row-pair view, persistent
    3x3, weight-stationary 1x1, few-channel 3x3): the launch's OUTPUT is bit-identical to the plain launch's and the
    sums it leaves equal a float64 evaluation of the definition on the stored values (2e-4 of the largest sum: fp32
    partial sums in another order) -- ragged tiles, residual (a joined gradient), channel tails;
  * the autograd path: a training step of a detector with the fusion on gives the gradients of the same step with the
    fusion off (YV4_BN_BWD_FUSE) to reassociation noise, with YV4_BN_BWD_CHECK-style verification of every fused link
    against the reduction pass (a ``grad_final`` that is wrong for the graph shows as a mismatch there), at toy size
    for fp16 / bf16 and at YOLOv4-L 608 and YOLOv5-L 640 depth in bf16;
  * most BatchNorms of YOLOv4-L take the fused path (count stated in the assertion)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

import mmdet_yolov4_amd as pkg
from mmdet_yolov4_amd import _lib
from mmdet_yolov4_amd import train_ops as T
from mmdet_yolov4_amd._lib import ConvDesc, check
from mmdet_yolov4_amd.ops import stream_ptr

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
DT = {'f16': torch.float16, 'bf16': torch.bfloat16}


def _mish_grad64(z):
    sp = torch.nn.functional.softplus(z, threshold=20.0)
    t = torch.tanh(sp)
    return t + z * (1 - t * t) * torch.sigmoid(z)


def _ref_sums(dz, y, mean, invstd, gamma, beta, act):
    """float64 definition on the STORED dz (N, C, H, W) and the BatchNorm input y."""
    dz, y = dz.double(), y.double()
    v = lambda t: t.double().view(1, -1, 1, 1)
    xh = (y - v(mean)) * v(invstd)
    z = xh * v(gamma) + v(beta)
    d = _mish_grad64(z) if act == _lib.ACT_MISH else torch.where(z >= 0, torch.ones_like(z), torch.full_like(z, 0.1))
    g = dz * d
    return g.sum((0, 2, 3)), (g * xh).sum((0, 2, 3))


def _bn_params(Cc, gen):
    mean = torch.randn(Cc, generator=gen).to(DEV) * 0.3
    invstd = (torch.rand(Cc, generator=gen).to(DEV) + 0.5)
    gamma = torch.randn(Cc, generator=gen).to(DEV)
    beta = torch.randn(Cc, generator=gen).to(DEV) * 0.5
    return mean, invstd, gamma, beta


def _bnred(y, mean, invstd, gamma, beta, act, rep, cstride=None, C_=None):
    br = _lib.BnRed()
    br.x, br.x_cstride, br.x_coff, br.C = y.data_ptr(), int(cstride or y.shape[1]), 0, int(C_ or y.shape[1])
    br.mean, br.invstd, br.gamma, br.beta = mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr()
    br.act, br.slope, br.sums = act, 0.1, rep.data_ptr()
    return br


CASES = [
    # name, N, H, W, Cin(launch), Cout(launch = BN channels), k, tile, residual
    ('generic128x64_3x3', 2, 21, 19, 64, 64, 3, 2, False),
    ('generic64x64_1x1_res', 3, 17, 13, 320, 72, 1, 3, True),          # Cin > 256: not the weight-stationary kernel
    ('generic_general_k', 2, 12, 20, 40, 48, 3, 2, True),
    ('pp3x3', 2, 30, 38, 128, 128, 3, 4, False),
    ('pp3x3_res_tails', 1, 35, 37, 64, 192, 3, 4, True),
    ('ws1x1', 2, 40, 36, 128, 64, 1, 6, False),
    ('ws1x1_res', 3, 33, 31, 64, 128, 1, 6, True),
    ('ws1x1_wide', 2, 24, 24, 256, 256, 1, 6, True),
    ('s3x3', 2, 40, 37, 64, 32, 3, 7, False),
    ('s3x3_res', 2, 33, 48, 32, 64, 3, 7, True),
]


@pytest.mark.parametrize('dt', ['bf16', 'f16'])
@pytest.mark.parametrize('case', CASES, ids=[c[0] for c in CASES])
def test_dgrad_launch_leaves_the_batchnorm_sums(case, dt):
    name, N, H, W, Cin, Cout, k, tile, with_res = case
    dtype = DT[dt]
    g = torch.Generator().manual_seed(hash(name) % 1000)
    x = torch.randn(N, Cin, H, W, generator=g).to(DEV).to(dtype).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).to(DEV)
    wp, cp = T.packed_weight(w, dtype)
    res = torch.randn(N, Cout, H, W, generator=g).to(DEV).to(dtype).contiguous(memory_format=torch.channels_last) \
        if with_res else None
    ybn = (torch.randn(N, Cout, H, W, generator=g) * 1.5).to(DEV).to(dtype).contiguous(memory_format=torch.channels_last)
    mean, invstd, gamma, beta = _bn_params(Cout, g)
    ones, zeros = T._identity_affine(x.device, Cout)
    d = ConvDesc()
    d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = N, H, W, cp, H, W, Cout
    d.KH = d.KW = k
    d.stride, d.pad = 1, k // 2
    d.x_cstride, d.y_cstride, d.r_cstride = cp, Cout, Cout
    d.tile = tile
    L = _lib.lib()
    code = T._DCODE[dtype]
    plain = torch.empty((N, Cout, H, W), device=DEV, dtype=dtype, memory_format=torch.channels_last)
    check(L.yv4_conv_bn_act_fwd_h16(C.byref(d), code, code, x.data_ptr(), wp.data_ptr(), ones.data_ptr(), zeros.data_ptr(),
                                    None, None, res.data_ptr() if with_res else None, plain.data_ptr(), stream_ptr()), 'plain')
    for act in (_lib.ACT_MISH, _lib.ACT_LEAKY):
        rep = torch.zeros(_lib.STATS_REPLICAS * 2 * Cout, dtype=torch.float64, device=DEV)
        out = torch.empty_like(plain)
        br = _bnred(ybn, mean, invstd, gamma, beta, act, rep)
        check(L.yv4_conv_dgrad_bnred_h16(C.byref(d), code, x.data_ptr(), wp.data_ptr(), ones.data_ptr(), zeros.data_ptr(),
                                         res.data_ptr() if with_res else None, out.data_ptr(), C.byref(br), stream_ptr()),
              'bnred')
        torch.cuda.synchronize()
        assert torch.equal(out, plain), f'{name}: the fused launch changed the stored gradient'
        got = rep.view(_lib.STATS_REPLICAS, 2, Cout).sum(0)
        db, dg = _ref_sums(out, ybn, mean, invstd, gamma, beta, act)
        scale = float(torch.maximum(db.abs().max(), dg.abs().max()))
        assert float((got[0] - db).abs().max()) <= 2e-4 * scale, (name, act, float((got[0] - db).abs().max()), scale)
        assert float((got[1] - dg).abs().max()) <= 2e-4 * scale, (name, act, float((got[1] - dg).abs().max()), scale)
        # the fold + apply half: dy and dgamma / dbeta equal the two-pass backward's
        dx1 = torch.empty_like(out)
        dx2 = torch.empty_like(out)
        g1, b1, g2, b2 = (torch.zeros(Cout, device=DEV) for _ in range(4))
        totals = torch.empty(2 * Cout, dtype=torch.float64, device=DEV)
        M = N * H * W
        check(L.yv4_bn_act_bwd_prereduced(ybn.data_ptr(), code, Cout, 0, out.data_ptr(), Cout, 0, mean.data_ptr(),
                                          invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), dx1.data_ptr(), Cout, 0,
                                          g1.data_ptr(), b1.data_ptr(), rep.data_ptr(), totals.data_ptr(), M, Cout, act, 0.1,
                                          0, stream_ptr()), 'prereduced')
        work = torch.empty(2 * Cout, dtype=torch.float64, device=DEV)
        check(L.yv4_bn_act_bwd_h16(ybn.data_ptr(), code, Cout, 0, out.data_ptr(), Cout, 0, mean.data_ptr(), invstd.data_ptr(),
                                   gamma.data_ptr(), beta.data_ptr(), dx2.data_ptr(), Cout, 0, g2.data_ptr(), b2.data_ptr(),
                                   work.data_ptr(), M, Cout, act, 0.1, stream_ptr()), 'two-pass')
        torch.cuda.synchronize()
        assert float(rep.abs().max()) == 0.0, 'the fold must leave the replica buffer clean'
        np.testing.assert_allclose(g1.cpu().numpy(), g2.cpu().numpy(), rtol=0, atol=2e-4 * scale)
        np.testing.assert_allclose(b1.cpu().numpy(), b2.cpu().numpy(), rtol=0, atol=2e-4 * scale)
        ulp = 2.0 ** -7 if dt == 'bf16' else 2.0 ** -10
        assert float((dx1.float() - dx2.float()).abs().max()) <= 2 * ulp * float(dx2.float().abs().max())


@pytest.mark.parametrize('dt', ['bf16', 'f16'])
@pytest.mark.parametrize('rowpair', [False, True])
def test_stride2_data_gradient_classes_leave_the_sums(dt, rowpair):
    """The parity classes (and the two row-pair launches: 2 C channels per pixel pair, channel c' -> c' % C) of a
    3x3 / stride-2 data gradient write disjoint positions; together they hold the sums of the whole tensor."""
    dtype = DT[dt]
    N, Cin, H, W, Cout = 2, (32 if rowpair else 64), 24, 28, 64
    g = torch.Generator().manual_seed(7)
    dy = torch.randn(N, Cout, H // 2, W // 2, generator=g).to(DEV).to(dtype).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cout * 9) ** 0.5).to(DEV)
    ybn = (torch.randn(N, Cin, H, W, generator=g) * 1.5).to(DEV).to(dtype).contiguous(memory_format=torch.channels_last)
    mean, invstd, gamma, beta = _bn_params(Cin, g)
    rep = torch.zeros(_lib.STATS_REPLICAS * 2 * Cin, dtype=torch.float64, device=DEV)
    link = T.BnLink(ybn, mean, invstd, gamma, beta, _lib.ACT_MISH, 0.0, rep)
    fn = T._dgrad_s2_rowpair if rowpair else T._dgrad_s2_parity
    plain = fn(dy, w, (N, Cin, H, W), dtype)
    fused = fn(dy, w, (N, Cin, H, W), dtype, link=link)
    torch.cuda.synchronize()
    assert link.reduced and torch.equal(plain, fused)
    got = rep.view(_lib.STATS_REPLICAS, 2, Cin).sum(0)
    db, dg = _ref_sums(fused, ybn, mean, invstd, gamma, beta, _lib.ACT_MISH)
    scale = float(torch.maximum(db.abs().max(), dg.abs().max()))
    assert float((got[0] - db).abs().max()) <= 2e-4 * scale and float((got[1] - dg).abs().max()) <= 2e-4 * scale


def _toy(dev):
    torch.manual_seed(0)
    det = pkg.build_detector(dict(
        type='SingleStageDetector',
        backbone=dict(type='DarknetCSP', scale=[['conv', 'bottleneck', 'csp', 'csp', 'csp', 'sppv4'], [None, 1, 1, 2, 2, 1],
                                                [8, 16, 32, 64, 128, 128]], out_indices=[3, 4, 5]),
        neck=dict(type='YOLOV4Neck', in_channels=[64, 128, 128], out_channels=[64, 128, 256], csp_repetition=2),
        bbox_head=dict(type='YOLOCSPHead', num_classes=80, in_channels=[64, 128, 256]), train_cfg=dict()))
    det.init_weights()
    return det.train().to(dev)


def _step_grads(det, data, fuse, verify, flat=None):
    """Gradients (fp32 arena or .grad) of one forward_train + backward with the fusion on / off."""
    T._BN_BWD_FUSE, T._BN_BWD_CHECK = fuse, verify
    T.bn_bwd_fuse_stats.update(fused=0, unfused=0)
    if flat is not None:
        flat.zero_grad()          # gradients live in the flat arena: dW / dgamma / dbeta are accumulated in place
    else:
        det.zero_grad()
    losses = det(**data)
    total, _ = det._parse_losses(losses)
    total.backward()
    torch.cuda.synchronize()
    return {n: p.grad.detach().clone() for n, p in det.named_parameters()}, dict(T.bn_bwd_fuse_stats)


def _ab(det, data, min_fused, flat=None):
    sd = {k: v.clone() for k, v in det.state_dict().items()}
    try:
        g_on, st_on = _step_grads(det, data, True, True, flat)
        det.load_state_dict(sd)
        g_off, st_off = _step_grads(det, data, False, False, flat)
    finally:
        T._BN_BWD_FUSE, T._BN_BWD_CHECK = True, False
    assert st_off['fused'] == 0 and st_on['fused'] >= min_fused, (st_on, st_off)
    worst = 0.0
    for n in g_on:
        a, b = g_on[n].double(), g_off[n].double()
        assert bool(torch.isfinite(a).all()), n
        worst = max(worst, float((a - b).norm() / (b.norm() + 1e-20)))
    return worst, st_on


@pytest.mark.parametrize('arena', [False, True])
@pytest.mark.parametrize('dt', ['bf16', 'f16'])
def test_toy_detector_step_fused_equals_unfused(dt, arena):
    det = _toy(DEV)
    pkg.wrap_fp16_model(det, DT[dt])
    flat = None
    if arena:
        from mmdet_yolov4_amd.flat_state import FlatState
        flat = FlatState(det)
    g = torch.Generator().manual_seed(3)
    img = torch.randn(3, 3, 128, 160, generator=g).to(DEV)
    boxes = [torch.tensor([[8., 10., 80., 90.], [30., 20., 120., 100.]], device=DEV)] * 3
    labels = [torch.tensor([1, 5], device=DEV)] * 3
    data = dict(img=img, img_metas=[dict()] * 3, gt_bboxes=boxes, gt_labels=labels)
    worst, st = _ab(det, data, min_fused=25, flat=flat)
    # the fused sums are fp32 partials of the same stored values in another order; every downstream rounding to 16
    # bits can flip a last bit: per-tensor relative L2 distance of the gradients (measured 1e-3 bf16 / 2e-4 fp16)
    assert worst <= (2e-2 if dt == 'bf16' else 4e-3), (worst, st)


@pytest.mark.parametrize('model,size', [('yolov4l', 608), ('yolov5l', 640)])
def test_fullsize_step_fused_equals_unfused(model, size):
    """BASELINE.json configs[2] / [4] depth and input size (batch 4): every fused link verified against the reduction
    pass inside the step (YV4_BN_BWD_CHECK), gradients against the unfused step; most BatchNorms fuse."""
    torch.manual_seed(0)
    det = pkg.build_detector(bench.model_cfg(model))
    det.init_weights()
    det.train().to(DEV)
    pkg.wrap_fp16_model(det, torch.bfloat16)
    B = 4
    img = bench.synthetic_images(B, size, 1000, DEV)
    gtb, gtl = bench.synthetic_gts(B, size, 2000, DEV)
    data = dict(img=img, img_metas=[dict() for _ in range(B)], gt_bboxes=gtb, gt_labels=gtl)
    n_bn = sum(1 for m in det.modules() if isinstance(m, torch.nn.BatchNorm2d))
    worst, st = _ab(det, data, min_fused=int(0.7 * n_bn))
    print(f'{model}: {st["fused"]} of {n_bn} BatchNorm backwards fused, worst per-tensor gradient distance {worst:.3e}')
    assert worst <= 5e-2, (worst, st)
