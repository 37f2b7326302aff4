"""BASELINE.json configs[2], [3] and [4] at their real input size on the GPU (the toy-size versions of
these tests live in test_gpu_h16.py / test_gpu_train_parity.py):

  configs[2]  YOLOv4-L 608x608 bf16 training step        (configs/yolov4/yolov4l_coco_mosaic.py:86-149)
  configs[3]  YOLOv4-S 416x416 fp16 inference batch 256  (configs/yolov4/yolov4s_coco_mosaic.py; "YOLOv4-tiny" of
              BASELINE.json does not exist in the reference)
  configs[4]  YOLOv5-L 640x640 bf16 training step        (configs/yolov5/yolov5l_coco_mosaic.py:1-6)

The reference has no bf16 path, so parity for the 16-bit configurations is defined against the fp32
arithmetic under a tolerance stated in each assertion:
  * the loss of the 16-bit step equals the fp32 CPU oracle's ``head_loss`` evaluated ON THE SAME (HIP) pred maps to
    2e-4 -- the loss kernels compute in fp32 whatever the operand type of the convs;
  * the 16-bit step's losses are within LOSS_TOL of the fp32 HIP step's on identical weights and batch;
  * every parameter receives a finite fp32 gradient whose norm is within the bounds stated in ``_check_norms`` of
    the fp32 step's, and the kernels behind it are checked against fp64 at the real layer shapes;
  * the loss kernels are bit-deterministic run to run (duplicate positives resolve by slot number).
The per-GPU batch is reduced from 64 to 8 (the CPU oracle and two full backward passes have to fit a test's time
budget); the input size, depth and channel widths are the configurations'.
"""
import os
import sys

import numpy as np
import pytest
import torch

import mmdet_yolov4_amd as pkg
from mmdet_yolov4_amd.calibrate import calibrate_bn
from oracle import yolov4_oracle as O

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

pytestmark = pytest.mark.gpu

LOSS_TOL = 0.03          # bf16 step vs fp32 step, each loss term (relative)
BATCH = 8


def _train_step(det, data):
    """forward_train with the pred maps kept: returns (loss dict as floats, raw pred maps, total loss tensor)."""
    feats = det.extract_feat(data['img'])
    raws = det.bbox_head.fwd_raw(feats)
    losses = det.bbox_head.loss(raws, data['gt_bboxes'], data['gt_labels'], data['img_metas'])
    total, log_vars = det._parse_losses(losses)
    return log_vars, raws, total, losses


def _oracle_losses(raws, data):
    maps = [r.dense().detach().float().cpu() for r in raws]
    out = O.head_loss(maps, [b.cpu() for b in data['gt_bboxes']], [l.cpu() for l in data['gt_labels']])
    return {k: float(sum(x.sum() for x in v)) for k, v in out.items() if k.startswith('loss')}


_NDIM = {}


def _grad_norms(det):
    out = {}
    for n, p in det.named_parameters():
        assert p.grad is not None and p.grad.dtype == torch.float32, n
        assert bool(torch.isfinite(p.grad).all()), n
        out[n] = float(p.grad.double().norm())
        _NDIM[n] = p.dim()
    return out


def _train_case(model, size, dev):
    torch.manual_seed(0)
    det = pkg.build_detector(bench.model_cfg(model))
    det.init_weights()
    det.train().to(dev)
    img = bench.synthetic_images(BATCH, size, 1000, dev)
    gtb, gtl = bench.synthetic_gts(BATCH, size, 2000, dev)
    data = dict(img=img, img_metas=[dict() for _ in range(BATCH)], gt_bboxes=gtb, gt_labels=gtl)
    sd0 = {k: v.clone() for k, v in det.state_dict().items()}

    # ---- fp32 step: the arithmetic that IS pinned against the reference (test_gpu_train_parity.py) ----
    log32, raws32, total32, _ = _train_step(det, data)
    ora32 = _oracle_losses(raws32, data)
    for k in ('loss_cls', 'loss_conf', 'loss_bbox'):
        np.testing.assert_allclose(log32[k], ora32[k], rtol=2e-4, err_msg=f'fp32 {k} vs oracle head_loss on the HIP maps')
    total32.backward()
    n32 = _grad_norms(det)
    det.zero_grad()
    del raws32, total32

    # ---- bf16 step on the same weights and batch ----
    det.load_state_dict(sd0)
    pkg.wrap_fp16_model(det, torch.bfloat16)
    log16, raws16, total16, losses16 = _train_step(det, data)
    assert all(r.raw.dtype == torch.bfloat16 for r in raws16)
    ora16 = _oracle_losses(raws16, data)
    for k in ('loss_cls', 'loss_conf', 'loss_bbox'):
        np.testing.assert_allclose(log16[k], ora16[k], rtol=2e-4, err_msg=f'bf16 {k} vs oracle head_loss on the HIP maps')
        np.testing.assert_allclose(log16[k], log32[k], rtol=LOSS_TOL, err_msg=f'bf16 {k} vs the fp32 step')
    # determinism of the loss kernels: same maps, same gts -> bit-identical losses and pred-map gradients
    rg = []
    for _ in range(2):
        leaves = [r.raw.detach().clone().requires_grad_(True) for r in raws16]
        from mmdet_yolov4_amd.yolocsp_head import RawPredMap
        maps = [RawPredMap(l, r.bias.detach(), r.A, r.attr) for l, r in zip(leaves, raws16)]
        ls = det.bbox_head.loss(maps, gtb, gtl, data['img_metas'])
        tot, _ = det._parse_losses(ls)
        tot.backward()
        rg.append(([float(x.detach()) for v in ls.values() if isinstance(v, list) for x in v], [l.grad.clone() for l in leaves]))
    assert rg[0][0] == rg[1][0]
    assert all(torch.equal(a, b) for a, b in zip(rg[0][1], rg[1][1]))
    total16.backward()
    n16 = _grad_norms(det)
    return n32, n16, log32, log16


def _check_norms(n32, n16, label):
    """Gradient norms of the bf16 step vs the fp32 step, per parameter tensor.

    What bf16 can and cannot promise here: a randomly initialised 110-layer network with batch-statistics BatchNorm
    amplifies a perturbation ~150x from the first layers to the pred maps (test_gpu_fullsize.py measures exactly that
    for fp32 against float64: 6e-8 -> 9e-6 mean).  One bf16 rounding per fused layer (2^-9) therefore arrives at the
    head as tens of percent of feature noise whatever the kernels do, and the gradient DIRECTION of a bf16 step
    decorrelates from the fp32 step's (cosine ~0.5 measured on every layer of YOLOv5-L at init; fp16 with its three
    extra bits: 0.96).  The kernels themselves are held to half an ulp against fp64 on identical operands at the real
    layer shapes by test_fullsize_layer_shapes_bf16 below.  What must still hold for the step as a whole are its
    STATISTICS -- the quantities the recipe's clip_grad_norm_(35) and SGD see.  Bounds = 2x the values measured on
    MI355X (YOLOv4-L 608 / YOLOv5-L 640, batch 8):
      * conv / head weights (>= 2-D): |g| of every tensor within 15 % (measured worst 5.7 % / 4.7 %), median within
        6 % (0.6 % / 3.0 %);
      * BatchNorm weight / bias and head bias (1-D): sums of signed terms over N*H*W positions that cancel to ~1e-3
        of their summed magnitude, so the noise shows amplified: every tensor within 60 % (31 % / 21 %), median within
        9 % (4.5 % / 4.3 %);
      * the global gradient norm within 6 % (0.6 % / 3.2 %)."""
    CONV_TOL, VEC_TOL = 0.15, 0.6
    rel = {k: abs(n16[k] - n32[k]) / (n32[k] + 1e-12) for k in n32 if n32[k] > 1e-8}
    groups = {'matrix': {k: v for k, v in rel.items() if _NDIM[k] >= 2},
              'vector': {k: v for k, v in rel.items() if _NDIM[k] < 2}}
    tot32 = float(np.sqrt(sum(v * v for v in n32.values())))
    tot16 = float(np.sqrt(sum(v * v for v in n16.values())))
    print(f'{label}: global gradient norm bf16 {tot16:.5g} vs fp32 {tot32:.5g}')
    stats = {}
    for gname, g in groups.items():
        worst = max(g, key=g.get)
        med = float(np.median(list(g.values())))
        p90 = float(np.quantile(list(g.values()), 0.9))
        stats[gname] = (med, g[worst], worst)
        print(f'  {gname}: {len(g)} tensors, rel diff of |g|: median {med:.3e}, p90 {p90:.3e}, worst {g[worst]:.3e} ({worst})')
    assert abs(tot16 - tot32) <= 0.06 * tot32
    assert stats['matrix'][0] <= 0.06 and stats['matrix'][1] <= CONV_TOL, stats['matrix']
    assert stats['vector'][0] <= 0.09 and stats['vector'][1] <= VEC_TOL, stats['vector']


# the FLOP-heaviest / most awkward conv shapes of YOLOv4-L 608 and YOLOv5-L 640, batch 8:
# (Cin, Cout, k, stride, H_in); the 255-channel head conv runs padded to 256 in training (YOLOCSPHead.fwd_raw)
LAYER_SHAPES = [(32, 64, 3, 2, 608), (64, 64, 3, 1, 152), (128, 128, 3, 1, 76), (256, 256, 3, 1, 38),
                (512, 512, 3, 1, 19), (128, 256, 3, 2, 76), (1024, 512, 1, 1, 19), (2048, 512, 1, 1, 19),
                (256, 256, 1, 1, 76), (128, 128, 1, 1, 76), (64, 32, 1, 1, 304), (8, 64, 6, 2, 640)]      # Focus: the 3-channel image is padded to 8 (darknetcsp.Focus)


@pytest.mark.parametrize('shape', LAYER_SHAPES, ids=lambda s: 'x'.join(map(str, s)))
def test_fullsize_layer_shapes_bf16(gpu_device, shape):
    """The bf16 training kernels (forward, data gradient, weight gradient) at the REAL layer shapes of configs[2] and
    [4] vs a float64-accumulated convolution of the SAME rounded operands (torch's own conv on the device is the
    checker here, in fp64 on the bf16-rounded tensors).  Output / dX: half a bf16 ulp of the largest value + fp32
    accumulation noise -> 1e-2 of the tensor's max; dW accumulates in fp32 over N*Ho*Wo terms -> 2e-3."""
    from mmdet_yolov4_amd import train_ops as T
    import torch.nn.functional as F
    Cin, Cout, k, s, H = shape
    N = 8 if H <= 152 else 2
    torch.manual_seed(0)
    x = torch.randn(N, Cin, H, H, device=gpu_device).bfloat16()
    w = torch.randn(Cout, Cin, k, k, device=gpu_device) * (Cin * k * k) ** -0.5
    pad = 2 if k == 6 else k // 2
    xr = x.clone().requires_grad_(Cin > 8)                      # the image needs no gradient
    wr = w.clone().requires_grad_(True)
    y = T.conv2d(xr, wr, s, pad, dtype=torch.bfloat16)
    gy = torch.randn(y.shape, device=gpu_device).bfloat16().contiguous(memory_format=torch.channels_last)
    y.backward(gy)
    x64 = x.double().requires_grad_(True)
    w64 = w.bfloat16().double().requires_grad_(True)
    y64 = F.conv2d(x64, w64, None, s, pad)
    y64.backward(gy.double())

    def rel(a, b):
        return float((a.double() - b).abs().max() / (b.abs().max() + 1e-30))
    assert y.shape == y64.shape
    e_y, e_w = rel(y, y64.detach()), rel(wr.grad, w64.grad)
    e_x = rel(xr.grad, x64.grad) if Cin > 8 else 0.0
    print(f'{shape}: y {e_y:.2e} dX {e_x:.2e} dW {e_w:.2e}')
    assert e_y <= 1e-2 and e_x <= 1e-2 and e_w <= 2e-3


def test_cfg2_yolov4l_608_bf16_train_step(gpu_device):
    n32, n16, l32, l16 = _train_case('yolov4l', 608, gpu_device)
    assert len(n32) > 300                                      # 108 BN (w, b) + 112 conv weights + 3 head (w, b)
    _check_norms(n32, n16, 'yolov4l 608 bf16')


def test_cfg4_yolov5l_640_bf16_train_step(gpu_device):
    n32, n16, l32, l16 = _train_case('yolov5l', 640, gpu_device)
    _check_norms(n32, n16, 'yolov5l 640 bf16')


@pytest.mark.parametrize('model,size', [('yolov4l', 608), ('yolov5l', 640)])
def test_cfg2_cfg4_bf16_batch64_forward_and_loss(gpu_device, model, size):
    """configs[2] / configs[4] at their REAL per-GPU batch of 64 (tile shapes, `prefer_w3` /
    `prefer_wide` and the weight-gradient chunking depend on the batch, and the batch-8 cases above do not exercise the
    batch-64 selection): one bf16 training forward + fused loss + backward.  The loss of the HIP maps must equal the
    ORACLE's head_loss (yolocsp_head.py:384-575 restated, CPU) on those same maps -- the CPU cost is the loss only --, the
    loss terms must sit where the batch-8 step's sit per image (the images are drawn the same way), and every parameter
    must receive a finite gradient whose global norm relates to the batch-8 bf16 step's as 1 / sqrt(batch) within a factor
    of two (a different batch: statistics, not equality)."""
    torch.manual_seed(0)
    det = pkg.build_detector(bench.model_cfg(model))
    det.init_weights()
    det.train().to(gpu_device)
    pkg.wrap_fp16_model(det, torch.bfloat16)
    sd0 = {k: v.clone() for k, v in det.state_dict().items()}
    res = {}
    for batch in (64, 8):
        det.load_state_dict(sd0)
        det.zero_grad()
        img = bench.synthetic_images(batch, size, 1000, gpu_device)
        gtb, gtl = bench.synthetic_gts(batch, size, 2000, gpu_device)
        data = dict(img=img, img_metas=[dict() for _ in range(batch)], gt_bboxes=gtb, gt_labels=gtl)
        log, raws, total, _ = _train_step(det, data)
        assert all(r.raw.dtype == torch.bfloat16 for r in raws)
        ora = _oracle_losses(raws, data)
        for k in ('loss_cls', 'loss_conf', 'loss_bbox'):
            np.testing.assert_allclose(log[k], ora[k], rtol=2e-4, err_msg=f'batch {batch} bf16 {k} vs oracle head_loss on the HIP maps')
        total.backward()
        norms = _grad_norms(det)
        res[batch] = (log, float(np.sqrt(sum(v * v for v in norms.values()))))
        del raws, total, img, data
        torch.cuda.empty_cache()
    (l64, g64), (l8, g8) = res[64], res[8]
    print(f'{model} bf16: batch 64 losses {l64} |g| {g64:.4g}; batch 8 losses {l8} |g| {g8:.4g}')
    for k in ('loss_cls', 'loss_conf', 'loss_bbox'):
        np.testing.assert_allclose(l64[k], l8[k], rtol=0.1, err_msg=f'{k}: batch 64 vs batch 8 (means over images / positives)')
    # a randomly initialised network's per-image gradients are nearly uncorrelated: the gradient of the batch MEAN shrinks
    # like 1 / sqrt(batch) (measured: 201 at batch 8, 61 at batch 64 = 1 / 3.3 against 1 / sqrt(8) = 1 / 2.83)
    assert 0.5 * g8 <= g64 * (64 / 8) ** 0.5 <= 2.0 * g8, (g64, g8)


# ---- configs[3]: YOLOv4-S 416x416 fp16 inference, batch 256 ------------------------------------------------------
@pytest.fixture(scope='module')
def v4s(gpu_device):
    torch.manual_seed(0)
    det = pkg.build_detector(bench.model_cfg('yolov4s'))
    det.init_weights()
    det.eval().to(gpu_device)
    img = bench.synthetic_images(256, 416, 1000, gpu_device)
    plan = det.compile(256, 416, 416, device=gpu_device, rescale=True)
    calibrate_bn(plan, img)
    ncand = bench.init_head(det, plan, img, 1500.0)
    assert 300 < ncand < 6000
    det._engines.clear()
    del plan
    torch.cuda.empty_cache()
    return det, img


def _pred_maps(plan):
    return [v.buf.tensor.view(v.N, v.H, v.W, v.C).permute(0, 3, 1, 2).float() for v in plan.pred_views]


def test_cfg3_yolov4s_416_fp16_b256_vs_oracle_emulation(v4s):
    """Image 0 of the batch-256 fp16 plan vs the CPU oracle rounding to fp16 where a single-rounding fused kernel
    rounds (oracle.precision 'fused') and vs the fp32 oracle, on logits scaled by 1 + |fp32 logit|."""
    det, img = v4s
    dev = img.device
    sd = {k: v.detach().cpu() for k, v in det.state_dict().items()}
    stages, reps = O.ARCH['v4s5p']
    one = img[:1].cpu()
    ref32, _ = O.forward_pred_maps(one, sd, stages, reps, [3, 4, 5], neck='v4')
    with O.precision(torch.float16, 'fused'):
        emu, _ = O.forward_pred_maps(one, sd, stages, reps, [3, 4, 5], neck='v4')
    plan = det.compile(256, 416, 416, device=dev, rescale=True, dtype=torch.float16)
    plan.run(img)
    got = [p[:1].cpu() for p in _pred_maps(plan)]
    for i in range(3):
        assert got[i].shape == (1, 255, 416 // (8 << i), 416 // (8 << i))
        scale = 1.0 + ref32[i].abs()
        e_emu = (got[i] - emu[i]).abs() / scale          # HIP fp16 vs the CPU emulation of fp16
        e_32 = (got[i] - ref32[i]).abs() / scale         # HIP fp16 vs fp32
        e_ref = (emu[i] - ref32[i]).abs() / scale        # the CPU emulation's own distance from fp32
        print(f'level {i}: fp16 HIP vs fused emulation max {float(e_emu.max()):.2e} mean {float(e_emu.mean()):.2e} | '
              f'HIP vs fp32 oracle max {float(e_32.max()):.2e} mean {float(e_32.mean()):.2e} | '
              f'emulation vs fp32 oracle max {float(e_ref.max()):.2e} mean {float(e_ref.mean()):.2e}')
        # two fp16 evaluations that round at the same points but sum in different orders decorrelate after a few
        # layers, so the HIP path cannot track the emulation element-wise at depth 79; what it must be is AS CLOSE TO
        # fp32 AS THE EMULATION IS (the same statement test_gpu_fullsize.py makes for fp32 against float64)
        assert float(e_32.mean()) <= 1.5 * float(e_ref.mean()) + 1e-5
        assert float(e_32.max()) <= 2.5 * float(e_ref.max()) + 1e-4
        assert float(e_emu.mean()) <= 2.5 * float(e_ref.mean()) + 1e-5
        assert float(e_32.max()) > 0
    # detections of image 0: the fp16 plan's selection vs the oracle's post-processing of the SAME fp16 pred maps
    full = [p.cpu() for p in _pred_maps(plan)]
    post = plan.post
    torch.cuda.synchronize()
    cnt = post['count'].cpu().numpy()
    assert (cnt >= 0).all() and (cnt <= 300).all() and cnt.sum() > 0
    for n in (0, 100, 255):
        ores = O.get_bboxes([p[n:n + 1] for p in full], [np.ones(4, dtype=np.float32)], 80, rescale=True)[0]
        k = int(cnt[n])
        assert k == ores[0].shape[0]
        np.testing.assert_array_equal(post['labels'][n, :k].cpu().numpy(), ores[1].numpy())
        np.testing.assert_allclose(post['dets'][n, :k].cpu().numpy(), ores[0].numpy(), rtol=1e-4, atol=1e-4)


def test_cfg3_batch_independence_and_nms_invariants(v4s):
    """An image's detections do not depend on its position in the batch of 256 nor on the batch size (the batch-2 plan
    may use other conv tiles; every tile walks K in the same order, so the result is bit-identical)."""
    det, img = v4s
    dev = img.device
    cfg = det.bbox_head.test_cfg
    plan = det.compile(256, 416, 416, device=dev, rescale=True, dtype=torch.float16)
    plan.run(img)
    torch.cuda.synchronize()
    d0, l0, c0 = plan.post['dets'].clone(), plan.post['labels'].clone(), plan.post['count'].clone()
    plan.run(img.flip(0))
    torch.cuda.synchronize()
    assert torch.equal(plan.post['count'].flip(0), c0)
    assert torch.equal(plan.post['dets'].flip(0), d0) and torch.equal(plan.post['labels'].flip(0), l0)
    det._engines.clear()
    small = det.compile(2, 416, 416, device=dev, rescale=True, dtype=torch.float16)
    small.run(img[[7, 200]])
    torch.cuda.synchronize()
    for j, n in enumerate((7, 200)):
        k = int(c0[n])
        assert int(small.post['count'][j]) == k
        assert torch.equal(small.post['labels'][j, :k], l0[n, :k])
        assert torch.equal(small.post['dets'][j, :k], d0[n, :k])
    # NMS invariants on every image of the batch
    dets, labels, cnt = d0.cpu().numpy(), l0.cpu().numpy(), c0.cpu().numpy()
    for n in range(0, 256, 17):
        k = int(cnt[n])
        d, l = dets[n, :k], labels[n, :k]
        assert (d[:, 4] > cfg['score_thr']).all() and (np.diff(d[:, 4]) <= 0).all()
        for c in np.unique(l):
            b = d[l == c, :4]
            if len(b) > 1:
                x1 = np.maximum(b[:, None, 0], b[None, :, 0]); y1 = np.maximum(b[:, None, 1], b[None, :, 1])
                x2 = np.minimum(b[:, None, 2], b[None, :, 2]); y2 = np.minimum(b[:, None, 3], b[None, :, 3])
                inter = np.clip(x2 - x1, 0, None) * np.clip(y2 - y1, 0, None)
                area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
                iou = inter / (area[:, None] + area[None, :] - inter)
                np.fill_diagonal(iou, 0)
                assert iou.max() <= cfg['nms']['iou_threshold'] + 1e-6


# ---- fp32 train step at full depth vs the CPU oracle's autograd (the pinned arithmetic) --------------------------
def _oracle_train_grads(sd0, img, gtb, gtl, arch, neck, out_indices, dtype):
    """forward_train + autograd of the CPU oracle (oracle/yolov4_oracle.py:717 = single_stage.py:51-79 +
    yolocsp_head.py:384-575 restated) in `dtype`; returns (loss terms, total, {name: grad as float64})."""
    stages, reps = O.ARCH[arch]
    sd, params = {}, {}
    for k, v in sd0.items():
        v = v.detach().cpu()
        if v.is_floating_point():
            v = v.to(dtype)
        if v.is_floating_point() and 'running_' not in k:
            sd[k] = v.clone().requires_grad_(True)
            params[k] = sd[k]
        else:
            sd[k] = v.clone()
    L = O.forward_train(img.cpu().to(dtype), sd, stages, reps, out_indices, [b.cpu().to(dtype) for b in gtb],
                        [l.cpu() for l in gtl], neck=neck)
    total = O.total_loss(L)
    total.backward()
    terms = {k: float(sum(x.sum() for x in v)) for k, v in L.items() if k.startswith('loss')}
    stats = {k: v.detach().double() for k, v in sd.items() if 'running_' in k}
    return terms, float(total), {k: p.grad.double() for k, p in params.items()}, stats


@pytest.mark.parametrize('model,size,arch,neck,outs', [('yolov4l', 608, 'v4l5p', 'v4', [3, 4, 5]),
                                                        ('yolov5l', 640, 'v5l5p', 'v5', [2, 3, 4])],
                         ids=['yolov4l_608', 'yolov5l_640'])
def test_fullsize_fp32_train_step_vs_oracle_autograd(gpu_device, model, size, arch, neck, outs):
    """The fp32 HIP training step (forward, fused loss, data / weight gradients, train-mode BN backward) at the real
    depth and input size of configs[2] / configs[4], batch 2, against the ORACLE's forward_train + torch autograd on
    the CPU -- the arithmetic test_oracle_train_golden.py pins to the reference's own step.  At this depth two fp32
    evaluations of the same gradient differ by reassociation noise amplified through ~110 batch-statistics layers, so
    the statement is relative to the truth (the oracle in float64): per parameter tensor, the HIP gradient's distance
    from the float64 gradient (L2, relative to the float64 norm) is at most K_REL x the fp32 CPU oracle's own distance
    (+ a floor for tensors whose fp32 CPU error happens to be tiny); loss terms within 2e-4 of the float64 oracle."""
    K_REL, FLOOR = 3.0, 1e-4       # measured on MI355X: HIP 7.8e-4 / 1.0e-4 (median, v4l / v5l) vs the fp32 CPU oracle's 1.45e-3 / 7.6e-5
    B = 2
    torch.manual_seed(0)
    det = pkg.build_detector(bench.model_cfg(model))
    det.init_weights()
    det.train().to(gpu_device)
    img = bench.synthetic_images(B, size, 1000, gpu_device)
    gtb, gtl = bench.synthetic_gts(B, size, 2000, gpu_device)
    data = dict(img=img, img_metas=[dict() for _ in range(B)], gt_bboxes=gtb, gt_labels=gtl)
    sd0 = {k: v.detach().clone() for k, v in det.state_dict().items()}
    log, raws, total, _ = _train_step(det, data)
    total.backward()
    got = {n: p.grad.detach().double().cpu() for n, p in det.named_parameters()}
    t32, tot32, g32, _ = _oracle_train_grads(sd0, img, gtb, gtl, arch, neck, outs, torch.float32)
    t64, tot64, g64, st64 = _oracle_train_grads(sd0, img, gtb, gtl, arch, neck, outs, torch.float64)
    assert list(got) == list(g64)                                  # same parameter set, same order
    for k in ('loss_cls', 'loss_conf', 'loss_bbox'):
        np.testing.assert_allclose(log[k], t64[k], rtol=2e-4, err_msg=f'{k}: HIP fp32 step vs float64 oracle')
        np.testing.assert_allclose(t32[k], t64[k], rtol=2e-4, err_msg=f'{k}: fp32 oracle vs float64 oracle')
    np.testing.assert_allclose(log['loss'], tot64, rtol=2e-4)
    worst = (0.0, None)
    e_hip, e_cpu = [], []
    for n, t in g64.items():
        den = float(t.norm()) + 1e-30
        eh = float((got[n] - t).norm()) / den
        ec = float((g32[n] - t).norm()) / den
        e_hip.append(eh); e_cpu.append(ec)
        ratio = eh / (K_REL * ec + FLOOR)
        if ratio > worst[0]:
            worst = (ratio, n, eh, ec)
    e_hip, e_cpu = np.array(e_hip), np.array(e_cpu)
    print(f'{model} {size} fp32 step, {len(g64)} parameter tensors, |g - g64| / |g64|: HIP median {np.median(e_hip):.2e} '
          f'p90 {np.quantile(e_hip, 0.9):.2e} max {e_hip.max():.2e} | CPU fp32 oracle median {np.median(e_cpu):.2e} '
          f'p90 {np.quantile(e_cpu, 0.9):.2e} max {e_cpu.max():.2e} | worst ratio {worst}')
    assert worst[0] <= 1.0, worst
    assert np.median(e_hip) <= 2.0 * np.median(e_cpu) + 1e-6
    # BatchNorm running statistics after the step (momentum 0.03 / SPP block 0.1, Q1; unbiased variance) vs the
    # float64 oracle's in-place updated buffers
    bufs = dict(det.named_buffers())
    assert len(st64) >= 190                                        # 2 x (108 BatchNorms of YOLOv4-L / 99 of YOLOv5-L)
    for n, t in st64.items():
        np.testing.assert_allclose(bufs[n].double().cpu().numpy(), t.numpy(), rtol=2e-4, atol=2e-5, err_msg=n)
