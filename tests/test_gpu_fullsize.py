"""YOLOv4-L at BASELINE.json's full size (608x608) on the GPU: parity against the CPU oracle on one
image (the oracle needs a few seconds for it) and the size-independent properties of the path
(determinism, batch-composition independence, plan == module-by-module composition, NMS invariants,
rescale consistency)."""
import os
import sys

import numpy as np
import pytest
import torch

import mmdet_yolov4_amd as pkg
from mmdet_yolov4_amd.calibrate import calibrate_bn
from oracle import yolov4_oracle as O

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (model config + synthetic inputs of the headline benchmark)

pytestmark = pytest.mark.gpu
SIZE = 608


def _make(gpu_device, batch, size):
    torch.manual_seed(0)
    det = pkg.build_detector(bench.model_cfg('yolov4l'))
    det.init_weights()
    det.eval().to(gpu_device)
    img = bench.synthetic_images(batch, size, 1000, gpu_device)
    plan = det.compile(batch, size, size, device=gpu_device, rescale=True)
    calibrate_bn(plan, img)                               # BN statistics fitted on the batch (modules updated)
    ncand = bench.init_head(det, plan, img, 1500.0)       # ~1500 NMS candidates per image
    assert 500 < ncand < 5000
    return det, img


@pytest.fixture(scope='module')
def v4l(gpu_device):
    return _make(gpu_device, 2, SIZE)


@pytest.fixture(scope='module')
def v4l_416(gpu_device):
    """BASELINE.json configs[0]'s workload: YOLOv4-L (CSPDarknet53-class backbone), 416x416, a single image."""
    return _make(gpu_device, 1, 416)


def _ious(b):
    x1 = np.maximum(b[:, None, 0], b[None, :, 0]); y1 = np.maximum(b[:, None, 1], b[None, :, 1])
    x2 = np.minimum(b[:, None, 2], b[None, :, 2]); y2 = np.minimum(b[:, None, 3], b[None, :, 3])
    inter = np.clip(x2 - x1, 0, None) * np.clip(y2 - y1, 0, None)
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    return inter / (area[:, None] + area[None, :] - inter)


def _assert_maps_float64_anchored(got, ref, ref64, size):
    """`got` (HIP), `ref` (fp32 CPU oracle), `ref64` (the oracle in float64): per level the HIP maps must be as close to
    float64 as the reference's own fp32 arithmetic is (mean <= 1.5 x, max <= 2.5 x), and within 5e-5 of the fp32 oracle
    in the mean of |d| / (1 + |logit|)."""
    for i, (a, b, t) in enumerate(zip(got, ref, ref64)):
        assert a.shape == b.shape == (1, 255, size // (8 << i), size // (8 << i))
        e_gpu = (a.double() - t).abs() / (1 + t.abs())
        e_cpu = (b.double() - t).abs() / (1 + t.abs())
        e_ab = (a - b).abs() / (1 + b.abs())
        print(f'level {i}: HIP vs fp64 max {float(e_gpu.max()):.2e} mean {float(e_gpu.mean()):.2e} | '
              f'CPU fp32 vs fp64 max {float(e_cpu.max()):.2e} mean {float(e_cpu.mean()):.2e} | '
              f'HIP vs CPU fp32 max {float(e_ab.max()):.2e} mean {float(e_ab.mean()):.2e}')
        assert float(e_gpu.mean()) <= 1.5 * float(e_cpu.mean()) + 1e-6
        assert float(e_gpu.max()) <= 2.5 * float(e_cpu.max()) + 1e-5
        assert float(e_ab.mean()) <= 5e-5


@pytest.mark.parametrize('case', ['608', '416_single_image'])
def test_fullsize_parity_with_the_oracle_on_one_image(request, case):
    """Case 416_single_image is BASELINE.json configs[0]'s workload (416x416, one image) through the HIP path.

    115 layers deep, two correct fp32 implementations no longer agree to 1e-4 on every logit: the
    rounding differences of each layer (summation order inside a K = 4608 dot product) are amplified by
    the layers after it.  The honest statement at full depth is therefore relative to the truth: the
    oracle evaluated in float64.  The HIP path must be as close to it as the fp32 CPU oracle (= the
    reference's own arithmetic) is; and what the north star names -- scores and box coordinates -- must
    agree with the fp32 oracle to 1e-4 for all but a vanishing fraction of entries."""
    det, img = request.getfixturevalue('v4l' if case == '608' else 'v4l_416')
    SIZE = img.shape[-1]
    sd = {k: v.detach().cpu() for k, v in det.state_dict().items()}
    sd64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in sd.items()}
    stages, reps = O.ARCH['v4l5p']
    one = img[:1]
    with torch.no_grad():
        got = [p.cpu() for p in det.forward_dummy(one)[0]]
    ref, _ = O.forward_pred_maps(one.cpu(), sd, stages, reps, [3, 4, 5], neck='v4')
    ref64, _ = O.forward_pred_maps(one.cpu().double(), sd64, stages, reps, [3, 4, 5], neck='v4')
    _assert_maps_float64_anchored(got, ref, ref64, SIZE)
    # scores and boxes (what the north star names), HIP decode of HIP maps vs oracle decode of oracle maps
    boxes_o, conf_o, cls_o = O.decode_maps(ref, 80)
    plan = pkg.Plan(img.device)
    views = [plan.add_input_nchw(*p.shape, name=f'p{i}', pad4=False) for i, p in enumerate(got)]
    det.bbox_head.emit_postprocess(plan, views, rescale=False, want_cls=True)
    plan.finalize()
    plan.run(*[p.to(img.device) for p in got])
    post = plan.post
    sc_h = (post['cls'][0] * post['conf'][0][:, None]).cpu()
    sc_o = cls_o[0] * conf_o[0][:, None]
    ds = (sc_h - sc_o).abs()
    print(f'scores: max diff {float(ds.max()):.2e}, fraction above 1e-4: {float((ds > 1e-4).float().mean()):.2e}')
    assert float(ds.max()) <= 3e-4 and float((ds > 1e-4).float().mean()) <= 1e-4, (float(ds.max()), float((ds > 1e-4).float().mean()))
    # boxes: w = (2 sigma(t))^2 * anchor turns a logit difference into up to 1.5 x the anchor size, so at
    # this depth coordinates differ by ~1e-3 of the box size between ANY two fp32 evaluations; measured
    # against the float64 decode, the HIP path must stay within 3x of the fp32 CPU oracle's own error
    boxes_t, _, _ = O.decode_maps(ref64, 80)
    bt = boxes_t[0]
    size = torch.maximum(bt[:, 2] - bt[:, 0], bt[:, 3] - bt[:, 1]).clamp_min(1.0)[:, None]
    e_h = (post['boxes'][0].cpu().double() - bt).abs() / size
    e_c = (boxes_o[0].double() - bt).abs() / size
    qs = torch.tensor([0.5, 0.99], dtype=torch.float64)
    qh, qc = torch.quantile(e_h.flatten(), qs), torch.quantile(e_c.flatten(), qs)
    print(f'boxes |diff|/size vs fp64: HIP median {float(qh[0]):.2e} p99 {float(qh[1]):.2e} max {float(e_h.max()):.2e} | '
          f'CPU fp32 median {float(qc[0]):.2e} p99 {float(qc[1]):.2e} max {float(e_c.max()):.2e}')
    assert float(qh[1]) <= 3.0 * float(qc[1]) + 1e-6 and float(e_h.max()) <= 3.0 * float(e_c.max()) + 1e-5
    assert float(qh[0]) <= 1e-4          # the typical coordinate is inside the north star's 1e-4
    # detections from the SAME pred maps: identical selection, boxes/scores within 1e-4
    metas = [dict(scale_factor=np.array([1.25, 1.5, 1.25, 1.5], dtype=np.float32))]
    res = det.bbox_head.get_bboxes([p.to(img.device) for p in ref], metas, rescale=True)[0]
    ores = O.get_bboxes(ref, [metas[0]['scale_factor']], 80, rescale=True)[0]
    np.testing.assert_array_equal(res[1].cpu().numpy(), ores[1].numpy())
    np.testing.assert_allclose(res[0].cpu().numpy(), ores[0].numpy(), rtol=1e-4, atol=1e-4)


def test_fullsize_determinism_and_batch_independence(v4l):
    det, img = v4l
    metas = [dict(scale_factor=np.ones(4, dtype=np.float32))] * 2
    r1 = det.simple_test(img, metas, rescale=True)
    r2 = det.simple_test(img, metas, rescale=True)
    swapped = det.simple_test(img.flip(0), metas, rescale=True)
    for n in range(2):
        for c in range(80):
            assert np.array_equal(r1[n][c], r2[n][c])                  # run-to-run bit-identical
            assert np.array_equal(r1[n][c], swapped[1 - n][c])         # an image's result ignores its neighbours


def test_fullsize_plan_equals_module_composition(v4l):
    det, img = v4l
    plan = det.compile(2, SIZE, SIZE, device=img.device, rescale=True)
    plan.run(img)
    fused = [v.buf.tensor.view(v.N, v.H, v.W, v.C).permute(0, 3, 1, 2).clone() for v in plan.pred_views]
    with torch.no_grad():
        step = det.bbox_head(det.neck(det.backbone(img)))[0]
    for a, b in zip(fused, step):
        assert torch.equal(a, b)          # same kernels, same K order, same tiles: bit-identical


def test_fullsize_nms_invariants(v4l):
    det, img = v4l
    cfg = det.bbox_head.test_cfg
    metas = [dict(scale_factor=np.ones(4, dtype=np.float32))] * 2
    res = det.simple_test(img, metas, rescale=False)
    for n in range(2):
        tot = 0
        for c in range(80):
            d = res[n][c]
            tot += len(d)
            if len(d) == 0:
                continue
            assert d.dtype == np.float32 and d.shape[1] == 5
            assert (d[:, 4] > cfg['score_thr']).all()
            assert (np.diff(d[:, 4]) <= 0).all()                       # per class sorted by score
            if len(d) > 1:
                iou = _ious(d[:, :4])
                np.fill_diagonal(iou, 0)
                assert iou.max() <= cfg['nms']['iou_threshold'] + 1e-6  # survivors do not suppress each other
        assert 0 < tot <= cfg['max_per_img']


def test_fullsize_rescale_divides_boxes_before_nms(v4l):
    det, img = v4l
    sf = np.array([2.0, 2.0, 2.0, 2.0], dtype=np.float32)            # uniform scale: IoUs unchanged -> same selection
    a = det.simple_test(img, [dict(scale_factor=np.ones(4, dtype=np.float32))] * 2, rescale=True)
    b = det.simple_test(img, [dict(scale_factor=sf)] * 2, rescale=True)
    for n in range(2):
        for c in range(80):
            assert a[n][c].shape == b[n][c].shape
            if len(a[n][c]):
                np.testing.assert_array_equal(a[n][c][:, 4], b[n][c][:, 4])
                np.testing.assert_allclose(a[n][c][:, :4] / 2.0, b[n][c][:, :4], rtol=1e-6)


def test_single_image_plan_splits_k_and_matches_the_batched_plan(v4l):
    """Batch-1 plans (the reference's benchmark protocol, tools/analysis_tools/benchmark.py:83-109) split the K loop of
    the layers with too few output tiles (yv4_conv_bn_act_fwd_splitk): the same convolution with another summation
    order -- pred maps within fp32 reassociation noise of the batch-2 plan's (measured 1.2e-5 of 1 + |logit| in the
    mean: what any two fp32 evaluations of 115 layers differ by, see the float64 test above), identical detections, and run-to-run bit-identical (slabs are added in order, no
    atomics)."""
    det, img = v4l
    p1 = det.compile(1, SIZE, SIZE, device=img.device, rescale=True)
    nsplit = sum(1 for o in p1.ops if o.kind == 'conv' and 'ksplit' in o.info['launch'])
    assert nsplit > 50
    p2 = det.compile(2, SIZE, SIZE, device=img.device, rescale=True)
    assert not any('ksplit' in o.info['launch'] for o in p2.ops if o.kind == 'conv')
    p2.run(img)
    ref = [v.buf.tensor.view(v.N, v.H, v.W, v.C)[:1].clone() for v in p2.pred_views]
    d2, l2, c2 = p2.post['dets'][0].clone(), p2.post['labels'][0].clone(), int(p2.post['count'][0])
    p1.run(img[:1])
    got = [v.buf.tensor.view(v.N, v.H, v.W, v.C).clone() for v in p1.pred_views]
    d1, l1, c1 = p1.post['dets'][0].clone(), p1.post['labels'][0].clone(), int(p1.post['count'][0])
    for a, b in zip(got, ref):
        e = (a - b).abs() / (1 + b.abs())
        print(f'split-K vs batched plan: max {float(e.max()):.2e} mean {float(e.mean()):.2e}')
        assert float(e.mean()) <= 5e-5 and float(e.max()) <= 2e-3      # the bounds HIP vs the fp32 CPU oracle is held to above
    # detections: scores move by ~1e-5, so near-ties may swap places or flip a borderline suppression; at least 90 %
    # of the 300 detections must have a partner with the same label, box within 0.01 px and score within 1e-4
    assert abs(c1 - c2) <= 3
    a, b = d1[:c1].cpu().numpy(), d2[:c2].cpu().numpy()
    la, lb = l1[:c1].cpu().numpy(), l2[:c2].cpu().numpy()
    matched = 0
    for i in range(c1):
        ok = (lb == la[i]) & (np.abs(b[:, :4] - a[i, :4]).max(1) <= 1e-2) & (np.abs(b[:, 4] - a[i, 4]) <= 1e-4)
        matched += bool(ok.any())
    print(f'detections with a partner: {matched} of {c1}')
    assert matched >= 0.9 * c1                      # measured 287 of 300: the max_per_img cut makes the tail of the list sensitive
    p1.run(img[:1])
    again = [v.buf.tensor.view(v.N, v.H, v.W, v.C) for v in p1.pred_views]
    assert all(torch.equal(a, b) for a, b in zip(again, got))


@pytest.mark.parametrize('dtype,bound', [(torch.float16, 2e-2), (torch.bfloat16, 1.5e-1)])
def test_single_image_16bit_plan_splits_k(v4l, dtype, bound):
    """The 16-bit form of the batch-1 protocol (yv4_conv_bn_act_fwd_h16_splitk): the single-image plan splits the deep
    layers' K loops, is run-to-run bit-identical, and agrees with the batch-2 plan of the same images to the noise ONE
    16-bit rounding flip per layer grows into over ~110 layers (the split sums fp32 partials in another order, so a few
    outputs per layer round the other way; mean of |diff| / (1 + |logit|) stated per type)."""
    det, img = v4l
    p1 = det.compile(1, SIZE, SIZE, device=img.device, rescale=True, dtype=dtype)
    nsplit = sum(1 for o in p1.ops if o.kind == 'conv' and 'ksplit' in o.info['launch'])
    assert nsplit > 40
    p2 = det.compile(2, SIZE, SIZE, device=img.device, rescale=True, dtype=dtype)
    assert not any('ksplit' in o.info['launch'] for o in p2.ops if o.kind == 'conv')
    p2.run(img)
    ref = [v.buf.tensor.view(v.N, v.H, v.W, v.C)[:1].float().clone() for v in p2.pred_views]
    p1.run(img[:1])
    got = [v.buf.tensor.view(v.N, v.H, v.W, v.C).float().clone() for v in p1.pred_views]
    for a, b in zip(got, ref):
        e = (a - b).abs() / (1 + b.abs())
        print(f'{dtype} split-K vs batched plan: max {float(e.max()):.2e} mean {float(e.mean()):.2e}')
        assert bool(torch.isfinite(a).all()) and float(e.mean()) <= bound
    p1.run(img[:1])
    again = [v.buf.tensor.view(v.N, v.H, v.W, v.C).float() for v in p1.pred_views]
    assert all(torch.equal(a, b) for a, b in zip(again, got))
    det._engines.clear()


def test_headline_config_at_its_batch_of_32(gpu_device):
    """BASELINE.json configs[1] as it is quoted: YOLOv4-L 608 x 608 fp32 at batch 32.  Tile choice depends on M (the wide
    kernels' fill rules, the 1x1 pick), so the batch-32 plan runs kernels the batch-2 fixture above never selects.
    (a) images 0 and 31 of the batch against the oracle, float64-anchored exactly as the batch-2 test; (b) images 0-1 of
    the batch-32 step bit-equal -- pred maps, detections, labels -- to a batch-2 plan of the same images pinned to the
    batch-32 plan's tile ids layer by layer (bench.py's output check, here under -m gpu)."""
    det, img = _make(gpu_device, 32, SIZE)
    plan = det.compile(32, SIZE, SIZE, device=gpu_device, rescale=True)
    plan.run(img)
    torch.cuda.synchronize()
    maps = [v.buf.tensor.view(v.N, v.H, v.W, v.C).permute(0, 3, 1, 2).clone() for v in plan.pred_views]
    dets, labels, count = plan.post['dets'].clone(), plan.post['labels'].clone(), plan.post['count'].clone()
    import ctypes
    pick = pkg._lib.lib().yv4_conv_pick_tile
    ids = [o.info['desc'].tile or pick(ctypes.byref(o.info['desc'])) for o in plan.ops if o.kind == 'conv' and 'desc' in o.info]
    names = sorted({pkg._lib.TILE_NAMES.get(t, str(t)) for t in ids})
    print('batch-32 plan, fp32 conv tile classes:', names)
    assert 'w3x3' in names and 'ws_1x1' in names          # the kernels the headline number is made of are the ones under test
    # (a) the oracle
    sd = {k: v.detach().cpu() for k, v in det.state_dict().items()}
    sd64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in sd.items()}
    stages, reps = O.ARCH['v4l5p']
    for n in (0, 31):
        one = img[n:n + 1].cpu()
        ref, _ = O.forward_pred_maps(one, sd, stages, reps, [3, 4, 5], neck='v4')
        ref64, _ = O.forward_pred_maps(one.double(), sd64, stages, reps, [3, 4, 5], neck='v4')
        print(f'image {n} of 32:')
        _assert_maps_float64_anchored([m[n:n + 1].cpu() for m in maps], ref, ref64, SIZE)
        if n == 0:
            ores = O.get_bboxes(ref, [[1.0, 1.0, 1.0, 1.0]], 80, rescale=True)[0]
            assert abs(int(ores[0].shape[0]) - int(count[0])) <= 2       # near-ties at the max_per_img cut
    # (b) the batch-2 plan, pinned
    small = det.compile(2, SIZE, SIZE, device=gpu_device, rescale=True)
    note = bench.pin_check_plan(plan, small)
    assert note == '', note
    small.run(img[:2])
    torch.cuda.synchronize()
    for v32, v2 in zip(maps, small.pred_views):
        assert torch.equal(v32[:2], v2.buf.tensor.view(v2.N, v2.H, v2.W, v2.C).permute(0, 3, 1, 2))
    for n in range(2):
        k = int(count[n])
        assert k == int(small.post['count'][n]) and k > 0
        assert torch.equal(small.post['dets'][n, :k], dets[n, :k])
        assert torch.equal(small.post['labels'][n, :k], labels[n, :k])
    det._engines.clear()
