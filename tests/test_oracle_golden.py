"""Pin the CPU oracle against the golden vectors generated from the reference
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import arch_from, state_dict_from
from oracle import yolov4_oracle as O

# The oracle and the reference both run torch CPU kernels; the only sources of difference
# are op ordering inside torch (none expected) -> tight tolerances.
TOL = dict(rtol=1e-5, atol=1e-5)


def test_mish_fixture_matches_c_restatement(golden):
    g = golden('mish')
    x, gr = g['x'], g['g']
    # reference C++ kernel (built from its own source, oracle/_ref) vs the plain-C restatement:
    # bit for bit, fp32 and fp64, forward and backward
    np.testing.assert_array_equal(O.mish_c(x), g['y'])
    np.testing.assert_array_equal(O.mish_bwd_c(gr, x), g['gin'])
    np.testing.assert_array_equal(O.mish_c(x.astype(np.float64)), g['y64'])
    np.testing.assert_array_equal(O.mish_bwd_c(gr.astype(np.float64), x.astype(np.float64)), g['gin64'])
    # torch statement used inside the model oracle
    np.testing.assert_allclose(O.mish(torch.from_numpy(x)).numpy(), g['y'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(O.mish_bwd(torch.from_numpy(gr), torch.from_numpy(x)).numpy(), g['gin'], rtol=1e-4, atol=5e-6)


def test_mish_known_points():
    x = np.array([0.0, 20.0, 25.0, -100.0], dtype=np.float32)
    y = O.mish_c(x)
    assert y[0] == 0.0 and y[1] == 20.0 and y[2] == 25.0 and abs(y[3]) < 1e-30


@pytest.mark.parametrize('name,neck', [('tiny_v4', 'v4'), ('tiny_v5', 'v5')])
def test_model_graph_matches_reference(golden, name, neck):
    g = golden(name)
    sd = state_dict_from(g)
    stages, reps, _ = arch_from(g)
    img = torch.from_numpy(g['img'])
    outi = [int(i) for i in g['meta_out_indices']]
    with torch.no_grad():
        all_stage = O.darknetcsp(img, sd, stages, reps, out_indices=range(len(stages)))
        for i, s in enumerate(all_stage):
            np.testing.assert_allclose(s.numpy(), g[f'stage{i}'], err_msg=f'stage{i}', **TOL)
        feats = O.darknetcsp(img, sd, stages, reps, out_indices=outi)
        for i, f in enumerate(feats):
            np.testing.assert_allclose(f.numpy(), g[f'feat{i}'], **TOL)
        nouts = (O.yolov4_neck if neck == 'v4' else O.yolov5_neck)(feats, sd)
        for i, f in enumerate(nouts):
            np.testing.assert_allclose(f.numpy(), g[f'neck{i}'], err_msg=f'neck{i}', **TOL)
        preds = O.head_forward(nouts, sd)
        for i, f in enumerate(preds):
            np.testing.assert_allclose(f.numpy(), g[f'pred{i}'], err_msg=f'pred{i}', **TOL)


@pytest.mark.parametrize('name', ['tiny_v4', 'tiny_v5'])
def test_get_bboxes_matches_reference(golden, name):
    g = golden(name)
    preds = [torch.from_numpy(g[f'pred{i}']) for i in range(3)]
    sf = g['scale_factors']
    boxes, conf, cls = O.decode_maps(preds, 80)
    np.testing.assert_array_equal(boxes.numpy(), g['dec_boxes'])
    res = O.get_bboxes(preds, sf, 80, rescale=True)
    res_nr = O.get_bboxes(preds, sf, 80, rescale=False)
    for n in range(2):
        # same torch ops on the same inputs: bit-exact
        np.testing.assert_array_equal(res[n][0].numpy(), g[f'dets{n}'])
        np.testing.assert_array_equal(res[n][1].numpy(), g[f'labels{n}'])
        np.testing.assert_array_equal(res_nr[n][0].numpy(), g[f'dets_norescale{n}'])
        np.testing.assert_array_equal(res_nr[n][1].numpy(), g[f'labels_norescale{n}'])
        assert res[n][1].dtype == torch.int64 and res[n][0].shape[1] == 5


def test_decode_c_crosscheck(golden):
    """decode_ref.c (plain C) vs the torch statement, one image of one fixture."""
    import ctypes
    g = golden('tiny_v4')
    preds = [torch.from_numpy(g[f'pred{i}']) for i in range(3)]
    boxes, conf, cls = O.decode_maps(preds, 80)
    base = O.base_anchors()
    off = 0
    for lvl, p in enumerate(preds):
        H, W = p.shape[-2:]
        nhwc = np.ascontiguousarray(p[0].permute(1, 2, 0).numpy())
        nb = H * W * 3
        ob = np.empty((nb, 4), np.float32); oc = np.empty(nb, np.float32); ocl = np.empty((nb, 80), np.float32)
        fp = ctypes.POINTER(ctypes.c_float)
        ba = np.ascontiguousarray(base[lvl].numpy())
        O.clib().oracle_decode_level(nhwc.ctypes.data_as(fp), H, W, 3, 80, O.DEFAULT_STRIDES[lvl], 0,
                                     ba.ctypes.data_as(fp), ob.ctypes.data_as(fp), oc.ctypes.data_as(fp),
                                     ocl.ctypes.data_as(fp), None)
        np.testing.assert_allclose(ob, boxes[0, off:off + nb].numpy(), rtol=1e-5, atol=1e-4)
        np.testing.assert_allclose(oc, conf[0, off:off + nb].numpy(), rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(ocl, cls[0, off:off + nb].numpy(), rtol=1e-6, atol=1e-7)
        off += nb


@pytest.mark.parametrize('tag', ['small', 'mid', 'split', 'empty'])
def test_multiclass_nms_fixture(golden, tag):
    g = golden('nms')
    b, s, thr = torch.from_numpy(g[f'{tag}_boxes']), torch.from_numpy(g[f'{tag}_scores']), float(g[f'{tag}_thr'])
    d, l, inds = O.multiclass_nms(b, s, thr, dict(type='nms', iou_threshold=0.65), 300, return_inds=True)
    np.testing.assert_array_equal(d.numpy(), g[f'{tag}_dets'])
    np.testing.assert_array_equal(l.numpy(), g[f'{tag}_labels'])
    np.testing.assert_array_equal(inds.numpy(), g[f'{tag}_inds'])
    if tag == 'empty':
        assert tuple(d.shape) == (0, 4) and l.dtype == torch.int64  # Q7


def test_nms_c_equals_numpy_statement():
    rng = np.random.RandomState(3)
    for n in (1, 2, 63, 64, 65, 700):
        b = rng.rand(n, 4).astype(np.float32) * 50
        b[:, 2:] = b[:, :2] + rng.rand(n, 2).astype(np.float32) * 30
        s = rng.rand(n).astype(np.float32)
        s[::3] = 0.5
        assert np.array_equal(O.nms_c(b, s, 0.4), O.nms_numpy(b, s, 0.4))
    assert O.nms_c(np.zeros((0, 4), np.float32), np.zeros(0, np.float32), 0.5).size == 0


def _pair_predicates(b, iou_thr):
    """Per boundary pair (rows 2k, 2k+1 of the fixture, class-offset applied): (ovr, division form, product form)
    in numpy fp32, the arithmetic of oracle/nms_ref.c restated once more."""
    a, c = b[0::2], b[1::2]
    f0 = np.float32(0)
    iw = np.maximum(f0, np.minimum(a[:, 2], c[:, 2]) - np.maximum(a[:, 0], c[:, 0]))
    ih = np.maximum(f0, np.minimum(a[:, 3], c[:, 3]) - np.maximum(a[:, 1], c[:, 1]))
    inter = iw * ih
    uni = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]) + (c[:, 2] - c[:, 0]) * (c[:, 3] - c[:, 1]) - inter
    ovr = inter / uni
    return ovr, ovr > iou_thr, inter > iou_thr * uni


@pytest.mark.parametrize('form,tag', [(0, 'div'), (1, 'mul')])
def test_nms_boundary_fixture(golden, form, tag):
    """Pairs whose IoU equals the fp32 threshold or straddles it by one rounding (tests/golden/nms_boundary.npz, made
    by the reference's multiclass_nms glue over the restated mmcv nms): the oracle under the documented definition
    (division = mmcv's CPU kernel) and under mmcv's CUDA-kernel predicate (product) returns what the fixture holds; the
    C loop and its numpy restatement agree; and every pair behaves as its category says -- including pairs on which
    the two mmcv kernels disagree in BOTH directions, i.e. a GPU run of the reference may differ from the documented
    definition exactly there."""
    g = golden('nms_boundary')
    b, s = torch.from_numpy(g['boxes']), torch.from_numpy(g['scores'])
    iou_thr = np.float32(g['iou_thr'])
    O.NMS_IOU_FORM = form
    try:
        d, l, inds = O.multiclass_nms(b, s, float(g['thr']), dict(type='nms', iou_threshold=float(g['iou_thr'])), -1,
                                      return_inds=True)
    finally:
        O.NMS_IOU_FORM = 0
    np.testing.assert_array_equal(d.numpy(), g[f'{tag}_dets'])
    np.testing.assert_array_equal(l.numpy(), g[f'{tag}_labels'])
    np.testing.assert_array_equal(inds.numpy(), g[f'{tag}_inds'])
    # category semantics on the class-offset boxes batched_nms actually compares
    names = [str(n) for n in g['category_names']]
    cat = g['category'][:-1:2]                                  # one entry per pair (the last box is the anchor box)
    labels = g['scores'][:-1, :-1].argmax(1)
    off = (labels.astype(np.float32) * (g['boxes'].max() + np.float32(1)))[:, None]
    ovr, div, mul = _pair_predicates(g['boxes'][:-1] + off, iou_thr)
    want = {'eq': (False, None), 'int_eq': (False, None), 'div_only': (True, False), 'mul_only': (False, True),
            'int_mul_only': (False, True), 'above': (True, True), 'below': (False, False)}
    for k, (wd, wm) in want.items():
        m = cat == names.index(k)
        assert m.sum() >= 15, (k, int(m.sum()))
        assert (div[m] == wd).all(), k
        if wm is not None:
            assert (mul[m] == wm).all(), k
        if 'eq' in k:
            assert (ovr[m] == iou_thr).all()
    assert (div != mul).sum() >= 60 and (div & ~mul).sum() >= 15 and (mul & ~div).sum() >= 40
    # the greedy loop in C and in numpy select identically on these boxes, either form
    flat = g['scores'][:, :-1].max(1)
    bo = np.concatenate([g['boxes'][:-1] + off, g['boxes'][-1:]], 0)
    assert np.array_equal(O.nms_c(bo, flat, iou_thr, form=form), O.nms_numpy(bo, flat, iou_thr, form=form))
    # B of a pair survives iff the form's predicate is false
    kept = set(O.nms_c(bo, flat, iou_thr, form=form).tolist())
    pred = mul if form else div
    for k in range(len(pred)):
        assert (2 * k + 1 in kept) == (not pred[k]) and 2 * k in kept


def test_base_anchor_known_answers():
    """Known answers of the shared base-anchor formula (the reference's own test for the v3
    generator, tests/test_utils/test_anchor.py:148-188, pins the same formula)."""
    ba = O.base_anchors([[(116, 90), (156, 198), (373, 326)], [(30, 61), (62, 45), (59, 119)],
                         [(10, 13), (16, 30), (33, 23)]], [32, 16, 8])
    exp0 = torch.tensor([[-42., -29., 74., 61.], [-62., -83., 94., 115.], [-170.5, -147., 202.5, 179.]])
    exp1 = torch.tensor([[-7., -22.5, 23., 38.5], [-23., -14.5, 39., 30.5], [-21.5, -51.5, 37.5, 67.5]])
    exp2 = torch.tensor([[-1., -2.5, 9., 10.5], [-4., -11., 12., 19.], [-12.5, -7.5, 20.5, 15.5]])
    for a, e in zip(ba, (exp0, exp1, exp2)):
        assert torch.equal(a, e)
    v4 = O.base_anchors()
    assert torch.equal(v4[0], torch.tensor([[-2., -4., 10., 12.], [-5.5, -14., 13.5, 22.], [-16., -10., 24., 18.]]))


@pytest.mark.parametrize('tag,agnostic', [('aware', False), ('agnostic', True)])
@pytest.mark.parametrize('nms_pre', [-1, 60, 250])
def test_get_bboxes_nms_pre_and_class_agnostic(golden, tag, agnostic, nms_pre):
    """yolocsp_head.py:349-360 variants (top-k by objectness, class-agnostic score) vs the reference."""
    g = golden('post_variants')
    preds = [torch.from_numpy(g[f'{tag}/pred{i}']) for i in range(3)]
    res = O.get_bboxes(preds, g['scale_factors'], int(g['num_classes']), score_thr=0.05, iou_threshold=0.5,
                       max_per_img=50, rescale=True, nms_pre=nms_pre, class_agnostic=agnostic)
    for n in range(2):
        np.testing.assert_array_equal(res[n][0].numpy(), g[f'{tag}/pre{nms_pre}/dets{n}'])
        np.testing.assert_array_equal(res[n][1].numpy(), g[f'{tag}/pre{nms_pre}/labels{n}'])
    if nms_pre == 60 and not agnostic:      # the pre-selection really changes the result
        assert not np.array_equal(g[f'{tag}/pre60/dets0'], g[f'{tag}/pre-1/dets0'])
