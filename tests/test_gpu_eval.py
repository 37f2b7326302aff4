"""Evaluation row, GPU: iou_coco / match_coco / eval_map_flexible through the C ABI
(yv4_iou_coco_batched, yv4_match_coco_batched) against the reference-made fixture (tests/golden/eval.npz, produced by the
reference's own compiled Cython ops in the build container; oracle/_ref itself does not travel) and the oracle.
Tolerance: none. IoU is compared bit for bit (fp32), matches and mAP tables exactly."""
import numpy as np
import pytest
import torch

import mmdet_yolov4_amd as pkg
from mmdet_yolov4_amd import eval_utils as EU
from _eval_data import REPORT, SCALES, THRS10, dataset, random_problem, result_table
from oracle import eval_oracle as E

pytestmark = pytest.mark.gpu


def test_ops_against_fixture(golden):
    z = golden('eval')
    thrs = z['thrs']
    for k in range(5):
        d, g, crowd, ign = (z[f'op{k}/{n}'] for n in ('det', 'gt', 'crowd', 'ignore'))
        iou = pkg.iou_coco(d, g, crowd)
        assert iou.dtype == np.float32 and iou.shape == (len(d), len(g))
        assert np.array_equal(iou, z[f'op{k}/iou'])
        m = pkg.match_coco(iou, thrs, ign, crowd)
        assert m.dtype == np.int32 and np.array_equal(m, z[f'op{k}/match'])
        tied = (np.round(iou * 4) / 4).astype(np.float32)
        assert np.array_equal(pkg.match_coco(tied, thrs, ign, crowd), z[f'op{k}/match_tied'])


def test_ops_random_against_oracle():
    """Random problems against the oracle's restatement (oracle/eval_oracle.py).  The reference's compiled Cython ops
    are NOT loaded on the GPU box: tests/golden/eval.npz (made from them by the committed generator) already pins both
    the oracle (tests/test_oracle_eval_golden.py) and the kernels (test above)."""
    rng = np.random.default_rng(11)
    thrs = np.array([0.1, 0.3, 0.5, 0.75, 0.9], np.float32)
    for it in range(60):
        d, g, crowd, ign = random_problem(rng, int(rng.integers(1, 300)), int(rng.integers(1, 40)), it % 4 == 0)
        iou = pkg.iou_coco(d, g, crowd)
        assert np.array_equal(iou, E.iou_coco(d, g, crowd))
        if it % 2:
            iou = (np.round(iou * 8) / 8).astype(np.float32)
        m = pkg.match_coco(iou, thrs, ign, crowd)
        if it < 20:
            assert np.array_equal(m, E.match_coco(iou, thrs, ign, crowd))


def test_empty_and_error_behaviour():
    e4 = np.zeros((0, 4), np.float32)
    b = np.array([[0, 0, 10, 10]], np.float32)
    assert pkg.iou_coco(e4, b, np.zeros(1, bool)).shape == (0, 1)
    assert pkg.iou_coco(b, e4, np.zeros(0, bool)).shape == (1, 0)
    m = pkg.match_coco(np.zeros((3, 0), np.float32), np.array([0.5], np.float32), np.zeros(0, bool), np.zeros(0, bool))
    assert np.array_equal(m, -np.ones((1, 3), np.int32))              # nothing to match: all -1
    assert pkg.match_coco(np.zeros((0, 2), np.float32), np.array([0.5], np.float32), np.zeros(2, bool),
                          np.zeros(2, bool)).shape == (1, 0)
    with pytest.raises(ValueError):                                   # the Cython buffer type check
        pkg.iou_coco(b.astype(np.float64), b, np.zeros(1, bool))
    with pytest.raises(ValueError):
        pkg.match_coco(np.zeros((1, 1)), np.array([0.5], np.float32), np.zeros(1, bool), np.zeros(1, bool))


def test_batched_equals_per_problem():
    """One launch over ragged problems (shared IoU blocks across breakdowns) == problem-by-problem calls."""
    rng = np.random.default_rng(5)
    dev = torch.device('cuda', 0)
    probs = [random_problem(rng, int(rng.integers(1, 50)), int(rng.integers(1, 12))) for _ in range(37)]
    nd = np.array([len(p[0]) for p in probs])
    ng = np.array([len(p[1]) for p in probs])
    det = torch.from_numpy(np.concatenate([p[0] for p in probs])).to(dev)
    gt = torch.from_numpy(np.concatenate([p[1] for p in probs])).to(dev)
    crowd = torch.from_numpy(np.concatenate([p[2] for p in probs])).to(dev)
    ign = torch.from_numpy(np.concatenate([p[3] for p in probs])).to(dev)
    det_off, gt_off = EU._offsets(nd), EU._offsets(ng)
    iou, iou_off = EU.iou_coco_batched(det, gt, crowd, det_off, gt_off)
    thrs = np.array(THRS10, np.float32)
    matched = EU.match_coco_batched(iou, det_off, gt_off, iou_off[:-1], thrs, ign, crowd).cpu().numpy()
    iou = iou.cpu().numpy()
    for k, (d, g, c, i) in enumerate(probs):
        ref_iou = E.iou_coco(d, g, c)
        assert np.array_equal(iou[iou_off[k]:iou_off[k + 1]].reshape(nd[k], ng[k]), ref_iou)
        got = matched[det_off[k] * 10:det_off[k + 1] * 10].reshape(10, nd[k])
        assert np.array_equal(got, E.match_coco(ref_iou, thrs, i, c))


def test_eval_map_flexible_against_fixture(golden):
    z = golden('eval')
    dets, annos, classes = dataset(z)
    fse = pkg.FlexibleStatisticsEval(classes, THRS10, [dict(type='ScaleBreakdown', scale_ranges=SCALES)],
                                     dict(type='IOU2DCoCo'), dict(type='MatcherCoCo'), 0)
    res = fse.statistics_eval(dets, annos)
    key, recall, ap = result_table(res, classes)
    assert np.array_equal(key, z['res/key'])
    assert np.array_equal(recall, z['res/recall'])
    assert np.array_equal(ap, z['res/mAP'])
    report = pkg.eval_map_flexible(dets, annos, iou_thrs=THRS10,
                                   breakdown=[dict(type='ScaleBreakdown', scale_ranges=SCALES)], classes=classes,
                                   report_config=REPORT, nproc=-1)
    assert list(report) == [n for n, _ in REPORT]
    for name, _ in REPORT:
        assert float(report[name]) == float(z[f'report/{name}'])
    # dataset.evaluate(metric='fast-bbox') (datasets/coco.py:464-496) = the same call with the recipe's constants
    fast = pkg.evaluate_fast_bbox(dets, annos, classes)
    assert list(fast) == list(report) and all(float(fast[k]) == float(report[k]) for k in fast)


def test_shared_tp_quirk_and_custom_matcher(golden):
    z = golden('eval')
    det, gt = [[z['quirk/det']]], [dict(gt_bboxes=z['quirk/gt'], gt_labels=np.array([0, 0]), gt_attrs={})]
    bk = [dict(type='ScaleBreakdown', scale_ranges=SCALES)]
    fse = pkg.FlexibleStatisticsEval(['a'], [0.95], bk, dict(type='IOU2DCoCo'), dict(type='MatcherCoCo'), 0)
    res = fse.statistics_eval(det, gt)
    assert np.array_equal(np.array([v['mAP'] for _, v in res], np.float32), z['quirk/mAP'])
    assert np.array_equal([v['num_det'] for _, v in res], z['quirk/num_det'])
    own = pkg.FlexibleStatisticsEval(['a'], [0.95], bk, dict(type='IOU2DCoCo'), dict(type='MatcherCoCo'), 0,
                                     shared_tp=False).statistics_eval(det, gt)
    assert own[0][1]['mAP'] == 1.0

    @pkg.EVAL_MATCHER.register_module()
    class PerProblemMatcher(pkg.MatcherCoCo):        # a user-registered matcher takes the per-problem route
        pass

    res2 = pkg.FlexibleStatisticsEval(['a'], [0.95], bk, dict(type='IOU2DCoCo'), dict(type='PerProblemMatcher'),
                                      0).statistics_eval(det, gt)
    assert [v for _, v in res2] == [v for _, v in res]


def test_full_size_properties():
    """COCO-val-sized problem table (5000 images x 80 classes): properties that need no oracle run --
    IoU in [0,1], a detection identical to a non-crowd gt is matched with IoU 1 at every threshold, a regular gt is
    matched at most once per threshold, and the result equals the oracle on a sampled subset of problems."""
    rng = np.random.default_rng(9)
    dev = torch.device('cuda', 0)
    P = 5000 * 80
    nd = rng.integers(0, 12, P)
    ng = rng.integers(0, 4, P)
    live = (nd > 0) & (ng > 0)
    nd, ng = nd[live], ng[live]
    P = len(nd)
    det_off, gt_off = EU._offsets(nd), EU._offsets(ng)
    xy = rng.uniform(0, 600, (int(gt_off[-1]), 2))
    gt = np.concatenate([xy, xy + rng.uniform(4, 200, xy.shape)], 1).astype(np.float32)
    xy = rng.uniform(0, 600, (int(det_off[-1]), 2))
    det = np.concatenate([xy, xy + rng.uniform(4, 200, xy.shape)], 1).astype(np.float32)
    det[det_off[:-1]] = gt[gt_off[:-1]]                       # first detection of a problem == its first gt
    crowd = rng.random(len(gt)) < 0.1
    crowd[gt_off[:-1]] = False
    ign = rng.random(len(gt)) < 0.1
    ign[gt_off[:-1]] = False
    t = lambda a: torch.from_numpy(a).to(dev)
    iou, iou_off = EU.iou_coco_batched(t(det), t(gt), t(crowd), det_off, gt_off)
    thrs = np.array(THRS10, np.float32)
    matched = EU.match_coco_batched(iou, det_off, gt_off, iou_off[:-1], thrs, t(ign), t(crowd)).cpu().numpy()
    iou = iou.cpu().numpy()
    assert iou.min() >= 0 and iou.max() <= 1
    assert np.all(iou[iou_off[:-1]] == 1.0)
    first = matched.reshape(-1)[(det_off[:-1] * 10)[:, None] + (np.arange(10) * nd[:, None])]
    assert np.all(first >= 0)                                 # an exact copy of gt 0 is matched at all thresholds,
    assert np.all(iou[iou_off[:-1, None] + first] == 1.0)     # to gt 0 or to a later (crowd) gt that also gives 1.0
    for k in rng.choice(P, 300, replace=False):
        blk = iou[iou_off[k]:iou_off[k + 1]].reshape(nd[k], ng[k])
        c, i = crowd[gt_off[k]:gt_off[k + 1]], ign[gt_off[k]:gt_off[k + 1]]
        assert np.array_equal(blk, E.iou_coco(det[det_off[k]:det_off[k + 1]], gt[gt_off[k]:gt_off[k + 1]], c))
        got = matched[det_off[k] * 10:det_off[k + 1] * 10].reshape(10, nd[k])
        assert np.array_equal(got, E.match_coco(blk, thrs, i, c))
        for row in got:
            hit = row[row >= 0]
            reg = hit[~c[hit]]
            assert len(np.unique(reg)) == len(reg)
