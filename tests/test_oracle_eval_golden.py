"""Evaluation row, CPU: the oracle restatement of iou_coco / match_coco / eval_map_flexible
against tests/golden/eval.npz (outputs of the reference's compiled Cython ops and of its
mean_ap_flexible.py) and, when oracle/_ref holds them, against the compiled ops on random problems."""
import numpy as np
import pytest

from _eval_data import REPORT, SCALES, THRS10, dataset, random_problem, result_table
from oracle import build_ref
from oracle import eval_oracle as E


def test_ops_against_fixture(golden):
    z = golden('eval')
    thrs = z['thrs']
    for k in range(5):
        d, g, crowd, ign = (z[f'op{k}/{n}'] for n in ('det', 'gt', 'crowd', 'ignore'))
        iou = E.iou_coco(d, g, crowd)
        assert iou.dtype == np.float32 and np.array_equal(iou, z[f'op{k}/iou'])          # bit-exact
        assert np.array_equal(E.match_coco(iou, thrs, ign, crowd), z[f'op{k}/match'])
        tied = (np.round(iou * 4) / 4).astype(np.float32)
        assert np.array_equal(E.match_coco(tied, thrs, ign, crowd), z[f'op{k}/match_tied'])
    assert z['op0/iou'][0, 0] == 1.0 and z['op0/iou'][3, 1] == 0.0


def test_eval_map_flexible_against_fixture(golden):
    z = golden('eval')
    dets, annos, classes = dataset(z)
    report, res = E.eval_map_flexible(dets, annos, THRS10, SCALES, classes, REPORT)
    key, recall, ap = result_table(res, classes)
    assert np.array_equal(key, z['res/key'])
    assert np.array_equal(recall, z['res/recall'])
    assert np.array_equal(ap, z['res/mAP'])
    for name, _ in REPORT:
        assert float(report[name]) == float(z[f'report/{name}'])


def test_shared_tp_quirk(golden):
    """The reference's statistics_single appends one in-place-updated cls_tp for every breakdown."""
    z = golden('eval')
    det, gt = [[z['quirk/det']]], [dict(gt_bboxes=z['quirk/gt'], gt_labels=np.array([0, 0]), gt_attrs={})]
    _, res = E.eval_map_flexible(det, gt, [0.95], SCALES, ['a'])
    assert np.array_equal(np.array([v['mAP'] for _, v in res], np.float32), z['quirk/mAP'])
    assert np.array_equal([v['num_det'] for _, v in res], z['quirk/num_det'])
    _, own = E.eval_map_flexible(det, gt, [0.95], SCALES, ['a'], shared_tp=False)
    assert own[0][1]['mAP'] == 1.0 and z['quirk/mAP'][0] == 0.5


def test_ops_against_compiled_reference():
    fns = build_ref.load_eval()
    if fns is None:
        pytest.skip('oracle/_ref holds no compiled eval ops and /root/reference is absent')
    iou_ref, match_ref = fns
    rng = np.random.default_rng(3)
    thrs = np.array([0.1, 0.3, 0.5, 0.75, 0.9], np.float32)
    for it in range(150):
        d, g, crowd, ign = random_problem(rng, int(rng.integers(1, 40)), int(rng.integers(1, 14)), it % 4 == 0)
        iou = iou_ref(d, g, crowd)
        assert np.array_equal(E.iou_coco(d, g, crowd), iou)
        if it % 2:
            iou = (np.round(iou * 8) / 8).astype(np.float32)
        assert np.array_equal(E.match_coco(iou, thrs, ign, crowd), match_ref(iou, thrs, ign, crowd))
