"""Generate tests/golden/eval.npz: outputs of the REFERENCE's evaluation path on a synthetic dataset.

Run in the build container only:   python tests/golden/make_golden_eval.py
  * ``iou_coco`` / ``match_coco``: the reference's Cython sources compiled where they lie by
    oracle/build_ref.py (cython + gcc -> oracle/_ref/), called directly on seeded problems;
  * ``eval_map_flexible``: mmdet/core/evaluation/mean_ap_flexible.py imported from /root/reference, bound
    to those compiled ops, configured as datasets/coco.py:469-495 ('fast-bbox': ten IoU thresholds,
    S/M/L scale breakdown).  mmcv (absent) is represented by a Registry/build_from_cfg/progress stand-in
    with no arithmetic; ``np.bool`` (removed from numpy 2) is aliased to ``bool`` for the import.
The fixture is data only.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import _ref_import as R  # noqa: E402
from oracle import build_ref  # noqa: E402


def import_reference_eval():
    iou_ref, match_ref = build_ref.load_eval()
    if not hasattr(np, 'bool'):
        np.bool = bool
    mmcv = R._mod('mmcv')
    R._mod('mmcv.utils', Registry=R._Registry, build_from_cfg=R._build_from_cfg, print_log=lambda *a, **k: None)
    R._mod('mmcv.utils.progressbar', track_iter_progress=lambda x: x,
           track_parallel_progress=lambda func, tasks, nproc, chunksize=1: [func(t) for t in tasks])
    R._mod('terminaltables', AsciiTable=object)
    m = os.path.join(R.REF, 'mmdet')
    R._pkg('mmdet', m)
    for sub in ('ops', 'ops/eval_utils', 'core', 'core/evaluation'):
        R._pkg('mmdet.' + sub.replace('/', '.'), os.path.join(m, sub))
    R._mod('mmdet.ops.eval_utils.iou', iou_coco=iou_ref)
    R._mod('mmdet.ops.eval_utils.match', match_coco=match_ref)
    R._mod('mmdet.core.evaluation.bbox_overlaps', bbox_overlaps=None)
    R._mod('mmdet.core.evaluation.class_names', get_classes=None)
    import importlib
    return iou_ref, match_ref, importlib.import_module('mmdet.core.evaluation.mean_ap_flexible')


def boxes(rng, n, scale=400., max_side=160.):
    xy = rng.uniform(0, scale, (n, 2))
    wh = rng.uniform(2, max_side, (n, 2)) * rng.choice([0.15, 0.5, 1.0], (n, 1))
    return np.concatenate([xy, xy + wh], 1).astype(np.float32)


def synth_dataset(rng, num_img, num_cls):
    """gt + detections that are noisy copies of gts (varied quality) plus clutter, with crowd/ignore flags."""
    dets, annos = [], []
    for i in range(num_img):
        ng = int(rng.integers(0, 9)) if i % 7 else 0
        gb = boxes(rng, ng)
        gl = rng.integers(0, num_cls, ng)
        crowd = rng.random(ng) < 0.15
        ignore = crowd | (rng.random(ng) < 0.1)
        anno = dict(gt_bboxes=gb, gt_labels=gl, gt_attrs=dict(iscrowd=crowd, ignore=ignore))
        if i % 5 == 0:
            anno['gt_attrs'] = dict(iscrowd=crowd)
        per_cls = []
        for c in range(num_cls):
            mine = gb[gl == c]
            reps = [mine + rng.normal(0, s, mine.shape).astype(np.float32) for s in (1.0, 6.0, 20.0)]
            cand = np.concatenate(reps + [boxes(rng, int(rng.integers(0, 4)))], 0)
            keep = rng.random(len(cand)) < 0.8
            cand = cand[keep]
            if i % 11 == 3:
                cand = cand[:0]
            sc = np.round(rng.random(len(cand)), 2).astype(np.float32)       # exact score ties happen
            per_cls.append(np.concatenate([cand, sc[:, None]], 1).astype(np.float32))
        dets.append(per_cls)
        annos.append(anno)
    return dets, annos


def main():
    iou_ref, match_ref, F = import_reference_eval()
    rng = np.random.default_rng(20260711)
    out = {}
    # ---- direct op vectors ------------------------------------------------------------------------
    cases = [(13, 7), (1, 1), (40, 3), (5, 25), (64, 64)]
    thrs = np.array([0.5 + 0.05 * x for x in range(10)], np.float32)
    for k, (nd, ng) in enumerate(cases):
        g = boxes(rng, ng)
        d = boxes(rng, nd)
        n = min(nd, ng)
        d[:n] = g[:n] + rng.normal(0, 5, (n, 4)).astype(np.float32)
        if k == 0:
            d[0] = g[0]                                   # IoU exactly 1
            d[1, 2:] = d[1, :2]                           # zero-area detection
            g[2, 2:] = g[2, :2]                           # zero-area ground truth
            d[3] = [g[1, 2], g[1, 1], g[1, 2] + 10, g[1, 3]]  # touching edge: tlx == brx -> 0
        crowd = rng.random(ng) < 0.3
        ign = rng.random(ng) < 0.3
        iou = iou_ref(d, g, crowd)
        out[f'op{k}/det'], out[f'op{k}/gt'], out[f'op{k}/crowd'], out[f'op{k}/ignore'] = d, g, crowd, ign
        out[f'op{k}/iou'] = iou
        out[f'op{k}/match'] = match_ref(iou, thrs, ign, crowd)
        tied = (np.round(iou * 4) / 4).astype(np.float32)            # many exact ties and exact-threshold values
        out[f'op{k}/match_tied'] = match_ref(tied, thrs, ign, crowd)
    out['thrs'] = thrs
    # ---- the whole evaluation -----------------------------------------------------------------------
    num_img, num_cls = 48, 5
    dets, annos = synth_dataset(rng, num_img, num_cls)
    classes = [f'c{i}' for i in range(num_cls)]
    iou_list = [0.5 + 0.05 * x for x in range(10)]
    scale_ranges = dict(Scale_S=(0, 32), Scale_M=(32, 96), Scale_L=(96, 10000))
    fse = F.FlexibleStatisticsEval(classes, iou_list, [dict(type='ScaleBreakdown', scale_ranges=scale_ranges)],
                                   dict(type='IOU2DCoCo'), dict(type='MatcherCoCo'), 0)
    res = fse.statistics_eval(dets, annos)
    report = fse.report(res, [
        ('map', lambda x: x['breakdown'] == 'All'),
        ('map50', lambda x: x['iou_threshold'] == 0.5 and x['breakdown'] == 'All'),
        ('map75', lambda x: x['iou_threshold'] == 0.75 and x['breakdown'] == 'All'),
        ('s_map', lambda x: x['breakdown'] == 'Scale_S'),
        ('m_map', lambda x: x['breakdown'] == 'Scale_M'),
        ('l_map', lambda x: x['breakdown'] == 'Scale_L')])
    out['ds/num_img'], out['ds/num_cls'] = np.int64(num_img), np.int64(num_cls)
    for i, (det, anno) in enumerate(zip(dets, annos)):
        for c in range(num_cls):
            out[f'ds/det/{i}/{c}'] = det[c]
        out[f'ds/gt_bboxes/{i}'] = anno['gt_bboxes']
        out[f'ds/gt_labels/{i}'] = anno['gt_labels']
        for k, v in anno['gt_attrs'].items():
            out[f'ds/attr/{k}/{i}'] = v
    order = ['All', 'Scale_S', 'Scale_M', 'Scale_L']
    tab = np.array([[classes.index(k['class_name']), order.index(k['breakdown']),
                     int(round((k['iou_threshold'] - 0.5) / 0.05)), v['num_det'], v['num_gt']] for k, v in res],
                   np.int64)
    out['res/key'] = tab
    out['res/recall'] = np.array([v['recall'] for _, v in res], np.float64)
    out['res/mAP'] = np.array([v['mAP'] for _, v in res], np.float32)
    for k, v in report.items():
        out[f'report/{k}'] = np.float64(v)
    # ---- a problem on which the shared cls_tp array of statistics_single changes the result -----------
    qgt = np.array([[0, 0, 100, 85], [0, 0, 100, 93]], np.float32)
    qdet = np.array([[0, 0, 100, 88.5, .9], [0, 0, 100, 93, .8]], np.float32)
    fq = F.FlexibleStatisticsEval(['a'], [0.95], [dict(type='ScaleBreakdown', scale_ranges=scale_ranges)],
                                  dict(type='IOU2DCoCo'), dict(type='MatcherCoCo'), 0)
    rq = fq.statistics_eval([[qdet]], [dict(gt_bboxes=qgt, gt_labels=np.array([0, 0]), gt_attrs={})])
    out['quirk/gt'], out['quirk/det'] = qgt, qdet
    out['quirk/mAP'] = np.array([v['mAP'] for _, v in rq], np.float32)
    out['quirk/num_det'] = np.array([v['num_det'] for _, v in rq], np.int64)
    path = os.path.join(HERE, 'eval.npz')
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), {k: float(v) for k, v in report.items()})


if __name__ == '__main__':
    main()
