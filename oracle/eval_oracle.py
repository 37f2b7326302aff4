"""CPU restatement of the reference's detection-evaluation path (TEST INFRASTRUCTURE ONLY).

Only tests/, ``__graft_entry__.smoke()`` and bench/tools CPU-baseline legs may import this file; the
product (``mmdet-yolov4_amd/eval_utils.py``) never does.

Restates
  * ``iou_coco``   -- mmdet/ops/eval_utils/iou/iou_coco.pyx:8-56
  * ``match_coco`` -- mmdet/ops/eval_utils/match/match_coco.pyx:8-57
  * ``average_precision`` (mode 'area') -- mmdet/core/evaluation/mean_ap.py:12-58
  * ``eval_map_flexible`` and its statistics -- mmdet/core/evaluation/mean_ap_flexible.py:19-302
Pinned by tests/golden/eval.npz (outputs of the reference's own Cython ops compiled here by
oracle/build_ref.py and of its ``eval_map_flexible`` imported from /root/reference), and, where
``oracle/_ref`` holds the compiled ops, against them directly on random problems.
"""
import numpy as np


def iou_coco(det, gt, is_crowd):
    """iou_coco.pyx:8-56, fp32 arithmetic in the reference's order (numpy float32 scalars round each step)."""
    det = np.asarray(det, np.float32)
    gt = np.asarray(gt, np.float32)
    nd, ng = det.shape[0], gt.shape[0]
    out = np.zeros((nd, ng), np.float32)
    if nd == 0 or ng == 0:
        return out
    d_area = (det[:, 2:4] - det[:, 0:2]).prod(axis=-1)
    g_area = (gt[:, 2:4] - gt[:, 0:2]).prod(axis=-1)
    tlx = np.maximum(det[:, None, 0], gt[None, :, 0])
    tly = np.maximum(det[:, None, 1], gt[None, :, 1])
    brx = np.minimum(det[:, None, 2], gt[None, :, 2])
    bry = np.minimum(det[:, None, 3], gt[None, :, 3])
    inter = ((brx - tlx) * (bry - tly)).astype(np.float32)
    crowd = np.asarray(is_crowd, bool)[None, :]
    union = np.where(crowd, d_area[:, None], (d_area[:, None] + g_area[None, :]) - inter).astype(np.float32)
    union = np.where(union <= 0, np.float32(1e-7), union)
    val = inter / union
    return np.where((tlx >= brx) | (tly >= bry), np.float32(0), val).astype(np.float32)


def match_coco(iou, iou_thrs, is_ignore, is_crowd):
    """match_coco.pyx:8-57: per threshold, detections in order greedily take the best available gt."""
    iou = np.asarray(iou, np.float32)
    thrs = np.asarray(iou_thrs, np.float32)
    ign = np.asarray(is_ignore, bool)
    crowd = np.asarray(is_crowd, bool)
    nd, ng = iou.shape
    out = np.empty((len(thrs), nd), np.int32)
    for t, thr in enumerate(thrs):
        used = np.zeros(ng, bool)
        for d in range(nd):
            best = best_ign = thr
            mg = -1
            for g in range(ng):
                if used[g] and not crowd[g]:
                    continue
                if mg > -1 and not ign[mg] and ign[g]:
                    continue
                v = iou[d, g]
                if v < (best_ign if ign[g] else best):
                    continue
                if ign[g]:
                    best_ign = v
                else:
                    best = v
                mg = g
            if mg != -1:
                used[mg] = True
            out[t, d] = mg
    return out


def average_precision(recalls, precisions):
    """mean_ap.py:12-58, mode 'area', one curve: area under the monotone precision envelope, fp32 result."""
    mrec = np.concatenate(([0.], recalls, [1.]))
    mpre = np.concatenate(([0.], precisions, [0.]))
    mpre = np.maximum.accumulate(mpre[::-1])[::-1]
    ind = np.where(mrec[1:] != mrec[:-1])[0]
    return np.float32(np.sum((mrec[ind + 1] - mrec[ind]) * mpre[ind + 1]))


def _breakdown_rows(boxes, attrs, area_ranges, with_ignore):
    """NoBreakdown + ScaleBreakdown flags stacked (mean_ap_flexible.py:49-96): row 0 'All', then one per range."""
    n = len(boxes)
    rows = np.ones((1 + len(area_ranges), n), bool)
    if attrs is not None and 'area' in attrs:
        area = attrs['area']
    else:
        wh = boxes[:, 2:] - boxes[:, :2]
        area = wh[:, 0] * wh[:, 1]
    for i, (lo, hi) in enumerate(area_ranges):
        rows[1 + i] = (area >= lo) & (area < hi)
    if with_ignore and attrs is not None and 'ignore' in attrs:
        rows[:, attrs['ignore']] = False
    return rows


def eval_map_flexible(det_results, annotations, iou_thrs=(0.5,), scale_ranges=None, classes=None,
                      report_config=(('map', lambda k: k['breakdown'] == 'All'),), shared_tp=True):
    """mean_ap_flexible.py:279-302 with the breakdown configured as datasets/coco.py:469-495 does
    (``ScaleBreakdown`` from ``scale_ranges`` name -> (min_side, max_side)), IOU2DCoCo + MatcherCoCo.
    Returns (report dict, list of (key, value)) like FlexibleStatisticsEval.statistics_eval + report.

    ``shared_tp`` restates a property of the reference, not of the metric: statistics_single fills ONE
    ``cls_tp`` array in place for every breakdown (mean_ap_flexible.py:172,191-192) and appends that same
    object each time (:199-202), so every breakdown of an (image, class) problem ends up with the
    true-positive flags of the LAST breakdown's matching.  ``shared_tp=False`` gives each breakdown its own."""
    scale_ranges = scale_ranges or {}
    names = ['All'] + list(scale_ranges)
    ranges = [(a * a, b * b) for a, b in scale_ranges.values()]
    thrs = np.array(iou_thrs, np.float32)
    nt = len(thrs)
    num_cls = len(det_results[0])
    acc = {}                                          # (cls, bkd) -> [num_gt, [scores], [tp], [msk]]
    for det, anno in zip(det_results, annotations):
        gtb, gtl, attrs = anno['gt_bboxes'], anno['gt_labels'], anno['gt_attrs']
        for c in range(num_cls):
            sc = det[c][:, -1]
            order = sc.argsort()[::-1]
            db, sc = det[c][order, :-1], sc[order]
            m = gtl == c
            gb = gtb[m]
            ga = {k: v[m] for k, v in attrs.items()}
            ign = ga['ignore'] if 'ignore' in ga else np.zeros(len(gb), bool)
            crowd = ga['iscrowd'] if 'iscrowd' in ga else np.zeros(len(gb), bool)
            det_bkd = _breakdown_rows(db, None, ranges, False)
            gt_bkd = _breakdown_rows(gb, ga, ranges, True)
            tp = np.zeros((nt, len(db)), bool)
            ious = None
            if len(gb) and len(db):
                ious = iou_coco(db, gb, crowd)
            last_tp = None
            if ious is not None and shared_tp:
                last_tp = match_coco(ious, thrs, ~gt_bkd[-1], crowd.astype(bool)) > -1
            for b, name in enumerate(names):
                slot = acc.setdefault((c, b), [0, [], [], []])
                slot[0] += int(np.count_nonzero(gt_bkd[b]))
                slot[1].append(sc)
                if ious is None:
                    slot[2].append(tp.copy())
                    slot[3].append(det_bkd[b:b + 1].repeat(nt, axis=0))
                    continue
                mg = match_coco(ious, thrs, ~gt_bkd[b], crowd.astype(bool))
                is_tp = mg > -1
                fp = det_bkd[b:b + 1] & (mg == -1)
                tpm = gt_bkd[b][mg] & (mg > -1)
                slot[2].append(is_tp if last_tp is None else last_tp)
                slot[3].append(fp | tpm)
    results = []
    for (c, b), (num_gt, scores, tps, msks) in acc.items():
        score = np.concatenate(scores)
        tp = np.concatenate(tps, axis=1)
        msk = np.concatenate(msks, axis=1)
        rank = score.argsort()[::-1]
        tp, msk = tp[:, rank], msk[:, rank]
        for t in range(nt):
            cum = tp[t, msk[t]].cumsum()
            n = len(cum)
            recall = cum / max(num_gt, 1e-7)
            precision = cum / np.arange(1, n + 1)
            key = dict(class_name=classes[c] if classes is not None else c, breakdown=names[b],
                       iou_threshold=iou_thrs[t])
            results.append((key, dict(num_det=n, num_gt=num_gt, recall=recall.max() if n else 0,
                                      mAP=average_precision(recall, precision))))
    report = {}
    for name, cond in report_config:
        vals = [v['mAP'] for k, v in results if cond(k) and v['num_gt'] > 0]
        report[name] = np.mean(vals)
    return report, results
