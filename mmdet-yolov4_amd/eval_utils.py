"""Detection evaluation: ``iou_coco`` / ``match_coco`` and ``eval_map_flexible`` under the reference's names.

Mirror of ``mmdet/ops/eval_utils/iou/iou_coco.pyx:58`` and ``match/match_coco.pyx:59`` (numpy in, numpy
out, one problem per call) and of ``mmdet/core/evaluation/mean_ap_flexible.py`` (registries
``EVAL_BREAKDOWN`` / ``EVAL_IOU_CALCULATOR`` / ``EVAL_MATCHER``, ``IOU2DCoCo``, ``MatcherCoCo``,
``ScaleBreakdown``, ``FlexibleStatisticsEval``, ``eval_map_flexible``; selected by
``metric='fast-bbox'`` in ``datasets/coco.py:464-496``).

Where the reference walks the (image, class) problems one by one (optionally over a process pool) and
calls the two Cython ops per problem, ``FlexibleStatisticsEval.statistics_eval`` here gathers every
problem of the dataset into one table and evaluates it with ONE ``yv4_iou_coco_batched`` launch and ONE
``yv4_match_coco_batched`` launch (all breakdowns and thresholds in parallel).  Sorting by score keeps
numpy's ``argsort()[::-1]`` on the host so that exact score ties order as in the reference.  There is no
CPU implementation of the two ops in this package.
"""
from collections import OrderedDict

import numpy as np
import torch

from . import _lib
from ._lib import check
from .ops import stream_ptr
from .registry import Registry, build_from_cfg

EVAL_BREAKDOWN = Registry('Evaluation Breakdown')
EVAL_IOU_CALCULATOR = Registry('Evaluation IOU calculator')
EVAL_MATCHER = Registry('Evaluation Matcher')


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError('eval_utils runs on the GPU through libyv4_hip.so; no GPU is visible '
                           '(there is no CPU fallback for this path)')
    return torch.device('cuda', torch.cuda.current_device())


def _dev(a, dtype, dev):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).to(dev, non_blocking=False)


def _offsets(counts):
    off = np.zeros(len(counts) + 1, np.int64)
    np.cumsum(counts, out=off[1:])
    return off


def iou_coco_batched(det, gt, is_crowd, det_off, gt_off):
    """IoU blocks of P problems.  det (D,4) / gt (G,4) float32 and is_crowd (G,) bool torch tensors on the
    GPU, det_off / gt_off (P+1,) int64 numpy.  Returns (iou flat float32 device tensor, iou_off numpy)."""
    nd, ng = np.diff(det_off), np.diff(gt_off)
    iou_off = _offsets(nd * ng)
    dev = det.device
    iou = torch.empty(int(iou_off[-1]), dtype=torch.float32, device=dev)
    if iou.numel():
        t_do, t_go, t_io = (_dev(o, np.int64, dev) for o in (det_off, gt_off, iou_off))
        crowd = is_crowd.to(torch.uint8)
        check(_lib.lib().yv4_iou_coco_batched(det.data_ptr(), gt.data_ptr(), crowd.data_ptr(), t_do.data_ptr(),
                                              t_go.data_ptr(), t_io.data_ptr(), len(nd), int(iou_off[-1]),
                                              iou.data_ptr(), stream_ptr()), 'yv4_iou_coco_batched')
    return iou, iou_off


def match_coco_batched(iou, det_off, gt_off, iou_off, iou_thrs, is_ignore, is_crowd):
    """Greedy COCO matching of Q problems over one flat IoU buffer.  det_off / gt_off are (Q+1,) int64
    cumulative tables (they place each problem's output and its is_ignore / is_crowd slice), iou_off (Q,)
    the start of each problem's IoU block (blocks may be shared between problems).  Returns the flat int32
    device tensor ``matched``: problem q's (num_thrs, num_det) block starts at det_off[q] * num_thrs."""
    dev = iou.device
    Q = len(det_off) - 1
    nt = len(iou_thrs)
    matched = torch.empty(int(det_off[-1]) * nt, dtype=torch.int32, device=dev)
    if Q == 0 or matched.numel() == 0:
        return matched
    work = torch.empty(max(int(gt_off[-1]) * nt, 1), dtype=torch.uint8, device=dev)
    t_do, t_go = _dev(det_off, np.int64, dev), _dev(gt_off, np.int64, dev)
    t_io = _dev(np.append(iou_off[:Q], 0), np.int64, dev)
    thr = _dev(iou_thrs, np.float32, dev)
    ign, crowd = is_ignore.to(torch.uint8), is_crowd.to(torch.uint8)
    if ign.numel() == 0:                                   # every problem has zero gts
        ign = crowd = torch.zeros(1, dtype=torch.uint8, device=dev)
    check(_lib.lib().yv4_match_coco_batched(iou.data_ptr(), t_do.data_ptr(), t_go.data_ptr(), t_io.data_ptr(),
                                            thr.data_ptr(), nt, ign.data_ptr(), crowd.data_ptr(), Q,
                                            work.data_ptr(), matched.data_ptr(), stream_ptr()),
          'yv4_match_coco_batched')
    return matched


def iou_coco(det_boxes, gt_boxes, is_crowd):
    """iou_coco.pyx:58: (num_det, 4), (num_gt, 4) float32 and (num_gt,) bool -> (num_det, num_gt) float32."""
    det_boxes, gt_boxes = np.asarray(det_boxes), np.asarray(gt_boxes)
    if det_boxes.dtype != np.float32 or gt_boxes.dtype != np.float32:
        raise ValueError("Buffer dtype mismatch, expected 'npy_float32'")      # the Cython signature's check
    if det_boxes.ndim != 2 or gt_boxes.ndim != 2:
        raise ValueError('Buffer has wrong number of dimensions (expected 2)')
    nd, ng = det_boxes.shape[0], gt_boxes.shape[0]
    if nd == 0 or ng == 0:
        return np.zeros((nd, ng), np.float32)
    dev = _device()
    iou, _ = iou_coco_batched(_dev(det_boxes[:, :4], np.float32, dev), _dev(gt_boxes[:, :4], np.float32, dev),
                              _dev(is_crowd, np.bool_, dev), np.array([0, nd]), np.array([0, ng]))
    return iou.view(nd, ng).cpu().numpy()


def match_coco(iou_mat, iou_thrs, is_ignore, is_crowd):
    """match_coco.pyx:59: (num_det, num_gt) float32 IoU, (num_thrs,) float32, two (num_gt,) bool ->
    (num_thrs, num_det) int32 index of the matched gt or -1."""
    iou_mat, iou_thrs = np.asarray(iou_mat), np.asarray(iou_thrs)
    if iou_mat.dtype != np.float32 or iou_thrs.dtype != np.float32:
        raise ValueError("Buffer dtype mismatch, expected 'npy_float32'")
    nd, ng = iou_mat.shape
    nt = iou_thrs.shape[0]
    if nd == 0 or nt == 0:
        return np.empty((nt, nd), np.int32)
    dev = _device()
    out = match_coco_batched(_dev(iou_mat, np.float32, dev), np.array([0, nd]), np.array([0, ng]), np.array([0]),
                             iou_thrs, _dev(is_ignore, np.bool_, dev), _dev(is_crowd, np.bool_, dev))
    return out.view(nt, nd).cpu().numpy()


def average_precision(recalls, precisions, mode='area'):
    """mean_ap.py:12-58: area under the monotone precision envelope ('area') or the 11-point average."""
    recalls, precisions = np.asarray(recalls), np.asarray(precisions)
    single = recalls.ndim == 1
    if single:
        recalls, precisions = recalls[None], precisions[None]
    assert recalls.shape == precisions.shape and recalls.ndim == 2
    ap = np.zeros(recalls.shape[0], np.float32)
    for i, (r, p) in enumerate(zip(recalls, precisions)):
        if mode == 'area':
            mrec = np.concatenate(([0], r, [1])).astype(recalls.dtype)
            mpre = np.concatenate(([0], p, [0])).astype(recalls.dtype)
            mpre = np.maximum.accumulate(mpre[::-1])[::-1]
            step = np.nonzero(mrec[1:] != mrec[:-1])[0]
            ap[i] = np.sum((mrec[step + 1] - mrec[step]) * mpre[step + 1])
        elif mode == '11points':
            for thr in np.arange(0, 1 + 1e-3, 0.1):
                sel = p[r >= thr]
                ap[i] += sel.max() if sel.size else 0
            ap[i] /= 11
        else:
            raise ValueError('Unrecognized mode, only "area" and "11points" are supported')
    return ap[0] if single else ap


@EVAL_IOU_CALCULATOR.register_module()
class IOU2DCoCo:
    """mean_ap_flexible.py:19-25."""

    def __call__(self, det_bboxes, gt_bboxes, gt_iscrowd=None):
        if gt_iscrowd is None:
            gt_iscrowd = np.zeros(gt_bboxes.shape[0], dtype=bool)
        return iou_coco(det_bboxes, gt_bboxes, gt_iscrowd)


@EVAL_MATCHER.register_module()
class MatcherCoCo:
    """mean_ap_flexible.py:28-36."""

    def __call__(self, ious, iou_thrs, gt_isignore=None, gt_iscrowd=None):
        if gt_iscrowd is None:
            gt_iscrowd = np.zeros(ious.shape[1], dtype=bool)
        if gt_isignore is None:
            gt_isignore = np.zeros(ious.shape[1], dtype=bool)
        return match_coco(ious, iou_thrs, gt_isignore, gt_iscrowd)


class NoBreakdown:
    """mean_ap_flexible.py:39-67: the single 'All' group; ignored gts drop out of it."""

    def __init__(self, classes, apply_to=None, *args, **kwargs):
        self.classes = classes
        self.apply_to = classes if apply_to is None else apply_to
        self.names = ['All']

    def breakdown_flags(self, boxes, attrs=None):
        flags = np.ones((1, len(boxes)), dtype=bool)
        if attrs is not None and 'ignore' in attrs:
            flags[:, attrs['ignore']] = False
        return flags

    def breakdown(self, boxes, label, attrs=None):
        flags = self.breakdown_flags(boxes, attrs)
        return flags if self.classes[label] in self.apply_to else flags[:0]

    def breakdown_names(self, label):
        return [f'{n}' for n in self.names] if self.classes[label] in self.apply_to else []


@EVAL_BREAKDOWN.register_module()
class ScaleBreakdown(NoBreakdown):
    """mean_ap_flexible.py:70-96: one group per (min_side, max_side) range, by box area."""

    def __init__(self, scale_ranges, classes, apply_to=None, *args, **kwargs):
        super().__init__(classes, apply_to, *args, **kwargs)
        self.names = list(scale_ranges)
        self.area_ranges = [(lo * lo, hi * hi) for lo, hi in scale_ranges.values()]

    def breakdown_flags(self, boxes, attrs=None):
        if attrs is not None and 'area' in attrs:
            area = attrs['area']
        else:
            wh = boxes[:, 2:] - boxes[:, :2]
            area = wh[:, 0] * wh[:, 1]
        flags = np.zeros((len(self.area_ranges), len(boxes)), dtype=bool)
        for i, (lo, hi) in enumerate(self.area_ranges):
            flags[i][(area >= lo) & (area < hi)] = True
        if attrs is not None and 'ignore' in attrs:
            flags[:, attrs['ignore']] = False
        return flags


class FlexibleStatisticsEval(object):
    """mean_ap_flexible.py:99-276.  ``nproc`` is accepted and ignored: the per-problem work the reference
    spreads over a process pool is one batched GPU launch here.  A custom ``iou_calculator`` / ``matcher``
    (anything but IOU2DCoCo / MatcherCoCo) is called per problem exactly as the reference does.

    ``shared_tp`` (default True = the reference's behaviour): statistics_single fills one ``cls_tp`` array
    in place per breakdown (:172,191-192) and appends that same object every time (:199-202), so all
    breakdowns of an (image, class) problem carry the true-positive flags of the LAST breakdown's matching.
    ``shared_tp=False`` gives every breakdown the flags of its own matching."""

    def __init__(self, classes, iou_thrs, breakdown, iou_calculator, matcher, nproc, shared_tp=True):
        self.shared_tp = shared_tp
        self.classes = classes
        self.iou_thrs = iou_thrs
        self.breakdown = [NoBreakdown(classes)]
        self.breakdown += [build_from_cfg(b, EVAL_BREAKDOWN, default_args=dict(classes=classes))
                           for b in breakdown]
        self.iou_calculator = build_from_cfg(iou_calculator, EVAL_IOU_CALCULATOR)
        self.matcher = build_from_cfg(matcher, EVAL_MATCHER)
        self.nproc = nproc

    # ---- host: the (image, class) problems ------------------------------------------------------------
    def _problems(self, det_results, annotations):
        """Per (image, class): score-sorted detections, the class's gts, flags and breakdown masks."""
        probs = []
        for det, anno in zip(det_results, annotations):
            gt_bboxes, gt_labels, gt_attrs = anno['gt_bboxes'], anno['gt_labels'], anno['gt_attrs']
            for cls in range(len(det)):
                scores = det[cls][:, -1]
                order = scores.argsort()[::-1]
                boxes = det[cls][order, :-1]
                scores = scores[order]
                msk = gt_labels == cls
                gtb = gt_bboxes[msk]
                attrs = {k: v[msk] for k, v in gt_attrs.items()}
                crowd = attrs['iscrowd'] if 'iscrowd' in attrs else np.zeros(len(gtb), dtype=bool)
                det_bkd = np.concatenate([f.breakdown(boxes, cls) for f in self.breakdown], axis=0)
                gt_bkd = np.concatenate([f.breakdown(gtb, cls, attrs) for f in self.breakdown], axis=0)
                names = sum([f.breakdown_names(cls) for f in self.breakdown], [])
                probs.append((cls, boxes, scores, gtb, np.asarray(crowd, dtype=bool), det_bkd, gt_bkd, names))
        return probs

    def _match_all(self, probs):
        """One IoU launch + one matching launch for every problem with both detections and gts.
        Returns {problem index: (num_bkd, num_thrs, num_det) int32}."""
        live = [i for i, p in enumerate(probs) if len(p[1]) and len(p[3])]
        if not live:
            return {}
        dev = _device()
        nt = len(self.iou_thrs)
        nd = np.array([len(probs[i][1]) for i in live], np.int64)
        ng = np.array([len(probs[i][3]) for i in live], np.int64)
        nb = np.array([probs[i][6].shape[0] for i in live], np.int64)
        det = _dev(np.concatenate([probs[i][1][:, :4] for i in live]), np.float32, dev)
        gt = _dev(np.concatenate([probs[i][3][:, :4] for i in live]), np.float32, dev)
        crowd = _dev(np.concatenate([probs[i][4] for i in live]), np.bool_, dev)
        iou, iou_off = iou_coco_batched(det, gt, crowd, _offsets(nd), _offsets(ng))
        # matching problems: (problem, breakdown); the breakdowns of a problem share its IoU block
        q_det, q_gt = np.repeat(nd, nb), np.repeat(ng, nb)
        q_iou = np.repeat(iou_off[:-1], nb)
        ignore = _dev(np.concatenate([~probs[i][6].reshape(-1) for i in live]), np.bool_, dev)
        q_crowd = _dev(np.concatenate([np.tile(probs[i][4], probs[i][6].shape[0]) for i in live]), np.bool_, dev)
        q_det_off = _offsets(q_det)
        matched = match_coco_batched(iou, q_det_off, _offsets(q_gt), q_iou,
                                     np.array(self.iou_thrs, dtype=np.float32), ignore, q_crowd).cpu().numpy()
        out, q = {}, 0
        for k, i in enumerate(live):
            lo = q_det_off[q] * nt
            out[i] = matched[lo:lo + nb[k] * nt * nd[k]].reshape(nb[k], nt, nd[k])
            q += nb[k]
        return out

    def _custom_match_all(self, probs):
        thrs = np.array(self.iou_thrs, dtype=np.float32)
        out = {}
        for i, (cls, boxes, scores, gtb, crowd, det_bkd, gt_bkd, names) in enumerate(probs):
            if len(boxes) and len(gtb):
                ious = self.iou_calculator(boxes, gtb, crowd)
                out[i] = np.stack([self.matcher(ious, thrs, ~m, crowd) for m in gt_bkd]) if len(gt_bkd) else \
                    np.empty((0, len(thrs), len(boxes)), np.int32)
        return out

    def statistics_eval(self, det_results, annotations):
        """mean_ap_flexible.py:119-208 (per problem) and :225-260 (accumulate over the dataset)."""
        nt = len(self.iou_thrs)
        probs = self._problems(det_results, annotations)
        batched = type(self.iou_calculator) is IOU2DCoCo and type(self.matcher) is MatcherCoCo
        matched = self._match_all(probs) if batched else self._custom_match_all(probs)
        groups = OrderedDict()                       # (class index, breakdown row) -> accumulators
        for i, (cls, boxes, scores, gtb, crowd, det_bkd, gt_bkd, names) in enumerate(probs):
            cls_name = self.classes[cls] if self.classes is not None else cls
            for b in range(gt_bkd.shape[0]):
                g = groups.setdefault((cls, b), [cls_name, names[b], 0, [], [], []])
                g[2] += int(np.count_nonzero(gt_bkd[b]))
                g[3].append(scores)
                if i in matched:
                    mg = matched[i][b]
                    g[4].append((matched[i][-1] if self.shared_tp else mg) > -1)
                    g[5].append((det_bkd[b:b + 1] & (mg == -1)) | (gt_bkd[b][mg] & (mg > -1)))
                else:
                    g[4].append(np.zeros((nt, len(boxes)), dtype=bool))
                    g[5].append(det_bkd[b:b + 1].repeat(nt, axis=0))
        results = []
        for cls_name, bkd, num_gt, scores, tps, msks in groups.values():
            results += self.statistics_accumulate((cls_name, bkd, num_gt, np.concatenate(scores, axis=0),
                                                   np.concatenate(tps, axis=1), np.concatenate(msks, axis=1)))
        return results

    def statistics_accumulate(self, input):
        """mean_ap_flexible.py:210-227."""
        cls, bkd, num_gt, score, tp, bkd_msk = input
        rank = score.argsort()[::-1]
        tp, bkd_msk = tp[:, rank], bkd_msk[:, rank]
        out = []
        for t, iou_thr in enumerate(self.iou_thrs):
            tpcumsum = tp[t, bkd_msk[t]].cumsum()
            num_det = len(tpcumsum)
            recall = tpcumsum / max(num_gt, 1e-7)
            precision = tpcumsum / np.arange(1, num_det + 1)
            out.append((dict(class_name=cls, breakdown=bkd, iou_threshold=iou_thr),
                        dict(num_det=num_det, num_gt=num_gt, recall=recall.max() if len(recall) > 0 else 0,
                             mAP=average_precision(recall, precision))))
        return out

    def report(self, eval_result_list, group_by):
        """mean_ap_flexible.py:265-276."""
        report_dict = OrderedDict()
        for name, cond in group_by:
            report_dict[name] = np.mean([v['mAP'] for k, v in eval_result_list if cond(k) and v['num_gt'] > 0])
        return report_dict


def eval_map_flexible(det_results, annotations, iou_thrs=[0.5], breakdown=[], iou_calculator=dict(type='IOU2DCoCo'),
                      matcher=dict(type='MatcherCoCo'), classes=None, logger=None,
                      report_config=[('map', lambda x: x['breakdown'] == 'All')], nproc=None):
    """mean_ap_flexible.py:279-302."""
    assert len(det_results) == len(annotations)
    fse = FlexibleStatisticsEval(classes, iou_thrs, breakdown, iou_calculator, matcher, nproc or 0)
    return fse.report(fse.statistics_eval(det_results, annotations), report_config)


def coco_test_annotation(ann_info, cat_ids, cat2label):
    """One image's COCO annotation records -> the dict ``eval_map_flexible`` consumes
    (``CocoDataset.get_ann_info_test``, datasets/coco.py:357-409, without the mask / seg-map fields the bbox
    evaluation never reads).  ``ann_info``: the image's ``annotations`` entries (``bbox`` = x, y, w, h);
    crowd boxes and boxes of unlisted categories are kept and flagged ``ignore``."""
    boxes, labels, ignore, crowd, area = [], [], [], [], []
    for ann in ann_info:
        is_crowd = ann.get('iscrowd', False)
        ignore.append(bool(ann.get('ignore', False) or is_crowd or ann['category_id'] not in cat_ids))
        crowd.append(bool(is_crowd))
        area.append(ann['area'])
        x, y, w, h = ann['bbox']
        boxes.append([x, y, x + w, y + h])
        labels.append(cat2label[ann['category_id']])
    return dict(gt_bboxes=np.array(boxes, dtype=np.float32).reshape(-1, 4), gt_labels=np.array(labels, dtype=np.int64),
                gt_attrs=dict(ignore=np.array(ignore, dtype=bool), iscrowd=np.array(crowd, dtype=bool),
                              area=np.array(area, dtype=np.float32)))


def evaluate_fast_bbox(results, annotations, classes, logger=None):
    """``dataset.evaluate(results, metric='fast-bbox')`` (datasets/coco.py:464-496): the ten COCO IoU thresholds, the
    three object-scale breakdowns, and the six-number report ``map, map50, map75, s_map, m_map, l_map``."""
    return eval_map_flexible(
        results, annotations, iou_thrs=[0.5 + 0.05 * x for x in range(10)],
        breakdown=[dict(type='ScaleBreakdown', scale_ranges=dict(Scale_S=(0, 32), Scale_M=(32, 96), Scale_L=(96, 10000)))],
        report_config=[('map', lambda x: x['breakdown'] == 'All'),
                       ('map50', lambda x: x['iou_threshold'] == 0.5 and x['breakdown'] == 'All'),
                       ('map75', lambda x: x['iou_threshold'] == 0.75 and x['breakdown'] == 'All'),
                       ('s_map', lambda x: x['breakdown'] == 'Scale_S'),
                       ('m_map', lambda x: x['breakdown'] == 'Scale_M'),
                       ('l_map', lambda x: x['breakdown'] == 'Scale_L')],
        classes=classes, iou_calculator=dict(type='IOU2DCoCo'), matcher=dict(type='MatcherCoCo'), nproc=-1, logger=logger)
