"""Target assignment of the YOLOv3 head: ``GridAssigner`` and ``PseudoSampler`` under the reference's
registry names (``mmdet/core/bbox/assigners/grid_assigner.py:8-156``,
``mmdet/core/bbox/samplers/pseudo_sampler.py:8-41``), expressed with torch tensor ops on whatever
device the boxes live on (a few thousand anchors x a handful of ground truths per image).
"""
import torch

from .losses import bbox_overlaps
from .registry import BBOX_ASSIGNERS, BBOX_SAMPLERS


class AssignResult:
    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts, self.gt_inds, self.max_overlaps, self.labels = num_gts, gt_inds, max_overlaps, labels


class SamplingResult:
    """core/bbox/samplers/sampling_result.py: the fields YOLOV3Head reads."""

    def __init__(self, pos_inds, neg_inds, bboxes, gt_bboxes, assign_result):
        self.pos_inds, self.neg_inds = pos_inds, neg_inds
        self.pos_bboxes = bboxes[pos_inds]
        self.neg_bboxes = bboxes[neg_inds]
        self.num_gts = gt_bboxes.shape[0]
        self.pos_assigned_gt_inds = assign_result.gt_inds[pos_inds] - 1
        if gt_bboxes.numel() == 0:
            self.pos_gt_bboxes = torch.empty_like(gt_bboxes).view(-1, 4)
        else:
            self.pos_gt_bboxes = gt_bboxes.view(-1, 4)[self.pos_assigned_gt_inds, :]


@BBOX_ASSIGNERS.register_module()
class GridAssigner:
    """-1 don't care, 0 negative, k > 0 assigned to gt k-1.  Steps (grid_assigner.py:73-156): (2) boxes
    whose max IoU <= neg_iou_thr become negatives; IoUs of boxes outside the responsible cells are set to
    -1; (3) responsible boxes with max IoU > pos_iou_thr take their argmax gt; (4) every gt whose best
    responsible IoU exceeds min_pos_iou claims all responsible boxes attaining it (gt order: later wins)."""

    def __init__(self, pos_iou_thr, neg_iou_thr, min_pos_iou=.0, gt_max_assign_all=True,
                 iou_calculator=dict(type='BboxOverlaps2D')):
        self.pos_iou_thr, self.neg_iou_thr = pos_iou_thr, neg_iou_thr
        self.min_pos_iou, self.gt_max_assign_all = min_pos_iou, gt_max_assign_all

    def assign(self, bboxes, box_responsible_flags, gt_bboxes, gt_labels=None):
        num_gts, num_bboxes = gt_bboxes.size(0), bboxes.size(0)
        overlaps = bbox_overlaps(gt_bboxes, bboxes)
        assigned = overlaps.new_full((num_bboxes,), -1, dtype=torch.long)
        if num_gts == 0 or num_bboxes == 0:
            if num_gts == 0:
                assigned[:] = 0
            return AssignResult(num_gts, assigned, overlaps.new_zeros((num_bboxes,)))
        max_overlaps, _ = overlaps.max(dim=0)
        if isinstance(self.neg_iou_thr, float):
            assigned[(max_overlaps >= 0) & (max_overlaps <= self.neg_iou_thr)] = 0
        else:
            assert len(self.neg_iou_thr) == 2
            assigned[(max_overlaps > self.neg_iou_thr[0]) & (max_overlaps <= self.neg_iou_thr[1])] = 0
        resp = box_responsible_flags.type(torch.bool)
        overlaps[:, ~resp] = -1.
        max_overlaps, argmax_overlaps = overlaps.max(dim=0)
        gt_max_overlaps, gt_argmax_overlaps = overlaps.max(dim=1)
        pos = (max_overlaps > self.pos_iou_thr) & resp
        assigned[pos] = argmax_overlaps[pos] + 1
        for i in range(num_gts):
            if gt_max_overlaps[i] > self.min_pos_iou:
                if self.gt_max_assign_all:
                    assigned[(overlaps[i, :] == gt_max_overlaps[i]) & resp] = i + 1
                elif box_responsible_flags[gt_argmax_overlaps[i]]:
                    assigned[gt_argmax_overlaps[i]] = i + 1
        return AssignResult(num_gts, assigned, max_overlaps)


@BBOX_SAMPLERS.register_module()
class PseudoSampler:
    """Every assigned box is a sample (pseudo_sampler.py:22-41)."""

    def __init__(self, **kwargs):
        pass

    def sample(self, assign_result, bboxes, gt_bboxes, **kwargs):
        pos_inds = torch.nonzero(assign_result.gt_inds > 0, as_tuple=False).squeeze(-1).unique()
        neg_inds = torch.nonzero(assign_result.gt_inds == 0, as_tuple=False).squeeze(-1).unique()
        return SamplingResult(pos_inds, neg_inds, bboxes, gt_bboxes, assign_result)
