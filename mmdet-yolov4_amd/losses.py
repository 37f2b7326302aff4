"""Loss side of ``YOLOCSPHead`` (training).

Mirror of ``mmdet/models/dense_heads/yolocsp_head.py:384-575`` (``loss``,
``loss_single_no_assigner``, ``get_targets_no_assigner``), ``GIoULoss`` / ``giou_loss``
(``mmdet/models/losses/iou_loss.py:85-102,330-366``), aligned GIoU of
``mmdet/core/bbox/iou_calculators/iou2d_calculator.py:74-260`` and sigmoid
``CrossEntropyLoss`` (``mmdet/models/losses/cross_entropy_loss.py:58-91,142-214``).

The training step's default configuration (sigmoid BCE + GIoU, no assigner) runs on the fused HIP loss
(``csrc/loss.hip`` via ``YoloLossFunction`` in ``yolocsp_head.py``).  The classes here are the registry
entries the configs name, the tensor-op statement of the same arithmetic (used for dense pred maps,
non-default loss settings and as the GPU-side cross-check in the tests) and YOLOV3Head's losses.
"""
import torch
import torch.nn.functional as F

from .registry import LOSSES


def bbox_overlaps_giou_aligned(b1, b2, eps=1e-6):
    area1 = (b1[..., 2] - b1[..., 0]) * (b1[..., 3] - b1[..., 1])
    area2 = (b2[..., 2] - b2[..., 0]) * (b2[..., 3] - b2[..., 1])
    lt = torch.max(b1[..., :2], b2[..., :2])
    rb = torch.min(b1[..., 2:], b2[..., 2:])
    wh = (rb - lt).clamp(min=0)
    overlap = wh[..., 0] * wh[..., 1]
    e = b1.new_tensor([eps])
    union = torch.max(area1 + area2 - overlap, e)
    ious = overlap / union
    ewh = (torch.max(b1[..., 2:], b2[..., 2:]) - torch.min(b1[..., :2], b2[..., :2])).clamp(min=0)
    earea = torch.max(ewh[..., 0] * ewh[..., 1], e)
    return ious - (earea - union) / earea


def bbox_overlaps(b1, b2, eps=1e-6):
    """Pairwise IoU, core/bbox/iou_calculators/iou2d_calculator.py:74-260 (mode='iou', is_aligned=False):
    (m,4) x (n,4) -> (m,n)."""
    area1 = (b1[..., 2] - b1[..., 0]) * (b1[..., 3] - b1[..., 1])
    area2 = (b2[..., 2] - b2[..., 0]) * (b2[..., 3] - b2[..., 1])
    lt = torch.max(b1[..., :, None, :2], b2[..., None, :, :2])
    rb = torch.min(b1[..., :, None, 2:], b2[..., None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    overlap = wh[..., 0] * wh[..., 1]
    union = area1[..., None] + area2[..., None, :] - overlap
    union = torch.max(union, union.new_tensor([eps]))
    return overlap / union


def weight_reduce_loss(loss, weight=None, reduction='mean', avg_factor=None):
    """losses/utils.py:27-54."""
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        return reduce_loss(loss, reduction)
    if reduction == 'mean':
        return loss.sum() / avg_factor
    if reduction != 'none':
        raise ValueError('avg_factor can not be used with reduction="sum"')
    return loss


def reduce_loss(loss, reduction):
    if reduction == 'none':
        return loss
    if reduction == 'mean':
        return loss.mean()
    if reduction == 'sum':
        return loss.sum()
    raise ValueError(reduction)


@LOSSES.register_module()
class GIoULoss(torch.nn.Module):

    def __init__(self, eps=1e-6, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.eps, self.reduction, self.loss_weight = eps, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        assert weight is None and avg_factor is None, 'weighted GIoU is not used by this head'
        reduction = reduction_override if reduction_override else self.reduction
        loss = 1 - bbox_overlaps_giou_aligned(pred, target, eps=self.eps)
        return self.loss_weight * reduce_loss(loss, reduction)


@LOSSES.register_module()
class CrossEntropyLoss(torch.nn.Module):
    """Only the ``use_sigmoid=True`` form the head uses (BCE with logits)."""

    def __init__(self, use_sigmoid=False, use_mask=False, reduction='mean', class_weight=None, loss_weight=1.0):
        super().__init__()
        if not use_sigmoid or use_mask:
            raise NotImplementedError('only CrossEntropyLoss(use_sigmoid=True) is built')
        self.use_sigmoid, self.reduction, self.loss_weight, self.class_weight = True, reduction, loss_weight, class_weight

    def forward(self, cls_score, label, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        """cross_entropy_loss.py:58-91 (element-wise ``weight``, ``avg_factor``) + :187-214."""
        reduction = reduction_override if reduction_override else self.reduction
        pw = cls_score.new_tensor(self.class_weight) if self.class_weight is not None else None
        loss = F.binary_cross_entropy_with_logits(cls_score, label.float(), pos_weight=pw, reduction='none')
        if weight is not None:
            weight = weight.float()
        return self.loss_weight * weight_reduce_loss(loss, weight, reduction=reduction, avg_factor=avg_factor)


@LOSSES.register_module()
class MSELoss(torch.nn.Module):
    """losses/mse_loss.py:8-50 (``@weighted_loss`` element-wise MSE)."""

    def __init__(self, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None):
        loss = F.mse_loss(pred, target, reduction='none')
        return self.loss_weight * weight_reduce_loss(loss, weight, reduction=self.reduction, avg_factor=avg_factor)


@LOSSES.register_module()
class SoftFocalLoss(torch.nn.Module):
    """yolocsp_head.py:21-50."""

    def __init__(self, raw_loss, gamma=1.5, alpha=0.25):
        super().__init__()
        from .registry import build_loss
        self.loss_fcn = build_loss(raw_loss)
        self.gamma, self.alpha = gamma, alpha
        self.reduction = self.loss_fcn.reduction
        self.loss_fcn.reduction = 'none'

    def forward(self, pred, gt, reduction_override=None):
        loss = self.loss_fcn(pred, gt)
        p = torch.sigmoid(pred)
        p_t = gt * p + (1 - gt) * (1 - p)
        loss = loss * (gt * self.alpha + (1 - gt) * (1 - self.alpha)) * (1.0 - p_t) ** self.gamma
        return reduce_loss(loss, reduction_override if reduction_override is not None else self.reduction)
