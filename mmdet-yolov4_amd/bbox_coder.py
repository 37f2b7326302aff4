"""``YOLOV4BBoxCoder`` under the reference's registry name.

Mirror of ``mmdet/core/bbox/coder/yolov4_bbox_coder.py:8-67``.  ``decode`` on CUDA
tensors runs through the HIP decode kernel's arithmetic only on the fused path
(``yv4_decode_filter``); this class-level ``decode`` is the host-side API form and uses
elementwise torch ops, which is what the reference's coder is.
"""
import torch

from .registry import BBOX_CODERS


@BBOX_CODERS.register_module()
class YOLOV4BBoxCoder:

    def __init__(self, eps=1e-6):
        self.eps = eps

    def encode(self, bboxes, gt_bboxes, stride):
        raise NotImplementedError

    def decode(self, bboxes, pred_bboxes, stride):
        assert pred_bboxes.size(0) == bboxes.size(0)
        assert pred_bboxes.size(-1) == bboxes.size(-1) == 4
        # the same fp32 operations per coordinate as the reference (centre = (lo + hi) * 0.5, size = hi - lo,
        # centre' = t_xy * stride + centre, size' = t_wh * size, corners = centre' -/+ size' / 2), on (…, 2) halves
        lo, hi = bboxes[..., :2], bboxes[..., 2:]
        centre = pred_bboxes[..., :2] * stride + (lo + hi) * 0.5
        half = pred_bboxes[..., 2:] * (hi - lo) / 2
        return torch.cat((centre - half, centre + half), dim=-1)
