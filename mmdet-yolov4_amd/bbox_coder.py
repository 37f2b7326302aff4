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
        xc = (bboxes[..., 0] + bboxes[..., 2]) * 0.5
        yc = (bboxes[..., 1] + bboxes[..., 3]) * 0.5
        w = bboxes[..., 2] - bboxes[..., 0]
        h = bboxes[..., 3] - bboxes[..., 1]
        xcp = pred_bboxes[..., 0] * stride + xc
        ycp = pred_bboxes[..., 1] * stride + yc
        wp = pred_bboxes[..., 2] * w
        hp = pred_bboxes[..., 3] * h
        return torch.stack((xcp - wp / 2, ycp - hp / 2, xcp + wp / 2, ycp + hp / 2), dim=-1)
