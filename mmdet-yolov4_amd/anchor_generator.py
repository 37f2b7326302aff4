"""``YOLOAnchorGenerator`` / ``YOLOV4AnchorGenerator`` under the reference's registry names.

Mirror of ``mmdet/core/anchor/anchor_generator.py:595-665`` (base anchors centred at
stride/2) and ``:207-270`` (row-major grid, index ``(y*W + x)*A + a``), and of
``mmdet/core/anchor/yolov4_anchor_generator.py:8``.  On the fused inference path no
anchor tensor is ever built -- ``yv4_decode_filter`` derives each anchor from
(level, y, x, a) with the same fp32 operations; ``grid_anchors`` exists for API parity
and as the host-side statement of what the kernel computes.
"""
import torch

from .registry import ANCHOR_GENERATORS


def _pair(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


@ANCHOR_GENERATORS.register_module()
class YOLOAnchorGenerator:

    def __init__(self, strides, base_sizes):
        self.strides = [_pair(s) for s in strides]
        self.centers = [(s[0] / 2., s[1] / 2.) for s in self.strides]
        self.base_sizes = []
        num_anchor_per_level = len(base_sizes[0])
        for per_level in base_sizes:
            assert num_anchor_per_level == len(per_level)
            self.base_sizes.append([_pair(b) for b in per_level])
        self.base_anchors = self.gen_base_anchors()

    @property
    def num_levels(self):
        return len(self.base_sizes)

    @property
    def num_base_anchors(self):
        return [b.size(0) for b in self.base_anchors]

    def gen_base_anchors(self):
        return [self.gen_single_level_base_anchors(per_level, self.centers[i])
                for i, per_level in enumerate(self.base_sizes)]

    def gen_single_level_base_anchors(self, base_sizes_per_level, center=None):
        x_c, y_c = center
        rows = []
        for w, h in base_sizes_per_level:
            # python-float arithmetic first, one rounding to fp32 (as torch.Tensor([...]) does)
            rows.append(torch.tensor([x_c - 0.5 * w, y_c - 0.5 * h, x_c + 0.5 * w, y_c + 0.5 * h],
                                     dtype=torch.float32))
        return torch.stack(rows, dim=0)

    def single_level_grid_anchors(self, base_anchors, featmap_size, stride=(16, 16), device='cuda'):
        feat_h, feat_w = featmap_size
        shift_x = torch.arange(0, feat_w, device=device) * stride[0]
        shift_y = torch.arange(0, feat_h, device=device) * stride[1]
        xx = shift_x.repeat(feat_h)
        yy = shift_y.view(-1, 1).repeat(1, feat_w).view(-1)
        shifts = torch.stack([xx, yy, xx, yy], dim=-1).type_as(base_anchors)
        return (base_anchors[None, :, :] + shifts[:, None, :]).view(-1, 4)

    def grid_anchors(self, featmap_sizes, device='cuda'):
        assert self.num_levels == len(featmap_sizes)
        return [self.single_level_grid_anchors(self.base_anchors[i].to(device), featmap_sizes[i],
                                               self.strides[i], device=device)
                for i in range(self.num_levels)]


@ANCHOR_GENERATORS.register_module()
class YOLOV4AnchorGenerator(YOLOAnchorGenerator):
    """yolov4_anchor_generator.py:8.  ``responsible_indices`` (training target
    assignment, :12-134) belongs to the training row of the scope table and is not
    built yet."""

    def responsible_indices(self, featmap_sizes, gt_bboxes_list, neighbor=3, shape_match_thres=4.,
                            device='cuda'):
        raise NotImplementedError(
            'YOLOV4AnchorGenerator.responsible_indices (training) is not built yet: see DESIGN.md scope')
