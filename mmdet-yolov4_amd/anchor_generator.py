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

    # ---- training targets of YOLOV3Head (anchor_generator.py:667-727) ---------------------------
    def responsible_flags(self, featmap_sizes, gt_bboxes, device='cuda'):
        """Per level, a uint8 flag per anchor box: 1 for the A boxes of every cell that holds a gt centre."""
        assert self.num_levels == len(featmap_sizes)
        return [self.single_level_responsible_flags(featmap_sizes[i], gt_bboxes, self.strides[i],
                                                    self.num_base_anchors[i], device=device)
                for i in range(self.num_levels)]

    def single_level_responsible_flags(self, featmap_size, gt_bboxes, stride, num_base_anchors, device='cuda'):
        feat_h, feat_w = featmap_size
        cx = ((gt_bboxes[:, 0] + gt_bboxes[:, 2]) * 0.5).to(device)
        cy = ((gt_bboxes[:, 1] + gt_bboxes[:, 3]) * 0.5).to(device)
        gx = torch.floor(cx / stride[0]).long()
        gy = torch.floor(cy / stride[1]).long()
        idx = gy * feat_w + gx                                  # row-major cell index
        grid = torch.zeros(feat_h * feat_w, dtype=torch.uint8, device=device)
        grid[idx] = 1
        return grid[:, None].expand(grid.size(0), num_base_anchors).contiguous().view(-1)


@ANCHOR_GENERATORS.register_module()
class YOLOV4AnchorGenerator(YOLOAnchorGenerator):
    """yolov4_anchor_generator.py:8 (+ training target assignment, :12-134)."""

    _NEIGHBOR_OFFSET = [[0, 0], [-1, 0], [0, -1], [1, 0], [0, 1], [-1, -1], [1, -1], [1, 1], [-1, 1]]

    def responsible_indices(self, featmap_sizes, gt_bboxes_list, neighbor=3, shape_match_thres=4.,
                            device='cuda'):
        """yolov4_anchor_generator.py:12-134: per level ``(img_idx, anchor_idx, gt_idx)`` of the
        anchors made responsible for each ground truth: shape test ``max(r, 1/r).max() < thres`` between
        the level's base anchors and the gt, then the gt's own cell plus the neighbour cells selected by
        ``neighbor`` (0 none, 1 nearest, 2 the two nearest as in YOLOv5, 3 all eight candidates)."""
        img_id = [g.new_full((g.shape[0],), i, dtype=torch.long) for i, g in enumerate(gt_bboxes_list)]
        gt = torch.cat(gt_bboxes_list, dim=0)
        img_id = torch.cat(img_id, dim=0).to(device)
        if gt.shape[0] == 0:
            e = torch.tensor([], device=device, dtype=torch.long)
            return [(e, e.clone(), e.clone()) for _ in range(self.num_levels)]
        gt_xy = (0.5 * (gt[:, 2:4] + gt[:, :2])).to(device)
        gt_wh = (gt[:, 2:4] - gt[:, :2]).to(device)
        noff = gt_xy.new_tensor(self._NEIGHBOR_OFFSET)
        out = []
        for i in range(self.num_levels):
            fh, fw = featmap_sizes[i]
            A = self.num_base_anchors[i]
            base = self.base_anchors[i].to(device)
            base_wh = base[:, 2:] - base[:, :2]
            dev = gt_wh[None, :, :] / base_wh[:, None, :]
            dev = torch.max(dev, 1. / dev).max(dim=2).values
            a_ind, g_ind = (dev < shape_match_thres).nonzero(as_tuple=True)
            feat = gt_xy.new_tensor([[fw, fh]])
            xy = gt_xy[g_ind] / gt_xy.new_tensor([self.strides[i]])
            inv = feat - xy
            if neighbor == 0:
                px, py = xy.long().T
                anchor = (py * fw + px) * A + a_ind
            else:
                left, up = ((xy % 1. < 0.5) & (xy > 1.)).T
                right, down = ((inv % 1. < 0.5) & (inv > 1.)).T
                rows = [torch.ones_like(left), left, up, right, down]
                if neighbor == 1:
                    ok = torch.stack(rows)
                    if ok.numel() > 0:
                        dist = torch.cat((xy, inv), dim=-1) % 1.
                        ok[1:] = ok[1:] & (dist == dist.min(dim=-1).values[:, None]).T
                elif neighbor == 2:
                    ok = torch.stack(rows)
                elif neighbor == 3:
                    ok = torch.stack(rows + [left & up, right & up, right & down, left & down])
                else:
                    raise NotImplementedError
                n = ok.shape[0]
                g_ind = g_ind.repeat((n, 1))[ok]
                a_ind = a_ind.repeat((n, 1))[ok]
                px, py = (xy[None, :, :] + noff[:n, None, :])[ok].long().T
                anchor = (py * fw + px) * A + a_ind
            out.append((img_id[g_ind], anchor, g_ind))
        return out
