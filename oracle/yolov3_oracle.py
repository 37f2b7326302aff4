"""CPU restatement of the reference's YOLOv3 inference path.  TEST INFRASTRUCTURE ONLY (imported by
tests/, never by the product package); pinned against tests/golden/tiny_v3.npz, which
tests/golden/make_golden_v3.py produced by running the reference's own modules.

Every function is plain torch-CPU arithmetic over a checkpoint-layout state dict and cites the
reference lines it follows (paths relative to /root/reference/mmdet/).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import yolov4_oracle as O

EPS = O.EPS_DEFAULT          # mmcv build_norm_layer default: the YOLOv3 modules pass norm_cfg without eps
ACT = 'LeakyReLU'            # negative_slope 0.1 everywhere (backbones/darknet.py:34, necks/yolo_neck.py:40)


def _cm(x, sd, p, stride=1, pad=None):
    """mmcv ConvModule = conv -> BN -> LeakyReLU(0.1)."""
    return O.conv_block(x, sd, p, stride=stride, pad=pad, eps=EPS, act=ACT)


def res_block(x, sd, p):
    """backbones/darknet.py:11-52: 1x1 (C -> C/2), 3x3 (C/2 -> C), + input."""
    return O._q(_cm(_cm(x, sd, p + '.conv1'), sd, p + '.conv2') + x, 'add')


def darknet(x, sd, layers, out_indices=(3, 4, 5), p='backbone'):
    """backbones/darknet.py:55-212: conv1, then per stage a 3x3 stride-2 ConvModule + n ResBlocks."""
    outs = []
    x = _cm(x, sd, p + '.conv1')
    if 0 in out_indices:
        outs.append(x)
    for i, n in enumerate(layers):
        q = f'{p}.conv_res_block{i + 1}'
        x = _cm(x, sd, q + '.conv', stride=2, pad=1)
        for r in range(n):
            x = res_block(x, sd, f'{q}.res{r}')
        if i + 1 in out_indices:
            outs.append(x)
    return tuple(outs)


def detection_block(x, sd, p):
    """necks/yolo_neck.py:11-61: 1x1, 3x3, 1x1, 3x3, 1x1."""
    for i in range(1, 6):
        x = _cm(x, sd, f'{p}.conv{i}')
    return x


def yolov3_neck(feats, sd, p='neck'):
    """necks/yolo_neck.py:64-137: top-down, nearest x2 upsample + cat((up, lateral))."""
    n = len(feats)
    outs = []
    out = detection_block(feats[-1], sd, f'{p}.detect1')
    outs.append(out)
    for i, x in enumerate(reversed(feats[:-1])):
        tmp = _cm(out, sd, f'{p}.conv{i + 1}')
        tmp = F.interpolate(tmp, scale_factor=2)
        out = detection_block(torch.cat((tmp, x), 1), sd, f'{p}.detect{i + 2}')
        outs.append(out)
    assert len(outs) == n
    return tuple(outs)


def yolov3_head(feats, sd, p='bbox_head'):
    """dense_heads/yolo_head.py:134-170: 3x3 bridge ConvModule + biased 1x1 prediction conv."""
    outs = []
    for i, x in enumerate(feats):
        x = _cm(x, sd, f'{p}.convs_bridge.{i}')
        outs.append(O._conv2d(x, sd[f'{p}.convs_pred.{i}.weight'], sd[f'{p}.convs_pred.{i}.bias'], head=True))
    return tuple(outs)


def yolo_bbox_decode(anchors, pred, stride):
    """core/bbox/coder/yolo_bbox_coder.py:61-89 (pred[..., :2] already sigmoid-ed, [..., 2:] raw)."""
    xc = (anchors[..., 0] + anchors[..., 2]) * 0.5
    yc = (anchors[..., 1] + anchors[..., 3]) * 0.5
    w = anchors[..., 2] - anchors[..., 0]
    h = anchors[..., 3] - anchors[..., 1]
    xcp = (pred[..., 0] - 0.5) * stride + xc
    ycp = (pred[..., 1] - 0.5) * stride + yc
    wp = torch.exp(pred[..., 2]) * w
    hp = torch.exp(pred[..., 3]) * h
    return torch.stack((xcp - wp / 2, ycp - hp / 2, xcp + wp / 2, ycp + hp / 2), dim=-1)


def yolo_bbox_encode(bboxes, gt, stride, eps=1e-6):
    """core/bbox/coder/yolo_bbox_coder.py:26-59."""
    xg = (gt[..., 0] + gt[..., 2]) * 0.5
    yg = (gt[..., 1] + gt[..., 3]) * 0.5
    wg = gt[..., 2] - gt[..., 0]
    hg = gt[..., 3] - gt[..., 1]
    xc = (bboxes[..., 0] + bboxes[..., 2]) * 0.5
    yc = (bboxes[..., 1] + bboxes[..., 3]) * 0.5
    w = bboxes[..., 2] - bboxes[..., 0]
    h = bboxes[..., 3] - bboxes[..., 1]
    wt = torch.log((wg / w).clamp(min=eps))
    ht = torch.log((hg / h).clamp(min=eps))
    xt = ((xg - xc) / stride + 0.5).clamp(eps, 1 - eps)
    yt = ((yg - yc) / stride + 0.5).clamp(eps, 1 - eps)
    return torch.stack([xt, yt, wt, ht], dim=-1)


V3_BASE_SIZES = [[(116, 90), (156, 198), (373, 326)], [(30, 61), (62, 45), (59, 119)], [(10, 13), (16, 30), (33, 23)]]
V3_STRIDES = [32, 16, 8]


def decode_maps_v3(pred_maps, num_classes, base_sizes=V3_BASE_SIZES, strides=V3_STRIDES):
    """yolo_head.py:254-279 per level: boxes (N,K,4), conf (N,K), cls (N,K,C)."""
    N = pred_maps[0].shape[0]
    attr = 5 + num_classes
    anchors = O.grid_anchors([p.shape[-2:] for p in pred_maps], base_sizes, strides)
    out = []
    for lvl, pm in enumerate(pred_maps):
        m = pm.permute(0, 2, 3, 1).reshape(N, -1, attr)
        pb = torch.cat([torch.sigmoid(m[..., :2]), m[..., 2:4]], dim=-1)
        boxes = yolo_bbox_decode(anchors[lvl].expand_as(pb), pb, strides[lvl])
        out.append((boxes, torch.sigmoid(m[..., 4]), torch.sigmoid(m[..., 5:])))
    return out


def get_bboxes_v3(pred_maps, scale_factors, num_classes, nms_pre=1000, score_thr=0.05, conf_thr=0.005,
                  iou_threshold=0.45, max_per_img=100, rescale=True, base_sizes=V3_BASE_SIZES, strides=V3_STRIDES):
    """yolo_head.py:210-391.  Per level: top-k by objectness when the level has more than nms_pre boxes
    (core/export/onnx_helper.py:45-78; ties towards the lower index, torch.topk leaves them open); per
    image: conf >= conf_thr, then multiclass_nms with score_factors = conf (cls > score_thr is tested
    BEFORE the multiplication, core/post_processing/bbox_nms.py:52-62)."""
    lv = decode_maps_v3(pred_maps, num_classes, base_sizes, strides)
    N = pred_maps[0].shape[0]
    out = []
    for n in range(N):
        bs, cs, ss = [], [], []
        for boxes, conf, cls in lv:
            b, c, s = boxes[n], conf[n], cls[n]
            if 0 < nms_pre < c.shape[0]:
                order = np.lexsort((np.arange(c.shape[0]), -c.numpy().astype(np.float64)))[:nms_pre]
                sel = torch.from_numpy(order.copy())          # topk order: descending objectness
                b, c, s = b[sel], c[sel], s[sel]
            bs.append(b); cs.append(c); ss.append(s)
        b, c, s = torch.cat(bs), torch.cat(cs), torch.cat(ss)
        if rescale:
            b = b / b.new_tensor(scale_factors[n])
        s = torch.cat([s, s.new_zeros(s.shape[0], 1)], dim=1)
        if conf_thr > 0:
            keep = (c >= conf_thr).nonzero(as_tuple=False).squeeze(1)
            b, s, c = b[keep], s[keep], c[keep]
        out.append(O.multiclass_nms(b, s, score_thr, dict(type='nms', iou_threshold=iou_threshold), max_per_img,
                                    score_factors=c))
    return out


# =====================================================================================
# Training side of YOLOV3Head -- yolo_head.py:393-560 (fp32, torch CPU)
# =====================================================================================
def responsible_flags(featmap_sizes, gt_bboxes, strides=V3_STRIDES, A=3):
    """core/anchor/anchor_generator.py:667-727: 1 for the A boxes of every cell that holds a gt centre."""
    out = []
    for (h, w), s in zip(featmap_sizes, strides):
        cx = (gt_bboxes[:, 0] + gt_bboxes[:, 2]) * 0.5
        cy = (gt_bboxes[:, 1] + gt_bboxes[:, 3]) * 0.5
        idx = torch.floor(cy / s).long() * w + torch.floor(cx / s).long()
        grid = torch.zeros(h * w, dtype=torch.uint8)
        grid[idx] = 1
        out.append(grid[:, None].expand(h * w, A).contiguous().view(-1))
    return out


def _iou(b1, b2, eps=1e-6):
    """core/bbox/iou_calculators/iou2d_calculator.py (mode 'iou'): (m,4),(n,4) -> (m,n)."""
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    lt = torch.max(b1[:, None, :2], b2[None, :, :2])
    rb = torch.min(b1[:, None, 2:], b2[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    ov = wh[..., 0] * wh[..., 1]
    union = torch.max(a1[:, None] + a2[None, :] - ov, ov.new_tensor([eps]))
    return ov / union


def grid_assign(bboxes, flags, gt_bboxes, pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.0):
    """core/bbox/assigners/grid_assigner.py:73-156 (gt_max_assign_all=True): -1 / 0 / gt index + 1."""
    G, B = gt_bboxes.size(0), bboxes.size(0)
    ov = _iou(gt_bboxes, bboxes)
    assigned = ov.new_full((B,), -1, dtype=torch.long)
    if G == 0:
        assigned[:] = 0
        return assigned
    mx, _ = ov.max(dim=0)
    assigned[(mx >= 0) & (mx <= neg_iou_thr)] = 0
    resp = flags.bool()
    ov[:, ~resp] = -1.
    mx, amx = ov.max(dim=0)
    gmx, _ = ov.max(dim=1)
    pos = (mx > pos_iou_thr) & resp
    assigned[pos] = amx[pos] + 1
    for i in range(G):
        if gmx[i] > min_pos_iou:
            assigned[(ov[i, :] == gmx[i]) & resp] = i + 1
    return assigned


def targets_v3(featmap_sizes, gt_bboxes_list, gt_labels_list, num_classes, base_sizes=V3_BASE_SIZES,
               strides=V3_STRIDES):
    """yolo_head.py:484-560: per level (N, A_l, 5+C) target maps and (N, A_l) uint8 negative maps."""
    anchors = O.grid_anchors(featmap_sizes, base_sizes, strides)
    cat = torch.cat(anchors)
    astride = torch.cat([torch.tensor(s).repeat(len(a)) for a, s in zip(anchors, strides)])
    tmaps, nmaps = [], []
    for gtb, gtl in zip(gt_bboxes_list, gt_labels_list):
        flags = torch.cat(responsible_flags(featmap_sizes, gtb, strides, anchors[0].shape[0] // (featmap_sizes[0][0] * featmap_sizes[0][1])))
        assigned = grid_assign(cat, flags, gtb)
        pos = torch.nonzero(assigned > 0, as_tuple=False).squeeze(-1).unique()
        neg = torch.nonzero(assigned == 0, as_tuple=False).squeeze(-1).unique()
        gi = assigned[pos] - 1
        t = cat.new_zeros(cat.size(0), 5 + num_classes)
        t[pos, :4] = yolo_bbox_encode(cat[pos], gtb[gi], astride[pos])
        t[pos, 4] = 1
        t[pos, 5:] = F.one_hot(gtl, num_classes=num_classes).float()[gi]
        n = cat.new_zeros(cat.size(0), dtype=torch.uint8)
        n[neg] = 1
        tmaps.append(t)
        nmaps.append(n)
    T_, N_ = torch.stack(tmaps, 0), torch.stack(nmaps, 0)
    out_t, out_n, start = [], [], 0
    for a in anchors:
        out_t.append(T_[:, start:start + a.size(0)])
        out_n.append(N_[:, start:start + a.size(0)])
        start += a.size(0)
    return out_t, out_n


def head_loss_v3(pred_maps, gt_bboxes, gt_labels, num_classes, w_cls=1.0, w_conf=1.0, w_xy=2.0, w_wh=2.0,
                 base_sizes=V3_BASE_SIZES, strides=V3_STRIDES):
    """yolo_head.py:395-482 with the loss configuration of configs/yolo/yolov3_d53_*: sigmoid BCE for cls /
    conf / xy and MSE for wh, all reduction='sum', element-wise weights pos_mask / pos_and_neg_mask."""
    pred_maps = [p.float() for p in pred_maps]
    sizes = [p.shape[-2:] for p in pred_maps]
    tmaps, nmaps = targets_v3(sizes, gt_bboxes, gt_labels, num_classes, base_sizes, strides)
    bce = lambda p, t, w: (F.binary_cross_entropy_with_logits(p, t, reduction='none') * w).sum()
    l_cls, l_conf, l_xy, l_wh = [], [], [], []
    for pm, t, n in zip(pred_maps, tmaps, nmaps):
        N = pm.shape[0]
        pm = pm.permute(0, 2, 3, 1).reshape(N, -1, 5 + num_classes)
        pos = t[..., 4]
        pn = (n.float() + pos).clamp(0., 1.)
        pos = pos.unsqueeze(-1)
        l_cls.append(w_cls * bce(pm[..., 5:], t[..., 5:], pos))
        l_conf.append(w_conf * bce(pm[..., 4], t[..., 4], pn))
        l_xy.append(w_xy * bce(pm[..., :2], t[..., :2], pos))
        l_wh.append(w_wh * (F.mse_loss(pm[..., 2:4], t[..., 2:4], reduction='none') * pos).sum())
    return dict(loss_cls=l_cls, loss_conf=l_conf, loss_xy=l_xy, loss_wh=l_wh)


def forward_train_v3(img, sd, layers, gt_bboxes, gt_labels, num_classes):
    """detectors/single_stage.py:51-79 in training mode: batch-statistics BN (momentum 0.1, the mmcv
    default every v3 module gets), loss dict out."""
    with O._TrainBN(momentum_cfg=0.1, momentum_default=0.1):
        preds = yolov3_head(yolov3_neck(darknet(img, sd, layers, (3, 4, 5)), sd), sd)
    return head_loss_v3(preds, gt_bboxes, gt_labels, num_classes)
