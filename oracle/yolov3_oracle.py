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
