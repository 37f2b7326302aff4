"""CPU restatement of the TRAIN-side input pipeline of configs/yolov4/*_coco_mosaic.py:22-69 (TEST INFRASTRUCTURE ONLY).

Only tests/ (and tools/aug_bench.py's `cpu_baseline` leg) may import this file.  Steps and what pins each of them:

  step                                    source                                            pinned by
  --------------------------------------  ------------------------------------------------  ---------------------------------
  per-image Resize(keep_ratio, 640)       transforms.py Resize -> mmcv.imrescale -> cv2     unpinned (mmcv / OpenCV absent);
                                                                                            = preprocess_oracle.resize_linear_u8
  MosaicPipeline stitch + box shift       transforms.py:1906-1983 (numpy only)              tests/golden/augment.npz: outputs of
                                                                                            the reference's own class
  Albu(PadIfNeeded 1920, RandomCrop 1280, albumentations (third party, absent)              unpinned: restated from the published
       RandomScale 0.5, CenterCrop 640,                                                     albumentations 1.x functional code
       HorizontalFlip; bbox_params)
  HueSaturationValueJitter                transforms.py:1986-2021: LUTs in numpy,            LUT lines: restated 1:1 (numpy);
                                          BGR<->HSV by cv2                                   cv2 8-bit colour conversion unpinned
  GtBBoxesFilter                          transforms.py:2024-2052 (numpy only)              tests/golden/augment.npz
  Normalize + ImageToTensor               mmcv.imnormalize                                  as preprocess_oracle

The random draws are INPUTS here (a dict of parameters per output image): the reference draws them from python's /
numpy's / albumentations' global generators, which nobody can replay on a GPU; parity is defined per parameter set.
"""
import numpy as np

from .preprocess_oracle import rescale_size, resize_linear_u8


# ---- MosaicPipeline (transforms.py:1906-1983) ---------------------------------------------------------------
def mosaic(images, boxes, labels, pad_val=114):
    """images: 4 uint8 (h_i, w_i, 3) arrays as the individual pipeline left them (resized); boxes: 4 float32 (k_i, 4);
    labels: 4 int64 (k_i,).  -> canvas (2c, 2c, 3) uint8, boxes (K, 4) float32, labels (K,), c."""
    shapes = [im.shape for im in images]
    cxy = max(shapes[0][0], shapes[1][0], shapes[0][1], shapes[2][1])          # transforms.py:1934
    canvas = np.full((cxy * 2, cxy * 2, shapes[0][2]), pad_val, dtype=np.uint8)
    out_boxes = []
    for i, im in enumerate(images):
        h, w = im.shape[:2]
        if i == 0:
            x1, y1, x2, y2 = cxy - w, cxy - h, cxy, cxy
        elif i == 1:
            x1, y1, x2, y2 = cxy, cxy - h, cxy + w, cxy
        elif i == 2:
            x1, y1, x2, y2 = cxy - w, cxy, cxy, cxy + h
        else:
            x1, y1, x2, y2 = cxy, cxy, cxy + w, cxy + h
        canvas[y1:y2, x1:x2] = im
        b = boxes[i].copy()
        b[:, 0::2] = b[:, 0::2] + x1
        b[:, 1::2] = b[:, 1::2] + y1
        out_boxes.append(b)
    return canvas, np.concatenate(out_boxes, axis=0), np.concatenate(labels, axis=0), cxy


def mosaic_origin(i, cxy, h, w):
    return [(cxy - w, cxy - h), (cxy, cxy - h), (cxy - w, cxy), (cxy, cxy)][i]


# ---- the Albu block (configs/yolov4/yolov4l_coco_mosaic.py:31-60), albumentations 1.x semantics, unpinned -------
def pad_if_needed_offsets(h, w, min_h=1920, min_w=1920):
    """PadIfNeeded(position=center): (top, left) padding; bottom/right take the remainder."""
    top = int((min_h - h) / 2.0) if h < min_h else 0
    left = int((min_w - w) / 2.0) if w < min_w else 0
    return top, left, max(h, min_h), max(w, min_w)


def random_crop_origin(h, w, crop_h, crop_w, h_start, w_start):
    """albumentations.augmentations.crops.functional.get_random_crop_coords (1.x)."""
    y1 = int((h - crop_h + 1) * h_start)
    x1 = int((w - crop_w + 1) * w_start)
    return y1, x1


def geometric(canvas, params, pad_val=114):
    """PadIfNeeded(1920) -> RandomCrop(1280) -> RandomScale -> CenterCrop(640) -> HorizontalFlip on the uint8 canvas.
    params: dict(h_start, w_start, scale, flip) (+ sizes).  Returns the (out, out, 3) uint8 image."""
    P, C, O = params.get('pad_to', 1920), params.get('crop', 1280), params.get('out', 640)
    h, w = canvas.shape[:2]
    top, left, H, W = pad_if_needed_offsets(h, w, P, P)
    big = np.full((H, W, 3), pad_val, np.uint8)
    big[top:top + h, left:left + w] = canvas
    y1, x1 = random_crop_origin(H, W, C, C, params['h_start'], params['w_start'])
    crop = big[y1:y1 + C, x1:x1 + C]
    S = int(C * params['scale'])                                   # F.scale: int(height * scale)
    scaled = resize_linear_u8(crop, S, S) if S != C else crop
    o = (S - O) // 2                                               # CenterCrop: get_center_crop_coords
    out = scaled[o:o + O, o:o + O]
    if params['flip']:
        out = out[:, ::-1]
    return np.ascontiguousarray(out)


def geometric_boxes(boxes, labels, canvas_hw, params, min_area=4.0, min_visibility=0.2):
    """The same chain on pascal_voc boxes with albumentations' BboxParams filter (min_area, min_visibility) applied
    once at the end (check_each_transform=False).  float64 like albumentations' python floats; float32 out."""
    P, C, O = params.get('pad_to', 1920), params.get('crop', 1280), params.get('out', 640)
    h, w = canvas_hw
    top, left, H, W = pad_if_needed_offsets(h, w, P, P)
    y1, x1 = random_crop_origin(H, W, C, C, params['h_start'], params['w_start'])
    S = int(C * params['scale'])
    o = (S - O) // 2
    b = boxes.astype(np.float64).copy()
    b[:, 0::2] += left - x1
    b[:, 1::2] += top - y1
    b *= S / float(C)                                              # normalised boxes are scale-invariant: px scale S / C
    b[:, 0::2] -= o
    b[:, 1::2] -= o
    if params['flip']:
        x_min, x_max = O - b[:, 2], O - b[:, 0]
        b[:, 0], b[:, 2] = x_min.copy(), x_max.copy()
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    c = np.clip(b, 0.0, float(O))
    carea = (c[:, 2] - c[:, 0]) * (c[:, 3] - c[:, 1])
    with np.errstate(divide='ignore', invalid='ignore'):
        keep = (area > 0) & (carea > 0) & (carea / area > min_visibility) & (carea > min_area)
    return c[keep].astype(np.float32), labels[keep]


# ---- HueSaturationValueJitter (transforms.py:1986-2021) ---------------------------------------------------------------
def hsv_luts(r):
    """transforms.py:2004-2008, line for line (numpy)."""
    x = np.arange(0, 256, dtype=np.int16)
    lut_hue = ((x * r[0]) % 180).astype(np.uint8)
    lut_sat = np.clip(x * r[1], 0, 255).astype(np.uint8)
    lut_val = np.clip(x * r[2], 0, 255).astype(np.uint8)
    return lut_hue, lut_sat, lut_val


_SHIFT = 12
_SDIV = np.array([0] + [int(round((255 << _SHIFT) / float(i))) for i in range(1, 256)], dtype=np.int64)
_HDIV180 = np.array([0] + [int(round((180 << _SHIFT) / (6.0 * i))) for i in range(1, 256)], dtype=np.int64)


def bgr2hsv_u8(img):
    """OpenCV cvtColor(COLOR_BGR2HSV) for CV_8U (modules/imgproc/src/color_hsv: RGB2HSV_b, hrange 180): integer
    arithmetic with the 12-bit division tables.  Restated, unpinned."""
    b, g, r = (img[..., i].astype(np.int64) for i in range(3))
    v = np.maximum(np.maximum(b, g), r)
    vmin = np.minimum(np.minimum(b, g), r)
    diff = v - vmin
    vr = v == r
    vg = v == g
    s = (diff * _SDIV[v] + (1 << (_SHIFT - 1))) >> _SHIFT
    h = np.where(vr, g - b, np.where(vg, b - r + 2 * diff, r - g + 4 * diff))
    h = (h * _HDIV180[diff] + (1 << (_SHIFT - 1))) >> _SHIFT
    h = h + np.where(h < 0, 180, 0)
    return np.stack([h, s, v], axis=-1).astype(np.uint8)


_SECTOR = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])


def hsv2bgr_u8(hsv):
    """OpenCV cvtColor(COLOR_HSV2BGR) for CV_8U: HSV2RGB_b converts to float (h, s/255, v/255), runs the float
    conversion (hscale = 6/180) and saturate_casts v*255 back (round half to even).  Restated, unpinned."""
    h = hsv[..., 0].astype(np.float32)
    s = hsv[..., 1].astype(np.float32) * np.float32(1.0 / 255.0)
    v = hsv[..., 2].astype(np.float32) * np.float32(1.0 / 255.0)
    hh = h * np.float32(6.0 / 180.0)
    sector = np.floor(hh).astype(np.int64)
    f = (hh - sector.astype(np.float32)).astype(np.float32)
    sector = sector % 6
    one = np.float32(1.0)
    tab = np.stack([v, (v * (one - s)).astype(np.float32), (v * (one - s * f)).astype(np.float32),
                    (v * (one - s * (one - f))).astype(np.float32)], axis=-1)
    idx = _SECTOR[sector]                                          # (..., 3): b, g, r table entries
    bgr = np.take_along_axis(tab, idx, axis=-1)
    bgr = np.where((s == 0)[..., None], v[..., None], bgr)
    return np.clip(np.rint(bgr * np.float32(255.0)), 0, 255).astype(np.uint8)


def hsv_jitter(img, r):
    """HueSaturationValueJitter.__call__ for gains r = (rh, rs, rv) (= uniform(-1, 1) * ratio + 1)."""
    lh, ls, lv = hsv_luts(np.asarray(r, dtype=np.float64))
    hsv = bgr2hsv_u8(img)
    out = np.stack([lh[hsv[..., 0]], ls[hsv[..., 1]], lv[hsv[..., 2]]], axis=-1)
    return hsv2bgr_u8(out)


# ---- GtBBoxesFilter (transforms.py:2024-2052) -------------------------------------------------------------------------
def gt_bboxes_filter(bboxes, labels, min_size=2, max_aspect_ratio=20):
    w = bboxes[:, 2] - bboxes[:, 0]
    h = bboxes[:, 3] - bboxes[:, 1]
    ar = np.maximum(w / (h + 1e-16), h / (w + 1e-16))
    valid = (w > min_size) & (h > min_size) & (ar < max_aspect_ratio)
    return bboxes[valid], labels[valid]


def normalize_chw(img, mean=(114, 114, 114), std=(255, 255, 255), to_rgb=True):
    a = img.astype(np.float32)
    if to_rgb:
        a = a[..., ::-1]
    out = (a - np.float64(mean).astype(np.float32)) * (1 / np.float64(std)).astype(np.float32)
    return np.ascontiguousarray(out.astype(np.float32).transpose(2, 0, 1))


def train_sample(sources, src_boxes, src_labels, params, scale=(640, 640), pad_val=114):
    """One training sample end to end: 4 decoded uint8 images + their boxes -> (3, 640, 640) float32, boxes, labels.
    params: dict(h_start, w_start, scale, flip, hsv=(rh, rs, rv))."""
    ims, bxs = [], []
    for im, b in zip(sources, src_boxes):
        h, w = im.shape[:2]
        nh, nw = rescale_size(h, w, scale)
        ims.append(resize_linear_u8(im, nh, nw))
        sf = np.array([nw / w, nh / h, nw / w, nh / h], dtype=np.float32)       # Resize._resize_bboxes
        bb = b * sf
        bb[:, 0::2] = np.clip(bb[:, 0::2], 0, nw)
        bb[:, 1::2] = np.clip(bb[:, 1::2], 0, nh)
        bxs.append(bb.astype(np.float32))
    canvas, boxes, labels, _ = mosaic(ims, bxs, src_labels, pad_val)
    img = geometric(canvas, params, pad_val)
    boxes, labels = geometric_boxes(boxes, labels, canvas.shape[:2], params)
    img = hsv_jitter(img, params['hsv'])
    boxes, labels = gt_bboxes_filter(boxes, labels)
    return normalize_chw(img), boxes, labels, img
