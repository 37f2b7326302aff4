"""CPU restatement of the test-time input pipeline (TEST INFRASTRUCTURE ONLY) -- **parity unpinned**.

Only tests/ may import this file.  The pipeline is
``Resize(keep_ratio=True) -> RandomFlip(no-op) -> Pad(size_divisor=32) -> Normalize -> ImageToTensor``
(configs/yolov4/yolov4l_coco_mosaic.py:70-84, mmdet/datasets/pipelines/transforms.py: ``Resize._resize_img``,
``Pad._pad_img``, ``Normalize.__call__``), but its arithmetic lives in mmcv 1.x (``imrescale`` / ``rescale_size``,
``impad_to_multiple``, ``imnormalize``) and OpenCV (``cv2.resize(..., INTER_LINEAR)`` on 8-bit images,
``cv2.subtract`` / ``cv2.multiply`` on float32), both third party and absent from the build image: there is
nothing here to run the reference's own code against, and no golden vector of it in /root/reference/tests.
What is restated, from the published sources:
  * mmcv ``rescale_size``: factor = min(max(scale) / max(h, w), min(scale) / min(h, w)),
    new size = int(dim * factor + 0.5);
  * OpenCV ``resize`` INTER_LINEAR for CV_8U (modules/imgproc/src/resize.cpp): sample position
    (float)((d + 0.5) * scale - 0.5) with scale = 1. / (dst / src) in double, taps clamped to the image,
    coefficients cvRound(c * 2048) as short, horizontal pass in int, vertical pass
    (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
  * mmcv ``imnormalize``: float32 image, BGR -> RGB swap, (img - mean) * (1 / std) with mean and 1 / std computed in
    float64 and applied by OpenCV in float32.
"""
import numpy as np


def rescale_size(h, w, scale):
    long_edge, short_edge = max(scale), min(scale)
    factor = min(long_edge / max(h, w), short_edge / min(h, w))
    return int(h * float(factor) + 0.5), int(w * float(factor) + 0.5)


def _coefs(dst, src):
    scale = 1.0 / (float(dst) / float(src))
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    s[lo], f[lo] = 0, 0
    hi = s >= src - 1
    s[hi], f[hi] = src - 1, 0
    s1 = np.minimum(s + 1, src - 1)
    a0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
    a1 = np.rint(f * np.float32(2048)).astype(np.int64)
    return s, s1, a0, a1


def resize_linear_u8(img, new_h, new_w):
    """cv2.resize(img, (new_w, new_h), interpolation=cv2.INTER_LINEAR) for an (h, w, 3) uint8 image."""
    h, w = img.shape[:2]
    x0, x1, ax0, ax1 = _coefs(new_w, w)
    y0, y1, ay0, ay1 = _coefs(new_h, h)
    src = img.astype(np.int64)
    hor = src[:, x0, :] * ax0[None, :, None] + src[:, x1, :] * ax1[None, :, None]          # (h, new_w, 3)
    r = (((ay0[:, None, None] * (hor[y0] >> 4)) >> 16) + ((ay1[:, None, None] * (hor[y1] >> 4)) >> 16) + 2) >> 2
    return np.clip(r, 0, 255).astype(np.uint8)


def pipeline(img, scale=(640, 640), size_divisor=32, mean=(114, 114, 114), std=(255, 255, 255), to_rgb=True,
             pad_val=0, pad_before_normalize=True):
    """-> (float32 (3, Hp, Wp) tensor data, meta dict like the reference's img_metas entry)."""
    h, w = img.shape[:2]
    nh, nw = rescale_size(h, w, scale)
    res = resize_linear_u8(img, nh, nw)
    hp = int(np.ceil(nh / size_divisor)) * size_divisor
    wp = int(np.ceil(nw / size_divisor)) * size_divisor
    mean64, stdinv64 = np.float64(mean), 1 / np.float64(std)

    def normalize(a):
        a = a.astype(np.float32)
        if to_rgb:
            a = a[..., ::-1]
        return ((a - mean64.astype(np.float32)) * stdinv64.astype(np.float32)).astype(np.float32)

    if pad_before_normalize:
        canvas = np.full((hp, wp, 3), pad_val, np.uint8)
        canvas[:nh, :nw] = res
        out = normalize(canvas)
    else:
        out = np.full((hp, wp, 3), pad_val, np.float32)
        out[:nh, :nw] = normalize(res)
    meta = dict(ori_shape=(h, w, 3), img_shape=(nh, nw, 3), pad_shape=(hp, wp, 3),
                scale_factor=np.array([nw / w, nh / h, nw / w, nh / h], dtype=np.float32), flip=False)
    return np.ascontiguousarray(out.transpose(2, 0, 1)), meta
