"""Test-time input pipeline on the GPU: ``Resize(keep_ratio=True) -> Pad(size_divisor) -> Normalize ->
ImageToTensor -> collate`` for a list of 8-bit HWC images, one ``yv4_letterbox_u8`` launch per image, straight
into the NCHW fp32 batch ``SingleStageDetector.simple_test`` takes, plus the ``img_metas`` the reference's pipeline
would have produced (``ori_shape``, ``img_shape``, ``pad_shape``, ``scale_factor``, ``flip``).

Mirrors the ``test_pipeline`` block of ``configs/yolov4/yolov4l_coco_mosaic.py:70-84`` (transform classes of
``mmdet/datasets/pipelines/transforms.py`` and the batch padding of mmcv's ``collate``).  **Parity unpinned**: the
arithmetic of those transforms is mmcv's and OpenCV's, third party and absent from the build image; the kernel is
tested bit for bit against ``oracle/preprocess_oracle.py``, a restatement of their published algorithms.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from ._lib import check
from .ops import stream_ptr


def rescale_size(h, w, scale):
    """mmcv ``rescale_size`` with a (long, short) tuple: the largest factor keeping both edges inside."""
    factor = min(max(scale) / max(h, w), min(scale) / min(h, w))
    return int(h * float(factor) + 0.5), int(w * float(factor) + 0.5)


class FusedTestPipeline:
    """``FusedTestPipeline(img_scale=(640, 640), size_divisor=32, mean=..., std=..., to_rgb=True)(images)`` ->
    ``(batch (N, 3, H, W) fp32 on the GPU, img_metas)``.  ``pad_before_normalize``: the order of ``Pad`` and
    ``Normalize`` in the config (the YOLOv4 configs pad first, so the border carries ``(0 - mean) / std``)."""

    def __init__(self, img_scale=(640, 640), size_divisor=32, mean=(114, 114, 114), std=(255, 255, 255), to_rgb=True,
                 pad_val=0, pad_before_normalize=True, device=None):
        self.img_scale, self.size_divisor = tuple(img_scale), int(size_divisor)
        self.mean = np.asarray(mean, dtype=np.float32)
        self.std = np.asarray(std, dtype=np.float32)
        self.to_rgb, self.pad_val, self.pad_first = bool(to_rgb), int(pad_val), bool(pad_before_normalize)
        self.device = device

    @classmethod
    def from_config(cls, pipeline, device=None):
        """Build from a reference ``test_pipeline`` list (``cfg.data.test.pipeline``): ``LoadImageFromFile`` and
        ``Collect`` / ``ImageToTensor`` / a no-flip ``RandomFlip`` carry no arithmetic and are accepted; the geometry
        comes from ``MultiScaleFlipAug(img_scale, flip=False)`` + ``Resize(keep_ratio=True)``, ``Pad(size_divisor)``
        and ``Normalize(mean, std, to_rgb)``, and their order decides whether the border is normalised."""
        kw, seen = dict(device=device), []

        def walk(items):
            for t in items:
                typ = t['type']
                if typ == 'MultiScaleFlipAug':
                    if t.get('flip', False):
                        raise NotImplementedError('test-time flip augmentation is not built')
                    scale = t['img_scale']
                    if isinstance(scale, list):
                        if len(scale) != 1:
                            raise NotImplementedError('multi-scale testing is not built')
                        scale = scale[0]
                    kw['img_scale'] = tuple(scale)
                    walk(t['transforms'])
                elif typ == 'Resize':
                    if not t.get('keep_ratio', False):
                        raise NotImplementedError('Resize(keep_ratio=False) is not built')
                    if 'img_scale' in t and t['img_scale'] is not None:
                        kw['img_scale'] = tuple(t['img_scale'])
                    seen.append(typ)
                elif typ == 'Pad':
                    if t.get('size', None) is not None:
                        raise NotImplementedError('Pad(size=...) is not built')
                    kw['size_divisor'] = t['size_divisor']
                    kw['pad_val'] = t.get('pad_val', 0)
                    seen.append(typ)
                elif typ == 'Normalize':
                    kw.update(mean=t['mean'], std=t['std'], to_rgb=t.get('to_rgb', True))
                    seen.append(typ)
                elif typ in ('LoadImageFromFile', 'RandomFlip', 'ImageToTensor', 'Collect', 'DefaultFormatBundle'):
                    continue
                else:
                    raise NotImplementedError(f'test pipeline transform {typ!r} is not built')
        walk(pipeline)
        if 'Resize' not in seen or 'Normalize' not in seen or 'img_scale' not in kw:
            raise ValueError('the test pipeline needs MultiScaleFlipAug / Resize and Normalize')
        if 'Pad' not in seen:
            kw['size_divisor'] = 1
        else:
            kw['pad_before_normalize'] = seen.index('Pad') < seen.index('Normalize')
        return cls(**kw)

    def __call__(self, images):
        if not torch.cuda.is_available():
            raise RuntimeError('FusedTestPipeline runs on the GPU through libyv4_hip.so (there is no CPU fallback)')
        dev = torch.device(self.device) if self.device is not None else torch.device('cuda', torch.cuda.current_device())
        geo = []
        for img in images:
            if img.dtype != np.uint8 or img.ndim != 3 or img.shape[2] != 3:
                raise ValueError('FusedTestPipeline takes (h, w, 3) uint8 images')
            h, w = img.shape[:2]
            nh, nw = rescale_size(h, w, self.img_scale)
            d = self.size_divisor
            geo.append((h, w, nh, nw, int(math.ceil(nh / d)) * d, int(math.ceil(nw / d)) * d))
        H = max(g[4] for g in geo)
        W = max(g[5] for g in geo)
        # collate pads every image of the batch to the largest padded shape with zeros (after Normalize)
        batch = torch.zeros((len(images), 3, H, W), dtype=torch.float32, device=dev)
        metas = []
        L = _lib.lib()
        mean_p = self.mean.ctypes.data_as(C.c_void_p)
        std_p = self.std.ctypes.data_as(C.c_void_p)
        keep = []
        for i, (img, (h, w, nh, nw, hp, wp)) in enumerate(zip(images, geo)):
            src = torch.from_numpy(np.ascontiguousarray(img)).to(dev, non_blocking=True)
            keep.append(src)
            slot = batch[i]
            if (hp, wp) == (H, W):
                check(L.yv4_letterbox_u8(src.data_ptr(), h, w, 3 * w, slot.data_ptr(), hp, wp, H * W, nh, nw, mean_p, std_p,
                                         int(self.to_rgb), self.pad_val, int(self.pad_first), stream_ptr()),
                      'yv4_letterbox_u8')
            else:       # a smaller image of a ragged batch: produce it densely, then place it in its zero-padded slot
                tmp = torch.empty((3, hp, wp), dtype=torch.float32, device=dev)
                check(L.yv4_letterbox_u8(src.data_ptr(), h, w, 3 * w, tmp.data_ptr(), hp, wp, hp * wp, nh, nw, mean_p, std_p,
                                         int(self.to_rgb), self.pad_val, int(self.pad_first), stream_ptr()),
                      'yv4_letterbox_u8')
                slot[:, :hp, :wp] = tmp
            metas.append(dict(ori_shape=(h, w, 3), img_shape=(nh, nw, 3), pad_shape=(hp, wp, 3),
                              scale_factor=np.array([nw / w, nh / h, nw / w, nh / h], dtype=np.float32), flip=False,
                              img_norm_cfg=dict(mean=self.mean, std=self.std, to_rgb=self.to_rgb)))
        return batch, metas
