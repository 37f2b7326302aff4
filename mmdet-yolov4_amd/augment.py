"""The train-side input pipeline of the YOLOv4 / YOLOv5 recipes as two launches per batch (``csrc/augment.hip``).

``FusedTrainPipeline.from_config(cfg.data.train.pipeline)`` reads the reference's ``train_pipeline`` block
(configs/yolov4/yolov4l_coco_mosaic.py:22-69): ``MosaicPipeline(individual_pipeline=[Load..., Resize(img_scale,
keep_ratio=True)], pad_val)``, the ``Albu`` block (``PadIfNeeded``, ``RandomCrop``, ``RandomScale``, ``CenterCrop``,
``HorizontalFlip`` + ``bbox_params``), ``HueSaturationValueJitter``, ``GtBBoxesFilter``, ``Normalize``; loading / format
transforms carry no arithmetic and are accepted.  Calling it with decoded source images that already live on the
device returns what the reference's collate hands the detector: ``img`` (N, 3, out, out) fp32, ``gt_bboxes`` /
``gt_labels`` lists.

What stays on the host: the random draws (which 3 other images join a mosaic is the dataset's business,
``dataset.batch_rand_others``; crop origin, scale, flip and the three colour gains are drawn here from a
``numpy.random.Generator``), the integer geometry derived from them, and the 3 x 256-byte colour LUTs
(transforms.py:2004-2008 verbatim).  One small descriptor table per batch goes to the device.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import AugImage, check
from .ops import stream_ptr


def rescale_size(h, w, scale):
    """mmcv ``rescale_size`` for ``Resize(keep_ratio=True)``."""
    long_edge, short_edge = max(scale), min(scale)
    factor = min(long_edge / max(h, w), short_edge / min(h, w))
    return int(h * float(factor) + 0.5), int(w * float(factor) + 0.5)


def hsv_luts(r):
    """HueSaturationValueJitter's three LUTs for gains r (transforms.py:2004-2008)."""
    x = np.arange(0, 256, dtype=np.int16)
    return (((x * r[0]) % 180).astype(np.uint8), np.clip(x * r[1], 0, 255).astype(np.uint8),
            np.clip(x * r[2], 0, 255).astype(np.uint8))


class FusedTrainPipeline:

    def __init__(self, img_scale=(640, 640), pad_val=114, pad_to=1920, crop=1280, scale_limit=0.5, out_size=640,
                 flip_p=0.5, min_area=4.0, min_visibility=0.2, hsv=(0.015, 0.7, 0.4), min_size=2, max_aspect_ratio=20,
                 mean=(114, 114, 114), std=(255, 255, 255), to_rgb=True, max_boxes=512, device=None):
        self.img_scale = tuple(img_scale)
        self.pad_val, self.pad_to, self.crop, self.out = int(pad_val), int(pad_to), int(crop), int(out_size)
        self.scale_limit, self.flip_p = float(scale_limit), float(flip_p)
        self.min_area, self.min_visibility = float(min_area), float(min_visibility)
        self.hsv = None if hsv is None else tuple(float(v) for v in hsv)
        self.min_size, self.max_ar = float(min_size), float(max_aspect_ratio)
        self.mean = np.asarray(mean, dtype=np.float32)
        self.std = np.asarray(std, dtype=np.float32)
        self.to_rgb = bool(to_rgb)
        self.max_boxes = int(max_boxes)
        self.device = device
        if int(self.crop * (1 - self.scale_limit)) < self.out:
            raise ValueError('CenterCrop larger than the smallest RandomScale output')

    @classmethod
    def from_config(cls, pipeline, device=None, **over):
        kw = dict(device=device)
        for t in pipeline:
            typ = t['type']
            if typ == 'MosaicPipeline':
                kw['pad_val'] = t.get('pad_val', 0)
                for s in t['individual_pipeline']:
                    if s['type'] == 'Resize':
                        if not s.get('keep_ratio', False):
                            raise NotImplementedError('MosaicPipeline: Resize(keep_ratio=False) is not built')
                        kw['img_scale'] = tuple(s['img_scale'])
                    elif s['type'] not in ('LoadImageFromFile', 'LoadAnnotations'):
                        raise NotImplementedError(f"MosaicPipeline individual transform {s['type']!r} is not built")
            elif typ == 'Albu':
                bp = t.get('bbox_params', {})
                kw['min_area'], kw['min_visibility'] = bp.get('min_area', 0.0), bp.get('min_visibility', 0.0)
                for a in t['transforms']:
                    at = a['type']
                    if at == 'PadIfNeeded':
                        assert a['min_height'] == a['min_width'] and a.get('border_mode', 0) == 0
                        kw['pad_to'] = a['min_height']
                    elif at == 'RandomCrop':
                        assert a['width'] == a['height']
                        kw['crop'] = a['width']
                    elif at == 'RandomScale':
                        if a.get('interpolation', 1) != 1:
                            raise NotImplementedError('RandomScale: only INTER_LINEAR is built')
                        kw['scale_limit'] = a['scale_limit']
                    elif at == 'CenterCrop':
                        assert a['width'] == a['height']
                        kw['out_size'] = a['width']
                    elif at == 'HorizontalFlip':
                        kw['flip_p'] = a.get('p', 0.5)
                    else:
                        raise NotImplementedError(f'Albu transform {at!r} is not built')
            elif typ == 'HueSaturationValueJitter':
                kw['hsv'] = (t.get('hue_ratio', 0.5), t.get('saturation_ratio', 0.5), t.get('value_ratio', 0.5))
            elif typ == 'GtBBoxesFilter':
                kw['min_size'], kw['max_aspect_ratio'] = t.get('min_size', 2), t.get('max_aspect_ratio', 20)
            elif typ == 'Normalize':
                kw.update(mean=t['mean'], std=t['std'], to_rgb=t.get('to_rgb', True))
            elif typ in ('DefaultFormatBundle', 'Collect', 'LoadImageFromFile', 'LoadAnnotations'):
                continue
            else:
                raise NotImplementedError(f'train pipeline transform {typ!r} is not built')
        kw.update(over)
        return cls(**kw)

    # ---- random draws (host) -----------------------------------------------------------------------------------------
    def draw_params(self, rng):
        """One sample's random parameters: RandomCrop's (h_start, w_start), RandomScale's factor, HorizontalFlip, the
        three colour gains ``uniform(-1, 1) * ratio + 1`` (transforms.py:1998-1999)."""
        p = dict(h_start=float(rng.random()), w_start=float(rng.random()),
                 scale=float(1.0 + rng.uniform(-self.scale_limit, self.scale_limit)),
                 flip=bool(rng.random() < self.flip_p), pad_to=self.pad_to, crop=self.crop, out=self.out)
        if self.hsv is not None:
            p['hsv'] = tuple(float(rng.uniform(-1.0, 1.0)) * r + 1.0 for r in self.hsv)
        return p

    def describe(self, sources, params):
        """The integer geometry of one output image -> filled ``AugImage``.  ``sources``: 4 uint8 (h, w, 3) device
        tensors (BGR as cv2 decodes)."""
        g = AugImage()
        for i, s in enumerate(sources):
            if s.dtype != torch.uint8 or s.dim() != 3 or s.shape[2] != 3 or not s.is_cuda:
                raise TypeError('source images must be uint8 (h, w, 3) tensors on the GPU')
            if s.stride(2) != 1 or s.stride(1) != 3:
                raise ValueError('source images must be dense along w and c')
            h, w = int(s.shape[0]), int(s.shape[1])
            nh, nw = rescale_size(h, w, self.img_scale)
            g.src[i] = s.data_ptr()
            g.sh[i], g.sw[i], g.pitch[i], g.rh[i], g.rw[i] = h, w, int(s.stride(0)), nh, nw
        g.cxy = max(g.rh[0], g.rh[1], g.rw[0], g.rw[2])
        side = 2 * g.cxy
        P, Cc, O = self.pad_to, self.crop, self.out
        top = int((P - side) / 2.0) if side < P else 0
        H = max(side, P)
        g.left = g.top = top
        g.y1 = int((H - Cc + 1) * params['h_start'])
        g.x1 = int((H - Cc + 1) * params['w_start'])
        g.C = Cc
        g.S = int(Cc * params['scale'])
        g.o = (g.S - O) // 2
        g.flip = 1 if params['flip'] else 0
        if 'hsv' in params and params['hsv'] is not None:
            g.hsv_on = 1
            for k, lut in enumerate(hsv_luts(np.asarray(params['hsv'], dtype=np.float64))):
                C.memmove(C.addressof(g.lut[k]), lut.ctypes.data, 256)
        return g

    # ---- device ------------------------------------------------------------------------------------------------------
    def __call__(self, samples, params=None, rng=None, return_u8=False):
        """``samples``: list (one per output image) of 4-tuples of (image uint8 (h,w,3) cuda tensor, boxes float32
        (k,4) numpy / tensor in that image's pixel coordinates, labels int (k,)).  ``params``: list of parameter dicts
        (default: drawn from ``rng``).  -> dict(img, gt_bboxes, gt_labels[, img_u8])."""
        N = len(samples)
        dev = samples[0][0][0].device
        if params is None:
            rng = rng or np.random.default_rng()
            params = [self.draw_params(rng) for _ in range(N)]
        table = (AugImage * N)()
        boxes, labels, tiles, seg = [], [], [], [0]
        for n, (four, prm) in enumerate(zip(samples, params)):
            if len(four) != 4:
                raise ValueError('a mosaic sample needs exactly 4 source images')
            table[n] = self.describe([f[0] for f in four], prm)
            for i, f in enumerate(four):
                b = np.asarray(f[1].cpu() if isinstance(f[1], torch.Tensor) else f[1], dtype=np.float32).reshape(-1, 4)
                l = np.asarray(f[2].cpu() if isinstance(f[2], torch.Tensor) else f[2]).reshape(-1)
                boxes.append(b)
                labels.append(l.astype(np.int32))
                tiles.append(np.full(len(b), i, np.int32))
            seg.append(seg[-1] + sum(len(f[1]) for f in four))
        raw = torch.frombuffer(bytearray(bytes(table)), dtype=torch.uint8)
        d_table = raw.to(dev, non_blocking=False)
        L = _lib.lib()
        O = self.out
        img = torch.empty((N, 3, O, O), dtype=torch.float32, device=dev)
        u8 = torch.empty((N, O, O, 3), dtype=torch.uint8, device=dev) if return_u8 else None
        mean, std = torch.from_numpy(self.mean), torch.from_numpy(self.std)
        check(L.yv4_mosaic_augment_u8(d_table.data_ptr(), N, O, u8.data_ptr() if return_u8 else None, img.data_ptr(),
                                      mean.data_ptr(), std.data_ptr(), int(self.to_rgb), self.pad_val, stream_ptr()),
              'yv4_mosaic_augment_u8')
        total = seg[-1]
        # a sample never keeps more boxes than its four tiles bring: size the output rows from the largest sample, so that
        # no ground truth is dropped silently (the reference's MosaicPipeline keeps every box of the four tiles;
        # `max_boxes` is only the floor of the allocation)
        cap = max(int(self.max_boxes), max((seg[n + 1] - seg[n] for n in range(N)), default=0), 1)
        ob = torch.empty((N, cap, 4), dtype=torch.float32, device=dev)
        ol = torch.empty((N, cap), dtype=torch.int32, device=dev)
        oc = torch.zeros((N,), dtype=torch.int32, device=dev)
        if total:
            hb = torch.from_numpy(np.concatenate(boxes, 0)).to(dev)
            hl = torch.from_numpy(np.concatenate(labels, 0)).to(dev)
            ht = torch.from_numpy(np.concatenate(tiles, 0)).to(dev)
            hs = torch.tensor(seg, dtype=torch.int64).to(dev)
            check(L.yv4_augment_boxes(d_table.data_ptr(), N, O, hb.data_ptr(), hl.data_ptr(), ht.data_ptr(), hs.data_ptr(),
                                      cap, self.min_area, self.min_visibility, self.min_size, self.max_ar, ob.data_ptr(),
                                      ol.data_ptr(), oc.data_ptr(), stream_ptr()), 'yv4_augment_boxes')
        counts = oc.tolist()
        out = dict(img=img, gt_bboxes=[ob[n, :counts[n]] for n in range(N)],
                   gt_labels=[ol[n, :counts[n]].to(torch.int64) for n in range(N)], params=params)
        if return_u8:
            out['img_u8'] = u8
        return out
