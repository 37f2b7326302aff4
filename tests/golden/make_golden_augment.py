"""Golden vectors of the train-side transforms the fork adds (mmdet/datasets/pipelines/transforms.py:1906-2052), made
by running the REFERENCE's own classes in the build container:

  * ``MosaicPipeline.__call__`` -- the 4-image stitch, canvas size, box shift and label concatenation -- on synthetic
    already-"loaded and resized" samples handed to it by a stand-in dataset / individual pipeline (those two are
    composition: which images join a mosaic and how they were decoded is not what is pinned here);
  * ``GtBBoxesFilter.__call__`` on boxes around its three thresholds.
``HueSaturationValueJitter`` calls ``cv2.cvtColor`` (OpenCV is absent from the image) and cannot be run: its LUT lines are
numpy and restated verbatim in oracle/augment_oracle.py; the colour conversion stays "parity unpinned".

transforms.py imports cv2 / mmcv / mmdet.core at module level; they are registered as empty shells (nothing under test
calls into them).  Output: tests/golden/augment.npz.
    python tests/golden/make_golden_augment.py
"""
import importlib
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import _ref_import  # noqa: E402
from oracle import build_ref  # noqa: E402


def import_transforms():
    _ref_import.install_shim(build_ref.load_ext())
    sys.modules['cv2'] = types.ModuleType('cv2')                       # absent; nothing below calls it
    core = sys.modules['mmdet.core']
    core.PolygonMasks = type('PolygonMasks', (), {})
    ev = types.ModuleType('mmdet.core.evaluation')
    bo = types.ModuleType('mmdet.core.evaluation.bbox_overlaps')
    bo.bbox_overlaps = lambda *a, **k: None
    sys.modules['mmdet.core.evaluation'] = ev
    sys.modules['mmdet.core.evaluation.bbox_overlaps'] = bo
    ds = os.path.join(_ref_import.REF, 'mmdet', 'datasets')
    _ref_import._pkg('mmdet.datasets', ds)
    _ref_import._pkg('mmdet.datasets.pipelines', os.path.join(ds, 'pipelines'))
    reg = _ref_import._Registry('pipeline')
    _ref_import._mod('mmdet.datasets.builder', PIPELINES=reg)

    class Compose:                                                     # composition only
        def __init__(self, transforms):
            self.transforms = transforms

        def __call__(self, data):
            for t in self.transforms:
                data = t(data)
            return data
    _ref_import._mod('mmdet.datasets.pipelines.compose', Compose=Compose)
    return importlib.import_module('mmdet.datasets.pipelines.transforms')


class FakeDataset:
    """What MosaicPipeline touches of a dataset (transforms.py:1917-1927)."""

    def __init__(self, samples):
        self.samples = samples
        self.data_infos = list(range(len(samples)))
        self.proposals = None

    def batch_rand_others(self, idx, n):
        return [(idx + k + 1) % len(self.samples) for k in range(n)]

    def get_ann_info(self, idx):
        return idx

    def pre_pipeline(self, results):
        pass


def main():
    if not _ref_import.available():
        print('reference not present: nothing to do')
        return
    T = import_transforms()
    rng = np.random.RandomState(2024)
    data = {}
    # (the stitch is size-independent: small images keep the fixture small)
    cases = {'square': [(96, 96), (96, 72), (64, 96), (96, 96)],
             'ragged': [(72, 96), (96, 54), (50, 75), (77, 96)],
             'small': [(37, 64), (64, 64), (64, 21), (50, 33)]}
    for tag, sizes in cases.items():
        samples = []
        for (h, w) in sizes:
            k = rng.randint(0, 6)
            xy = rng.rand(k, 2) * [w, h]
            wh = rng.rand(k, 2) * [w / 2, h / 2] + 1
            b = np.concatenate([xy, np.minimum(xy + wh, [w, h])], 1).astype(np.float32)
            samples.append(dict(img=rng.randint(0, 256, (h, w, 3)).astype(np.uint8), gt_bboxes=b,
                                gt_labels=rng.randint(0, 80, k).astype(np.int64)))
        ds = FakeDataset(samples)

        def individual(results):                                       # stands in for Load + Resize
            s = samples[results['_idx']]
            results.update(img=s['img'].copy(), gt_bboxes=s['gt_bboxes'].copy(), gt_labels=s['gt_labels'].copy(),
                           img_shape=s['img'].shape, pad_shape=s['img'].shape, img_fields=['img'],
                           bbox_fields=['gt_bboxes'])
            return results
        mp = object.__new__(T.MosaicPipeline)
        mp.individual_pipeline = individual
        mp.pad_val = 114
        out = mp(dict(_idx=0, dataset=ds, img_info=0, ann_info=0))
        for i, s in enumerate(samples):
            data[f'{tag}/img{i}'], data[f'{tag}/boxes{i}'], data[f'{tag}/labels{i}'] = s['img'], s['gt_bboxes'], s['gt_labels']
        data[f'{tag}/canvas'], data[f'{tag}/out_boxes'], data[f'{tag}/out_labels'] = out['img'], out['gt_bboxes'], out['gt_labels']
        data[f'{tag}/img_shape'] = np.array(out['img_shape'])
        print(tag, 'canvas', out['img'].shape, 'boxes', out['gt_bboxes'].shape)
    # GtBBoxesFilter around its thresholds (w or h == min_size, aspect ratio == 20, zero-size, huge ratio)
    f = T.GtBBoxesFilter(min_size=2, max_aspect_ratio=20)
    b = np.array([[0, 0, 2, 10], [0, 0, 2.0001, 10], [5, 5, 7.5, 55], [5, 5, 7.5, 55.0001], [5, 5, 7.5, 54.9999],
                  [1, 1, 1, 1], [0, 0, 100, 3], [0, 0, 100, 5.0], [0, 0, 100, 5.001], [10, 10, 300, 200],
                  [3, 3, 5, 5], [3, 3, 5.5, 5.5], [0, 0, 0.5, 400]], dtype=np.float32)
    b = np.concatenate([b, (rng.rand(200, 4) * 40).astype(np.float32)], 0)
    b[13:, 2:] = b[13:, :2] + (rng.rand(200, 2) ** 3 * 60).astype(np.float32)
    lab = np.arange(len(b)).astype(np.int64)
    res = f(dict(gt_bboxes=b.copy(), gt_labels=lab.copy()))
    data['filter/boxes'], data['filter/labels'] = b, lab
    data['filter/out_boxes'], data['filter/out_labels'] = res['gt_bboxes'], res['gt_labels']
    print('GtBBoxesFilter', len(b), '->', len(res['gt_bboxes']))
    np.savez_compressed(os.path.join(HERE, 'augment.npz'), **data)
    print('wrote augment.npz', os.path.getsize(os.path.join(HERE, 'augment.npz')) // 1024, 'KB')


if __name__ == '__main__':
    main()
