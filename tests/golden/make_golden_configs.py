"""Fixture N1 -- "configs/yolov4/* run unchanged" (BASELINE.json north_star).

Parses the reference's 12 YOLOv4 / YOLOv5 config FILES where they lie (/root/reference/configs/{yolov4,yolov5,
yolov5_ddp}/*.py, python-file configs with ``_base_`` inheritance) with THIS package's ``Config.fromfile``, builds every
detector twice -- from the reference's own module classes (imported through _ref_import.py) and from this package's
registry -- and stores, per config: the parsed ``model`` block, the optimizer / optimizer_config / lr_config /
custom_hooks blocks, samples_per_gpu, and from the REFERENCE build the parameter counts per part and a digest of the
state-dict layout (key order + shapes).  tests/test_host_logic.py rebuilds every detector from the stored blocks and
compares.  Run in the build container only (the reference does not travel):
    python tests/golden/make_golden_configs.py
"""
import glob
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import _ref_import  # noqa: E402
from oracle import build_ref  # noqa: E402


def layout_digest(module):
    h = hashlib.sha256()
    for k, v in module.state_dict().items():
        h.update(f'{k}:{tuple(v.shape)}:{v.dtype}\n'.encode())
    return h.hexdigest()


def plain(x):
    if isinstance(x, dict):
        return {k: plain(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [plain(v) for v in x]
    return x


def main():
    if not _ref_import.available():
        print('reference not present: nothing to do')
        return
    import mmdet_yolov4_amd as pkg
    ref = _ref_import.install_shim(build_ref.load_ext())
    REFCLS = {'DarknetCSP': ref.darknetcsp.DarknetCSP, 'YOLOV4Neck': ref.neck.YOLOV4Neck,
              'YOLOV5Neck': ref.neck.YOLOV5Neck, 'YOLOCSPHead': ref.head.YOLOCSPHead}
    out = {}
    files = sorted(glob.glob(os.path.join(_ref_import.REF, 'configs', 'yolov4', '*.py')) +
                   glob.glob(os.path.join(_ref_import.REF, 'configs', 'yolov5', '*.py')) +
                   glob.glob(os.path.join(_ref_import.REF, 'configs', 'yolov5_ddp', '*.py')))
    for f in files:
        cfg = pkg.Config.fromfile(f)
        model = plain(cfg.model.to_dict())
        parts = {}
        for part in ('backbone', 'neck', 'bbox_head'):
            args = dict(model[part])
            cls = REFCLS[args.pop('type')]
            if part == 'bbox_head':
                # train_cfg=dict() of the configs makes the reference head build a PseudoSampler from mmdet's sampler
                # registry (yolocsp_head.py:124-131), which the import shim does not carry; it owns no parameters
                args.update(train_cfg=None, test_cfg=ref.ConfigDict(model.get('test_cfg') or {}))
            m = cls(**args)
            parts[part] = dict(params=sum(p.numel() for p in m.parameters()),
                               tensors=len(list(m.parameters())), buffers=len(list(m.buffers())),
                               layout_sha256=layout_digest(m))
        det = pkg.build_detector(cfg.model)
        ours = {part: sum(p.numel() for p in getattr(det, part).parameters()) for part in parts}
        assert ours == {k: v['params'] for k, v in parts.items()}, (f, ours, parts)
        key = os.path.relpath(f, os.path.join(_ref_import.REF, 'configs'))
        out[key] = dict(model=model, reference_parts=parts,
                        optimizer=plain(cfg.optimizer.to_dict()), optimizer_config=plain(cfg.optimizer_config.to_dict()),
                        lr_config=plain(cfg.lr_config.to_dict()),
                        fp16=plain(cfg.fp16.to_dict()) if 'fp16' in cfg else None,
                        custom_hooks=plain([h.to_dict() for h in cfg.custom_hooks]),
                        samples_per_gpu=cfg.data['samples_per_gpu'],
                        test_pipeline=plain([p.to_dict() for p in cfg.data['test']['pipeline']]),
                        total_epochs=cfg.get('total_epochs', cfg.get('runner', {}).get('max_epochs')
                                             if isinstance(cfg.get('runner'), dict) else None),
                        dist_params=plain(cfg.dist_params.to_dict()) if 'dist_params' in cfg else None)
        print(key, {k: v['params'] for k, v in parts.items()}, 'total', sum(v['params'] for v in parts.values()))
    with open(os.path.join(HERE, 'configs_n1.json'), 'w') as fo:
        json.dump(out, fo, indent=1, sort_keys=True)
    print('wrote configs_n1.json with', len(out), 'configs')


if __name__ == '__main__':
    main()
