#!/usr/bin/env python
"""the ACTUAL reference timed on this container's host cores (build container only --
/root/reference does not travel to the GPU box; bench.py's `cpu_baseline` is the port that does).

The reference's YOLOv4-L modules (DarknetCSP v4l5p -> YOLOV4Neck -> YOLOCSPHead -> get_bboxes -> multiclass_nms) are
imported from /root/reference through tests/golden/_ref_import.py (mmcv composition surface shimmed, all arithmetic
torch's; mmcv's nms is the restated one) and timed in the two variants BASELINE.md names:
  (a) as-is: the reference's own C++ CPU Mish (mmdet/ops/mish_cuda/src/kernel/mish_cpu.cc:6-15, a single-threaded
      scalar loop, compiled from where it lies by oracle/build_ref.py) -- SURVEY Q11;
  (b) with torch.nn.functional.mish substituted, so the baseline is not flattered by that loop.
3 warm-ups + 5 timed runs, median.  Writes profiles/r02_reference_cpu.json.
    python tools/reference_cpu_time.py [--quick]
"""
import argparse
import json
import os
import statistics
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import _ref_import  # noqa: E402
import bench  # noqa: E402
from oracle import build_ref  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--quick', action='store_true', help='1 warm-up + 2 runs (smoke)')
    ap.add_argument('--out', default=os.path.join(ROOT, 'profiles', 'r02_reference_cpu.json'))
    a = ap.parse_args()
    if not _ref_import.available():
        sys.exit('the reference checkout is not present (this script runs in the build container only)')
    ref = _ref_import.install_shim(build_ref.load_ext())
    threads = bench.host_cpu_budget()
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    m = bench.MODELS['yolov4l']
    backbone = ref.darknetcsp.DarknetCSP(scale=m['scale'], out_indices=[3, 4, 5])
    neck = ref.neck.YOLOV4Neck(in_channels=m['neck_in'], out_channels=m['neck_out'], csp_repetition=m['csp_rep'])
    head = ref.head.YOLOCSPHead(num_classes=80, in_channels=m['neck_out'], train_cfg=None,
                                test_cfg=ref.ConfigDict(min_bbox_size=0, nms_pre=-1, score_thr=0.001,
                                                        nms=dict(type='nms', iou_threshold=0.65), max_per_img=300))
    for mod in (backbone, neck, head):
        torch.nn.Module.train(mod, False)                 # DarknetCSP.train() returns None (Q3)
    with torch.no_grad():
        for conv in head.convs_pred:
            conv.weight.normal_(0, 0.02)
            conv.bias.view(3, 85)[:, 4:] = 0.0
    mish_probe = sys.modules['mmdet.ops.mish_cuda.mish']
    fast = mish_probe.Mish.forward
    mish_probe.Mish.forward = lambda self, x: F.mish(x)    # (set-up only: the timed variants choose their own below)
    with torch.no_grad():
        # a head that passes ~2000 (box, class) candidates per image, the regime bench.py puts the GPU path in: the
        # objectness bias is bisected on the raw maps of one image (logits are linear in the bias)
        outs = head(neck(backbone(bench.synthetic_images(1, 416, 99, 'cpu'))))[0]
        raw = [o.view(1, 3, 85, -1) for o in outs]
        lo, hi = -30.0, 10.0
        for _ in range(40):
            mid = 0.5 * (lo + hi)
            n = sum(int(((r[:, :, 4:5] + mid).sigmoid() * r[:, :, 5:].sigmoid() > 0.001).sum()) for r in raw)
            lo, hi = (lo, mid) if n > 2000 else (mid, hi)
        for conv in head.convs_pred:
            conv.bias.view(3, 85)[:, 4] = lo
        print('objectness bias', round(lo, 3), 'candidates', n, flush=True)
    mish_probe.Mish.forward = fast
    mish_mod = sys.modules['mmdet.ops.mish_cuda.mish']
    as_is = mish_mod.Mish.forward

    def run(img, metas):
        with torch.no_grad():
            outs = head(neck(backbone(img)))
            return head.get_bboxes(*outs, metas, rescale=True)

    warm, reps = (1, 2) if a.quick else (3, 5)
    rows = []
    for variant in ('as-is (reference C++ serial Mish)', 'F.mish substituted'):
        mish_mod.Mish.forward = as_is if variant.startswith('as-is') else (lambda self, x: F.mish(x))
        for size, batch in ((416, 1), (608, 1), (608, 4)):
            if a.quick and batch > 1:
                continue
            img = bench.synthetic_images(batch, size, 99, 'cpu')
            metas = [dict(scale_factor=[1.0, 1.0, 1.0, 1.0]) for _ in range(batch)]
            for _ in range(warm):
                res = run(img, metas)
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                res = run(img, metas)
                ts.append(time.perf_counter() - t0)
            med = statistics.median(ts)
            rows.append(dict(variant=variant, input=f'{size}x{size}', batch=batch, seconds_per_forward=round(med, 3),
                             images_per_sec=round(batch / med, 4), runs=reps, warmups=warm,
                             detections_image0=int(res[0][0].shape[0])))
            print(rows[-1], flush=True)
    mish_mod.Mish.forward = as_is
    out = dict(what='the reference (zhanggefan/mmdet-yolov4) YOLOv4-L forward + get_bboxes + multiclass_nms on the host CPU',
               host=dict(threads=threads, cpu=open('/proc/cpuinfo').read().split('model name')[1].split('\n')[0].strip(': \t')
                         if os.path.exists('/proc/cpuinfo') else None, torch=torch.__version__),
               rows=rows,
               note='build container (no GPU); mmcv composition surface shimmed, batched_nms = the restated definition; '
                    'random weights, head biased to a realistic candidate count')
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, 'w') as f:
        json.dump(out, f, indent=1)
    print('wrote', a.out)


if __name__ == '__main__':
    main()
