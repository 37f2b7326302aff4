#!/usr/bin/env python
"""Compare step time of the end-to-end plan: eager launches with events, eager, hipGraph replay."""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import mmdet_yolov4_amd as pkg
from mmdet_yolov4_amd.calibrate import calibrate_bn

dev = torch.device('cuda:0')
torch.manual_seed(0)
det = pkg.build_detector(bench.model_cfg('yolov4l')); det.init_weights(); det.eval().to(dev)
img = bench.synthetic_images(32, 608, 1000, dev)
plan = det.compile(32, 608, 608, device=dev, rescale=True)
calibrate_bn(plan, img)
bench.init_head(det, plan, img, 2000.0)
stream = torch.cuda.current_stream(); sptr = ctypes.c_void_p(stream.cuda_stream)
plan.inputs[0]['src'] = img

def eager(ev):
    for op in plan.ops:
        if ev and op.kind == 'conv':
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(stream); op.fn(sptr); e1.record(stream)
        else:
            op.fn(sptr)

def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

print('eager + events  ms/step', round(timeit(lambda: eager(True)), 3))
print('eager           ms/step', round(timeit(lambda: eager(False)), 3))
plan.capture()
print('hipGraph replay ms/step', round(timeit(lambda: plan.run(img)), 3))
