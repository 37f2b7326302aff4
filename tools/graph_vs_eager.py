#!/usr/bin/env python
"""hipGraph replay vs eager launch list for one inference plan.  Usage: graph_vs_eager.py [f32|bf16|f16] [batch]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import mmdet_yolov4_amd as pkg
from mmdet_yolov4_amd.calibrate import calibrate_bn

dt = {'f32': torch.float32, 'bf16': torch.bfloat16, 'f16': torch.float16}[sys.argv[1] if len(sys.argv) > 1 else 'f32']
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device('cuda:0')
torch.manual_seed(0)
det = pkg.build_detector(bench.model_cfg('yolov4l'))
det.init_weights()
det.eval().to(dev)
img = bench.synthetic_images(batch, 608, 1000, dev)
plan = det.compile(batch, 608, 608, device=dev, rescale=True)
calibrate_bn(plan, img)
bench.init_head(det, plan, img, 2000.0)
res = {}
for graph in (False, True):
    det._engines.clear()
    p = det.compile(batch, 608, 608, device=dev, rescale=True, dtype=dt, graph=graph, autotune=True)
    for _ in range(5):
        p.run(img)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        p.run(img)
    torch.cuda.synchronize()
    res['graph' if graph else 'eager'] = (time.perf_counter() - t0) / 30 * 1e3
print({k: round(v, 3) for k, v in res.items()}, 'ms/step', {k: round(batch / v * 1e3, 1) for k, v in res.items()}, 'img/s')
