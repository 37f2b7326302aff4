#!/usr/bin/env python
"""Full-size sanity run of the registered detectors (GPU box): builds yolov4{s,m,l,x} and yolov5l style
configs with random weights, runs simple_test at batch 1 and 4, prints timings and detection counts."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mmdet_yolov4_amd as pkg
from mmdet_yolov4_amd.calibrate import calibrate_bn

TEST = dict(min_bbox_size=0, nms_pre=-1, score_thr=0.001, nms=dict(type='nms', iou_threshold=0.65), max_per_img=300)
CFGS = {
  'yolov4s': dict(bb=dict(type='DarknetCSP', scale='v4s5p', out_indices=[3, 4, 5]), neck=dict(type='YOLOV4Neck', in_channels=[128, 256, 256], out_channels=[128, 256, 512], csp_repetition=1), head=[128, 256, 512], size=416),
  'yolov4m': dict(bb=dict(type='DarknetCSP', scale='v4m5p', out_indices=[3, 4, 5]), neck=dict(type='YOLOV4Neck', in_channels=[192, 384, 384], out_channels=[192, 384, 768], csp_repetition=1), head=[192, 384, 768], size=608),
  'yolov4l': dict(bb=dict(type='DarknetCSP', scale='v4l5p', out_indices=[3, 4, 5]), neck=dict(type='YOLOV4Neck', in_channels=[256, 512, 512], out_channels=[256, 512, 1024], csp_repetition=2), head=[256, 512, 1024], size=608),
  'yolov4x': dict(bb=dict(type='DarknetCSP', scale='v4x5p', out_indices=[3, 4, 5]), neck=dict(type='YOLOV4Neck', in_channels=[320, 640, 640], out_channels=[320, 640, 1280], csp_repetition=3), head=[320, 640, 1280], size=608),
  'yolov5l': dict(bb=dict(type='DarknetCSP', scale='v5l5p', out_indices=[2, 3, 4]), neck=dict(type='YOLOV5Neck', in_channels=[256, 512, 1024], out_channels=[256, 512, 1024], csp_repetition=3), head=[256, 512, 1024], size=640),
}
dev = torch.device('cuda:0')
for name, c in CFGS.items():
    torch.manual_seed(0)
    det = pkg.build_detector(dict(type='SingleStageDetector', backbone=c['bb'], neck=c['neck'],
                                  bbox_head=dict(type='YOLOCSPHead', num_classes=80, in_channels=c['head']),
                                  train_cfg=dict(), test_cfg=TEST))
    det.init_weights(); det.eval().to(dev)
    S = c['size']
    for B in (1, 4):
        img = torch.randn(B, 3, S, S, device=dev)
        plan = det.compile(B, S, S, device=dev, rescale=True)
        calibrate_bn(plan, img)
        det._engines.clear()
        metas = [dict(scale_factor=np.ones(4, np.float32)) for _ in range(B)]
        res = det.simple_test(img, metas, rescale=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            res = det.simple_test(img, metas, rescale=True)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        ndet = [sum(len(r) for r in im) for im in res]
        gf = sum(o.flops for o in det.compile(B, S, S, device=dev, rescale=True).ops) / 1e9
        print(f'{name:8s} {S}x{S} batch {B}: {dt*1e3:7.2f} ms/step {B/dt:7.1f} img/s  {gf/dt/1e3:6.1f} TFLOP/s  dets/img {ndet}', flush=True)
