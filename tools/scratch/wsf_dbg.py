import sys, os, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import mmdet_yolov4_amd as pkg
from mmdet_yolov4_amd import _lib as L
import test_gpu_parity as T
dev = torch.device('cuda:0')
for two in (False, True):
  for act in (0, 1):
    outs = [T._conv_case(dev, 2, 19, 19, 128, 128, 1, 1, 0, act=act, tile=t, two_stage=two, y_off=4, raw=True) for t in (9, 5, 6, 7)]
    print('two', two, 'act', act, [float((outs[0] - o).abs().max()) for o in outs[1:]], float((outs[1]-outs[2]).abs().max()), int((outs[0] != outs[1]).sum()), outs[0].numel())
