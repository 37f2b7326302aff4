import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
from torch.profiler import profile, ProfilerActivity
import bench
import mmdet_yolov4_amd as pkg
from mmdet_yolov4_amd import hooks as H
from mmdet_yolov4_amd.optim import build_optimizer
dev = torch.device('cuda:0')
B = 32
torch.manual_seed(0)
det = pkg.build_detector(bench.model_cfg('yolov4l')); det.init_weights(); det.train().to(dev)
pkg.wrap_fp16_model(det, torch.bfloat16)
img = bench.synthetic_images(B, 608, 1000, dev)
gtb, gtl = bench.synthetic_gts(B, 608, 2000, dev)
data = dict(img=img, img_metas=[dict() for _ in range(B)], gt_bboxes=gtb, gt_labels=gtl)
opt = build_optimizer(det, dict(type='SGD', lr=0.01, momentum=0.937, weight_decay=0.0005, nesterov=True,
                                paramwise_cfg=dict(bias_decay_mult=0., norm_decay_mult=0.)))
runner = H.Runner(det, opt, max_epochs=1)
runner.log_buffer = None
runner.register_hook_from_cfg(dict(type='DetailedLinearWarmUpHook', warmup_iters=10000, priority='NORMAL'))
runner.register_hook_from_cfg(dict(type='StateEMAHook', momentum=0.9999, interval=1, warm_up=10000, priority='HIGH'))
runner.register_hook(H.Fp16GradAccumulateOptimizerHook(accumulation=1, grad_clip=dict(max_norm=35, norm_type=2), loss_scale='dynamic'), 'ABOVE_NORMAL')
runner.data_loader = H.BatchSource([data], B)
runner.call_hook('before_run'); runner.call_hook('before_train_epoch')
def step():
    runner.call_hook('before_train_iter')
    runner.outputs = det.train_step(data, opt)
    runner.call_hook('after_train_iter')
    runner.iter += 1
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
# aggregate non-yv4 device kernels/memcpys by the top python frame inside the package
import collections
agg = collections.defaultdict(lambda: [0.0, 0])
for ev in prof.events():
    if ev.device_type.name != 'CUDA' and getattr(ev, 'device_time_total', 0) == 0:
        continue
for ev in prof.key_averages(group_by_stack_n=8):
    dt = getattr(ev, 'device_time_total', None) or getattr(ev, 'cuda_time_total', 0)
    if dt <= 0: continue
    if ev.key.startswith('yv4'): continue
    stack = [s for s in (ev.stack or []) if 'mmdet-yolov4_amd' in s or 'mmdet_yolov4_amd' in s or 'tools/' in s]
    where = stack[0].strip()[-90:] if stack else '(no package frame)'
    agg[(ev.key[:40], where)][0] += dt; agg[(ev.key[:40], where)][1] += ev.count
tot = sum(v[0] for v in agg.values())
print('non-yv4 device time per step (us):', round(tot))
for (k, w), (t, c) in sorted(agg.items(), key=lambda x: -x[1][0])[:40]:
    print(f'{t:9.0f}us x{c:4d}  {k:40s} {w}')
