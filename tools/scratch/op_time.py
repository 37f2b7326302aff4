import sys, os, ctypes as C, torch
sys.path.insert(0, '/root/repo')
import bench as B
import mmdet_yolov4_amd as pkg
from mmdet_yolov4_amd.calibrate import calibrate_bn
dev = torch.device('cuda:0')
torch.manual_seed(0)
det = pkg.build_detector(B.model_cfg('yolov4l')); det.init_weights(); det.eval().to(dev)
img = B.synthetic_images(32, 608, 1000, dev)
plan = det.compile(32, 608, 608, device=dev, rescale=True)
calibrate_bn(plan, img)
det._engines.clear()
plan = det.compile(32, 608, 608, device=dev, rescale=True, dtype=torch.bfloat16)
plan.run(img); torch.cuda.synchronize()
sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
convs = [o for o in plan.ops if o.kind == 'conv']
for idx in range(50, 60):
    op = convs[idx]
    ts = []
    for rep in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); op.fn(sp); e1.record(); torch.cuda.synchronize()
        ts.append(round(e0.elapsed_time(e1) * 1e3, 1))
    i = op.info
    L = i['launch']
    print(idx, op.name, i['Cin'], i['Cout'], i['k'], i['stride'], i['H'], 'x', L['x'].name if hasattr(L['x'], 'name') else '', hex(L['x'].ptr() % (1 << 24)), 'y', hex(L['y'].ptr() % (1 << 24)), ts)
    t = L['x'].tensor.float()
    print('    x finite', bool(torch.isfinite(t).all()), 'absmax', float(t.abs().max()), 'subnormal frac', float(((t != 0) & (t.abs() < 1e-30)).float().mean()))
