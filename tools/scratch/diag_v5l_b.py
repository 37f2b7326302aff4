import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch, numpy as np
import mmdet_yolov4_amd as pkg
import bench
dev = torch.device('cuda:0')
B = 8
model, size = 'yolov5l', 640
torch.manual_seed(0)
det = pkg.build_detector(bench.model_cfg(model)); det.init_weights(); det.train().to(dev)
img = bench.synthetic_images(B, size, 1000, dev)
gtb, gtl = bench.synthetic_gts(B, size, 2000, dev)
data = dict(img=img, img_metas=[dict() for _ in range(B)], gt_bboxes=gtb, gt_labels=gtl)
sd0 = {k: v.clone() for k, v in det.state_dict().items()}
res = {}
for dt in (torch.float32, torch.bfloat16, torch.float16):
    det.load_state_dict(sd0); det.zero_grad()
    pkg.wrap_fp16_model(det, dt)
    out = det.train_step(data, None)
    out['loss'].backward()
    res[dt] = {n: p.grad.double().clone() for n, p in det.named_parameters()}
for n in res[torch.float32]:
    a, b, c = res[torch.float32][n], res[torch.bfloat16][n], res[torch.float16][n]
    cosb = float((a * b).sum() / (a.norm() * b.norm() + 1e-30)); cosc = float((a * c).sum() / (a.norm() * c.norm() + 1e-30))
    print(f'{float(b.norm() / a.norm()) - 1:+.4f} cos {cosb:.4f} | fp16 {float(c.norm() / a.norm()) - 1:+.4f} cos {cosc:.4f} | {float(a.norm()):.3e} {n}')
