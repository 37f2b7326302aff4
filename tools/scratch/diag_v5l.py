import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch, numpy as np
import mmdet_yolov4_amd as pkg
import bench
dev = torch.device('cuda:0')
B = 8
for model, size in (('yolov5l', 640), ('yolov4l', 608)):
    torch.manual_seed(0)
    det = pkg.build_detector(bench.model_cfg(model)); det.init_weights(); det.train().to(dev)
    img = bench.synthetic_images(B, size, 1000, dev)
    gtb, gtl = bench.synthetic_gts(B, size, 2000, dev)
    metas = [dict() for _ in range(B)]
    sd0 = {k: v.clone() for k, v in det.state_dict().items()}
    res = {}
    for dt in (torch.float32, torch.bfloat16):
        det.load_state_dict(sd0); det.zero_grad()
        pkg.wrap_fp16_model(det, dt)
        feats = det.extract_feat(img)
        print(model, dt, 'feature rms', [float(f.float().pow(2).mean().sqrt()) for f in feats])
        raws = det.bbox_head.fwd_raw(feats)
        dense = [r.dense() for r in raws]
        for l, d in enumerate(dense):
            N, C, H, W = d.shape
            x = d.view(N, 3, 85, H, W)
            print(f'  level {l}: raw rms {float(raws[l].raw.float().pow(2).mean().sqrt()):.5f} mean sigmoid(conf) {float(x[:, :, 4].sigmoid().double().mean()):.6e} '
                  f'mean sigmoid(cls) {float(x[:, :, 5:].sigmoid().double().mean()):.6e}')
        losses = det.bbox_head.loss(raws, gtb, gtl, metas)
        tot, lv = det._parse_losses(losses)
        tot.backward()
        res[dt] = {n: p.grad.clone() for n, p in det.bbox_head.named_parameters()}
        # the same through the tensor-op loss on the dense maps (autograd), for the bias gradient
        os.environ['YV4_FUSED_LOSS'] = '0'
        leaves = [d.detach().clone().requires_grad_(True) for d in dense]
        l2 = det.bbox_head.loss(leaves, gtb, gtl, metas)
        t2, lv2 = det._parse_losses(l2)
        t2.backward()
        os.environ['YV4_FUSED_LOSS'] = '1'
        for l in range(3):
            gb = leaves[l].grad.sum(dim=(0, 2, 3))
            fb = res[dt][f'convs_pred.{l}.bias']
            print(f'  level {l}: |dbias| fused {float(fb.norm()):.6e} autograd-dense {float(gb.norm()):.6e} conf entries fused {fb.view(3,85)[:,4].tolist()} dense {gb.view(3,85)[:,4].tolist()}')
        print('  losses', {k: round(v, 5) for k, v in lv.items()}, 'dense-path', {k: round(v, 5) for k, v in lv2.items()})
