"""YOLOv3 row: Darknet / YOLOV3Neck / YOLOV3Head on the fused path vs the fixture produced
by the reference's own modules (tests/golden/tiny_v3.npz) and vs the oracle restatement."""
import numpy as np
import pytest
import torch

import mmdet_yolov4_amd as pkg
from conftest import state_dict_from
from oracle import yolov3_oracle as V3

pytestmark = pytest.mark.gpu


class TinyDarknet(pkg.Darknet):
    """The fixture's run-time narrowed arch (tests/golden/make_golden_v3.py)."""
    arch_settings = {53: ((1, 1, 2, 2, 1), ((32, 16), (16, 32), (32, 32), (32, 64), (64, 64)))}


TEST_CFG = dict(nms_pre=40, min_bbox_size=0, score_thr=0.05, conf_thr=0.005, nms=dict(type='nms', iou_threshold=0.45),
                max_per_img=100)


def build(g, dev):
    det = pkg.YOLOV3(backbone=dict(type='Darknet', depth=53, out_indices=(3, 4, 5)),
                     neck=dict(type='YOLOV3Neck', num_scales=3, in_channels=[1024, 512, 256],
                               out_channels=[512, 256, 128]),
                     bbox_head=dict(type='YOLOV3Head', num_classes=6, in_channels=[512, 256, 128],
                                    out_channels=[1024, 512, 256]), test_cfg=TEST_CFG)
    det.backbone = TinyDarknet(depth=53, out_indices=(3, 4, 5))
    det.neck = pkg.YOLOV3Neck(num_scales=3, in_channels=[64, 64, 32], out_channels=[64, 32, 16])
    det.bbox_head = pkg.YOLOV3Head(num_classes=6, in_channels=[64, 32, 16], out_channels=[96, 64, 32], test_cfg=TEST_CFG)
    sd = state_dict_from(g)
    assert list(det.state_dict().keys()) == list(sd.keys())          # checkpoint layout = the reference's
    det.load_state_dict(sd, strict=True)
    return det.to(dev).eval()


def close(got, ref, tol, what):
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else got
    err = np.abs(got - ref) / (1 + np.abs(ref))
    assert err.max() <= tol, f'{what}: max rel err {err.max():.3e}'


def test_v3_detector_against_reference_golden(golden, gpu_device):
    g = golden('tiny_v3')
    det = build(g, gpu_device)
    img = torch.from_numpy(g['img']).to(gpu_device)
    x = img
    for i, name in enumerate(det.backbone.cr_blocks):      # module-level API, stage by stage
        x = getattr(det.backbone, name)(x) if name == 'conv1' else _run_seq(det.backbone, name, x)
        close(x, g[f'stage{i}'], 1e-4, f'stage{i}')
    feats = det.backbone(img)
    nouts = det.neck(feats)
    for i, f in enumerate(nouts):
        close(f, g[f'neck{i}'], 1e-4, f'neck{i}')
    outs = det.bbox_head(nouts)
    assert isinstance(outs, tuple) and len(outs) == 1 and len(outs[0]) == 3
    for i, f in enumerate(outs[0]):
        close(f, g[f'pred{i}'], 1e-4, f'pred{i}')
    # get_bboxes on the REFERENCE's pred maps: same selection, boxes/scores within 1e-4
    ref_preds = [torch.from_numpy(g[f'pred{i}']).to(gpu_device) for i in range(3)]
    metas = [dict(scale_factor=g['scale_factors'][i]) for i in range(2)]
    for rescale, tag in ((True, ''), (False, '_norescale')):
        res = det.bbox_head.get_bboxes(ref_preds, metas, rescale=rescale)
        for n in range(2):
            d, l = res[n]
            assert d.dtype == torch.float32 and l.dtype == torch.int64
            np.testing.assert_array_equal(l.cpu().numpy(), g[f'labels{tag}{n}'])
            np.testing.assert_allclose(d.cpu().numpy(), g[f'dets{tag}{n}'], rtol=1e-4, atol=1e-4)
    cfg2 = pkg.registry.ConfigDict(nms_pre=-1, min_bbox_size=0, score_thr=0.3, conf_thr=-1,
                                   nms=dict(type='nms', iou_threshold=0.6), max_per_img=30)
    res = det.bbox_head.get_bboxes(ref_preds, metas, cfg=cfg2, rescale=True)
    for n in range(2):
        np.testing.assert_array_equal(res[n][1].cpu().numpy(), g[f'cfg2/labels{n}'])
        np.testing.assert_allclose(res[n][0].cpu().numpy(), g[f'cfg2/dets{n}'], rtol=1e-4, atol=1e-4)
    # bit-exact selection against the oracle on the same inputs
    ores = V3.get_bboxes_v3([p.cpu() for p in ref_preds], g['scale_factors'], 6, nms_pre=40, score_thr=0.05,
                            conf_thr=0.005, iou_threshold=0.45, max_per_img=100, rescale=True)
    res = det.bbox_head.get_bboxes(ref_preds, metas, rescale=True)
    for n in range(2):
        np.testing.assert_array_equal(res[n][1].cpu().numpy(), ores[n][1].numpy())
    # end to end through the compiled plan
    out = det.simple_test(img, metas, rescale=True)
    assert len(out) == 2 and len(out[0]) == 6
    for n in range(2):
        tot = sum(len(r) for r in out[n])
        assert abs(tot - len(g[f'dets{n}'])) <= 3


def _run_seq(backbone, name, x):
    blk = getattr(backbone, name)
    for m in blk:                   # nn.Sequential of plan-backed modules
        x = m(x)
    return x


def test_v3_per_level_topk_keys(gpu_device):
    """yv4_conf_topk_levels: the k-th (conf desc, index asc) key of every (image, level) segment; levels
    with <= k boxes admit everything."""
    from mmdet_yolov4_amd import _lib
    L = _lib.lib()
    torch.manual_seed(3)
    N, A, ncls = 2, 3, 4
    sizes = [(2, 3), (4, 6), (8, 12)]
    preds = [torch.randn(N, h, w, A * (5 + ncls), device=gpu_device) for h, w in sizes]
    levels = (_lib.LevelDesc * 3)()
    for i, (p, (h, w)) in enumerate(zip(preds, sizes)):
        levels[i].pred = p.data_ptr()
        levels[i].H, levels[i].W, levels[i].stride = h, w, 32 >> i
    boxes = [h * w * A for h, w in sizes]
    total = sum(boxes)
    work = torch.empty(L.yv4_conf_topk_levels_work(N, total, 3), dtype=torch.uint8, device=gpu_device)
    out = torch.zeros(N * 3, dtype=torch.int64, device=gpu_device)
    k = 40
    _lib.check(L.yv4_conf_topk_levels(levels, 3, N, A, ncls, k, work.data_ptr(), out.data_ptr(),
                                      torch.cuda.current_stream().cuda_stream), 'topk_levels')
    got = out.cpu().numpy().astype(np.uint64).reshape(N, 3)
    base = np.cumsum([0] + boxes)
    for n in range(N):
        for l in range(3):
            conf = preds[l][n].reshape(-1, 5 + ncls)[:, 4].sigmoid().cpu().numpy()
            if boxes[l] <= k:
                assert got[n, l] == np.uint64(0xFFFFFFFFFFFFFFFF)
            else:
                order = np.lexsort((np.arange(boxes[l]), -conf.astype(np.float64)))
                assert int(got[n, l] & np.uint64(0xffffffff)) == base[l] + order[k - 1]


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
def test_v3_runs_in_16_bit(golden, gpu_device, dtype):
    g = golden('tiny_v3')
    det = build(g, gpu_device)
    img = torch.from_numpy(g['img']).to(gpu_device)
    pkg.wrap_fp16_model(det, dtype)
    with torch.no_grad():
        preds = det.forward_dummy(img)[0]
    tol = 2e-2 if dtype == torch.float16 else 1.5e-1
    for i, p in enumerate(preds):
        close(p, g[f'pred{i}'], tol, f'{dtype} pred{i}')


def test_v3_train_step_matches_reference(golden, gpu_device):
    """One training-mode forward + backward of YOLOV3 on the HIP training ops vs the reference: GridAssigner
    targets exact, the four losses 1e-4, every parameter gradient (norms 1 %)."""
    g = golden('tiny_v3')
    train_cfg = dict(assigner=dict(type='GridAssigner', pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0))
    det = build(g, gpu_device)
    sd = det.state_dict()
    det.bbox_head = pkg.YOLOV3Head(
        num_classes=6, in_channels=[64, 32, 16], out_channels=[96, 64, 32],
        loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0, reduction='sum'),
        loss_conf=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0, reduction='sum'),
        loss_xy=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=2.0, reduction='sum'),
        loss_wh=dict(type='MSELoss', loss_weight=2.0, reduction='sum'), train_cfg=train_cfg, test_cfg=TEST_CFG)
    det.load_state_dict(sd)
    det.to(gpu_device)
    det.training = True
    for m in (det.backbone, det.neck, det.bbox_head):
        torch.nn.Module.train(m, True)     # batch-statistics BN everywhere, like the fixture (Darknet.train()'s norm_eval bypassed)
    img = torch.from_numpy(g['img']).to(gpu_device)
    gtb = [torch.from_numpy(g[f'train/gt_bboxes{i}']).to(gpu_device) for i in range(2)]
    gtl = [torch.from_numpy(g[f'train/gt_labels{i}']).to(gpu_device) for i in range(2)]
    # targets
    sizes = [g[f'pred{i}'].shape[-2:] for i in range(3)]
    anchors = det.bbox_head.anchor_generator.grid_anchors(sizes, gpu_device)
    flags = [det.bbox_head.anchor_generator.responsible_flags(sizes, b, gpu_device) for b in gtb]
    tm, nm = det.bbox_head.get_targets([anchors, anchors], flags, gtb, gtl)
    for i in range(3):
        np.testing.assert_allclose(tm[i].cpu().numpy(), g[f'train/target_map{i}'], rtol=1e-6, atol=1e-6)
        np.testing.assert_array_equal(nm[i].cpu().numpy(), g[f'train/neg_map{i}'])
    out = det.train_step(dict(img=img, img_metas=[dict(), dict()], gt_bboxes=gtb, gt_labels=gtl), None)
    np.testing.assert_allclose(out['log_vars']['loss'], float(g['train/loss_total']), rtol=1e-4)
    for k in ('loss_cls', 'loss_conf', 'loss_xy', 'loss_wh'):
        np.testing.assert_allclose(out['log_vars'][k], float(g['train/' + k].sum()), rtol=1e-4, err_msg=k)
    out['loss'].backward()
    params = dict(det.named_parameters())
    names = [str(n) for n in g['train/grad_names']]
    assert names == list(params)
    for i, n in enumerate(names):
        gr = params[n].grad.double()
        np.testing.assert_allclose([float(gr.abs().sum()), float(gr.pow(2).sum().sqrt())], g['train/grad_sums'][i][1:],
                                   rtol=1e-2, atol=1e-5, err_msg=n)
    for k in ('bbox_head.convs_pred.0.bias', 'bbox_head.convs_pred.2.weight', 'backbone.conv1.conv.weight'):
        ref = g['train/grad/' + k]
        np.testing.assert_allclose(params[k].grad.cpu().numpy(), ref, rtol=1e-2, atol=2e-3 * float(np.abs(ref).max()) + 2e-5)
    # norm_eval (the Darknet default): BN statistics stay untouched by a training step
    det.train()
    bns = [m for m in det.backbone.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    assert not any(m.training for m in bns)
    rm = [m.running_mean.clone() for m in bns]
    det.train_step(dict(img=img, img_metas=[dict(), dict()], gt_bboxes=gtb, gt_labels=gtl), None)['loss'].backward()
    assert all(torch.equal(a, m.running_mean) for a, m in zip(rm, bns))
