"""The oracle's YOLOv3 restatement (oracle/yolov3_oracle.py) against the fixture produced by the
reference's own Darknet / YOLOV3Neck / YOLOV3Head / YOLOBBoxCoder (tests/golden/tiny_v3.npz)."""
import numpy as np
import torch

from conftest import state_dict_from
from oracle import yolov3_oracle as V3

SIZES = [[(116, 90), (156, 198), (373, 326)], [(30, 61), (62, 45), (59, 119)], [(10, 13), (16, 30), (33, 23)]]


def test_v3_graph_matches_reference(golden):
    g = golden('tiny_v3')
    sd = state_dict_from(g)
    img = torch.from_numpy(g['img'])
    layers = [int(v) for v in g['meta_layers']]
    feats = V3.darknet(img, sd, layers, (3, 4, 5))
    for i, f in zip((3, 4, 5), feats):
        np.testing.assert_allclose(f.numpy(), g[f'stage{i}'], rtol=1e-5, atol=1e-5)
    nouts = V3.yolov3_neck(feats, sd)
    for i, f in enumerate(nouts):
        np.testing.assert_allclose(f.numpy(), g[f'neck{i}'], rtol=1e-5, atol=1e-5)
    preds = V3.yolov3_head(nouts, sd)
    for i, f in enumerate(preds):
        np.testing.assert_allclose(f.numpy(), g[f'pred{i}'], rtol=1e-5, atol=2e-5)


def test_v3_get_bboxes_matches_reference(golden):
    g = golden('tiny_v3')
    preds = [torch.from_numpy(g[f'pred{i}']) for i in range(3)]
    sf = g['scale_factors']
    for rescale, tag in ((True, ''), (False, '_norescale')):
        res = V3.get_bboxes_v3(preds, sf, 6, nms_pre=40, score_thr=0.05, conf_thr=0.005, iou_threshold=0.45,
                               max_per_img=100, rescale=rescale)
        for n in range(2):
            np.testing.assert_array_equal(res[n][0].numpy(), g[f'dets{tag}{n}'])
            np.testing.assert_array_equal(res[n][1].numpy(), g[f'labels{tag}{n}'])
    res = V3.get_bboxes_v3(preds, sf, 6, nms_pre=-1, score_thr=0.3, conf_thr=-1, iou_threshold=0.6, max_per_img=30,
                           rescale=True)
    for n in range(2):
        np.testing.assert_array_equal(res[n][0].numpy(), g[f'cfg2/dets{n}'])
        np.testing.assert_array_equal(res[n][1].numpy(), g[f'cfg2/labels{n}'])


def test_yolo_bbox_coder_known_answers(golden):
    g = golden('tiny_v3')
    b, p = torch.from_numpy(g['coder/bboxes']), torch.from_numpy(g['coder/pred'])
    np.testing.assert_array_equal(V3.yolo_bbox_decode(b, p, 32).numpy(), g['coder/decode_s32'])
    np.testing.assert_array_equal(V3.yolo_bbox_encode(b, torch.from_numpy(g['coder/encode_gt']), 32).numpy(),
                                  g['coder/encode_s32'])
    # the reference's own expected values (tests/test_utils/test_coder.py:8-24), stride 32
    bboxes = torch.Tensor([[-42., -29., 74., 61.], [-10., -29., 106., 61.], [22., -29., 138., 61.], [54., -29., 170., 61.]])
    pred = torch.Tensor([[0.4709, 0.6152, 0.1690, -0.4056], [0.5399, 0.6653, 0.1162, -0.4162],
                         [0.4654, 0.6618, 0.1548, -0.4301], [0.4786, 0.6197, 0.1896, -0.4479]])
    expect = torch.Tensor([[-53.6102, -10.3096, 83.7478, 49.6824], [-15.8700, -8.3901, 114.4236, 50.9693],
                           [11.1822, -8.0924, 146.6034, 50.4476], [41.2068, -8.9232, 181.4236, 48.5840]])
    assert V3.yolo_bbox_decode(bboxes, pred, 32).allclose(expect, atol=1e-3)


def test_v3_training_targets_and_losses_match_reference(golden):
    """GridAssigner targets and the four v3 losses / gradients of one training step vs the reference."""
    g = golden('tiny_v3')
    gtb = [torch.from_numpy(g['train/gt_bboxes0']), torch.from_numpy(g['train/gt_bboxes1'])]
    gtl = [torch.from_numpy(g['train/gt_labels0']), torch.from_numpy(g['train/gt_labels1'])]
    sizes = [g[f'pred{i}'].shape[-2:] for i in range(3)]
    tm, nm = V3.targets_v3(sizes, gtb, gtl, 6)
    for i in range(3):
        np.testing.assert_array_equal(tm[i].numpy(), g[f'train/target_map{i}'])
        np.testing.assert_array_equal(nm[i].numpy(), g[f'train/neg_map{i}'])
    sd = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and 'running' not in k else v.clone())
          for k, v in state_dict_from(g).items()}
    layers = [int(v) for v in g['meta_layers']]
    losses = V3.forward_train_v3(torch.from_numpy(g['img']), sd, layers, gtb, gtl, 6)
    for k in ('loss_cls', 'loss_conf', 'loss_xy', 'loss_wh'):
        got = torch.stack([x.reshape(()) for x in losses[k]]).detach().numpy()
        np.testing.assert_allclose(got, g['train/' + k], rtol=2e-5, atol=1e-5, err_msg=k)
    total = sum(sum(x.mean() for x in v) for v in losses.values())
    np.testing.assert_allclose(float(total), float(g['train/loss_total']), rtol=2e-5)
    total.backward()
    names = [str(n) for n in g['train/grad_names']]
    sums = g['train/grad_sums']
    for i, n in enumerate(names):
        gr = sd[n].grad.double()
        np.testing.assert_allclose([float(gr.abs().sum()), float(gr.pow(2).sum().sqrt())], sums[i][1:], rtol=2e-3,
                                   atol=1e-6, err_msg=n)
    np.testing.assert_allclose(sd['bbox_head.convs_pred.0.bias'].grad.numpy(), g['train/grad/bbox_head.convs_pred.0.bias'],
                               rtol=1e-4, atol=1e-5)
