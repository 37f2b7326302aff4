"""Pin the training-side oracle (target assignment, losses, train-mode graph, gradients) against the
golden vectors produced by the reference (tests/golden/train_v4.npz).  CPU only."""
import numpy as np
import torch

from conftest import arch_from, state_dict_from
from oracle import yolov4_oracle as O


def test_responsible_indices_match_reference(golden):
    g = golden('train_v4')
    gts = [torch.from_numpy(g['resp_gt0']), torch.from_numpy(g['resp_gt1'])]
    sizes = [(8, 12), (4, 6), (2, 3)]
    for nb in (0, 2, 3):
        res = O.responsible_indices(sizes, gts, neighbor=nb, shape_match_thres=4.)
        for lvl, (img, anc, gi) in enumerate(res):
            ref = g[f'resp/n{nb}/l{lvl}']
            np.testing.assert_array_equal(torch.stack([img, anc, gi]).numpy(), ref)   # same entries, same order
    empty = O.responsible_indices(sizes, [torch.zeros((0, 4)), torch.zeros((0, 4))])
    assert all(t.numel() == 0 and t.dtype == torch.int64 for lvl in empty for t in lvl)


def test_loss_on_reference_pred_maps(golden):
    g = golden('train_v4')
    preds = [torch.from_numpy(g[f'pred{i}']) for i in range(3)]
    gtb = [torch.from_numpy(g['gt_bboxes0']), torch.from_numpy(g['gt_bboxes1'])]
    gtl = [torch.from_numpy(g['gt_labels0']), torch.from_numpy(g['gt_labels1'])]
    L = O.head_loss(preds, gtb, gtl)
    for k in ('loss_cls', 'loss_conf', 'loss_bbox'):
        got = torch.stack([x.reshape(()) for x in L[k]]).numpy()
        np.testing.assert_allclose(got, g['loss/' + k], rtol=1e-6, atol=1e-7)
    assert float(L['num_gts']) == float(g['loss/num_gts']) == 2.5
    np.testing.assert_allclose(float(O.total_loss(L)), float(g['loss_total']), rtol=1e-6)


def test_train_forward_backward_matches_reference(golden):
    g = golden('train_v4')
    stages, reps, _ = arch_from(g)
    sd = state_dict_from(g)
    params = {}
    for k, v in sd.items():
        if v.is_floating_point() and not ('running_' in k):
            sd[k] = v.clone().requires_grad_(True)
            params[k] = sd[k]
        else:
            sd[k] = v.clone()
    img = torch.from_numpy(g['img'])
    gtb = [torch.from_numpy(g['gt_bboxes0']), torch.from_numpy(g['gt_bboxes1'])]
    gtl = [torch.from_numpy(g['gt_labels0']), torch.from_numpy(g['gt_labels1'])]
    L = O.forward_train(img, sd, stages, reps, [3, 4, 5], gtb, gtl)
    total = O.total_loss(L)
    np.testing.assert_allclose(float(total), float(g['loss_total']), rtol=2e-5)
    total.backward()
    names = [str(n) for n in g['grad_names']]
    assert names == [k for k in params]                         # same parameter set, same order
    sums = g['grad_sums']
    for i, n in enumerate(names):
        gr = params[n].grad.double()
        got = np.array([float(gr.sum()), float(gr.abs().sum()), float(gr.pow(2).sum().sqrt())])
        np.testing.assert_allclose(got[1:], sums[i][1:], rtol=2e-3, atol=1e-6, err_msg=n)
    for k in g.files:
        if k.startswith('grad/'):
            # weights feeding a batch-stat BN get gradients that are differences of large sums:
            # absolute tolerance relative to the tensor's scale
            np.testing.assert_allclose(params[k[5:]].grad.numpy(), g[k], rtol=5e-3,
                                       atol=5e-4 * float(np.abs(g[k]).max()) + 2e-5, err_msg=k)
        if k.startswith('after/'):                              # running statistics after the step (momentum, Q1)
            np.testing.assert_allclose(sd[k[6:]].numpy(), g[k], rtol=1e-5, atol=1e-6, err_msg=k)
