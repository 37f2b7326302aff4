"""Generate tests/golden/*.npz by running the REFERENCE (imported from /root/reference).

Run in the build container only:   python tests/golden/make_golden.py
The fixtures are data (inputs + the reference's outputs); the reference's source never
enters this repository.  See _ref_import.py for how the reference is imported without
mmcv-full, and for the one part that is NOT the reference's arithmetic (batched_nms).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import _ref_import  # noqa: E402
from oracle import build_ref  # noqa: E402


def randomize(module, gen):
    """Non-trivial BN statistics / affine so that folding is exercised."""
    with torch.no_grad():
        for m in module.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.copy_(torch.empty_like(m.weight).uniform_(0.5, 1.5, generator=gen))
                m.bias.copy_(torch.empty_like(m.bias).normal_(0, 0.2, generator=gen))
                m.running_mean.copy_(torch.empty_like(m.running_mean).normal_(0, 0.2, generator=gen))
                m.running_var.copy_(torch.empty_like(m.running_var).uniform_(0.5, 1.5, generator=gen))
            elif isinstance(m, torch.nn.Conv2d):
                fan_in = m.weight[0].numel()
                m.weight.copy_(torch.empty_like(m.weight).normal_(0, (1.5 / fan_in) ** 0.5, generator=gen))


def quantize_fp16(module):
    """Round every float entry to an fp16-representable value so the fixture can store the
    checkpoint losslessly in half the bytes (the reference then RUNS on these fp32 values)."""
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            if t.is_floating_point():
                t.copy_(t.half().float())


def sd_np(prefix, module):
    out = {}
    for k, v in module.state_dict().items():
        a = v.detach().cpu().numpy()
        if a.dtype == np.float32:
            assert np.array_equal(a.astype(np.float16).astype(np.float32), a), k
            a = a.astype(np.float16)
        out[f'sd/{prefix}.{k}'] = a
    return out


def make_mish(ref, out):
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(4096, generator=gen) * 6
    special = torch.tensor([20.0, -20.0, 19.999, 20.001, 88.0, -88.0, 0.0, -0.0, 1e-40, -1e-40, 50.0, -50.0, 9.0, 17.0])
    x[:special.numel()] = special
    g = torch.randn(4096, generator=gen)
    act = ref.mish.Mish()
    xr = x.clone().requires_grad_(True)
    y = act(xr)
    y.backward(g)
    x64 = x.double().clone().requires_grad_(True)
    y64 = act(x64)
    y64.backward(g.double())
    np.savez_compressed(out, x=x.numpy(), g=g.numpy(), y=y.detach().numpy(), gin=xr.grad.numpy(),
                        y64=y64.detach().numpy(), gin64=x64.grad.numpy())
    print('mish', out)


def run_detector(ref, name, scale, backbone_out, neck_type, neck_in, neck_out, csp_rep, img_hw, seed, out,
                 obj_bias, cls_bias, head_std):
    gen = torch.Generator().manual_seed(seed)
    dk, nk, hd = ref.darknetcsp, ref.neck, ref.head
    backbone = dk.DarknetCSP(scale=scale, out_indices=backbone_out)
    Neck = nk.YOLOV4Neck if neck_type == 'v4' else nk.YOLOV5Neck
    neck = Neck(in_channels=neck_in, out_channels=neck_out, csp_repetition=csp_rep)
    test_cfg = ref.ConfigDict(min_bbox_size=0, nms_pre=-1, score_thr=0.001,
                              nms=dict(type='nms', iou_threshold=0.65), max_per_img=300)
    head = hd.YOLOCSPHead(num_classes=80, in_channels=neck_out, train_cfg=None, test_cfg=test_cfg)
    for m in (backbone, neck, head):
        randomize(m, gen)
        torch.nn.Module.eval(m)          # DarknetCSP.train() returns None (Q3)
    N = 2
    img = (torch.randint(0, 256, (N, 3) + img_hw, generator=gen).float() - 114.0) / 255.0
    for m in (backbone, neck):
        quantize_fp16(m)
    # head: logits = bias + std * noise; std is bisected so that about `head_std` (a target
    # candidate count per image) scores pass the 0.001 threshold
    from oracle import yolov4_oracle as O
    base_w = [torch.empty_like(c.weight).normal_(0, 1, generator=gen) for c in head.convs_pred]
    with torch.no_grad():
        nfeat = neck(backbone(img))
        lo, hi = 1e-4, 1.0
        for _ in range(30):
            std = (lo * hi) ** 0.5
            for conv, bw in zip(head.convs_pred, base_w):
                conv.weight.copy_(bw * std)
                b = conv.bias.view(3, 85)
                b.zero_()
                b[:, 4] = obj_bias
                b[:, 5:] = cls_bias
            quantize_fp16(head)
            _, cf, cl = O.decode_maps(head(nfeat)[0], 80)
            cnt = float(((cl * cf[:, :, None]) > 0.001).sum()) / N
            if cnt > head_std:
                hi = std
            else:
                lo = std
    scale_factors = np.array([[1.0, 1.0, 1.0, 1.0], [1.5, 1.25, 1.5, 1.25]], dtype=np.float32)
    img_metas = [dict(scale_factor=scale_factors[i]) for i in range(N)]
    data = {'img': img.numpy(), 'scale_factors': scale_factors}
    with torch.no_grad():
        stage_outs = []
        x = img
        for lname in backbone.layers:
            x = getattr(backbone, lname)(x)
            stage_outs.append(x)
        feats = backbone(img)
        nouts = neck(feats)
        preds = head(nouts)[0]
        dets = head.get_bboxes(preds, img_metas, rescale=True)
        dets_norescale = head.get_bboxes(preds, img_metas, rescale=False)
    for i, s in enumerate(stage_outs):
        data[f'stage{i}'] = s.numpy()
    for i, f in enumerate(feats):
        data[f'feat{i}'] = f.numpy()
    for i, f in enumerate(nouts):
        data[f'neck{i}'] = f.numpy()
    for i, f in enumerate(preds):
        data[f'pred{i}'] = f.numpy()
    for i, (d, l) in enumerate(dets):
        data[f'dets{i}'] = d.numpy()
        data[f'labels{i}'] = l.numpy()
        print(f'  {name} img{i}: {d.shape[0]} dets')
    for i, (d, l) in enumerate(dets_norescale):
        data[f'dets_norescale{i}'] = d.numpy()
        data[f'labels_norescale{i}'] = l.numpy()
    # candidate counts (to know which NMS path the fixture exercises)
    boxes, conf, cls = O.decode_maps(preds, 80)
    data['dec_boxes'] = boxes.numpy()
    data['dec_conf'] = conf.numpy()
    data['dec_cls_first8'] = cls[:, :, :8].numpy()
    cnt = [(int(((cls[n] * conf[n][:, None]) > 0.001).sum())) for n in range(N)]
    data['num_candidates'] = np.array(cnt)
    print(f'  {name} candidates per image: {cnt}')
    data.update(sd_np('backbone', backbone))
    data.update(sd_np('neck', neck))
    data.update(sd_np('bbox_head', head))
    data['meta_stages'] = np.array(scale[0] if not isinstance(scale, str) else [scale])
    data['meta_reps'] = np.array([-1 if r is None else r for r in scale[1]]) if not isinstance(scale, str) else np.array([0])
    data['meta_channels'] = np.array(scale[2]) if not isinstance(scale, str) else np.array([0])
    data['meta_out_indices'] = np.array(backbone_out)
    data['meta_neck_in'] = np.array(neck_in)
    data['meta_neck_out'] = np.array(neck_out)
    data['meta_csp_rep'] = np.array(csp_rep)
    data['state_keys'] = np.array([k[3:] for k in data if k.startswith('sd/')])
    np.savez_compressed(out, **data)
    print(name, out, f'{os.path.getsize(out) / 1e6:.2f} MB')


def make_nms(ref, out):
    """multiclass_nms (reference glue, bbox_nms.py) over synthetic clustered candidates."""
    rng = np.random.RandomState(7)
    data = {}

    def case(tag, K, C, thr, spread, ties):
        centers = rng.rand(40, 2) * 300
        which = rng.randint(0, 40, K)
        cxy = centers[which] + rng.randn(K, 2) * spread
        wh = np.abs(rng.randn(K, 2)) * 20 + 10
        b = np.concatenate([cxy - wh / 2, cxy + wh / 2], 1).astype(np.float32)
        s = (rng.rand(K, C) ** 6).astype(np.float32)
        if ties:
            s[::5] = np.float32(0.25)          # exact-tie scores
            b[1::9] = b[0:-1:9][:b[1::9].shape[0]]  # duplicate boxes -> IoU exactly 1
        sc = np.concatenate([s, np.zeros((K, 1), np.float32)], 1)
        d, l, inds = ref.nms.multiclass_nms(torch.from_numpy(b), torch.from_numpy(sc), thr,
                                            dict(type='nms', iou_threshold=0.65), 300, return_inds=True)
        data[f'{tag}_boxes'], data[f'{tag}_scores'] = b, sc
        data[f'{tag}_thr'] = np.float32(thr)
        data[f'{tag}_dets'], data[f'{tag}_labels'], data[f'{tag}_inds'] = d.numpy(), l.numpy(), inds.numpy()
        n = int((s > thr).sum())
        print(f'  nms case {tag}: {n} candidates -> {d.shape[0]} dets')
    case('small', 600, 5, 0.05, 6.0, True)          # ~3000 candidates > thr? (single-call path)
    case('mid', 1500, 8, 0.3, 8.0, True)
    case('split', 3000, 8, 0.001, 10.0, False)      # >= 10000 candidates -> per-class path
    case('empty', 50, 3, 2.0, 5.0, False)           # nothing passes (Q7 shapes)
    np.savez_compressed(out, **data)
    print('nms', out)



def find_boundary_pairs(thr, rng, want=48):
    """Box pairs whose IoU sits ON the fp32 threshold, found by search (all arithmetic in numpy fp32, in the order of
    oracle/nms_ref.c).  Integer coordinates: areas, intersection and union are exact, so the only roundings are the one
    division (mmcv's CPU kernel: ``inter / union > thr``) or the one product (mmcv's CUDA kernel:
    ``inter > thr * union``).  Returns dict category -> list of (boxA, boxB):
      'eq'       inter / union == thr exactly (division keeps B; the product form may or may not)
      'div_only' division suppresses, product keeps
      'mul_only' product suppresses, division keeps
      'above'    one fp32 ulp above thr in both forms (suppressed), 'below' one ulp below (kept)"""
    thr = np.float32(thr)
    cats = {k: [] for k in ('eq', 'div_only', 'mul_only', 'above', 'below')}
    tries = 0
    while min(len(v) for v in cats.values()) < want and tries < 4000:
        tries += 1
        n = 20000
        w1 = rng.randint(20, 400, n); h1 = rng.randint(20, 400, n)
        w2 = rng.randint(20, 400, n); h2 = rng.randint(20, 400, n)
        dx = rng.randint(0, 60, n); dy = rng.randint(0, 60, n)
        a = np.stack([np.zeros(n), np.zeros(n), w1, h1], 1).astype(np.float32)
        b = np.stack([dx, dy, dx + w2, dy + h2], 1).astype(np.float32)
        iw = np.maximum(np.float32(0), np.minimum(a[:, 2], b[:, 2]) - np.maximum(a[:, 0], b[:, 0]))
        ih = np.maximum(np.float32(0), np.minimum(a[:, 3], b[:, 3]) - np.maximum(a[:, 1], b[:, 1]))
        inter = iw * ih
        uni = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]) + (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]) - inter
        ovr = inter / uni
        rhs = thr * uni
        div, mul = ovr > thr, inter > rhs
        up, dn = np.nextafter(thr, np.float32(1)), np.nextafter(thr, np.float32(0))
        sel = {'eq': ovr == thr, 'div_only': div & ~mul, 'mul_only': mul & ~div,
               'above': (ovr == up) & div & mul, 'below': (ovr == dn) & ~div & ~mul}
        for k, m in sel.items():
            for i in np.nonzero(m)[0]:
                if len(cats[k]) < want:
                    cats[k].append((a[i].copy(), b[i].copy()))
    return cats


def _fp32_predicates(a, b, thr):
    """oracle/nms_ref.c's arithmetic on arrays of boxes (n,4) fp32: (ovr, division form, product form)."""
    f0 = np.float32(0)
    iw = np.maximum(f0, np.minimum(a[:, 2], b[:, 2]) - np.maximum(a[:, 0], b[:, 0]))
    ih = np.maximum(f0, np.minimum(a[:, 3], b[:, 3]) - np.maximum(a[:, 1], b[:, 1]))
    inter = iw * ih
    uni = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]) + (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]) - inter
    ovr = inter / uni
    return ovr, ovr > thr, inter > thr * uni


def find_float_boundary_pairs(thr, rng, cells, want=24):
    """The categories integer coordinates cannot reach ('div_only': the quotient rounds above thr while the product
    rounds to >= inter -- needs |IoU - thr| < 4e-8): for a random pair placed IN its final grid cell, bisect B's x1 to
    the threshold crossing, then walk the 8192 consecutive fp32 values of x1 around it and classify each."""
    thr = np.float32(thr)
    up, dn = np.nextafter(thr, np.float32(1)), np.nextafter(thr, np.float32(0))
    cats = {k: [] for k in ('div_only', 'mul_only', 'eq', 'above', 'below')}
    used = 0                                                     # one pair per cell: a cell is consumed on success only
    for attempt in range(60000):
        if min(len(v) for v in cats.values()) >= want or used >= len(cells):
            break
        ox, oy = cells[used]
        w1, h1 = rng.uniform(40, 200, 2)                         # the pair stays inside its 520 x 520 cell
        ax, ay = ox + rng.uniform(5, 60), oy + rng.uniform(5, 60)
        a = np.array([ax, ay, ax + w1, ay + h1], np.float32)
        w2, h2 = w1 * rng.uniform(0.9, 1.1), h1 * rng.uniform(0.9, 1.1)
        by = np.float32(ay + rng.uniform(-0.05, 0.05) * h1)

        def boxes_for(x1s):
            x1s = x1s.astype(np.float32)
            b = np.stack([x1s, np.full_like(x1s, by), x1s + np.float32(w2), np.full_like(x1s, by + np.float32(h2))], 1)
            return np.repeat(a[None], len(x1s), 0), b.astype(np.float32)
        lo, hi = float(ax), float(ax + 0.6 * w1)                 # IoU decreases as B moves right
        A, B = boxes_for(np.array([lo, hi]))
        o = _fp32_predicates(A, B, thr)[0]
        if not (o[0] > thr > o[1]):
            continue
        for _ in range(60):
            mid = 0.5 * (lo + hi)
            A, B = boxes_for(np.array([mid]))
            if _fp32_predicates(A, B, thr)[0][0] > thr:
                lo = mid
            else:
                hi = mid
        c = np.float32(lo)
        xs = [c]
        for _ in range(4096):
            xs.append(np.nextafter(xs[-1], np.float32(1e9)))
        x = c
        for _ in range(4096):
            x = np.nextafter(x, np.float32(-1e9))
            xs.append(x)
        A, B = boxes_for(np.array(xs, np.float32))
        ovr, div, mul = _fp32_predicates(A, B, thr)
        sel = {'div_only': div & ~mul, 'mul_only': mul & ~div, 'eq': ovr == thr,
               'above': (ovr == up) & div & mul, 'below': (ovr == dn) & ~div & ~mul}
        avail = [k for k, m in sel.items() if m.any() and len(cats[k]) < want]
        if not avail:
            continue
        k = min(avail, key=lambda n: len(cats[n]))               # the category that still needs pairs most
        i0 = np.nonzero(sel[k])[0][0]
        cats[k].append((A[i0].copy(), B[i0].copy()))
        used += 1
    return cats


def make_nms_boundary(ref, out):
    """multiclass_nms on candidates whose pairwise IoU equals the threshold in fp32 or straddles it by one rounding:
    the only inputs on which the definition's ``>`` vs ``>=`` and its division vs product form matter
    (mmdet/core/post_processing/bbox_nms.py:84 -> mmcv nms).  Stored: what the documented definition (division, mmcv's
    CPU kernel) returns AND what the product form (mmcv's CUDA kernel) returns, through the reference's own
    multiclass_nms glue.  A GPU run of the reference (CUDA kernel) may differ from the division form on exactly the
    'div_only' / 'mul_only' pairs below.
    Layout: every pair sits alone in a 520 x 520 cell of a 12-wide grid.  Integer-coordinate pairs ('int_*') cycle
    through the 4 classes (class offsets are integers: all arithmetic stays exact); float-coordinate pairs are class 0
    (offset 0) so that batched_nms' ``boxes + label * (max + 1)`` does not re-round them."""
    from oracle import yolov4_oracle as O
    rng = np.random.RandomState(65)
    thr = 0.65
    C = 4
    icats = find_boundary_pairs(thr, rng)
    ncell_int = sum(len(v) for v in icats.values())
    cells = [((c % 12) * 520.0, (c // 12) * 520.0) for c in range(ncell_int, ncell_int + 160)]
    fcats = find_float_boundary_pairs(thr, rng, cells)
    print('  integer-coordinate pairs:', {k: len(v) for k, v in icats.items()})
    print('  float-coordinate pairs  :', {k: len(v) for k, v in fcats.items()})
    boxes, scores, cat_of, pair_of, names = [], [], [], [], []
    cell = 0

    def add(a, b, cls, cname):
        nonlocal cell
        if cname not in names:
            names.append(cname)
        for j, bx in enumerate((a, b)):
            boxes.append(bx)
            sc = np.zeros(C + 1, np.float32)
            sc[cls] = np.float32(0.9 - 0.001 * (cell % 50)) if j == 0 else np.float32(0.5 - 0.001 * (cell % 50))
            scores.append(sc)
            cat_of.append(names.index(cname))
            pair_of.append(cell)
        cell += 1
    for k, v in icats.items():
        for (a, b) in v:
            ox, oy = (cell % 12) * 520.0, (cell // 12) * 520.0
            sh = np.array([ox, oy, ox, oy], np.float32)
            add(a + sh, b + sh, cell % C, 'int_' + k)
    for k, v in fcats.items():
        for (a, b) in v:
            add(a, b, 0, k)                                   # already placed in their cells by the search
    # one far-away low-score box pins boxes.max() + 1 to 32768 = 2^15: the class offsets label * 32768 are then exact
    # for the integer-coordinate pairs
    boxes.append(np.array([32700, 32700, 32767, 32767], np.float32))
    sc = np.zeros(C + 1, np.float32)
    sc[0] = 0.06
    scores.append(sc)
    names.append('anchor_box')
    cat_of.append(names.index('anchor_box'))
    pair_of.append(cell)
    b = np.stack(boxes).astype(np.float32)
    sc = np.stack(scores).astype(np.float32)
    data = dict(boxes=b, scores=sc, thr=np.float32(0.05), iou_thr=np.float32(thr), category=np.array(cat_of),
                pair=np.array(pair_of), category_names=np.array(names))
    for form, tag in ((0, 'div'), (1, 'mul')):
        O.NMS_IOU_FORM = form
        try:
            d, l, inds = ref.nms.multiclass_nms(torch.from_numpy(b), torch.from_numpy(sc), 0.05,
                                                dict(type='nms', iou_threshold=thr), -1, return_inds=True)
        finally:
            O.NMS_IOU_FORM = 0
        data[f'{tag}_dets'], data[f'{tag}_labels'], data[f'{tag}_inds'] = d.numpy(), l.numpy(), inds.numpy()
        print(f'  nms boundary ({tag}): {b.shape[0]} candidates -> {d.shape[0]} kept')
    np.savez_compressed(out, **data)
    print('nms_boundary', out)


def make_train(ref, out):
    """One training-mode forward + backward of the reference on a tiny v4 detector: loss dict,
    gradients (full for selected tensors, 3 checksums for every parameter), BN running statistics
    after the step, and responsible_indices for hand-placed ground truths."""
    gen = torch.Generator().manual_seed(11)
    dk, nk, hd = ref.darknetcsp, ref.neck, ref.head
    scale = [['conv', 'bottleneck', 'csp', 'csp', 'csp', 'sppv4'], [None, 1, 1, 2, 1, 1], [4, 8, 16, 32, 64, 64]]
    backbone = dk.DarknetCSP(scale=scale, out_indices=[3, 4, 5])
    neck = nk.YOLOV4Neck(in_channels=[32, 64, 64], out_channels=[32, 64, 128], csp_repetition=1)
    head = hd.YOLOCSPHead(num_classes=80, in_channels=[32, 64, 128], train_cfg=None,
                          test_cfg=ref.ConfigDict(nms_pre=-1, score_thr=0.001, nms=dict(type='nms', iou_threshold=0.65),
                                                  max_per_img=300))
    for m in (backbone, neck, head):
        randomize(m, gen)
    with torch.no_grad():
        for conv in head.convs_pred:
            conv.weight.normal_(0, 0.05, generator=gen)
            conv.bias.normal_(-2.0, 0.5, generator=gen)
    for m in (backbone, neck, head):
        quantize_fp16(m)
        torch.nn.Module.train(m, True)
    N = 2
    img = (torch.randint(0, 256, (N, 3, 64, 96), generator=gen).float() - 114.0) / 255.0
    gt_bboxes = [torch.tensor([[8.0, 10.0, 40.0, 44.0], [50.5, 5.0, 95.0, 30.0], [20.0, 30.0, 28.0, 62.0]]),
                 torch.tensor([[0.0, 0.0, 30.0, 20.0], [60.0, 20.0, 90.0, 63.5]])]
    gt_labels = [torch.tensor([3, 17, 79]), torch.tensor([0, 41])]
    metas = [dict() for _ in range(N)]
    data = {'img': img.numpy()}
    sd_before = {}
    for pre, m in (('backbone', backbone), ('neck', neck), ('bbox_head', head)):
        sd_before.update(sd_np(pre, m))
    feats = backbone(img)
    nouts = neck(feats)
    preds = head(nouts)[0]
    losses = head.loss(preds, gt_bboxes, gt_labels, metas)
    total = sum(sum(x.mean() for x in v) for k, v in losses.items() if 'loss' in k)
    total.backward()
    for k, v in losses.items():
        data['loss/' + k] = (torch.stack([x.reshape(()) for x in v]).detach().numpy() if isinstance(v, list)
                             else v.detach().numpy())
    data['loss_total'] = total.detach().numpy()
    for i, f in enumerate(preds):
        data[f'pred{i}'] = f.detach().numpy()
    names, sums = [], []
    keep_full = ('backbone.conv0.conv.weight', 'backbone.conv0.bn.weight', 'backbone.conv0.bn.bias',
                 'backbone.csp3.conv_csp.bn.weight', 'backbone.csp3.conv_csp.conv3.weight',
                 'backbone.csp2.conv_downscale.conv.weight', 'backbone.sppv45.spp.conv5.conv.weight',
                 'neck.downsample_convs.0.conv.weight', 'neck.out_convs.1.bn.bias',
                 'bbox_head.convs_pred.0.weight', 'bbox_head.convs_pred.2.bias')
    for pre, m in (('backbone', backbone), ('neck', neck), ('bbox_head', head)):
        for n, prm in m.named_parameters():
            full = f'{pre}.{n}'
            g = prm.grad
            assert g is not None, full
            names.append(full)
            sums.append([float(g.double().sum()), float(g.double().abs().sum()), float(g.double().pow(2).sum().sqrt())])
            if full in keep_full:
                data['grad/' + full] = g.numpy()
        for n, b in m.named_buffers():
            if n.endswith('running_mean') or n.endswith('running_var'):
                full = f'{pre}.{n}'
                if full.split('.running')[0] in ('backbone.conv0.bn', 'backbone.sppv45.spp.conv1.bn',
                                                 'backbone.csp3.conv_csp.bn', 'neck.out_convs.2.bn'):
                    data['after/' + full] = b.detach().numpy()
    data['grad_names'] = np.array(names)
    data['grad_sums'] = np.array(sums)
    data.update(sd_before)
    for i, (b, l) in enumerate(zip(gt_bboxes, gt_labels)):
        data[f'gt_bboxes{i}'] = b.numpy()
        data[f'gt_labels{i}'] = l.numpy()
    data['meta_stages'] = np.array(scale[0])
    data['meta_reps'] = np.array([-1 if r is None else r for r in scale[1]])
    data['meta_channels'] = np.array(scale[2])
    # responsible_indices on hand-placed gts: cell centres, cell edges, image borders, extreme aspect
    ag = head.anchor_generator
    gts = [torch.tensor([[12.0, 12.0, 20.0, 20.0], [0.0, 0.0, 7.9, 8.1], [88.0, 56.0, 96.0, 64.0], [30.0, 2.0, 34.0, 62.0],
                         [15.9, 16.0, 48.1, 47.9]]),
           torch.tensor([[40.0, 24.0, 56.0, 40.0], [1.0, 30.0, 95.0, 34.0]])]
    sizes = [(8, 12), (4, 6), (2, 3)]
    for nb in (0, 2, 3):
        res = ag.responsible_indices(sizes, gts, neighbor=nb, shape_match_thres=4., device='cpu')
        for lvl, (a, b, c) in enumerate(res):
            data[f'resp/n{nb}/l{lvl}'] = torch.stack([a, b, c]).numpy()
    for i, gbox in enumerate(gts):
        data[f'resp_gt{i}'] = gbox.numpy()
    np.savez_compressed(out, **data)
    print('train', out, f'{os.path.getsize(out) / 1e6:.2f} MB', 'loss', float(total))


def make_post(ref, out):
    """get_bboxes variants the YOLOv4 configs leave at their defaults but the head supports
    (yolocsp_head.py:349-360): nms_pre top-k pre-selection and the class-agnostic head, on
    seeded random pred maps (N=2, featmaps 8x12 / 4x6 / 2x3)."""
    hd = ref.head
    gen = torch.Generator().manual_seed(21)
    sizes = [(8, 12), (4, 6), (2, 3)]
    data = {}
    metas = [dict(scale_factor=np.array([1.25, 1.5, 1.25, 1.5], dtype=np.float32)),
             dict(scale_factor=np.array([0.75, 0.75, 0.75, 0.75], dtype=np.float32))]
    data['scale_factors'] = np.stack([m['scale_factor'] for m in metas])
    for tag, agnostic, ncls in (('aware', False, 6), ('agnostic', True, 6)):
        attr = 5 if agnostic else 5 + ncls
        preds = []
        for (h, w) in sizes:
            p = torch.randn(2, 3 * attr, h, w, generator=gen) * 1.5
            p.view(2, 3, attr, h, w)[:, :, 4] += 1.0           # objectness well above the threshold
            preds.append(p)
            data[f'{tag}/pred{len(preds) - 1}'] = p.numpy()
        for nms_pre in (-1, 60, 250):
            head = hd.YOLOCSPHead(num_classes=ncls, in_channels=[8, 8, 8], class_agnostic=agnostic, train_cfg=None,
                                  test_cfg=ref.ConfigDict(nms_pre=nms_pre, score_thr=0.05,
                                                          nms=dict(type='nms', iou_threshold=0.5), max_per_img=50))
            res = head.get_bboxes([p.clone() for p in preds], metas, rescale=True)
            for i, (d, l) in enumerate(res):
                data[f'{tag}/pre{nms_pre}/dets{i}'] = d.numpy()
                data[f'{tag}/pre{nms_pre}/labels{i}'] = l.numpy()
    data['num_classes'] = np.array(6)
    np.savez_compressed(out, **data)
    print('post', out, f'{os.path.getsize(out) / 1e3:.1f} kB',
          {k: v.shape for k, v in data.items() if k.endswith('dets0')})


def make_softfocal(ref, out):
    """SoftFocalLoss (yolocsp_head.py:21-50) around the reference's sigmoid CrossEntropyLoss: soft targets, the three
    reductions, two (gamma, alpha) settings; outputs and input gradients."""
    gen = torch.Generator().manual_seed(11)
    pred = torch.randn(48, 6, generator=gen) * 3
    gt = torch.rand(48, 6, generator=gen)
    gt[::5] = 0.
    gt[1::7] = 1.
    res = dict(pred=pred.numpy(), gt=gt.numpy())
    for tag, (gamma, alpha, red, weight) in dict(a=(1.5, 0.25, 'mean', 1.0), b=(2.0, 0.5, 'sum', 64.0),
                                                  c=(1.5, 0.25, 'none', 1.0)).items():
        crit = ref.head.SoftFocalLoss(ref.ConfigDict(type='CrossEntropyLoss', use_sigmoid=True, reduction=red,
                                                     loss_weight=weight), gamma=gamma, alpha=alpha)
        x = pred.clone().requires_grad_(True)
        y = crit(x, gt)
        y.sum().backward()
        res[f'{tag}/out'] = y.detach().numpy()
        res[f'{tag}/grad'] = x.grad.numpy()
        res[f'{tag}/cfg'] = np.array([gamma, alpha, weight])
        res[f'{tag}/reduction'] = np.array(red)
    np.savez_compressed(out, **res)
    print('softfocal', out)


def main():
    if not _ref_import.available():
        print('reference not present: nothing to do')
        return
    ext = build_ref.load_ext()
    ref = _ref_import.install_shim(ext)
    torch.manual_seed(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'post':      # only the newest fixture
        make_post(ref, os.path.join(HERE, 'post_variants.npz'))
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'nms_boundary':
        make_nms_boundary(ref, os.path.join(HERE, 'nms_boundary.npz'))
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'softfocal':
        make_softfocal(ref, os.path.join(HERE, 'softfocal.npz'))
        return
    make_mish(ref, os.path.join(HERE, 'mish.npz'))
    v4 = [['conv', 'bottleneck', 'csp', 'csp', 'csp', 'sppv4'], [None, 1, 1, 2, 2, 1], [4, 8, 16, 32, 64, 64]]
    run_detector(ref, 'tiny_v4', v4, [3, 4, 5], 'v4', [32, 64, 64], [32, 64, 128], 2, (64, 96), 3,
                 os.path.join(HERE, 'tiny_v4.npz'), obj_bias=-4.0, cls_bias=-4.5, head_std=2500)
    v5 = [['focus', 'csp', 'csp', 'csp', 'sppv5'], [None, 1, 2, 2, 1], [8, 16, 32, 64, 128]]
    run_detector(ref, 'tiny_v5', v5, [2, 3, 4], 'v5', [32, 64, 128], [32, 64, 128], 1, (64, 64), 5,
                 os.path.join(HERE, 'tiny_v5.npz'), obj_bias=-3.5, cls_bias=-4.0, head_std=12000)
    make_nms(ref, os.path.join(HERE, 'nms.npz'))
    make_nms_boundary(ref, os.path.join(HERE, 'nms_boundary.npz'))
    make_train(ref, os.path.join(HERE, 'train_v4.npz'))
    make_post(ref, os.path.join(HERE, 'post_variants.npz'))
    make_softfocal(ref, os.path.join(HERE, 'softfocal.npz'))


if __name__ == '__main__':
    main()
