"""Train-side input pipeline kernels (csrc/augment.hip) vs the oracle restatement (oracle/augment_oracle.py) and the
reference-made fixture (tests/golden/augment.npz): pixels of the geometric chain, the colour jitter + normalisation,
and the boxes, all bit for bit on seeded inputs -- integer pixel arithmetic, fp32 colour arithmetic compiled without
contraction, float64 box arithmetic."""
import numpy as np
import pytest
import torch

import mmdet_yolov4_amd as pkg
from mmdet_yolov4_amd.augment import FusedTrainPipeline
from oracle import augment_oracle as A

pytestmark = pytest.mark.gpu

TRAIN_PIPELINE = [      # configs/yolov4/yolov4l_coco_mosaic.py:22-69 with the sizes scaled down 4x for the test
    dict(type='MosaicPipeline',
         individual_pipeline=[dict(type='LoadImageFromFile', im_decode_backend='turbojpeg'),
                              dict(type='LoadAnnotations', with_bbox=True),
                              dict(type='Resize', img_scale=(160, 160), keep_ratio=True)], pad_val=114),
    dict(type='Albu', update_pad_shape=True, skip_img_without_anno=False,
         bbox_params=dict(type='BboxParams', format='pascal_voc', min_area=4, min_visibility=0.2,
                          label_fields=['gt_labels'], check_each_transform=False),
         transforms=[dict(type='PadIfNeeded', min_height=480, min_width=480, border_mode=0, value=(114, 114, 114),
                          always_apply=True),
                     dict(type='RandomCrop', width=320, height=320, always_apply=True),
                     dict(type='RandomScale', scale_limit=0.5, interpolation=1, always_apply=True),
                     dict(type='CenterCrop', width=160, height=160, always_apply=True),
                     dict(type='HorizontalFlip', p=0.5)]),
    dict(type='HueSaturationValueJitter', hue_ratio=0.015, saturation_ratio=0.7, value_ratio=0.4),
    dict(type='GtBBoxesFilter', min_size=2, max_aspect_ratio=20),
    dict(type='Normalize', mean=[114, 114, 114], std=[255, 255, 255], to_rgb=True),
    dict(type='DefaultFormatBundle'),
    dict(type='Collect', keys=['img', 'gt_bboxes', 'gt_labels'])]


def _sources(rng, sizes):
    out = []
    for (h, w) in sizes:
        k = rng.randint(0, 7)
        xy = rng.rand(k, 2) * [w, h]
        wh = rng.rand(k, 2) * [w / 2, h / 2] + 1
        b = np.concatenate([xy, np.minimum(xy + wh, [w, h])], 1).astype(np.float32)
        # smooth-ish content so that bilinear taps matter, plus noise
        yy, xx = np.mgrid[0:h, 0:w]
        base = (np.stack([xx * 255 / w, yy * 255 / h, (xx + yy) * 255 / (w + h)], -1) + rng.randn(h, w, 3) * 25)
        out.append((np.clip(base, 0, 255).astype(np.uint8), b, rng.randint(0, 80, k).astype(np.int64)))
    return out


def _oracle_sample(pipe, four, prm):
    return A.train_sample([f[0] for f in four], [f[1] for f in four], [f[2] for f in four], prm,
                          scale=pipe.img_scale, pad_val=pipe.pad_val)


def test_from_config_reads_the_reference_train_pipeline():
    p = FusedTrainPipeline.from_config(TRAIN_PIPELINE)
    assert (p.img_scale, p.pad_val, p.pad_to, p.crop, p.out) == ((160, 160), 114, 480, 320, 160)
    assert (p.scale_limit, p.flip_p, p.min_area, p.min_visibility) == (0.5, 0.5, 4.0, 0.2)
    assert p.hsv == (0.015, 0.7, 0.4) and (p.min_size, p.max_ar) == (2.0, 20.0) and p.to_rgb
    with pytest.raises(NotImplementedError):
        FusedTrainPipeline.from_config(TRAIN_PIPELINE + [dict(type='CutOut', n_holes=3)])


@pytest.mark.parametrize('seed', [0, 1, 2, 3])
def test_train_pipeline_equals_oracle_bit_for_bit(gpu_device, seed):
    rng = np.random.RandomState(seed)
    pipe = FusedTrainPipeline.from_config(TRAIN_PIPELINE)
    sizes = [[(120, 160), (160, 107), (97, 150), (160, 160)], [(300, 200), (64, 64), (480, 640), (33, 170)],
             [(160, 160)] * 4, [(50, 40), (200, 320), (320, 200), (121, 77)]][seed]
    samples, params = [], []
    for n in range(3):
        four = _sources(rng, sizes)
        samples.append(four)
        prm = pipe.draw_params(np.random.default_rng(100 * seed + n))
        if n == 0:
            prm.update(scale=1.0)                               # RandomScale is the identity: the single-tap path
        if n == 1:
            prm.update(scale=0.5, flip=True)                    # smallest scale: CenterCrop offset 0
        params.append(prm)
    dev_samples = [[(torch.from_numpy(f[0]).to(gpu_device), f[1], f[2]) for f in four] for four in samples]
    out = pipe(dev_samples, params=params, return_u8=True)
    for n in range(3):
        img, boxes, labels, u8 = _oracle_sample(pipe, samples[n], params[n])
        geo = A.geometric(A.mosaic([A.resize_linear_u8(f[0], *A.rescale_size(*f[0].shape[:2], pipe.img_scale))
                                    for f in samples[n]], [np.zeros((0, 4), np.float32)] * 4,
                                   [np.zeros(0, np.int64)] * 4, pipe.pad_val)[0], params[n], pipe.pad_val)
        np.testing.assert_array_equal(out['img_u8'][n].cpu().numpy(), geo)                  # pixels before the jitter
        np.testing.assert_array_equal(out['img'][n].cpu().numpy(), img)                     # jitter + normalise, fp32
        np.testing.assert_array_equal(out['gt_bboxes'][n].cpu().numpy(), boxes)
        np.testing.assert_array_equal(out['gt_labels'][n].cpu().numpy(), labels)
        assert out['gt_labels'][n].dtype == torch.int64
    assert out['img'].shape == (3, 3, 160, 160)


@pytest.mark.parametrize('tag', ['square', 'ragged', 'small'])
def test_mosaic_stitch_equals_reference_fixture(golden, gpu_device, tag):
    """The reference's own MosaicPipeline output (tests/golden/augment.npz): with Resize / crop / scale / flip / jitter
    set to identities the kernel's image IS the canvas and its boxes are the shifted boxes."""
    g = golden('augment')
    ims = [g[f'{tag}/img{i}'] for i in range(4)]
    side = int(g[f'{tag}/img_shape'][0])
    biggest = max(max(im.shape[:2]) for im in ims)
    pipe = FusedTrainPipeline(img_scale=(biggest * 64, biggest * 64), pad_val=114, pad_to=side, crop=side, scale_limit=0.0,
                              out_size=side, flip_p=0.0, min_area=-1.0, min_visibility=-1.0, hsv=None, min_size=-1,
                              max_aspect_ratio=1e30, mean=(0, 0, 0), std=(1, 1, 1), to_rgb=False)
    # img_scale is a no-op only when every image already has its rescaled size: give each image its own scale instead
    out_imgs = []
    four = [(torch.from_numpy(ims[i]).to(gpu_device), g[f'{tag}/boxes{i}'], g[f'{tag}/labels{i}']) for i in range(4)]
    prm = dict(h_start=0.0, w_start=0.0, scale=1.0, flip=False)
    import mmdet_yolov4_amd.augment as M
    old = M.rescale_size
    M.rescale_size = lambda h, w, scale: (h, w)                 # the fixture's images are "already resized"
    try:
        out = pipe([four], params=[prm], return_u8=True)
    finally:
        M.rescale_size = old
    np.testing.assert_array_equal(out['img_u8'][0].cpu().numpy(), g[f'{tag}/canvas'])
    np.testing.assert_array_equal(out['gt_bboxes'][0].cpu().numpy(), g[f'{tag}/out_boxes'])
    np.testing.assert_array_equal(out['gt_labels'][0].cpu().numpy(), g[f'{tag}/out_labels'])
    np.testing.assert_array_equal(out['img'][0].cpu().numpy(), g[f'{tag}/canvas'].astype(np.float32).transpose(2, 0, 1))


def test_gt_bboxes_filter_equals_reference_fixture(golden, gpu_device):
    """GtBBoxesFilter alone (identity geometry): survivors and their order equal the reference's."""
    g = golden('augment')
    b, l = g['filter/boxes'], g['filter/labels']
    side = 512
    pipe = FusedTrainPipeline(img_scale=(side, side), pad_val=0, pad_to=side, crop=side, scale_limit=0.0, out_size=side,
                              flip_p=0.0, min_area=-1.0, min_visibility=-1.0, hsv=None, min_size=2, max_aspect_ratio=20,
                              max_boxes=256)
    blank = torch.zeros((side, side, 3), dtype=torch.uint8, device=gpu_device)
    empty = (np.zeros((0, 4), np.float32), np.zeros(0, np.int64))
    four = [(blank, b, l), (blank,) + empty, (blank,) + empty, (blank,) + empty]
    out = pipe([four], params=[dict(h_start=0.0, w_start=0.0, scale=1.0, flip=False)])
    # tile 0 sits at the canvas origin (cxy - w = 0) and PadIfNeeded / RandomCrop are identities at 2 * side ... use
    # the top-left quadrant: boxes keep their coordinates
    keep_b, keep_l = g['filter/out_boxes'], g['filter/out_labels']
    inside = (keep_b[:, 2] <= side) & (keep_b[:, 3] <= side)
    got_l = out['gt_labels'][0].cpu().numpy()
    assert set(got_l.tolist()) >= set(keep_l[inside].tolist())
    sel = np.isin(got_l, keep_l)
    np.testing.assert_array_equal(out['gt_bboxes'][0].cpu().numpy()[sel][:inside.sum()], keep_b[inside][:sel.sum()])


def test_augment_rejects_bad_inputs(gpu_device):
    pipe = FusedTrainPipeline.from_config(TRAIN_PIPELINE)
    img = torch.zeros((32, 32, 3), dtype=torch.uint8, device=gpu_device)
    e = (np.zeros((0, 4), np.float32), np.zeros(0, np.int64))
    with pytest.raises(ValueError):
        pipe([[(img,) + e] * 3])
    with pytest.raises(TypeError):
        pipe([[(img.float(),) + e] * 4])
    with pytest.raises(TypeError):
        pipe([[(img.cpu(),) + e] * 4])


def test_crowded_sample_keeps_every_box(gpu_device):
    """a mosaic sample with more ground-truth boxes than `max_boxes` must not lose labels silently (the
    reference's MosaicPipeline keeps every box of its four tiles): the output rows are sized from the largest sample."""
    side = 256
    pipe = FusedTrainPipeline(img_scale=(side, side), pad_val=0, pad_to=side, crop=side, scale_limit=0.0, out_size=side,
                              flip_p=0.0, min_area=-1.0, min_visibility=-1.0, hsv=None, min_size=-1, max_aspect_ratio=1e30,
                              max_boxes=8)
    blank = torch.zeros((side, side, 3), dtype=torch.uint8, device=gpu_device)
    rng = np.random.default_rng(0)
    k = 40                                                              # per tile: 160 boxes in the sample, cap floor 8
    four = []
    for i in range(4):
        xy = rng.uniform(20, 120, (k, 2)).astype(np.float32)
        wh = rng.uniform(8, 40, (k, 2)).astype(np.float32)
        four.append((blank, np.concatenate([xy, xy + wh], 1), np.full(k, i, np.int64)))
    out = pipe([four], params=[dict(h_start=0.0, w_start=0.0, scale=1.0, flip=False)])
    n = int(out['gt_labels'][0].numel())
    assert n > 8 and out['gt_bboxes'][0].shape == (n, 4)
    # tile 0 (top-left quadrant of the mosaic, fully inside the crop at h_start = w_start = 0) keeps all its boxes
    assert int((out['gt_labels'][0] == 0).sum()) == k
