"""The train-side transform oracle (oracle/augment_oracle.py) against what the REFERENCE's own classes produced
(tests/golden/augment.npz, tests/golden/make_golden_augment.py): MosaicPipeline's stitch / box shift / labels and
GtBBoxesFilter, bit for bit.  The OpenCV / albumentations steps of the chain have no reference output to compare with
(third party, absent: parity unpinned); what is checked for them here are exact identities of the restatement."""
import numpy as np
import pytest

from oracle import augment_oracle as A


@pytest.mark.parametrize('tag', ['square', 'ragged', 'small'])
def test_mosaic_stitch_equals_reference(golden, tag):
    g = golden('augment')
    ims = [g[f'{tag}/img{i}'] for i in range(4)]
    bxs = [g[f'{tag}/boxes{i}'] for i in range(4)]
    lbs = [g[f'{tag}/labels{i}'] for i in range(4)]
    canvas, boxes, labels, cxy = A.mosaic(ims, bxs, lbs, pad_val=114)
    assert canvas.shape == tuple(g[f'{tag}/img_shape']) == (2 * cxy, 2 * cxy, 3)
    np.testing.assert_array_equal(canvas, g[f'{tag}/canvas'])
    np.testing.assert_array_equal(boxes, g[f'{tag}/out_boxes'])
    np.testing.assert_array_equal(labels, g[f'{tag}/out_labels'])
    assert boxes.dtype == np.float32 and labels.dtype == np.int64


def test_gt_bboxes_filter_equals_reference(golden):
    g = golden('augment')
    b, l = A.gt_bboxes_filter(g['filter/boxes'], g['filter/labels'])
    np.testing.assert_array_equal(b, g['filter/out_boxes'])
    np.testing.assert_array_equal(l, g['filter/out_labels'])
    kept = set(l.tolist())
    assert 0 not in kept and 1 in kept          # w == min_size is dropped (strict >), w = 2.0001 kept
    assert 2 not in kept and 4 in kept          # aspect ratio == 20 dropped (strict <), 19.99.. kept
    assert 5 not in kept                        # zero-size box


def test_hsv_restatement_identities():
    """Gains of 1 give identity LUTs, and OpenCV's 8-bit BGR -> HSV -> BGR round trip is exact on greys and on
    saturated primaries; the LUT lines are the reference's (transforms.py:2004-2008)."""
    lh, ls, lv = A.hsv_luts(np.array([1.0, 1.0, 1.0]))
    assert (lh[:180] == np.arange(180)).all() and (lh[180:] == np.arange(76)).all()      # x % 180
    assert (ls == np.arange(256)).all() and (lv == np.arange(256)).all()
    lh, ls, lv = A.hsv_luts(np.array([1.015, 1.7, 0.6]))
    assert ls.max() == 255 and lv.max() == int(255 * 0.6) and lh[100] == int((100 * 1.015) % 180)
    grey = np.repeat(np.arange(256, dtype=np.uint8)[:, None, None], 3, axis=2)
    hsv = A.bgr2hsv_u8(grey)
    assert (hsv[..., 1] == 0).all() and (hsv[..., 2] == grey[..., 0]).all()
    np.testing.assert_array_equal(A.hsv2bgr_u8(hsv), grey)
    prim = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 0], [0, 255, 255], [255, 0, 255]]], np.uint8)
    hsv = A.bgr2hsv_u8(prim)
    assert hsv[0, :, 0].tolist() == [120, 60, 0, 90, 30, 150]                         # OpenCV hue / 2 of B, G, R, C, Y, M
    np.testing.assert_array_equal(A.hsv2bgr_u8(hsv), prim)
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, (32, 32, 3)).astype(np.uint8)
    back = A.hsv2bgr_u8(A.bgr2hsv_u8(img)).astype(int)
    assert np.abs(back - img.astype(int)).max() <= 4                                  # 8-bit HSV quantisation


def test_geometric_chain_identities():
    """scale 1 / no flip is a pure crop of the padded canvas; flip mirrors it; boxes follow the pixels."""
    rng = np.random.RandomState(1)
    canvas = rng.randint(0, 256, (160, 160, 3)).astype(np.uint8)
    p = dict(h_start=0.25, w_start=0.75, scale=1.0, flip=False, pad_to=240, crop=200, out=100)
    out = A.geometric(canvas, p)
    top, left, H, W = A.pad_if_needed_offsets(160, 160, 240, 240)
    y1, x1 = A.random_crop_origin(H, W, 200, 200, 0.25, 0.75)
    big = np.full((240, 240, 3), 114, np.uint8)
    big[top:top + 160, left:left + 160] = canvas
    np.testing.assert_array_equal(out, big[y1 + 50:y1 + 150, x1 + 50:x1 + 150])
    np.testing.assert_array_equal(A.geometric(canvas, dict(p, flip=True)), out[:, ::-1])
    boxes = np.array([[60, 60, 100, 90], [0, 0, 5, 5], [150, 150, 160, 160]], np.float32)
    b, l = A.geometric_boxes(boxes, np.arange(3), (160, 160), p)
    sx, sy = left - x1 - 50, top - y1 - 50
    want = np.clip(boxes[0] + [sx, sy, sx, sy], 0, 100)
    np.testing.assert_allclose(b[l == 0][0], want)
    bf, lf = A.geometric_boxes(boxes, np.arange(3), (160, 160), dict(p, flip=True))
    np.testing.assert_allclose(bf[lf == 0][0], [100 - want[2], want[1], 100 - want[0], want[3]])
    small = A.geometric(canvas, dict(p, scale=0.5))                  # 200 -> 100: CenterCrop offset 0
    assert small.shape == (100, 100, 3)
