"""CPU tests of the oracle (no GPU): hand-derivable known answers, an independent NumPy restatement,
and the committed golden vectors.

The reference ships no fixture for the hot path and cannot run here (TensorFlow absent), so the
oracle is pinned by (1) known answers derived by hand from the reference's formulas, (2) a second
restatement written independently in NumPy (tests/np_restatement.py) that must agree bit for bit,
(3) regression vectors in tests/golden/.  Against real TensorFlow the path stays "parity unpinned".
"""

import os

import numpy as np
import pytest

import oracle
import synth
from tests import np_restatement as npr

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
f32 = np.float32


def bits_equal(a, b):
    a, b = np.asarray(a, f32), np.asarray(b, f32)
    return a.shape == b.shape and bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | (a == b) | (np.isnan(a) & np.isnan(b))))


# ---- hand-derivable known answers ------------------------------------------------------
def test_base_anchor_known_answers():
    hp = synth.hyper_params()
    base = oracle.generate_base_anchors(hp)
    assert base.shape == (9, 4)
    np.testing.assert_array_equal(base[0], f32([-0.128, -0.128, 0.128, 0.128]))  # scale 128 / ratio 1
    w = np.sqrt(f32(0.256 ** 2 / 2.0))  # scale 128 / ratio 2 (tall): w = sqrt(0.032768), h = 2 w
    np.testing.assert_array_equal(base[1], f32([-w, -w / 2, w, w / 2]))
    assert abs(float(w) - 0.181019) < 1e-6


@pytest.mark.parametrize("backbone,count", [("mobilenet_v2", 9216), ("vgg16", 8649)])
def test_anchor_counts_and_rows(backbone, count):
    hp = synth.hyper_params(backbone)
    a = oracle.generate_anchors(hp)
    assert a.shape == (count, 4)
    assert a.min() >= 0.0 and a.max() <= 1.0
    G = hp["feature_map_shape"]
    c0 = f32(0.0 / G + (1 / G) / 2)
    np.testing.assert_array_equal(a[0], np.clip(f32([-0.128, -0.128, 0.128, 0.128]) + c0, 0, 1))
    cl = f32((G - 1) / G + (1 / G) / 2)
    last = oracle.generate_base_anchors(hp)[8] + cl
    np.testing.assert_array_equal(a[-1], np.clip(last, 0, 1).astype(f32))
    # cell-major, anchor-minor ordering: anchor 9 is the first template of cell (gy=0, gx=1)
    c1 = f32(1.0 / G + (1 / G) / 2)
    np.testing.assert_array_equal(a[9], np.clip(f32([-0.128, -0.128, 0.128, 0.128]) + f32([c0, c1, c0, c1]), 0, 1))


def test_exp_log_known_answers():
    assert oracle.expf(0.0) == 1.0  # exp(0) == 1 exactly: zero deltas keep the anchor size
    assert abs(oracle.expf(1.0) - np.e) < 3e-7
    assert abs(oracle.logf(1.0)) == 0.0
    assert oracle.logf(0.0) == -np.inf and np.isnan(oracle.logf(-1.0))
    xs = np.random.default_rng(0).normal(0, 3, 4000).astype(f32)
    ref = np.exp(xs.astype(np.float64))
    assert np.max(np.abs(oracle.exp_array(xs) - ref) / ref) < 2e-7  # Cephes expf: < 1 ulp-ish


def test_decode_zero_deltas_and_known_box():
    a = f32([[0.25, 0.25, 0.75, 0.5]])
    np.testing.assert_array_equal(oracle.get_bboxes_from_deltas(a, np.zeros((1, 1, 4), f32))[0], a)
    # dy = 1 moves the centre by one height; dh = ln 2 doubles the height (to fp32 accuracy)
    out = oracle.get_bboxes_from_deltas(f32([[0.0, 0.0, 1.0, 1.0]]), f32([[[1.0, 0.0, np.log(2.0), 0.0]]]))[0, 0]
    np.testing.assert_allclose(out, [0.5, 0.0, 2.5, 1.0], rtol=1e-6)


def test_encode_known_answers():
    bb = f32([[0.0, 0.0, 1.0, 1.0]])
    gt = f32([[[0.5, 0.5, 1.5, 1.5]]])
    np.testing.assert_array_equal(oracle.get_deltas_from_bboxes(bb, gt)[0, 0], f32([0.5, 0.5, 0.0, 0.0]))
    # zero-size gt -> zero deltas; zero-size bbox dims -> 1e-3
    np.testing.assert_array_equal(oracle.get_deltas_from_bboxes(bb, np.zeros((1, 1, 4), f32))[0, 0], np.zeros(4, f32))
    out = oracle.get_deltas_from_bboxes(f32([[0.5, 0.5, 0.5, 0.5]]), f32([[[0.4, 0.4, 0.6, 0.6]]]))[0, 0]
    assert out[0] == 0.0 and abs(out[2] - np.log(0.2 / 1e-3)) < 1e-5
    # variances divide the result (utils/train_utils.py:200)
    np.testing.assert_array_equal(oracle.get_deltas_from_bboxes(bb, gt, [0.1, 0.1, 0.2, 0.2])[0, 0],
                                  f32([0.5, 0.5, 0, 0]) / f32([0.1, 0.1, 0.2, 0.2]))


def test_iou_map_known_answers():
    bb = f32([[0, 0, 1, 1]])
    gt = f32([[[0, 0, 1, 1], [0, 0.5, 1, 1.5], [2, 2, 3, 3], [0, 0, 0, 0]]])
    iou = oracle.generate_iou_map(bb, gt)[0, 0]
    assert iou[0] == 1.0 and iou[1] == f32(0.5) / f32(1.5) and iou[2] == 0.0 and iou[3] == 0.0  # identical, 1/3, disjoint, pad
    # no epsilon: zero-area box vs zero-area gt is 0/0 = NaN (unlike the NMS-internal IoU)
    assert np.isnan(oracle.generate_iou_map(np.zeros((1, 4), f32), np.zeros((1, 1, 4), f32))[0, 0, 0])


def test_top_k_ties_lower_index_first():
    s = f32([[0.5, 0.9, 0.5, 0.9, 0.1]])
    vals, idx = oracle.top_k(s, 4)
    np.testing.assert_array_equal(idx[0], [1, 3, 0, 2])
    np.testing.assert_array_equal(vals[0], f32([0.9, 0.9, 0.5, 0.5]))
    with pytest.raises(ValueError):
        oracle.top_k(s, 6)


def test_nms_known_answers():
    boxes = f32([[[[0, 0, 1, 1]], [[0, 0, 1, 0.72]], [[2, 2, 3, 3]]]])  # IoU(b0,b1) = 0.72 > 0.7; b2 disjoint
    scores = f32([[[0.9], [0.8], [0.7]]])
    b, s, c, v = oracle.combined_non_max_suppression(boxes, scores, 3, 3, 0.7)
    assert v[0] == 2
    np.testing.assert_array_equal(s[0], f32([0.9, 0.7, 0.0]))
    np.testing.assert_array_equal(b[0, 1], f32([1, 1, 1, 1]))  # output clipped to [0,1]
    np.testing.assert_array_equal(b[0, 2], np.zeros(4, f32))   # zero padding
    # IoU exactly equal to the threshold does NOT suppress (strict >)
    boxes = f32([[[[0, 0, 1, 1]], [[0, 0, 1, 0.5]]]])
    _, _, _, v = oracle.combined_non_max_suppression(boxes, f32([[[0.9], [0.8]]]), 2, 2, 0.5)
    assert v[0] == 2
    # equal scores pop in ascending index (this build's tie rule; TF leaves it unspecified)
    boxes = f32([[[[0, 0, .5, .5]], [[.5, .5, 1, 1]]]])
    b, _, _, _ = oracle.combined_non_max_suppression(boxes, f32([[[0.6], [0.6]]]), 2, 2)
    np.testing.assert_array_equal(b[0], boxes[0, :, 0])
    # score threshold is strict; classes are emitted as fp32 indices; cross-class order by score
    boxes = f32([[[[0, 0, .4, .4]], [[.5, .5, 1, 1]]]])
    scores = f32([[[0.5, 0.8], [0.9, 0.5]]])
    b, s, c, v = oracle.combined_non_max_suppression(boxes, scores, 2, 4, 0.5, 0.5)
    assert v[0] == 2
    np.testing.assert_array_equal(s[0, :2], f32([0.9, 0.8]))
    np.testing.assert_array_equal(c[0, :2], f32([0.0, 1.0]))


def test_crop_and_resize_known_answers():
    ramp = f32([[[[0.0], [1.0]], [[0.0], [1.0]]]])  # 2x2 map, box (0,0,1,1) -> linear ramp i/6 along x
    out = oracle.crop_and_resize(ramp, f32([[0, 0, 1, 1]]), [0], (7, 7))
    np.testing.assert_allclose(out[0, 3, :, 0], np.arange(7) / 6.0, rtol=1e-6)
    fm = synth.feature_map(0, 2, 4, 3)
    # a zero-padded RoI (0,0,0,0) samples pixel (0,0) for all 49 cells
    out = oracle.roi_pooling(fm, np.zeros((2, 1, 4), f32), (7, 7))
    for b in range(2):
        np.testing.assert_array_equal(out[b, 0], np.broadcast_to(fm[b, 0, 0], (7, 7, 3)))
    # outside the image -> extrapolation value
    out = oracle.crop_and_resize(fm, f32([[-1, -1, -0.5, -0.5]]), [1], (2, 2), extrapolation_value=7.0)
    assert (out == 7.0).all()
    with pytest.raises(ValueError):
        oracle.crop_and_resize(fm, f32([[0, 0, 1, 1]]), [2], (2, 2))


def test_select_mask_known_answers():
    mask = np.array([[True, False, True, True, True]])
    rnd = np.array([[5, 9, 7, 7, 1]], np.int32)
    # keys = [5,0,7,7,1]; descending order with ties by index: idx 2,3,0,4,1 ; keep rank < 2
    np.testing.assert_array_equal(oracle.randomly_select_xyz_mask(mask, [2], rnd), [[False, False, True, True, False]])
    np.testing.assert_array_equal(oracle.randomly_select_xyz_mask(mask, [10], rnd), mask)
    np.testing.assert_array_equal(oracle.randomly_select_xyz_mask(mask, [0], rnd), np.zeros_like(mask))


def test_rpn_targets_properties():
    hp = synth.hyper_params(feature_map_shape=8)
    anchors = oracle.generate_anchors(hp)
    gt, labels = synth.gt_boxes(3, 4, 6)
    rp, rn = synth.sampling_ints(3, 4, anchors.shape[0])
    d, l, miou, arow, acol = oracle.calculate_rpn_actual_outputs(anchors, gt, labels, hp, (rp, rn), debug=True)
    assert l.shape == (4, 8, 8, 9) and d.shape == (4, 576, 4)
    lf = l.reshape(4, -1)
    iou = oracle.generate_iou_map(anchors, gt)
    np.testing.assert_array_equal(miou, iou.max(-1))
    np.testing.assert_array_equal(arow, iou.argmax(-1))
    np.testing.assert_array_equal(acol, iou.argmax(1))
    assert ((lf == 1).sum(1) <= 128).all() and ((lf == 1).sum(1) + (lf == 0).sum(1) <= 256).all()
    assert (miou[lf == 0] < 0.3).all()                      # negatives have IoU < 0.3
    assert (np.abs(d).sum(-1)[lf != 1] == 0).all()          # only positives carry regression targets
    for b in range(4):                                       # every valid gt's best anchor is positive (few gts here)
        for m in np.nonzero(labels[b] != -1)[0]:
            assert lf[b, acol[b, m]] == 1


# ---- independent NumPy restatement must agree bit for bit ----------------------------------------------
def test_numpy_restatement_exp_log():
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.normal(0, 3, 5000), rng.uniform(-87, 88, 2000), [0.0, 88.5, -88.5, 100, -100]]).astype(f32)
    assert bits_equal(oracle.exp_array(x), npr.cephes_exp(x))
    y = np.concatenate([np.exp(rng.uniform(-40, 40, 5000)), [1.0, 0.5, 2.0, 1e-38, 3e38]]).astype(f32)
    assert bits_equal(oracle.log_array(y), npr.cephes_log(y))


def test_numpy_restatement_anchors_decode_iou():
    for hp in (synth.hyper_params("vgg16"), synth.hyper_params(img_size=1000, feature_map_shape=63)):
        assert bits_equal(oracle.generate_anchors(hp), npr.generate_anchors(hp))
    hp = synth.hyper_params(feature_map_shape=8)
    anchors = oracle.generate_anchors(hp)
    deltas, _ = synth.rpn_outputs(0, 3, 8)
    scaled = (deltas.reshape(3, -1, 4) * f32(hp["variances"])).astype(f32)
    assert bits_equal(oracle.get_bboxes_from_deltas(anchors, deltas.reshape(3, -1, 4), hp["variances"]),
                      npr.get_bboxes_from_deltas(anchors, scaled))
    gt, _ = synth.gt_boxes(0, 3, 7)
    assert bits_equal(oracle.generate_iou_map(anchors, gt), npr.generate_iou_map(anchors, gt))


def test_numpy_restatement_crop_and_nms():
    rng = np.random.default_rng(2)
    img = rng.normal(size=(2, 5, 6, 3)).astype(f32)
    boxes = np.sort(rng.random((6, 2, 2)).astype(f32), axis=1).reshape(6, 4)
    boxes[0] = [0, 0, 1, 1]
    boxes[1] = [-0.2, 0.2, 1.2, 0.8]
    ind = rng.integers(0, 2, 6)
    for crop in [(7, 7), (1, 3), (4, 1)]:
        assert bits_equal(oracle.crop_and_resize(img, boxes, ind, crop, 0.5), npr.crop_and_resize(img, boxes, ind, crop, 0.5))
    c = rng.random((60, 2))
    hw = rng.uniform(0.05, 0.25, (60, 2))
    bx = np.concatenate([c - hw, c + hw], -1).astype(f32)
    sc = rng.random(60).astype(f32)
    sel = npr.single_class_nms(bx, sc, 20, 0.3, 0.2)
    b, s, _, v = oracle.combined_non_max_suppression(bx[None, :, None], sc[None, :, None], 20, 20, 0.3, 0.2, clip_boxes=False)
    assert v[0] == len(sel)
    assert bits_equal(b[0, :len(sel)], bx[sel]) and bits_equal(s[0, :len(sel)], sc[sel])


def test_crop_and_resize_agrees_with_torch_grid_sample():
    """An implementation nobody here wrote: torch.nn.functional.grid_sample(bilinear, align_corners=True) samples the same
    grid as tf.image.crop_and_resize for boxes inside the image (pixel i of H at i / (H - 1)).  Evaluated in fp64 it must
    agree with the fp32 oracle to a few ulps -- a cross-check of the sampling coordinates and the lerp weights (the
    extrapolation rule differs between the two and is not exercised: every sample lies inside)."""
    torch = pytest.importorskip("torch")
    import torch.nn.functional as F

    rng = np.random.default_rng(0)
    B, H, W, C, R = 2, 9, 11, 5, 6
    fm = rng.normal(0, 1, (B, H, W, C)).astype(np.float32)
    rois = np.sort(rng.random((B, R, 2, 2)).astype(np.float32), axis=2).reshape(B, R, 4)
    want = oracle.roi_pooling(fm, rois, (7, 7))
    t = torch.from_numpy(fm).permute(0, 3, 1, 2).double()
    for b in range(B):
        for r in range(R):
            y1, x1, y2, x2 = (float(v) for v in rois[b, r])
            ys = torch.linspace(0, 1, 7, dtype=torch.float64) * (y2 - y1) + y1
            xs = torch.linspace(0, 1, 7, dtype=torch.float64) * (x2 - x1) + x1
            gy, gx = torch.meshgrid(ys * 2 - 1, xs * 2 - 1, indexing="ij")
            o = F.grid_sample(t[b:b + 1], torch.stack([gx, gy], -1)[None], mode="bilinear", padding_mode="zeros",
                              align_corners=True)
            np.testing.assert_allclose(want[b, r], o[0].permute(1, 2, 0).numpy(), rtol=0, atol=1e-5)


def test_top_k_agrees_with_torch_topk():
    """torch.topk(sorted=True) on rows of distinct values: same values, same indices as the oracle's tf.nn.top_k (tie order
    is the one thing the two libraries do not share, hence distinct values)."""
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(4)
    s = rng.permutation(3 * 5000).reshape(3, 5000).astype(np.float32) / np.float32(16384)
    vals, idx = oracle.top_k(s, 700)
    tv, ti = torch.topk(torch.from_numpy(s), 700, dim=1, largest=True, sorted=True)
    assert bits_equal(vals, tv.numpy()) and (idx == ti.numpy()).all()


# ---- committed golden vectors -----------------------------------------------------------------------------
def test_golden_inference_path():
    g = np.load(os.path.join(GOLDEN, "inference_path.npz"))
    hp = synth.hyper_params(feature_map_shape=8, pre_nms_topn=64, test_nms_topn=16)
    assert bits_equal(oracle.generate_anchors(hp), g["anchors"])
    rois, sc, valid = oracle.roi_bbox(g["deltas"], g["scores"], g["anchors"], hp["variances"], 64, 16, 0.7, return_all=True)
    assert bits_equal(rois, g["rois"]) and bits_equal(sc, g["roi_scores"]) and (valid == g["roi_valid"]).all()
    assert bits_equal(oracle.roi_pooling(g["fm"], g["rois"], (7, 7)), g["pooled"])
    b, l, s, v = oracle.decoder(g["rois"], g["head_deltas"], g["head_probs"], hp["variances"], 5, return_valid=True)
    assert bits_equal(b, g["det_boxes"]) and bits_equal(l, g["det_labels"]) and bits_equal(s, g["det_scores"])
    assert (v == g["det_valid"]).all()


def test_golden_rpn_targets():
    g = np.load(os.path.join(GOLDEN, "rpn_targets.npz"))
    hp = synth.hyper_params(feature_map_shape=8, total_pos_bboxes=8, total_neg_bboxes=8)
    assert bits_equal(oracle.generate_iou_map(g["anchors"], g["gt_boxes"]), g["iou_map"])
    d, l, miou, arow, acol = oracle.calculate_rpn_actual_outputs(g["anchors"], g["gt_boxes"], g["gt_labels"], hp,
                                                                 (g["rnd_pos"], g["rnd_neg"]), debug=True)
    assert bits_equal(d, g["bbox_deltas"]) and bits_equal(l, g["bbox_labels"]) and bits_equal(miou, g["max_iou"])
    assert (arow == g["argmax_row"]).all() and (acol == g["argmax_col"]).all()


def test_golden_roi_delta():
    g = np.load(os.path.join(GOLDEN, "roi_delta.npz"))
    hp = synth.hyper_params(total_labels=5, total_pos_bboxes=4, total_neg_bboxes=4)
    d, l = oracle.roi_delta(g["rois"], g["gt_boxes"], g["gt_labels"], hp, (g["rnd_pos"], g["rnd_neg"]))
    assert bits_equal(d, g["roi_bbox_deltas"]) and bits_equal(l, g["roi_bbox_labels"])
    assert l[..., 1:].sum() > 0 and l[..., 0].sum() > 0  # the fixture holds positives and negatives


def test_roi_delta_known_answers():
    """models/faster_rcnn.py:215-234 by hand: one positive (IoU 0.8 with gt 0, label 2), one negative (IoU 0.25), one ignored
    RoI (disjoint), one zero-padded RoI against a zero-padded gt (IoU NaN -> ignored)."""
    hp = synth.hyper_params(total_labels=3, total_pos_bboxes=4, total_neg_bboxes=4)
    gt = np.array([[[0.1, 0.1, 0.6, 0.5], [0.0, 0.0, 0.0, 0.0]]], np.float32)
    labels = np.array([[2, -1]], np.int32)
    rois = np.array([[[0.1, 0.1, 0.5, 0.5], [0.1, 0.1, 0.225, 0.5], [0.7, 0.7, 0.9, 0.9], [0, 0, 0, 0]]], np.float32)
    rnd = np.ones((1, 4), np.int32)
    d, l = oracle.roi_delta(rois, gt, labels, hp, (rnd, rnd))
    assert l.tolist() == [[[0, 0, 1], [1, 0, 0], [0, 0, 0], [0, 0, 0]]]
    d = d.reshape(4, 3, 4)
    # positive: dy = (0.35 - 0.3) / 0.4 / 0.1, dh = log(0.5 / 0.4) / 0.2, dx = dw = 0, in class slot 2 only
    np.testing.assert_allclose(d[0, 2], [1.25, 0.0, np.log(1.25) / 0.2, 0.0], rtol=2e-6, atol=1e-6)
    assert not d[0, :2].any() and not d[1:].any()


def test_roi_delta_matches_numpy_restatement():
    rng = np.random.default_rng(12)
    B, R, M, L = 3, 200, 9, 7
    gt, labels = synth.gt_boxes(4, B, M)
    labels = np.where(labels >= 0, labels % (L - 1) + 1, labels).astype(np.int32)
    rois = np.sort(rng.random((B, R, 2, 2)).astype(np.float32), axis=2).reshape(B, R, 4)
    rois[:, :2 * M] = np.clip(np.repeat(gt, 2, axis=1) + rng.normal(0, 0.04, (B, 2 * M, 4)), 0, 1)
    rois[:, -3:] = 0.0
    rp = rng.integers(1, 100, (B, R), dtype=np.int32)
    rn = rng.integers(1, 100, (B, R), dtype=np.int32)
    hp = synth.hyper_params(total_labels=L, total_pos_bboxes=10, total_neg_bboxes=10)
    d, l = oracle.roi_delta(rois, gt, labels, hp, (rp, rn))
    nd, nl = npr.roi_delta(rois, gt, labels, L, 10, 10, hp["variances"], rp, rn)
    assert bits_equal(l, nl) and bits_equal(d, nd)
    assert l[..., 1:].sum() > 0 and l[..., 0].sum() > 0


def test_oracle_threads_agree():
    hp = synth.hyper_params(feature_map_shape=8, pre_nms_topn=200, test_nms_topn=32)
    anchors = oracle.generate_anchors(hp)
    deltas, scores = synth.rpn_outputs(5, 4, 8)
    oracle.set_threads(1)
    a = oracle.roi_bbox(deltas, scores, anchors, hp["variances"], 200, 32, 0.7)
    oracle.set_threads(4)
    b = oracle.roi_bbox(deltas, scores, anchors, hp["variances"], 200, 32, 0.7)
    oracle.set_threads(1)
    assert bits_equal(a, b)
