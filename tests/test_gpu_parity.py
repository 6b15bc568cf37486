"""GPU parity: every HIP entry point (through the C ABI) against the CPU oracle on the same inputs.

Bar (north_star): index sets and integer outputs bit-exact; fp32 boxes / IoUs / pooled features are
compared bit-exactly too (the kernels and the oracle evaluate the same separately-rounded fp32
operations), which is stricter than the 1e-5 relative tolerance the north_star allows -- where a test
falls back to the tolerance it says so.
"""

import numpy as np
import pytest

import oracle
import synth

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu

RTOL = 1e-5  # north_star tolerance for boxes / IoUs / pooled features


@pytest.fixture(scope="module")
def pkg():
    import tf_faster_rcnn_amd as p
    return p


def dev(x):
    return torch.as_tensor(x).cuda()


def assert_bits(actual, expected, what=""):
    a = actual.detach().cpu().numpy() if hasattr(actual, "detach") else np.asarray(actual)
    e = np.asarray(expected)
    assert a.shape == e.shape, f"{what}: shape {a.shape} vs {e.shape}"
    if a.dtype.kind == "f":
        same = (a.view(np.uint32) == e.astype(np.float32).view(np.uint32)) | (np.isnan(a) & np.isnan(e))
        # +0 / -0 are the same value; accept them as equal
        same |= (a == e)
    else:
        same = a == e
    if not same.all():
        bad = np.argwhere(~same)
        i = tuple(bad[0])
        raise AssertionError(f"{what}: {len(bad)} of {a.size} elements differ; first at {i}: got {a[i]!r} want {e[i]!r}")


# ------------------------------------------------------------------------------------------------
# anchors / codec
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("backbone,over", [("mobilenet_v2", {}), ("vgg16", {}),
                                           ("mobilenet_v2", dict(img_size=1000, feature_map_shape=63))])
def test_generate_anchors(pkg, backbone, over):
    hp = pkg.train_utils.get_hyper_params(backbone, **over)
    got = pkg.bbox_utils.generate_anchors(hp)
    want = oracle.generate_anchors(hp)
    assert_bits(pkg.bbox_utils.generate_base_anchors(hp), oracle.generate_base_anchors(hp), "base anchors")
    assert_bits(got, want, "anchors")


def test_exp_log_restatement_bits(pkg):
    """The device Cephes exp/log restatements agree bit for bit with the oracle's, via decode/encode."""
    rng = np.random.default_rng(7)
    n = 20000
    d = np.zeros((1, n, 4), np.float32)
    d[0, :, 2] = rng.uniform(-20, 20, n)
    d[0, :, 3] = rng.normal(0, 3, n)
    d[0, :5, 2] = [0.0, 88.5, -88.5, 100.0, -100.0]
    a = np.tile(np.array([[0.0, 0.0, 1.0, 1.0]], np.float32), (n, 1))
    got = pkg.bbox_utils.get_bboxes_from_deltas(dev(a), dev(d))
    assert_bits(got, oracle.get_bboxes_from_deltas(a, d), "decode(exp)")
    # log through encode: gt sizes over many magnitudes
    g = np.zeros((1, n, 4), np.float32)
    g[0, :, 2] = np.exp(rng.uniform(-30, 3, n)).astype(np.float32)
    g[0, :, 3] = rng.uniform(1e-6, 4, n).astype(np.float32)
    got = pkg.bbox_utils.get_deltas_from_bboxes(dev(a), dev(g))
    assert_bits(got, oracle.get_deltas_from_bboxes(a, g), "encode(log)")


@pytest.mark.parametrize("B,G", [(2, 8), (16, 32)])
def test_decode_rpn(pkg, B, G):
    hp = synth.hyper_params(feature_map_shape=G)
    anchors = oracle.generate_anchors(hp)
    deltas, _ = synth.rpn_outputs(0, B, G)
    deltas = deltas.reshape(B, -1, 4)
    got = pkg.bbox_utils.get_bboxes_from_deltas(dev(anchors), dev(deltas), variances=hp["variances"])
    assert_bits(got, oracle.get_bboxes_from_deltas(anchors, deltas, hp["variances"]), "decode")
    # plain function, caller-side variance multiply like the reference (models/faster_rcnn.py:149)
    scaled = (deltas * np.asarray(hp["variances"], np.float32)).astype(np.float32)
    got = pkg.bbox_utils.get_bboxes_from_deltas(dev(anchors), dev(scaled))
    assert_bits(got, oracle.get_bboxes_from_deltas(anchors, scaled), "decode pre-scaled")


def test_decode_same_shape_and_zero_deltas(pkg):
    rng = np.random.default_rng(1)
    rois = rng.random((2, 5, 3, 4)).astype(np.float32)
    d = rng.normal(0, 0.3, (2, 5, 3, 4)).astype(np.float32)
    got = pkg.bbox_utils.get_bboxes_from_deltas(dev(rois), dev(d))
    want = oracle.get_bboxes_from_deltas(rois.reshape(-1, 4), d.reshape(1, -1, 4)).reshape(d.shape)
    assert_bits(got, want, "decode same-shape")
    # exp(0) == 1 exactly: zero deltas give back the anchor up to the centre round trip
    a = np.array([[0.25, 0.25, 0.75, 0.5]], np.float32)
    got = pkg.bbox_utils.get_bboxes_from_deltas(dev(a), dev(np.zeros((1, 1, 4), np.float32))).cpu().numpy()
    np.testing.assert_array_equal(got[0], a)


def test_encode(pkg):
    hp = synth.hyper_params(feature_map_shape=8)
    anchors = oracle.generate_anchors(hp)
    rng = np.random.default_rng(3)
    gt = rng.random((3, anchors.shape[0], 4)).astype(np.float32)
    gt[..., 2:] += gt[..., :2]
    gt[0, :50] = 0.0  # zero-size gt -> zero deltas
    got = pkg.bbox_utils.get_deltas_from_bboxes(dev(anchors), dev(gt), variances=hp["variances"])
    assert_bits(got, oracle.get_deltas_from_bboxes(anchors, gt, hp["variances"]), "encode")
    # zero-size bbox dims -> 1e-3
    bb = np.array([[0.5, 0.5, 0.5, 0.5]], np.float32)
    g1 = np.array([[[0.4, 0.4, 0.6, 0.7]]], np.float32)
    assert_bits(pkg.bbox_utils.get_deltas_from_bboxes(dev(bb), dev(g1)), oracle.get_deltas_from_bboxes(bb, g1), "encode eps")


# ------------------------------------------------------------------------------------------------
# IoU map
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,G,M", [(2, 8, 4), (3, 8, 7), (8, 32, 100), (1, 4, 1)])
def test_iou_map(pkg, B, G, M):
    hp = synth.hyper_params(feature_map_shape=G)
    anchors = oracle.generate_anchors(hp)
    gt, _ = synth.gt_boxes(0, B, M)
    assert_bits(pkg.bbox_utils.generate_iou_map(dev(anchors), dev(gt)), oracle.generate_iou_map(anchors, gt), "iou_map")


def test_iou_map_batched_and_nan(pkg):
    rng = np.random.default_rng(5)
    bb = rng.random((2, 33, 4)).astype(np.float32)
    bb[..., 2:] += bb[..., :2]
    gt = rng.random((2, 8, 4)).astype(np.float32)
    gt[..., 2:] += gt[..., :2]
    bb[0, 0] = 0.0
    gt[0, 0] = 0.0  # 0/0 -> NaN, like the reference (no epsilon)
    got = pkg.bbox_utils.generate_iou_map(dev(bb), dev(gt))
    want = oracle.generate_iou_map(bb, gt)
    assert np.isnan(want[0, 0, 0])
    assert_bits(got, want, "iou_map batched")


def _fastdiv_boxes(rng, shape, kind):
    """Boxes that stress the IoU arithmetic: overlaps of a few ulps, equal edges, magnitudes from 2^-60 to 2^60, negative
    areas, non-finite coordinates."""
    n = int(np.prod(shape))
    y1 = rng.random(n).astype(np.float32)
    x1 = rng.random(n).astype(np.float32)
    h = rng.random(n).astype(np.float32)
    w = rng.random(n).astype(np.float32)
    if kind == "slivers":        # overlaps and sides down to a few ulps
        h = np.float32(2.0) ** rng.integers(-22, 1, n).astype(np.float32)
        w = np.float32(2.0) ** rng.integers(-22, 1, n).astype(np.float32)
    elif kind == "grid":         # coordinates on a coarse grid: many equal edges, zero-width overlaps, identical boxes
        y1, x1 = (np.round(y1 * 8) / 8).astype(np.float32), (np.round(x1 * 8) / 8).astype(np.float32)
        h, w = (np.round(h * 4) / 8).astype(np.float32), (np.round(w * 4) / 8).astype(np.float32)
    elif kind == "extremes":     # both ends of the nice range, and just outside it
        scale = np.float32(2.0) ** rng.integers(-21, 3, n).astype(np.float32)
        y1, x1, h, w = y1 * scale, x1 * scale, h * scale, w * scale
    elif kind == "rough":        # negative areas, huge and tiny magnitudes, infinities: the exact divide must take over
        scale = np.float32(2.0) ** rng.integers(-60, 60, n).astype(np.float32)
        y1, x1, h, w = y1 * scale, x1 * scale, (h - np.float32(0.3)) * scale, (w - np.float32(0.3)) * scale
    b = np.stack([y1, x1, y1 + h, x1 + w], axis=-1).astype(np.float32)
    return b.reshape(*shape, 4)


@pytest.mark.parametrize("kind", ["plain", "slivers", "grid", "extremes", "rough"])
def test_iou_map_range_stress(pkg, kind):
    """Every quotient of the materialised map is the correctly rounded one (= the oracle's) over the whole range of
    operands, NaN / infinity handling included (the wave-uniform fast paths of csrc/iou_map.hip must not change a bit)."""
    rng = np.random.default_rng({"plain": 1, "slivers": 2, "grid": 3, "extremes": 4, "rough": 5}[kind])
    B, N, M = 6, 4096, 100
    bb = _fastdiv_boxes(rng, (B, N), kind)
    gt = _fastdiv_boxes(rng, (B, M), kind)
    gt[:, 60:] = 0.0                       # zero padding, as the data pipeline produces
    if kind == "grid":
        gt[:, :20] = bb[:, :20]            # identical boxes: IoU exactly 1
    if kind == "rough":
        bb[1, 7] = [0.1, 0.1, np.inf, 0.5]
        gt[2, 3] = [np.nan, 0.1, 0.4, 0.5]
        bb[3] = _fastdiv_boxes(rng, (N,), "plain")   # a nice image between rough ones
        gt[3] = _fastdiv_boxes(rng, (M,), "plain")
    want = oracle.generate_iou_map(bb, gt)
    assert_bits(pkg.bbox_utils.generate_iou_map(dev(bb), dev(gt)), want, f"iou_map {kind} (batched boxes)")
    want1 = oracle.generate_iou_map(bb[0], gt)
    assert_bits(pkg.bbox_utils.generate_iou_map(dev(bb[0]), dev(gt)), want1, f"iou_map {kind} (shared boxes)")


@pytest.mark.parametrize("split", [0, 1])
@pytest.mark.parametrize("kind", ["plain", "slivers", "grid", "extremes", "rough"])
def test_iou_map_reduce_range_stress(pkg, knobs, kind, split):
    """frcnn_iou_map_reduce (the fused reductions of utils/train_utils.py:177-179) through the C ABI on the same inputs:
    row max / first argmax / first column argmax must equal those of the oracle's materialised map -- with one box per
    lane walking every gt, and with the small-batch form (64 boxes x 4 gt quarters per workgroup)."""
    from tf_faster_rcnn_amd import _lib

    knobs.setenv("FRCNN_IOU_REDUCE_SPLIT", split)

    rng = np.random.default_rng({"plain": 11, "slivers": 12, "grid": 13, "extremes": 14, "rough": 15}[kind])
    B, N, M = 3, 1500, 37
    bb = _fastdiv_boxes(rng, (B, N), kind)
    gt = _fastdiv_boxes(rng, (B, M), kind)
    gt[:, 30:] = 0.0
    if kind == "rough":
        bb[1] = _fastdiv_boxes(rng, (N,), "plain")
        gt[1] = _fastdiv_boxes(rng, (M,), "plain")
        bb[1, 100:164] = 0.0               # one wave of zero boxes (0/0 = NaN against the padded gts) inside a nice image
    iou = oracle.generate_iou_map(bb, gt)
    filled = np.where(np.isnan(iou), -np.inf, iou)
    want_max = filled.max(axis=2)
    want_max = np.where(np.isinf(want_max), np.float32(-3.402823466e38), want_max).astype(np.float32)
    want_row = filled.argmax(axis=2).astype(np.int32)
    want_col = filled.argmax(axis=1).astype(np.int32)
    lib = _lib.lib()
    d_bb, d_gt = dev(bb), dev(gt)
    mx = torch.empty((B, N), dtype=torch.float32, device="cuda")
    ar = torch.empty((B, N), dtype=torch.int32, device="cuda")
    ac = torch.empty((B, M), dtype=torch.int32, device="cuda")
    ws = torch.empty(max(1, lib.frcnn_iou_reduce_workspace_bytes(B, N, M)), dtype=torch.uint8, device="cuda")
    _lib.check(lib.frcnn_iou_map_reduce(d_bb.data_ptr(), 1, d_gt.data_ptr(), B, N, M, mx.data_ptr(), ar.data_ptr(), ac.data_ptr(),
                                        ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert_bits(mx, want_max, f"iou_map_reduce {kind}: row max")
    assert_bits(ar, want_row, f"iou_map_reduce {kind}: row argmax")
    assert_bits(ac, want_col, f"iou_map_reduce {kind}: column argmax")


@pytest.mark.parametrize("split", [0, 1])
def test_iou_map_reduce_ties_and_ragged_shapes(pkg, knobs, split):
    """Equal maxima (the earliest gt / the lowest box index wins), gt counts that do not divide by four, fewer gts than
    quarters, box counts that are not a multiple of the workgroup: both forms of the fused reductions."""
    from tf_faster_rcnn_amd import _lib

    knobs.setenv("FRCNN_IOU_REDUCE_SPLIT", split)
    rng = np.random.default_rng(5)
    lib = _lib.lib()
    for B, N, M in ((2, 130, 9), (1, 64, 3), (3, 257, 50), (2, 1000, 1), (1, 77, 101)):
        gt = np.sort(rng.random((B, M, 2, 2)).astype(np.float32), axis=2).reshape(B, M, 4)
        gt[:, M // 2:] = gt[:, : M - M // 2]      # duplicated gts: equal IoUs in different quarters
        bb = np.sort(rng.random((B, N, 2, 2)).astype(np.float32), axis=2).reshape(B, N, 4)
        bb[:, N // 3: 2 * (N // 3)] = bb[:, : N // 3]  # duplicated boxes: equal column maxima
        iou = oracle.generate_iou_map(bb, gt)
        filled = np.where(np.isnan(iou), -np.inf, iou)
        mx = torch.empty((B, N), dtype=torch.float32, device="cuda")
        ar = torch.empty((B, N), dtype=torch.int32, device="cuda")
        ac = torch.empty((B, M), dtype=torch.int32, device="cuda")
        ws = torch.empty(max(1, lib.frcnn_iou_reduce_workspace_bytes(B, N, M)), dtype=torch.uint8, device="cuda")
        d_bb, d_gt = dev(bb), dev(gt)
        _lib.check(lib.frcnn_iou_map_reduce(d_bb.data_ptr(), 1, d_gt.data_ptr(), B, N, M, mx.data_ptr(), ar.data_ptr(),
                                            ac.data_ptr(), ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        assert_bits(mx, filled.max(axis=2).astype(np.float32), f"row max B={B} N={N} M={M}")
        assert_bits(ar, filled.argmax(axis=2).astype(np.int32), f"row argmax B={B} N={N} M={M}")
        assert_bits(ac, filled.argmax(axis=1).astype(np.int32), f"column argmax B={B} N={N} M={M}")


# ------------------------------------------------------------------------------------------------
# top-k
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,N,K", [(2, 576, 64), (3, 1000, 1000), (16, 9216, 6000), (2, 35721, 12000), (1, 70000, 300), (2, 100, 1)])
def test_top_k(pkg, B, N, K):
    rng = np.random.default_rng(N + K)
    s = rng.random((B, N)).astype(np.float32)
    vals, idx = pkg.bbox_utils.top_k(dev(s), K)
    wv, wi = oracle.top_k(s, K)
    assert_bits(idx, wi, "top_k indices")
    assert_bits(vals, wv, "top_k values")


def test_top_k_score_distributions(pkg, knobs):
    """The full-sort kernel orders its slots with a bucket sort (buckets linear in the score, then in the key's bit pattern)
    and falls back to the stable radix sort when a bucket overflows: distributions that take each of the three routes,
    non-finite scores included, with and without the bucket sort."""
    rng = np.random.default_rng(77)
    B, N, K = 3, 9000, 6000
    dists = {
        "uniform": lambda: rng.random((B, N)),
        "log_uniform": lambda: np.exp(-40.0 * rng.random((B, N))),
        "sigmoid_saturated": lambda: 1.0 / (1.0 + np.exp(-8.0 * rng.normal(0, 1, (B, N)))),
        "signed_wide": lambda: rng.normal(0, 1e10, (B, N)),
        "constant": lambda: np.full((B, N), 0.5),
        "few_values": lambda: rng.choice(rng.random(20), (B, N)),
        "tight_cluster": lambda: 0.25 + 1e-7 * rng.random((B, N)),
        "with_inf_nan": lambda: np.where(rng.random((B, N)) < 0.01, rng.choice([np.inf, -np.inf, np.nan], (B, N)), rng.random((B, N))),
        "huge_range": lambda: rng.choice([1e38, -1e38, 1e-38, 0.0], (B, N)) * rng.random((B, N)),
    }
    for name, draw in dists.items():
        s = draw().astype(np.float32)
        wv, wi = oracle.top_k(s, K)
        for radix_only in (0, 1):
            knobs.setenv("FRCNN_SORT_RADIX_ONLY", radix_only)
            vals, idx = pkg.bbox_utils.top_k(dev(s), K)
            assert_bits(idx, wi, f"top_k indices, scores={name}, radix_only={radix_only}")
            assert_bits(vals, wv, f"top_k values, scores={name}, radix_only={radix_only}")


def test_top_k_ties_and_errors(pkg):
    rng = np.random.default_rng(11)
    s = rng.integers(0, 8, size=(4, 3000)).astype(np.float32) / 8.0  # heavy ties -> ascending index
    s[0, :10] = [0.0, -0.0, 0.0, -0.0, 1.0, 1.0, -1.0, -1.0, 0.5, 0.5]
    vals, idx = pkg.bbox_utils.top_k(dev(s), 2000)
    wv, wi = oracle.top_k(s, 2000)
    assert_bits(idx, wi, "top_k ties idx")
    # ties in a segment that needs the radix select (n > LDS capacity)
    s2 = rng.integers(0, 50, size=(2, 40000)).astype(np.float32)
    _, idx2 = pkg.bbox_utils.top_k(dev(s2), 5000)
    _, wi2 = oracle.top_k(s2, 5000)
    assert_bits(idx2, wi2, "top_k select ties idx")
    with pytest.raises(ValueError):
        pkg.bbox_utils.top_k(dev(s), 3001)


# ------------------------------------------------------------------------------------------------
# NMS
# ------------------------------------------------------------------------------------------------
def _rand_boxes(rng, shape, spread=0.25):
    c = rng.random(shape + (2,))
    hw = rng.uniform(0.05, spread, shape + (2,))
    return np.concatenate([c - hw, c + hw], -1).astype(np.float32)


@pytest.mark.parametrize("B,n,q,c,mpc,mt,iou,thr", [
    (2, 64, 1, 1, 16, 16, 0.7, float("-inf")),
    (2, 300, 5, 5, 20, 30, 0.5, 0.5),
    (3, 300, 1, 4, 200, 200, 0.5, 0.3),
    (2, 1000, 1, 1, 1000, 1000, 0.3, float("-inf")),  # dense suppression, several chunks, no early exit
    (1, 700, 3, 3, 50, 40, 0.6, 0.2),
])
def test_combined_nms(pkg, B, n, q, c, mpc, mt, iou, thr):
    rng = np.random.default_rng(n * 7 + c)
    boxes = _rand_boxes(rng, (B, n, q))
    boxes[0, : n // 4] = boxes[0, n // 4: 2 * (n // 4)]  # exact duplicates (IoU 1)
    scores = rng.random((B, n, c)).astype(np.float32)
    got = pkg.bbox_utils.non_max_suppression(dev(boxes), dev(scores), max_output_size_per_class=mpc, max_total_size=mt,
                                            iou_threshold=iou, score_threshold=thr)
    want = oracle.combined_non_max_suppression(boxes, scores, mpc, mt, iou, thr)
    for g, w, name in zip(got, want, ("boxes", "scores", "classes", "valid")):
        assert_bits(g, w, f"nms {name}")


def test_nms_known_answers(pkg):
    # three boxes: b1 overlaps b0 with IoU just above 0.7, b2 is disjoint
    boxes = np.array([[[[0.0, 0.0, 1.0, 1.0]], [[0.0, 0.0, 1.0, 0.72]], [[2.0, 2.0, 3.0, 3.0]]]], np.float32)
    scores = np.array([[[0.9], [0.8], [0.7]]], np.float32)
    b, s, c, v = pkg.bbox_utils.non_max_suppression(dev(boxes), dev(scores), max_output_size_per_class=3, max_total_size=3,
                                                    iou_threshold=0.7)
    assert int(v[0]) == 2
    np.testing.assert_array_equal(s.cpu().numpy()[0], np.array([0.9, 0.7, 0.0], np.float32))
    np.testing.assert_array_equal(b.cpu().numpy()[0, 1], np.array([1, 1, 1, 1], np.float32))  # clipped to [0,1]
    # IoU exactly equal to the threshold must NOT suppress (strict >): IoU(b0, half box) == 0.5
    boxes = np.array([[[[0.0, 0.0, 1.0, 1.0]], [[0.0, 0.0, 1.0, 0.5]]]], np.float32)
    scores = np.array([[[0.9], [0.8]]], np.float32)
    _, _, _, v = pkg.bbox_utils.non_max_suppression(dev(boxes), dev(scores), max_output_size_per_class=2, max_total_size=2,
                                                    iou_threshold=0.5)
    assert int(v[0]) == 2
    # equal scores: lower index first (this build's documented tie rule)
    boxes = np.array([[[[0.0, 0.0, 0.5, 0.5]], [[0.5, 0.5, 1.0, 1.0]]]], np.float32)
    scores = np.array([[[0.6], [0.6]]], np.float32)
    b, _, _, _ = pkg.bbox_utils.non_max_suppression(dev(boxes), dev(scores), max_output_size_per_class=2, max_total_size=2)
    np.testing.assert_array_equal(b.cpu().numpy()[0], boxes[0, :, 0])
    # zero-area and inverted boxes: area <= 0 -> IoU 0 -> never suppressed
    boxes = np.array([[[[0.2, 0.2, 0.2, 0.9]], [[0.2, 0.2, 0.2, 0.9]], [[0.9, 0.9, 0.1, 0.1]], [[0.1, 0.1, 0.9, 0.9]]]], np.float32)
    scores = np.array([[[0.9], [0.8], [0.7], [0.6]]], np.float32)
    got = pkg.bbox_utils.non_max_suppression(dev(boxes), dev(scores), max_output_size_per_class=4, max_total_size=4)
    want = oracle.combined_non_max_suppression(boxes, scores, 4, 4)
    for g, w in zip(got, want):
        assert_bits(g, w, "nms degenerate")
    # non-finite coordinates take the reference formulation (std::min/std::max NaN behaviour and all)
    rng = np.random.default_rng(33)
    boxes = _rand_boxes(rng, (2, 200, 1))
    boxes[0, 5, 0] = [np.nan, 0.1, 0.5, 0.6]
    boxes[0, 9, 0] = [0.1, 0.1, np.inf, 0.6]
    boxes[1, 3, 0] = [-np.inf, -np.inf, np.inf, np.inf]
    boxes[1, 7, 0] = [0.2, np.nan, np.nan, 0.3]
    scores = rng.random((2, 200, 1)).astype(np.float32)
    got = pkg.bbox_utils.non_max_suppression(dev(boxes), dev(scores), max_output_size_per_class=200, max_total_size=200,
                                            iou_threshold=0.3)
    want = oracle.combined_non_max_suppression(boxes, scores, 200, 200, 0.3)
    for g, w in zip(got, want):
        assert_bits(g, w, "nms non-finite")
    # thresholds at the edges of the exact compare: 0, negative, > 1
    for thr in (0.0, -0.5, 1.5, 1.0):
        got = pkg.bbox_utils.non_max_suppression(dev(boxes[:, 20:]), dev(scores[:, 20:]), max_output_size_per_class=50,
                                                max_total_size=50, iou_threshold=thr)
        want = oracle.combined_non_max_suppression(boxes[:, 20:], scores[:, 20:], 50, 50, thr)
        for g, w in zip(got, want):
            assert_bits(g, w, f"nms iou_threshold={thr}")


# ------------------------------------------------------------------------------------------------
# layers
# ------------------------------------------------------------------------------------------------
def _roi_bbox_case(pkg, seed, B, G, K, R, mode="inference", thr=0.7, unique=True):
    hp = pkg.train_utils.get_hyper_params("mobilenet_v2")
    hp.update(feature_map_shape=G, pre_nms_topn=K, nms_iou_threshold=thr)
    hp["train_nms_topn" if mode == "training" else "test_nms_topn"] = R
    anchors = oracle.generate_anchors(hp)
    deltas, scores = synth.rpn_outputs(seed, B, G, unique_scores=unique)
    layer = pkg.faster_rcnn.RoIBBox(dev(anchors), mode, hp, name="roi_bboxes")
    got = layer([dev(deltas), dev(scores)])
    want = oracle.roi_bbox(deltas, scores, anchors, hp["variances"], K, R, thr)
    return got, want


@pytest.mark.parametrize("B,G,K,R,thr", [(2, 8, 64, 16, 0.7), (2, 16, 2304, 300, 0.7), (2, 16, 2000, 2000, 0.3),
                                         (3, 8, 576, 600, 0.7)])
def test_roi_bbox_small(pkg, B, G, K, R, thr):
    got, want = _roi_bbox_case(pkg, 0, B, G, K, R, thr=thr)
    assert_bits(got, want, "RoIBBox")


def test_roi_bbox_config2(pkg):
    """BASELINE config 2: mobilenet_v2, B=16, N=9216, K=6000, R=300."""
    got, want = _roi_bbox_case(pkg, 1, 16, 32, 6000, 300)
    assert_bits(got, want, "RoIBBox cfg2")


def test_roi_bbox_fused_kernel_large_r(pkg, knobs):
    """The one-workgroup-per-image kernel at a post_nms_topn the bitmask schedule would otherwise take."""
    knobs.setenv("FRCNN_ROI_BBOX_BITMASK_MIN_R", "0")
    got, want = _roi_bbox_case(pkg, 2, 2, 31, 6000, 2000, mode="training")
    assert_bits(got, want, "RoIBBox fused kernel, R=2000")


def test_roi_bbox_train_mode_and_ties(pkg):
    got, want = _roi_bbox_case(pkg, 2, 4, 31, 6000, 2000, mode="training")
    assert_bits(got, want, "RoIBBox cfg3-like (vgg16 grid, R=2000)")
    got, want = _roi_bbox_case(pkg, 3, 2, 16, 1000, 100, unique=False, thr=0.5)
    assert_bits(got, want, "RoIBBox non-unique scores")


def test_roi_bbox_stress_shape(pkg):
    """BASELINE config 5 shape: G=63 (35721 anchors), pre_nms_topn 12000 -> radix select path."""
    got, want = _roi_bbox_case(pkg, 4, 2, 63, 12000, 300)
    assert_bits(got, want, "RoIBBox cfg5")


def test_roi_bbox_unfused_path(pkg, knobs):
    """The generic sort -> NMS pipeline behind frcnn_roi_bbox (taken when the fused kernel's LDS budget is exceeded)."""
    knobs.setenv("FRCNN_ROI_BBOX_UNFUSED", "1")
    got, want = _roi_bbox_case(pkg, 7, 2, 16, 2304, 1500, mode="training")
    assert_bits(got, want, "RoIBBox unfused large R")
    got, want = _roi_bbox_case(pkg, 5, 4, 32, 6000, 300)
    assert_bits(got, want, "RoIBBox unfused cfg2")
    got, want = _roi_bbox_case(pkg, 6, 2, 63, 12000, 300)
    assert_bits(got, want, "RoIBBox unfused cfg5")


def test_roi_bbox_bitmask_path(pkg, knobs):
    """The whole-chip bitmask schedule (taken for post_nms_topn >= 512), forced on every shape: one and several phases,
    early finish, exhausted candidates, ties, K not a multiple of 64, R larger than K."""
    knobs.setenv("FRCNN_ROI_BBOX_BITMASK_MIN_R", "1")
    for threads in (256, 1024):  # the scans' workgroup size: 4 or 16 waves per image, same results
        knobs.setenv("FRCNN_BITMASK_THREADS", threads)
        for args, kw in (((10, 2, 8, 64, 16), {}), ((11, 2, 16, 2304, 300), {}), ((12, 4, 32, 6000, 300), {}),
                         ((13, 3, 31, 6000, 2000), dict(mode="training")), ((14, 2, 16, 1000, 100), dict(unique=False, thr=0.5)),
                         ((15, 2, 63, 12000, 1500), dict(mode="training")), ((16, 2, 8, 570, 600), {}),
                         ((17, 2, 16, 2304, 1500), dict(thr=0.1))):
            got, want = _roi_bbox_case(pkg, *args, **kw)
            assert_bits(got, want, f"RoIBBox bitmask ({threads} threads) {args} {kw}")
        test_roi_bbox_dense_suppression(pkg)


def test_roi_bbox_dense_suppression(pkg):
    """Heavily overlapping proposals (tiny deltas, IoU threshold 0.1): several slices and chunks are consumed."""
    hp = pkg.train_utils.get_hyper_params("mobilenet_v2")
    hp.update(feature_map_shape=16, pre_nms_topn=2304, test_nms_topn=300, nms_iou_threshold=0.1)
    anchors = oracle.generate_anchors(hp)
    deltas, scores = synth.rpn_outputs(8, 3, 16)
    deltas = (deltas * 0.05).astype(np.float32)
    got = pkg.faster_rcnn.RoIBBox(dev(anchors), "inference", hp)([dev(deltas), dev(scores)])
    want, _, valid = oracle.roi_bbox(deltas, scores, anchors, hp["variances"], 2304, 300, 0.1, return_all=True)
    assert (valid < 300).all(), "case should exhaust the candidates"
    assert_bits(got, want, "RoIBBox dense")


@pytest.mark.parametrize("thr", [0.3, 0.5, 0.7, 0.9])
def test_roi_bbox_pairs_at_the_iou_threshold(pkg, thr):
    """Pairs whose IoU lies within a few ulps of the threshold, on both routes of the packed band filter (nms_core.hpp,
    union-free form): 64 partners that meet their box in the KEPT LIST (candidates 64..127 against 0..63) and 48 pairs
    inside one chunk's tile.  Equal-size boxes shifted by dx have IoU (w - dx) / (w + dx); dx = w (1 - thr) / (1 + thr) in
    double, then the fp32 roundings of the decode scatter the IoUs around thr.  Any pair the filter decides wrongly flips a
    keep / suppress and shows as a mismatch with the oracle."""
    G = 5
    N = G * G * 9  # 225 "anchors": the boxes themselves (zero deltas)
    hp = pkg.train_utils.get_hyper_params("mobilenet_v2")
    hp.update(feature_map_shape=G, pre_nms_topn=N, test_nms_topn=N, nms_iou_threshold=thr)
    rng = np.random.default_rng(int(thr * 100))
    inside = near = 0
    for rep in range(24):
        cells = rng.permutation(144)[:112]  # 12 x 12 cells of 1/12: no overlap between cells
        cy, cx = (cells // 12) / 12.0, (cells % 12) / 12.0
        h = rng.uniform(0.02, 0.07, 112)
        w = rng.uniform(0.015, 0.035, 112)
        y1, x1 = cy + rng.uniform(0.0, 0.005, 112), cx + rng.uniform(0.0, 0.005, 112)
        dx = w * (1.0 - thr) / (1.0 + thr) * (1.0 + rng.integers(-8, 9, 112) * 2.0 ** -23)
        base = np.stack([y1, x1, y1 + h, x1 + w], -1)
        partner = np.stack([y1, x1 + dx, y1 + h, x1 + dx + w], -1)
        boxes = np.zeros((N, 4), np.float64)
        boxes[0:64] = base[:64]
        boxes[64:128] = partner[:64]
        boxes[128:224:2] = base[64:]
        boxes[129:224:2] = partner[64:]
        boxes[224] = [0.99, 0.99, 0.995, 0.995]
        anchors = boxes.astype(np.float32)
        deltas = np.zeros((1, G, G, 36), np.float32)
        scores = np.linspace(0.99, 0.01, N).astype(np.float32).reshape(1, G, G, 9)
        got = pkg.faster_rcnn.RoIBBox(dev(anchors), "inference", hp)([dev(deltas), dev(scores)])
        want, _, valid = oracle.roi_bbox(deltas, scores, anchors, hp["variances"], N, N, thr, return_all=True)
        assert_bits(got, want, f"RoIBBox threshold pairs thr={thr} rep={rep}")
        # how close the cases really are (fp32 boxes as the kernel sees them, IoU in double)
        a, b = anchors[0:64].astype(np.float64), anchors[64:128].astype(np.float64)
        iw = np.minimum(a[:, 3], b[:, 3]) - np.maximum(a[:, 1], b[:, 1])
        inter = iw * (a[:, 2] - a[:, 0])
        iou = inter / ((a[:, 3] - a[:, 1]) * (a[:, 2] - a[:, 0]) + (b[:, 3] - b[:, 1]) * (b[:, 2] - b[:, 0]) - inter)
        inside += int((np.abs(iou / thr - 1.0) < 2.0 ** -21).sum())
        near += int((np.abs(iou / thr - 1.0) < 2.0 ** -17).sum())
        assert 0 < int(valid[0]) < N
    assert near >= 24 * 32, near  # the construction does put the pairs at the threshold ...
    assert inside > 0, inside     # ... and some inside the filter's band (they take the exact test)


def test_roi_bbox_more_anchors_than_the_key_cache_and_member_masks_hold(pkg, knobs):
    """N = 72,900 anchors: no LDS key cache in any form and more elements per thread than the fast slices' 64-bit member
    masks cover, so the fused kernel runs on exact-rank slices with keys from global memory."""
    G, K, R = 90, 3000, 100
    hp = pkg.train_utils.get_hyper_params("mobilenet_v2")
    hp.update(feature_map_shape=G, pre_nms_topn=K, test_nms_topn=R)
    anchors = oracle.generate_anchors(hp)
    deltas, scores = synth.rpn_outputs(41, 1, G)
    knobs.setenv("FRCNN_ROI_BBOX_BITMASK_MIN_R", "0")
    got = pkg.faster_rcnn.RoIBBox(dev(anchors), "inference", hp)([dev(deltas), dev(scores)])
    want = oracle.roi_bbox(deltas, scores, anchors, hp["variances"], K, R, 0.7)
    assert_bits(got, want, "RoIBBox N=72900")


def test_roi_bbox_k_gt_n_raises(pkg):
    with pytest.raises(ValueError):
        _roi_bbox_case(pkg, 0, 1, 4, 1000, 10)


@pytest.mark.parametrize("B,G,C,R", [(2, 8, 8, 16), (2, 5, 6, 7), (16, 32, 576, 300), (2, 31, 512, 64)])
def test_roi_pooling(pkg, B, G, C, R):
    hp = synth.hyper_params()
    fm = synth.feature_map(0, B, G, C)
    rng = np.random.default_rng(9)
    rois = np.sort(rng.random((B, R, 2, 2)).astype(np.float32), axis=2).reshape(B, R, 4)
    rois[:, -2:] = 0.0                     # zero-padded RoIs sample pixel (0,0)
    rois[0, 0] = [0.0, 0.0, 1.0, 1.0]      # touches the border exactly
    rois[0, 1] = [-0.2, 0.3, 1.3, 0.6]     # extrapolated rows
    rois[0, 2] = [0.3, -0.5, 0.6, 1.5]     # extrapolated columns
    layer = pkg.faster_rcnn.RoIPooling(hp, name="roi_pooling")
    got = layer([dev(fm), dev(rois)])
    want = oracle.roi_pooling(fm, rois, hp["pooling_size"])
    assert_bits(got, want, "RoIPooling")


def test_roi_pooling_channel_slice_kernel(pkg, knobs):
    """The kernel used for feature maps larger than an XCD's L2 (each XCD owns an eighth of the channels): forced on small
    shapes (incl. a depth that leaves the last slices empty), and taken by itself on a 63x63x576 map."""
    knobs.setenv("FRCNN_CROP_SLICE", "1")
    test_roi_pooling(pkg, 2, 8, 8, 16)
    test_roi_pooling(pkg, 2, 31, 512, 64)
    test_roi_pooling(pkg, 1, 9, 20, 33)
    knobs.delenv("FRCNN_CROP_SLICE")
    test_roi_pooling(pkg, 1, 63, 576, 40)


def test_roi_pooling_in_two_calls(pkg, knobs):
    """frcnn_roi_pooling_prepare (pre-pass on one stream) + frcnn_roi_pooling_prepared (crop on another, ordered by an event),
    through the C ABI: same bits as the oracle on a 63x63x576 map (taken by itself) and on small forced shapes; a shape that
    takes another schedule reports prepared = 0 and launches nothing; the second call without the schedule is an error."""
    import ctypes
    lib = pkg._lib.lib()
    side = torch.cuda.Stream()
    for (B, G, C, R, force) in [(2, 63, 576, 40, False), (3, 9, 20, 33, True), (1, 5, 8, 300, True)]:
        if force:
            knobs.setenv("FRCNN_CROP_STAGE", 1)
        else:
            knobs.delenv("FRCNN_CROP_STAGE")
        fm = synth.feature_map(1, B, G, C)
        rng = np.random.default_rng(5 + B)
        rois = np.sort(rng.random((B, R, 2, 2)).astype(np.float32), axis=2).reshape(B, R, 4)
        rois[:, -2:] = 0.0
        rois[0, 0] = [-0.2, 0.3, 1.3, 0.6]
        d_fm, d_rois = dev(fm), dev(rois)
        out = torch.full((B, R, 7, 7, C), float("nan"), device="cuda")
        ws = torch.empty(lib.frcnn_roi_pooling_workspace_bytes(B, R), dtype=torch.uint8, device="cuda")
        prepared = ctypes.c_int(-1)
        torch.cuda.synchronize()
        pkg._lib.check(lib.frcnn_roi_pooling_prepare(d_fm.data_ptr(), B, G, G, C, d_rois.data_ptr(), R, 7, 7, out.data_ptr(),
                                                     ws.data_ptr(), ws.numel(), ctypes.byref(prepared), side.cuda_stream))
        assert prepared.value == 1
        torch.cuda.current_stream().wait_stream(side)
        pkg._lib.check(lib.frcnn_roi_pooling_prepared(d_fm.data_ptr(), B, G, G, C, d_rois.data_ptr(), R, 7, 7, out.data_ptr(),
                                                      ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream))
        assert_bits(out, oracle.roi_pooling(fm, rois, (7, 7)), f"RoIPooling in two calls B={B} G={G} C={C} R={R}")
    # three prepared batches in one launch of the crop kernel
    B, G, C, R = 2, 11, 24, 50
    fms = [synth.feature_map(10 + g, B, G, C) for g in range(3)]
    rs = [np.sort(np.random.default_rng(20 + g).random((B, R, 2, 2)).astype(np.float32), axis=2).reshape(B, R, 4) for g in range(3)]
    d_fms, d_rs = [dev(f) for f in fms], [dev(r) for r in rs]
    outs = [torch.full((B, R, 7, 7, C), float("nan"), device="cuda") for _ in range(3)]
    wss = [torch.empty(lib.frcnn_roi_pooling_workspace_bytes(B, R), dtype=torch.uint8, device="cuda") for _ in range(3)]
    for g in range(3):
        pkg._lib.check(lib.frcnn_roi_pooling_prepare(d_fms[g].data_ptr(), B, G, G, C, d_rs[g].data_ptr(), R, 7, 7, outs[g].data_ptr(),
                                                     wss[g].data_ptr(), wss[g].numel(), ctypes.byref(prepared), side.cuda_stream))
        assert prepared.value == 1
    torch.cuda.current_stream().wait_stream(side)
    tab = pkg._lib.ptr_table
    pkg._lib.check(lib.frcnn_roi_pooling_prepared_multi(tab(d_fms), tab(d_rs), tab(outs), tab(wss), wss[0].numel(), 3, B, G, G, C, R,
                                                        7, 7, torch.cuda.current_stream().cuda_stream))
    for g in range(3):
        assert_bits(outs[g], oracle.roi_pooling(fms[g], rs[g], (7, 7)), f"RoIPooling, three prepared batches in one launch: batch {g}")
    knobs.delenv("FRCNN_CROP_STAGE")
    # a feature map that fits an XCD's L2 takes the box kernel: nothing to prepare
    B, G, C, R = 2, 8, 8, 16
    d_fm, d_rois = dev(synth.feature_map(0, B, G, C)), dev(np.zeros((B, R, 4), np.float32))
    out = torch.empty((B, R, 7, 7, C), device="cuda")
    ws = torch.empty(lib.frcnn_roi_pooling_workspace_bytes(B, R), dtype=torch.uint8, device="cuda")
    prepared = ctypes.c_int(-1)
    pkg._lib.check(lib.frcnn_roi_pooling_prepare(d_fm.data_ptr(), B, G, G, C, d_rois.data_ptr(), R, 7, 7, out.data_ptr(),
                                                 ws.data_ptr(), ws.numel(), ctypes.byref(prepared), 0))
    assert prepared.value == 0
    rc = lib.frcnn_roi_pooling_prepared(d_fm.data_ptr(), B, G, G, C, d_rois.data_ptr(), R, 7, 7, out.data_ptr(), ws.data_ptr(),
                                        ws.numel(), 0)
    assert rc != 0 and b"prepared" in lib.frcnn_last_error()


def test_crop_and_resize_generic(pkg):
    rng = np.random.default_rng(13)
    img = rng.normal(size=(3, 6, 9, 5)).astype(np.float32)
    boxes = np.sort(rng.random((11, 2, 2)).astype(np.float32), axis=1).reshape(11, 4)
    ind = rng.integers(0, 3, 11).astype(np.int32)
    for crop in [(7, 7), (1, 1), (1, 4), (3, 1), (20, 17)]:
        got = pkg.bbox_utils.crop_and_resize(dev(img), dev(boxes), dev(ind), crop, extrapolation_value=-1.5)
        assert_bits(got, oracle.crop_and_resize(img, boxes, ind, crop, -1.5), f"crop {crop}")
    # known answer: 2x2 map, box (0,0,1,1) -> linear ramp i/6
    ramp = np.array([[[[0.0], [1.0]], [[0.0], [1.0]]]], np.float32)
    got = pkg.bbox_utils.crop_and_resize(dev(ramp), dev(np.array([[0, 0, 1, 1]], np.float32)), dev(np.array([0], np.int32)), (7, 7))
    np.testing.assert_allclose(got.cpu().numpy()[0, 3, :, 0], np.arange(7) / 6.0, rtol=RTOL)
    with pytest.raises(ValueError):
        pkg.bbox_utils.crop_and_resize(dev(img), dev(boxes), dev(np.full(11, 3, np.int32)), (7, 7))


@pytest.mark.parametrize("B,R,L", [(2, 16, 5), (16, 300, 21), (3, 300, 21)])
def test_decoder(pkg, B, R, L):
    hp = synth.hyper_params()
    rng = np.random.default_rng(R + L)
    rois = np.sort(rng.random((B, R, 2, 2)).astype(np.float32), axis=2).reshape(B, R, 4)
    rois[:, -3:] = 0.0
    deltas, probs = synth.head_outputs(0, B, R, L)
    layer = pkg.faster_rcnn.Decoder(hp["variances"], L, name="faster_rcnn_decoder")
    got = layer([dev(rois), dev(deltas), dev(probs)])
    want = oracle.decoder(rois, deltas, probs, hp["variances"], L)
    for g, w, name in zip(got, want, ("boxes", "labels", "scores")):
        assert_bits(g, w, f"Decoder {name}")


def test_decoder_unfused_path(pkg, knobs):
    """The generic sort -> NMS -> merge pipeline behind frcnn_decoder (taken when the fused kernel does not fit)."""
    knobs.setenv("FRCNN_DECODER_UNFUSED", "1")
    test_decoder(pkg, 16, 300, 21)
    test_decoder_unnormalised_scores(pkg)


def test_decoder_many_candidates_per_class(pkg):
    """More than 64 candidates in one class (several NMS tiles per class) and more than 1024 candidates per image
    (radix-sort branch of the fused kernel), with heavy suppression."""
    hp = synth.hyper_params()
    B, R, L = 2, 300, 6
    rng = np.random.default_rng(77)
    c = rng.random((B, R, 2)) * 0.5 + 0.25
    hw = rng.uniform(0.1, 0.2, (B, R, 2))
    rois = np.concatenate([c - hw, c + hw], -1).astype(np.float32)
    deltas = rng.normal(0, 0.3, (B, R, L * 4)).astype(np.float32)
    probs = rng.random((B, R, L)).astype(np.float32)  # ~ R*(L-1)/2*... > 1024 candidates above 0.2 per image
    for thr, T in ((0.2, 200), (0.9, 50)):
        layer = pkg.faster_rcnn.Decoder(hp["variances"], L, max_total_size=T, score_threshold=thr)
        got = layer([dev(rois), dev(deltas), dev(probs)])
        want = oracle.decoder(rois, deltas, probs, hp["variances"], L, T, thr)
        for g, w, name in zip(got, want, ("boxes", "labels", "scores")):
            assert_bits(g, w, f"Decoder dense thr={thr} {name}")


def test_decoder_unnormalised_scores(pkg):
    """Un-normalised scores: several classes (incl. class 0) above threshold per RoI, cross-class ties."""
    hp = synth.hyper_params()
    B, R, L = 2, 120, 6
    rng = np.random.default_rng(21)
    rois = np.sort(rng.random((B, R, 2, 2)).astype(np.float32), axis=2).reshape(B, R, 4)
    deltas = rng.normal(0, 0.5, (B, R, L * 4)).astype(np.float32)
    probs = (rng.integers(0, 10, (B, R, L)) / 10.0).astype(np.float32)
    layer = pkg.faster_rcnn.Decoder(hp["variances"], L, max_total_size=50, score_threshold=0.35)
    got = layer([dev(rois), dev(deltas), dev(probs)])
    want = oracle.decoder(rois, deltas, probs, hp["variances"], L, 50, 0.35)
    for g, w, name in zip(got, want, ("boxes", "labels", "scores")):
        assert_bits(g, w, f"Decoder un-normalised {name}")


@pytest.mark.parametrize("unfused", [False, True])
def test_decoder_negative_score_threshold(pkg, knobs, unfused):
    """score_threshold < 0: the rows the reference zeroes (background argmax, models/faster_rcnn.py:83) pass `0 > thr`
    and are candidates with score 0 -- on both schedules (the fused kernel hands such thresholds to the generic path)."""
    if unfused:
        knobs.setenv("FRCNN_DECODER_UNFUSED", "1")
    hp = synth.hyper_params()
    B, R, L = 2, 40, 5
    rng = np.random.default_rng(33)
    rois = np.sort(rng.random((B, R, 2, 2)).astype(np.float32), axis=2).reshape(B, R, 4)
    deltas, probs = synth.head_outputs(3, B, R, L)
    for T in (200, 30):
        layer = pkg.faster_rcnn.Decoder(hp["variances"], L, max_total_size=T, score_threshold=-0.25)
        got = layer([dev(rois), dev(deltas), dev(probs)])
        want = oracle.decoder(rois, deltas, probs, hp["variances"], L, T, -0.25)
        assert ((want[2] > 0) & (want[2] < 0.5)).any()  # candidates the default threshold would have dropped
        for g, w, name in zip(got, want, ("boxes", "labels", "scores")):
            assert_bits(g, w, f"Decoder thr<0 T={T} {name}")


# ------------------------------------------------------------------------------------------------
# RPN targets
# ------------------------------------------------------------------------------------------------
def test_select_mask(pkg):
    rng = np.random.default_rng(17)
    B, N = 5, 3000
    mask = rng.random((B, N)) < 0.3
    mask[1] = False
    mask[2] = True
    rnd = rng.integers(1, 1280, (B, N), dtype=np.int32)
    rnd[3] = rng.integers(-5, 5, N)  # out-of-contract integers (<= 0) still follow the argsort definition
    for sel in ([128], [0, 50, 4000, 7, 128]):
        got = pkg.train_utils.randomly_select_xyz_mask(dev(mask), dev(np.asarray(sel, np.int32)), random_mask=dev(rnd))
        want = oracle.randomly_select_xyz_mask(mask, sel, rnd)
        assert_bits(got, want, f"select_mask {sel}")
    # without injected integers: at most `select` kept, all inside the mask
    got = pkg.train_utils.randomly_select_xyz_mask(dev(mask), dev(np.asarray([128], np.int32))).cpu().numpy()
    assert (got <= mask).all() and (got.sum(1) == np.minimum(mask.sum(1), 128)).all()
    # the in-kernel draw (frcnn_select_mask_seeded: max(select) taken on the device, no host sync): per-image quotas,
    # deterministic in the seed, a different sample for another seed, and nothing selected where tf.random.uniform raises
    sel = np.asarray([0, 50, 4000, 7, 128], np.int32)
    a = pkg.train_utils.randomly_select_xyz_mask(dev(mask), dev(sel), seed=5).cpu().numpy()
    b = pkg.train_utils.randomly_select_xyz_mask(dev(mask), dev(sel), seed=5).cpu().numpy()
    c = pkg.train_utils.randomly_select_xyz_mask(dev(mask), dev(sel), seed=6).cpu().numpy()
    assert (a <= mask).all() and (a.sum(1) == np.minimum(mask.sum(1), sel)).all()
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    none = pkg.train_utils.randomly_select_xyz_mask(dev(mask), dev(np.asarray([0], np.int32)), seed=1).cpu().numpy()
    assert not none.any()


@pytest.mark.parametrize("B,G,M", [(2, 8, 4), (8, 32, 100), (3, 31, 17)])
def test_rpn_targets(pkg, B, G, M):
    hp = pkg.train_utils.get_hyper_params("mobilenet_v2")
    hp.update(feature_map_shape=G)
    anchors = oracle.generate_anchors(hp)
    N = anchors.shape[0]
    gt, labels = synth.gt_boxes(1, B, M)
    rp, rn = synth.sampling_ints(1, B, N)
    got_d, got_l = pkg.train_utils.calculate_rpn_actual_outputs(dev(anchors), dev(gt), dev(labels), hp,
                                                                random_masks=(dev(rp), dev(rn)))
    want_d, want_l = oracle.calculate_rpn_actual_outputs(anchors, gt, labels, hp, (rp, rn))
    assert_bits(got_l, want_l, "rpn labels")
    assert_bits(got_d, want_d, "rpn deltas")
    # the wrapper's extra arguments for feeder loops: caller-provided outputs and an explicit stream
    side = torch.cuda.Stream()
    out = (torch.full((B, N, 4), float("nan"), device="cuda"), torch.full((B, N), float("nan"), device="cuda"))
    d_in = (dev(anchors), dev(gt), dev(labels), dev(rp), dev(rn))
    torch.cuda.synchronize()
    d3, l3 = pkg.train_utils.calculate_rpn_actual_outputs(d_in[0], d_in[1], d_in[2], hp, random_masks=(d_in[3], d_in[4]), out=out,
                                                          stream=side)
    side.synchronize()
    assert d3 is out[0] and l3.data_ptr() == out[1].data_ptr()
    assert_bits(l3, want_l, "rpn labels (out=, stream=)")
    assert_bits(d3, want_d, "rpn deltas (out=, stream=)")
    with pytest.raises(ValueError):
        pkg.train_utils.calculate_rpn_actual_outputs(d_in[0], d_in[1], d_in[2], hp, out=(out[0], out[1][:, :-1]))
    # in-kernel draw: structural properties
    d2, l2 = pkg.train_utils.calculate_rpn_actual_outputs(dev(anchors), dev(gt), dev(labels), hp, seed=5)
    l2 = l2.cpu().numpy().reshape(B, -1)
    assert ((l2 == 1).sum(1) <= hp["total_pos_bboxes"]).all()
    assert ((l2 == 1).sum(1) + (l2 == 0).sum(1) <= hp["total_pos_bboxes"] + hp["total_neg_bboxes"]).all()
    assert set(np.unique(l2)) <= {-1.0, 0.0, 1.0}
    nz = np.abs(d2.cpu().numpy()).sum(-1).reshape(B, -1) != 0
    assert (nz <= (l2 == 1)).all()


@pytest.mark.parametrize("total_pos,total_neg,bad_ints", [(128, 128, False), (4, 6, False), (128, 128, True), (3, 500, True),
                                                         (0, 20, False), (20000, 20000, False)])
def test_rpn_targets_selection_regimes(pkg, total_pos, total_neg, bad_ints):
    """The sampling of calculate_rpn_actual_outputs in every regime of the selection: fewer candidates than the quota (the
    radix selection is skipped), more candidates than the quota, quotas of 0 and beyond N, and out-of-contract random
    integers (<= 0 on masked anchors: they rank among the unmasked ones, so the shortcut must not be taken)."""
    B, G, M = 4, 16, 12
    hp = pkg.train_utils.get_hyper_params("mobilenet_v2")
    hp.update(feature_map_shape=G, total_pos_bboxes=total_pos, total_neg_bboxes=total_neg)
    anchors = oracle.generate_anchors(hp)
    N = anchors.shape[0]
    gt, labels = synth.gt_boxes(9, B, M)
    gt[0, :3] = anchors[[100, 700, 1500]]  # exact anchor hits: IoU 1 positives besides the forced ones
    labels[0, :3] = 1
    rp, rn = synth.sampling_ints(9, B, N, max(total_pos, 1), max(total_neg, 1))
    if bad_ints:
        rng = np.random.default_rng(2)
        rp[rng.random(rp.shape) < 0.3] = 0
        rp[rng.random(rp.shape) < 0.1] = -7
        rn[rng.random(rn.shape) < 0.3] = 0
        rn[rng.random(rn.shape) < 0.1] = -2
    got_d, got_l = pkg.train_utils.calculate_rpn_actual_outputs(dev(anchors), dev(gt), dev(labels), hp,
                                                                random_masks=(dev(rp), dev(rn)))
    want_d, want_l = oracle.calculate_rpn_actual_outputs(anchors, gt, labels, hp, (rp, rn))
    assert_bits(got_l, want_l, "rpn labels")
    assert_bits(got_d, want_d, "rpn deltas")


@pytest.mark.parametrize("int_range", ["ties_1_3", "below_4096", "straddle_4096", "wide", "with_zero_and_negative"])
@pytest.mark.parametrize("total_pos,total_neg", [(16, 40), (128, 128), (700, 900)])
def test_rpn_targets_counting_selection(pkg, int_range, total_pos, total_neg):
    """The selections of rpn_targets_kernel count over the 4096 key values when every key lies in [0, 4096) and fall
    back to the radix passes otherwise: random integers with heavy ties, at the top of the range, straddling it, far
    beyond it, and with zeros / negatives on masked anchors; quotas below, around and above the number of candidates."""
    B, G, M = 3, 16, 20
    hp = pkg.train_utils.get_hyper_params("mobilenet_v2")
    hp.update(feature_map_shape=G, total_pos_bboxes=total_pos, total_neg_bboxes=total_neg)
    anchors = oracle.generate_anchors(hp)
    N = anchors.shape[0]
    gt, labels = synth.gt_boxes(21, B, M)
    gt[:, :6] = anchors[[50, 400, 900, 1300, 1700, 2100]]  # exact anchor hits: positives with IoU 1 and their neighbours
    labels[:, :6] = 2
    rng = np.random.default_rng(5)
    lo, hi = {"ties_1_3": (1, 4), "below_4096": (3000, 4096), "straddle_4096": (4000, 4200), "wide": (1, 2 ** 30),
              "with_zero_and_negative": (-3, 50)}[int_range]
    rp = rng.integers(lo, hi, (B, N)).astype(np.int32)
    rn = rng.integers(lo, hi, (B, N)).astype(np.int32)
    got_d, got_l = pkg.train_utils.calculate_rpn_actual_outputs(dev(anchors), dev(gt), dev(labels), hp,
                                                                random_masks=(dev(rp), dev(rn)))
    want_d, want_l = oracle.calculate_rpn_actual_outputs(anchors, gt, labels, hp, (rp, rn))
    assert (want_l == 1).any() and (want_l == 0).any()
    assert_bits(got_l, want_l, f"rpn labels ({int_range})")
    assert_bits(got_d, want_d, f"rpn deltas ({int_range})")


# ------------------------------------------------------------------------------------------------
# RoIDelta (second-stage training targets)
# ------------------------------------------------------------------------------------------------
def _roi_delta_case(seed, B, R, M, L, G=16):
    """RoIs near the gts (so that positives exist), zero-padded RoI rows, padded gts."""
    rng = np.random.default_rng(seed)
    gt, labels = synth.gt_boxes(seed, B, M)
    labels = np.where(labels >= 0, labels % max(L - 1, 1) + 1, labels).astype(np.int32)
    rois = np.sort(rng.random((B, R, 2, 2)).astype(np.float32), axis=2).reshape(B, R, 4)
    for b in range(B):  # a third of the RoIs are jittered copies of valid gts -> IoU > 0.5 / in (0.1, 0.5)
        valid = np.flatnonzero(labels[b] >= 0)
        if len(valid):
            pick = rng.choice(valid, R // 3)
            jit = rng.normal(0, 0.04, (R // 3, 4)).astype(np.float32)
            rois[b, : R // 3] = np.clip(gt[b, pick] + jit, 0, 1)
    rois[:, -5:] = 0.0  # NMS padding rows: IoU with padded gts is 0/0 = NaN
    return rois, gt, labels


@pytest.mark.parametrize("B,R,M,L,pos,neg", [(2, 64, 4, 5, 8, 8), (4, 1500, 100, 21, 128, 128), (3, 300, 17, 21, 128, 128),
                                             (2, 2000, 8, 21, 0, 3000)])
def test_roi_delta(pkg, B, R, M, L, pos, neg):
    hp = pkg.train_utils.get_hyper_params("mobilenet_v2")
    hp.update(total_labels=L, total_pos_bboxes=pos, total_neg_bboxes=neg)
    rois, gt, labels = _roi_delta_case(R + M, B, R, M, L)
    rng = np.random.default_rng(5)
    rp = rng.integers(1, max(10 * pos, 2), (B, R), dtype=np.int32)
    rn = rng.integers(1, max(10 * neg, 2), (B, R), dtype=np.int32)
    layer = pkg.faster_rcnn.RoIDelta(hp, name="roi_deltas")
    got_d, got_l = layer([dev(rois), dev(gt), dev(labels)], random_masks=(dev(rp), dev(rn)))
    want_d, want_l = oracle.roi_delta(rois, gt, labels, hp, (rp, rn))
    assert want_l.sum() > 0  # the case exercises positives / negatives
    assert_bits(got_l, want_l, "RoIDelta labels")
    assert_bits(got_d, want_d, "RoIDelta deltas")
    # in-kernel draw: structural properties
    d2, l2 = layer([dev(rois), dev(gt), dev(labels)], seed=11)
    l2 = l2.cpu().numpy()
    assert set(np.unique(l2)) <= {0.0, 1.0} and (l2.sum(-1) <= 1).all()
    assert (l2[..., 1:].sum((1, 2)) <= pos).all() and (l2[..., 0].sum(1) <= neg).all()
    nz = np.abs(d2.cpu().numpy()).reshape(B, R, L, 4).sum(-1) != 0
    assert (nz <= (l2 == 1)).all()


def test_roi_delta_errors(pkg):
    hp = pkg.train_utils.get_hyper_params("mobilenet_v2")
    hp.update(total_labels=5)
    layer = pkg.faster_rcnn.RoIDelta(hp)
    rois = dev(np.zeros((2, 8, 4), np.float32))
    with pytest.raises(ValueError):
        layer([rois, dev(np.zeros((2, 0, 4), np.float32)), dev(np.zeros((2, 0), np.int32))])
    with pytest.raises(ValueError):
        layer([rois, dev(np.zeros((3, 4, 4), np.float32)), dev(np.zeros((3, 4), np.int32))])
    assert layer.get_config()["hyper_params"] is hp


# ------------------------------------------------------------------------------------------------
# other callers of the same kernels
# ------------------------------------------------------------------------------------------------
def test_rpn_predictor_call_pattern(pkg):
    """rpn_predictor.py:58-65: reshape, * variances, decode, top_k(scores, 10), gather."""
    B, G = 4, 32
    hp = pkg.train_utils.get_hyper_params("mobilenet_v2")
    anchors = oracle.generate_anchors(hp)
    deltas, scores = synth.rpn_outputs(3, B, G)
    var = torch.tensor(hp["variances"], dtype=torch.float32).cuda()
    d = dev(deltas).reshape(B, -1, 4) * var
    boxes = pkg.bbox_utils.get_bboxes_from_deltas(dev(anchors), d)
    _, top = pkg.bbox_utils.top_k(dev(scores).reshape(B, -1), 10)
    got = torch.gather(boxes, 1, top.long()[..., None].expand(-1, -1, 4))
    wd = (deltas.reshape(B, -1, 4) * np.asarray(hp["variances"], np.float32)).astype(np.float32)
    wboxes = oracle.get_bboxes_from_deltas(anchors, wd)
    _, wtop = oracle.top_k(scores.reshape(B, -1), 10)
    assert_bits(top, wtop, "top-10 indices")
    assert_bits(got, np.take_along_axis(wboxes, wtop[..., None].astype(np.int64), axis=1), "top-10 boxes")


def test_eval_match_predictions_and_update_stats(pkg):
    """utils/eval_utils.py:69-72 (device) and :73-97 (host loop) against the oracle's map + NumPy reductions."""
    B, R, M, L = 3, 200, 7, 6
    rois, gt, labels = _roi_delta_case(9, B, R, M, L)
    merged, arg, order = pkg.eval_utils.match_predictions(dev(rois), dev(gt))
    iou = oracle.generate_iou_map(rois, gt)
    filled = np.where(np.isnan(iou), -np.inf, iou)
    wm = filled.max(-1)
    wm = np.where(np.isinf(wm), np.float32(-3.402823466e38), wm).astype(np.float32)
    assert_bits(merged, wm, "merged_iou_map")
    assert_bits(arg, filled.argmax(-1).astype(np.int32), "max_indices_each_gt")
    _, worder = oracle.top_k(wm, R)
    assert_bits(order, worder, "sorted_ids")
    rng = np.random.default_rng(2)
    pl = rng.integers(0, L, (B, R)).astype(np.int32)
    ps = rng.random((B, R)).astype(np.float32)
    stats = pkg.eval_utils.update_stats(dev(rois), dev(pl), dev(ps), dev(gt), dev(labels),
                                        pkg.eval_utils.init_stats([f"c{i}" for i in range(L)]))
    # restated by hand from the reference's loop
    want = {i: {"total": 0, "tp": [], "fp": []} for i in range(1, L)}
    for v in labels.reshape(-1):
        if v != -1:
            want[int(v)]["total"] += 1
    for b in range(B):
        seen = []
        for r in worder[b]:
            c = int(pl[b, r])
            if c == 0:
                continue
            g = int(filled[b, r].argmax())
            ok = wm[b, r] >= 0.5 and c == int(labels[b, g]) and g not in seen
            want[c]["tp"].append(int(ok))
            want[c]["fp"].append(int(not ok))
            if ok:
                seen.append(g)
    for c in want:
        assert stats[c]["total"] == want[c]["total"] and stats[c]["tp"] == want[c]["tp"] and stats[c]["fp"] == want[c]["fp"]
    assert sum(sum(s["tp"]) for s in stats.values()) > 0


def test_generators_prefetch_targets_on_side_stream(pkg):
    """utils/train_utils.py:91-136: endless feeders; targets of the next batch are computed on a side stream."""
    import itertools

    hp = pkg.train_utils.get_hyper_params("mobilenet_v2", feature_map_shape=16)
    anchors = pkg.bbox_utils.generate_anchors(hp)
    data = []
    for i in range(3):
        gt, lab = synth.gt_boxes(i, 2, 6)
        data.append((torch.full((2, 4), float(i)), dev(gt), dev(lab)))
    gen = pkg.train_utils.faster_rcnn_generator(data, anchors, hp, prefetch=2, seed=100)
    items = list(itertools.islice(gen, 7))  # more than two passes over the 3-batch dataset
    for i, ((img, gt, lab, deltas, labels), empty) in enumerate(items):
        assert empty == () and float(img[0, 0]) == float(i % 3)
        wd, wl = pkg.train_utils.calculate_rpn_actual_outputs(anchors, data[i % 3][1], data[i % 3][2], hp, seed=100 + i)
        assert_bits(deltas, wd.cpu().numpy(), f"generator batch {i} deltas")
        assert_bits(labels, wl.cpu().numpy(), f"generator batch {i} labels")
    img, (deltas, labels) = next(pkg.train_utils.rpn_generator(data, anchors, hp, prefetch=0, seed=100))
    assert_bits(deltas, items[0][0][3].cpu().numpy(), "rpn_generator deltas")
    with pytest.raises(ValueError):
        next(pkg.train_utils.rpn_generator([], anchors, hp))


@pytest.mark.parametrize("prefetch", [0, 1, 2])
@pytest.mark.parametrize("carrier", ["callable", "dataset"])
def test_generators_match_oracle_with_injected_sampling_integers(pkg, prefetch, carrier):
    """utils/train_utils.py:91-136 against the ORACLE: the sampling integers of every batch are supplied (by a per-batch
    callable, or riding with the dataset item), so the feeders' targets are comparable with
    oracle.calculate_rpn_actual_outputs bit for bit -- over more than two passes of the dataset, at prefetch 0 / 1 / 2."""
    import itertools

    G, B, M = 16, 3, 9
    hp = pkg.train_utils.get_hyper_params("mobilenet_v2", feature_map_shape=G)
    anchors = oracle.generate_anchors(hp)
    N = anchors.shape[0]
    raw = [synth.gt_boxes(40 + i, B, M) for i in range(3)]
    ints = {i: synth.sampling_ints(500 + i, B, N) for i in range(12)}  # a different draw for every batch, passes included
    if carrier == "callable":
        data = [(torch.full((B, 2), float(i)), dev(gt), dev(lab)) for i, (gt, lab) in enumerate(raw)]
        kw = dict(random_masks=lambda i, b, n: (assert_eq((b, n), (B, N)), ints[i])[1])
    else:
        class Passes:  # a re-iterable whose items carry their integers (host arrays: the feeder moves them to the device)
            def __init__(self):
                self.count = 0

            def __iter__(self):
                for i, (gt, lab) in enumerate(raw):
                    rp, rn = ints[self.count]
                    self.count += 1
                    yield torch.full((B, 2), float(i)), gt, lab, rp, rn
        data, kw = Passes(), {}
    gen = pkg.train_utils.faster_rcnn_generator(data, dev(anchors), hp, prefetch=prefetch, **kw)
    torch.cuda._sleep(3_000_000)  # the consumer's stream is busy: hand-over must still order the targets before their use
    for i, ((img, gt, lab, deltas, labels), empty) in enumerate(itertools.islice(gen, 7)):
        assert empty == () and float(img[0, 0]) == float(i % 3)
        want_d, want_l = oracle.calculate_rpn_actual_outputs(anchors, raw[i % 3][0], raw[i % 3][1], hp, ints[i])
        assert_bits(labels, want_l, f"{carrier} prefetch={prefetch} batch {i} labels")
        assert_bits(deltas, want_d, f"{carrier} prefetch={prefetch} batch {i} deltas")
        assert_bits(gt, raw[i % 3][0], "gt boxes handed through")
    if carrier == "callable":
        img, (deltas, labels) = next(pkg.train_utils.rpn_generator(data, dev(anchors), hp, prefetch=prefetch, **kw))
        want_d, want_l = oracle.calculate_rpn_actual_outputs(anchors, raw[0][0], raw[0][1], hp, ints[0])
        assert_bits(labels, want_l, "rpn_generator labels")
        assert_bits(deltas, want_d, "rpn_generator deltas")


def assert_eq(a, b):
    assert a == b, (a, b)


def test_config1_batch1_full_path(pkg):
    """BASELINE config 1 shape: mobilenet_v2, 500x500, batch 1 -- RoIBBox -> RoIPooling -> Decoder against the oracle."""
    hp = pkg.train_utils.get_hyper_params("mobilenet_v2")
    hp["total_labels"] = 21
    anchors = oracle.generate_anchors(hp)
    deltas, scores = synth.rpn_outputs(21, 1, 32)
    fm = synth.feature_map(21, 1, 32, 576)
    hd, hpb = synth.head_outputs(21, 1, 300, 21)
    rois = pkg.faster_rcnn.RoIBBox(dev(anchors), "inference", hp)([dev(deltas), dev(scores)])
    pooled = pkg.faster_rcnn.RoIPooling(hp)([dev(fm), rois])
    dets = pkg.faster_rcnn.Decoder(hp["variances"], 21)([rois, dev(hd), dev(hpb)])
    wrois = oracle.roi_bbox(deltas, scores, anchors, hp["variances"], 6000, 300, 0.7)
    assert_bits(rois, wrois, "cfg1 rois")
    assert_bits(pooled, oracle.roi_pooling(fm, wrois, (7, 7)), "cfg1 pooled")
    for g, w, name in zip(dets, oracle.decoder(wrois, hd, hpb, hp["variances"], 21), ("boxes", "labels", "scores")):
        assert_bits(g, w, f"cfg1 Decoder {name}")


# ------------------------------------------------------------------------------------------------
# committed golden vectors (tests/golden/*.npz, written by tests/golden/make_golden.py): the HIP path on the committed
# inputs against the committed outputs -- no oracle call in between
# ------------------------------------------------------------------------------------------------
def test_golden_vectors_through_the_hip_path(pkg):
    import os

    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g = np.load(os.path.join(golden, "inference_path.npz"))
    hp = synth.hyper_params(feature_map_shape=8, pre_nms_topn=64, test_nms_topn=16)
    assert_bits(pkg.bbox_utils.generate_anchors(hp), g["anchors"], "golden anchors")
    rois = pkg.faster_rcnn.RoIBBox(dev(g["anchors"]), "inference", hp)([dev(g["deltas"]), dev(g["scores"])])
    assert_bits(rois, g["rois"], "golden RoIBBox")
    pooled = pkg.faster_rcnn.RoIPooling(hp)([dev(g["fm"]), rois])
    assert_bits(pooled, g["pooled"], "golden RoIPooling")
    boxes, labels, scores = pkg.faster_rcnn.Decoder(hp["variances"], 5)([rois, dev(g["head_deltas"]), dev(g["head_probs"])])
    assert_bits(boxes, g["det_boxes"], "golden Decoder boxes")
    assert_bits(labels, g["det_labels"], "golden Decoder labels")
    assert_bits(scores, g["det_scores"], "golden Decoder scores")

    t = np.load(os.path.join(golden, "rpn_targets.npz"))
    hp_t = synth.hyper_params(feature_map_shape=8, total_pos_bboxes=8, total_neg_bboxes=8)
    assert_bits(pkg.bbox_utils.generate_iou_map(dev(t["anchors"]), dev(t["gt_boxes"])), t["iou_map"], "golden iou_map")
    d, l = pkg.train_utils.calculate_rpn_actual_outputs(dev(t["anchors"]), dev(t["gt_boxes"]), dev(t["gt_labels"]), hp_t,
                                                        random_masks=(dev(t["rnd_pos"]), dev(t["rnd_neg"])))
    assert_bits(d, t["bbox_deltas"], "golden rpn deltas")
    assert_bits(l, t["bbox_labels"], "golden rpn labels")

    r = np.load(os.path.join(golden, "roi_delta.npz"))
    hp_r = synth.hyper_params(total_labels=5, total_pos_bboxes=4, total_neg_bboxes=4)
    rd, rl = pkg.faster_rcnn.RoIDelta(hp_r)([dev(r["rois"]), dev(r["gt_boxes"]), dev(r["gt_labels"])],
                                            random_masks=(dev(r["rnd_pos"]), dev(r["rnd_neg"])))
    assert_bits(rd, r["roi_bbox_deltas"], "golden RoIDelta deltas")
    assert_bits(rl, r["roi_bbox_labels"], "golden RoIDelta labels")


# ------------------------------------------------------------------------------------------------
# stage pipeline over streams
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("depth,heads_callable,crop_group", [(1, False, 1), (3, False, 1), (4, True, 1), (4, False, 2),
                                                             (3, True, 3), (4, True, 4)])
def test_pipeline_matches_oracle_and_sequential(pkg, depth, heads_callable, crop_group):
    """Batches in flight on three streams give, per batch, exactly the sequential layers' (= the oracle's) results --
    also when the crops of `crop_group` batches share one launch (7 batches: the last group is a partial one)."""
    from tf_faster_rcnn_amd import pipeline

    B, G, C, L = 4, 16, 32, 6
    hp = pkg.train_utils.get_hyper_params("mobilenet_v2")
    hp.update(feature_map_shape=G, pre_nms_topn=1500, test_nms_topn=64, total_labels=L)
    anchors = oracle.generate_anchors(hp)
    roi_bbox = pkg.faster_rcnn.RoIBBox(dev(anchors), "inference", hp)
    roi_pool = pkg.faster_rcnn.RoIPooling(hp)
    decoder = pkg.faster_rcnn.Decoder(hp["variances"], L)
    pipe = pipeline.RoIPathPipeline(roi_bbox, roi_pool, decoder, depth=depth, crop_group=crop_group,
                                    host_outputs=heads_callable)  # the pinned host copies ride along on half of the cases
    batches = []
    for seed in range(7):
        deltas, scores = synth.rpn_outputs(seed, B, G)
        fm = synth.feature_map(seed, B, G, C)
        hd, hpb = synth.head_outputs(seed, B, 64, L)
        batches.append((deltas, scores, fm, hd, hpb))
    dev_batches = [tuple(dev(a) for a in b) for b in batches]
    got = {}
    for i, b in enumerate(dev_batches):
        heads = (lambda pooled, rois, b=b: (b[3] + 0.0, b[4] + 0.0)) if heads_callable else (b[3], b[4])
        t = pipe.submit(b[0], b[1], b[2], heads)
        assert t == i
        if i >= depth - 1:  # consume the oldest batch still in flight
            j = i - (depth - 1)
            res = pipe.result(j)
            got[j] = {k: v.clone() for k, v in res.items() if k != "host"}
            if heads_callable:  # host_outputs: the pinned copies are complete once result(wait="host") returns
                for name, h in res["host"].items():
                    assert_bits(h, res[name].cpu().numpy(), f"pipeline batch {j} host copy of {name}")
    for j in range(len(batches) - (depth - 1), len(batches)):
        got[j] = {k: v.clone() for k, v in pipe.result(j, wait="stream").items() if k != "host"}
    pipe.synchronize()
    with pytest.raises(ValueError):
        pipe.result(len(batches))
    for i, b in enumerate(batches):
        rois = oracle.roi_bbox(b[0], b[1], anchors, hp["variances"], 1500, 64, hp["nms_iou_threshold"])
        pooled = oracle.roi_pooling(b[2], rois, hp["pooling_size"])
        boxes, labels, sc = oracle.decoder(rois, b[3], b[4], hp["variances"], L)
        seq = pipeline.run_sequential(roi_bbox, roi_pool, decoder, dev_batches[i])
        for name, want in (("roi_bboxes", rois), ("roi_pooled", pooled), ("bboxes", boxes), ("labels", labels), ("scores", sc)):
            assert_bits(got[i][name], want, f"pipeline batch {i} {name} vs oracle")
            assert_bits(got[i][name], seq[name].cpu().numpy(), f"pipeline batch {i} {name} vs sequential")


@pytest.mark.parametrize("depth,crop_group", [(1, 1), (4, 1), (4, 2), (4, 4), (3, 3)])
def test_pipeline_with_prepared_roi_pooling(pkg, knobs, depth, crop_group):
    """The stage pipeline when RoIPooling takes the source-row schedule (forced here on a small feature map; taken by itself
    beyond 4 MiB per image): the pre-pass rides on the proposal stream into a per-slot workspace, the pooling stream runs ONE
    grouped crop launch (frcnn_roi_pooling_prepared_multi).  Same bits as the oracle, inference and training order."""
    knobs.setenv("FRCNN_CROP_STAGE", 1)
    test_pipeline_matches_oracle_and_sequential(pkg, depth, depth % 2 == 0, crop_group)
    test_train_pipeline_matches_oracle_and_sequential(pkg, depth, crop_group, 96, depth % 2 == 1)


@pytest.mark.parametrize("depth,crop_group,R,with_heads", [(1, 1, 96, False), (3, 1, 600, True), (4, 2, 96, True), (3, 3, 600, False)])
def test_train_pipeline_matches_oracle_and_sequential(pkg, depth, crop_group, R, with_heads):
    """Train mode of the stage pipeline (models/faster_rcnn.py:327-335: RoIBBox -> RoIPooling, RoIDelta beside the crop):
    per batch exactly the sequential layers' and the oracle's results, with the sampling integers injected -- on the fused
    proposal kernel (R = 96) and on the whole-chip bitmask schedule (R = 600 >= 512), with and without grouped launches."""
    from tf_faster_rcnn_amd import pipeline

    B, G, C, L, M = 3, 16, 32, 6, 9
    hp = pkg.train_utils.get_hyper_params("vgg16")
    hp.update(feature_map_shape=G, pre_nms_topn=1500, train_nms_topn=R, total_labels=L, total_pos_bboxes=16, total_neg_bboxes=24)
    anchors = oracle.generate_anchors(hp)
    roi_bbox = pkg.faster_rcnn.RoIBBox(dev(anchors), "training", hp)
    roi_pool = pkg.faster_rcnn.RoIPooling(hp)
    roi_delta = pkg.faster_rcnn.RoIDelta(hp)
    seen = []
    pipe = pipeline.RoIPathPipeline(roi_bbox, roi_pool, depth=depth, crop_group=crop_group, roi_delta=roi_delta, host_outputs=True,
                                    gather=lambda t, outs, q: seen.append((t, sorted(outs))))
    with pytest.raises(ValueError):
        pipeline.RoIPathPipeline(roi_bbox, roi_pool)  # neither decoder nor roi_delta
    rng = np.random.default_rng(5)
    batches, masks = [], []
    for seed in range(5):
        deltas, scores = synth.rpn_outputs(seed, B, G)
        fm = synth.feature_map(seed, B, G, C)
        gt, lab = synth.gt_boxes(seed, B, M, L)
        batches.append((deltas, scores, fm, gt, lab))
        masks.append((rng.integers(1, 160, (B, R)).astype(np.int32), rng.integers(1, 240, (B, R)).astype(np.int32)))
    dev_batches = [tuple(dev(a) for a in b) for b in batches]
    dev_masks = [tuple(dev(a) for a in m) for m in masks]
    calls = []
    heads = (lambda pooled, rois: calls.append(tuple(pooled.shape))) if with_heads else None
    with pytest.raises(ValueError):
        pipe.submit(*dev_batches[0][:3])  # train mode needs gt
    got = {}
    for i, b in enumerate(dev_batches):
        assert pipe.submit(b[0], b[1], b[2], heads, gt=(b[3], b[4]), random_masks=dev_masks[i]) == i
        if i >= depth - 1:
            j = i - (depth - 1)
            res = pipe.result(j)
            got[j] = {k: v.clone() for k, v in res.items() if k != "host"}
            assert_bits(res["host"]["roi_bboxes"], res["roi_bboxes"].cpu().numpy(), f"train pipeline batch {j} host copy of the RoIs")
    for j in range(len(batches) - (depth - 1), len(batches)):
        got[j] = {k: v.clone() for k, v in pipe.result(j, wait="stream").items() if k != "host"}
    pipe.synchronize()
    assert [t for t, _ in seen] == list(range(len(batches))) and all(names == ["roi_bboxes"] for _, names in seen)
    assert len(calls) == (len(batches) if with_heads else 0)
    for i, b in enumerate(batches):
        rois = oracle.roi_bbox(b[0], b[1], anchors, hp["variances"], 1500, R, hp["nms_iou_threshold"])
        pooled = oracle.roi_pooling(b[2], rois, hp["pooling_size"])
        rd, rl = oracle.roi_delta(rois, b[3], b[4], hp, masks[i])
        seq = pipeline.run_sequential_train(roi_bbox, roi_pool, roi_delta, dev_batches[i], random_masks=dev_masks[i])
        for name, want in (("roi_bboxes", rois), ("roi_pooled", pooled), ("roi_bbox_deltas", rd), ("roi_bbox_labels", rl)):
            assert_bits(got[i][name], want, f"train pipeline batch {i} {name} vs oracle")
            assert_bits(got[i][name], seq[name].cpu().numpy(), f"train pipeline batch {i} {name} vs sequential")


@pytest.mark.parametrize("nbytes", [16, 4096 + 7, 154_000, 3])
def test_gather_copy_into_pinned_host_memory(pkg, nbytes):
    """frcnn_gather_copy: a kernel that copies device memory into device-mapped pinned host memory (and device to device),
    any byte count."""
    from tf_faster_rcnn_amd import _lib

    rng = np.random.default_rng(nbytes)
    data = rng.integers(0, 256, nbytes, dtype=np.uint8)
    src = torch.from_numpy(data).cuda()
    host = torch.zeros(nbytes + 16, dtype=torch.uint8, pin_memory=True)
    devdst = torch.zeros(nbytes + 16, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    _lib.check(_lib.lib().frcnn_gather_copy(src.data_ptr(), host.data_ptr(), nbytes, stream))
    _lib.check(_lib.lib().frcnn_gather_copy(src.data_ptr(), devdst.data_ptr(), nbytes, stream))
    torch.cuda.synchronize()
    assert (host.numpy()[:nbytes] == data).all() and not host.numpy()[nbytes:].any()
    assert (devdst.cpu().numpy()[:nbytes] == data).all() and not devdst.cpu().numpy()[nbytes:].any()


def test_pipeline_grouped_launches_at_config2_size(pkg):
    """bench.py's default schedule (2 batches per launch of each stage, 8 in flight) at the full BASELINE config 2 size:
    every output of every batch is bit-identical to the three layers called in order (whose results the other tests pin
    to the oracle at this size)."""
    from tf_faster_rcnn_amd import pipeline

    B, G, C, L, R = 16, 32, 576, 21, 300
    hp = pkg.train_utils.get_hyper_params("mobilenet_v2")
    hp["total_labels"] = L
    anchors = pkg.bbox_utils.generate_anchors(hp)
    roi_bbox = pkg.faster_rcnn.RoIBBox(anchors, "inference", hp)
    roi_pool = pkg.faster_rcnn.RoIPooling(hp)
    decoder = pkg.faster_rcnn.Decoder(hp["variances"], L)
    pipe = pipeline.RoIPathPipeline(roi_bbox, roi_pool, decoder, depth=8, crop_group=2)
    batches = []
    for seed in range(3):
        deltas, scores = synth.rpn_outputs(70 + seed, B, G)
        hd, hpb = synth.head_outputs(70 + seed, B, R, L)
        batches.append(tuple(dev(a) for a in (deltas, scores, synth.feature_map(70 + seed, B, G, C), hd, hpb)))
    for b in batches:
        pipe.submit(b[0], b[1], b[2], (b[3], b[4]))
    pipe.synchronize()
    for i, b in enumerate(batches):
        got = pipe.result(i)
        seq = pipeline.run_sequential(roi_bbox, roi_pool, decoder, b)
        for name in ("roi_bboxes", "roi_pooled", "bboxes", "labels", "scores"):
            assert torch.equal(got[name].view(torch.int32), seq[name].view(torch.int32)), f"config 2 grouped batch {i} {name}"
        assert int((got["roi_bboxes"].abs().sum(-1) > 0).sum()) > B * R // 2  # real proposals, not padding


def _small_pipeline(pkg, depth=2, crop_group=1, B=2, G=12, C=16, L=4, R=24):
    from tf_faster_rcnn_amd import pipeline

    hp = pkg.train_utils.get_hyper_params("mobilenet_v2")
    hp.update(feature_map_shape=G, pre_nms_topn=600, test_nms_topn=R, total_labels=L)
    anchors = oracle.generate_anchors(hp)
    layers = (pkg.faster_rcnn.RoIBBox(dev(anchors), "inference", hp), pkg.faster_rcnn.RoIPooling(hp),
              pkg.faster_rcnn.Decoder(hp["variances"], L))
    pipe = pipeline.RoIPathPipeline(*layers, depth=depth, crop_group=crop_group)
    batches = []
    for seed in range(4):
        deltas, scores = synth.rpn_outputs(90 + seed, B, G)
        hd, hpb = synth.head_outputs(90 + seed, B, R, L)
        batches.append(tuple(dev(a) for a in (deltas, scores, synth.feature_map(90 + seed, B, G, C), hd, hpb)))
    return pipeline, layers, pipe, batches


@pytest.mark.parametrize("callable_heads", [False, True])
@pytest.mark.parametrize("crop_group", [1, 2])
def test_pipeline_head_outputs_that_need_a_conversion(pkg, callable_heads, crop_group):
    """Head outputs that are not contiguous fp32 device tensors (fp64, a transposed view, host memory): the copy that
    normalises them is ordered before the Decoder stage on whichever stream it runs (tuple heads: the caller's stream,
    before the input event; callable heads: the pooling stream).  Results as the sequential layers on the converted
    values.  A large spin kernel in front makes a missing ordering visible."""
    pipeline, layers, pipe, batches = _small_pipeline(pkg, depth=4, crop_group=crop_group)
    L = 4
    for i, b in enumerate(batches):
        hd, hp_ = b[3], b[4]
        awkward = [
            (hd.double(), hp_.double()),                                                      # dtype conversion
            (hd.reshape(hd.shape[0], hd.shape[1], L, 4).permute(0, 1, 3, 2).contiguous().permute(0, 1, 3, 2), # non-contiguous view
             hp_.transpose(1, 2).contiguous().transpose(1, 2)),
            (hd.cpu(), hp_.cpu()),                                                            # host tensors
            (hd.half().float().half(), hp_),                                                  # fp16 (values change: compared below)
        ][i]
        torch.cuda.synchronize()  # the awkward tensors exist; what is under test is the ordering of their CONVERSION
        if callable_heads:
            heads = lambda pooled, rois, a=awkward: (torch.cuda._sleep(2_000_000), a)[1]  # noqa: E731
        else:
            torch.cuda._sleep(2_000_000)
            heads = awkward
        pipe.submit(b[0], b[1], b[2], heads)
    pipe.synchronize()
    for i, b in enumerate(batches):
        got = pipe.result(i)
        hd = b[3].half().float() if i == 3 else b[3]
        seq = pipeline.run_sequential(*layers, (b[0], b[1], b[2], hd, b[4]))
        for name in ("roi_bboxes", "roi_pooled", "bboxes", "labels", "scores"):
            assert torch.equal(got[name].view(torch.int32), seq[name].view(torch.int32)), f"batch {i} {name}"


def test_pipeline_failed_group_is_reported_not_returned_stale(pkg):
    """A head callable that returns the wrong shape: flush raises, result() of that batch raises too (never the slot's
    previous contents), and the pipeline keeps working for later batches."""
    pipeline, layers, pipe, batches = _small_pipeline(pkg, depth=2, crop_group=1)
    b = batches[0]
    t0 = pipe.submit(b[0], b[1], b[2], (b[3], b[4]))
    pipe.result(t0)
    t1 = pipe.submit(b[0], b[1], b[2], (b[3], b[4]))
    with pytest.raises(ValueError):
        pipe.submit(b[0], b[1], b[2], lambda pooled, rois: (b[3][:, :5], b[4]))  # ticket 2 reuses slot 0
    with pytest.raises(RuntimeError):
        pipe.result(2)
    pipe.result(t1)
    with pytest.raises(ValueError):  # tuple heads are checked before anything is touched: no ticket is consumed
        pipe.submit(b[0], b[1], b[2], (b[3][:, :5], b[4]))
    c = batches[1]
    t3 = pipe.submit(c[0], c[1], c[2], (c[3], c[4]))
    assert t3 == 3
    got = pipe.result(t3)
    seq = pipeline.run_sequential(*layers, c)
    for name in ("roi_bboxes", "roi_pooled", "bboxes", "labels", "scores"):
        assert torch.equal(got[name].view(torch.int32), seq[name].view(torch.int32)), name


@pytest.mark.parametrize("schedule", ["unfused", "bitmask"])
def test_pipeline_groups_on_schedules_without_a_grouped_form(pkg, knobs, schedule):
    """crop_group > 1 where frcnn_roi_bbox_multi / frcnn_decoder_multi have no grouped launch (the generic sort -> NMS
    pipelines, the whole-chip bitmask NMS): they run one call per batch on the shared workspace; results as sequential."""
    from tf_faster_rcnn_amd import pipeline

    B, G, C, L = 2, 12, 16, 4
    R = 24 if schedule == "unfused" else 600
    if schedule == "unfused":
        knobs.setenv("FRCNN_ROI_BBOX_UNFUSED", "1")
        knobs.setenv("FRCNN_DECODER_UNFUSED", "1")
    hp = pkg.train_utils.get_hyper_params("mobilenet_v2")
    hp.update(feature_map_shape=G, pre_nms_topn=1000, test_nms_topn=R, total_labels=L)
    anchors = oracle.generate_anchors(hp)
    roi_bbox = pkg.faster_rcnn.RoIBBox(dev(anchors), "inference", hp)
    roi_pool = pkg.faster_rcnn.RoIPooling(hp)
    decoder = pkg.faster_rcnn.Decoder(hp["variances"], L)
    pipe = pipeline.RoIPathPipeline(roi_bbox, roi_pool, decoder, depth=4, crop_group=2)
    batches = []
    for seed in range(3):
        deltas, scores = synth.rpn_outputs(40 + seed, B, G)
        hd, hpb = synth.head_outputs(40 + seed, B, R, L)
        batches.append(tuple(dev(a) for a in (deltas, scores, synth.feature_map(40 + seed, B, G, C), hd, hpb)))
    for b in batches:
        pipe.submit(b[0], b[1], b[2], (b[3], b[4]))
    got = [{k: v.clone() for k, v in pipe.result(i).items()} for i in range(3)]
    pipe.synchronize()
    for i, b in enumerate(batches):
        seq = pipeline.run_sequential(roi_bbox, roi_pool, decoder, b)
        for name in ("roi_bboxes", "roi_pooled", "bboxes", "labels", "scores"):
            assert_bits(got[i][name], seq[name].cpu().numpy(), f"grouped pipeline ({schedule}) batch {i} {name}")
        want = oracle.roi_bbox(b[0].cpu().numpy(), b[1].cpu().numpy(), anchors, hp["variances"], 1000, R, hp["nms_iou_threshold"])
        assert_bits(got[i]["roi_bboxes"], want, f"grouped pipeline ({schedule}) batch {i} rois vs oracle")


@pytest.mark.parametrize("groups,C,H", [(1, 32, 9), (2, 576, 12), (3, 20, 7), (4, 64, 5), (2, 6, 8)])
def test_crop_and_resize_multi_matches_oracle(pkg, groups, C, H):
    """frcnn_crop_and_resize_multi through the C ABI: `groups` (image, boxes, out) triples in one call, each bit-identical
    to the oracle's roi_pooling of that triple (C % 4 != 0 takes the one-launch-per-triple route)."""
    import ctypes

    from tf_faster_rcnn_amd import _lib

    rng = np.random.default_rng(100 + groups)
    B, R, W = 3, 11, H + 2
    fms = [rng.normal(0, 1, (B, H, W, C)).astype(np.float32) for _ in range(groups)]
    rois = [rng.uniform(-0.1, 1.1, (B, R, 4)).astype(np.float32) for _ in range(groups)]
    rois[0][0, 0] = 0.0  # a zero-padded RoI
    d_fm, d_rois = [dev(a) for a in fms], [dev(a) for a in rois]
    outs = [torch.full((B, R, 7, 7, C), float("nan"), dtype=torch.float32, device="cuda") for _ in range(groups)]
    arr = ctypes.c_void_p * groups
    stream = torch.cuda.current_stream().cuda_stream
    _lib.check(_lib.lib().frcnn_crop_and_resize_multi(arr(*[t.data_ptr() for t in d_fm]), arr(*[t.data_ptr() for t in d_rois]),
                                                     arr(*[t.data_ptr() for t in outs]), groups, B, H, W, C, B * R, R, 7, 7,
                                                     0.0, None, 0, stream))
    torch.cuda.synchronize()
    for g in range(groups):
        assert_bits(outs[g], oracle.roi_pooling(fms[g], rois[g], (7, 7)), f"crop multi triple {g} of {groups} C={C}")
    with pytest.raises(ValueError):
        _lib.check(_lib.lib().frcnn_crop_and_resize_multi(arr(*[t.data_ptr() for t in d_fm]), arr(*[t.data_ptr() for t in d_rois]),
                                                         arr(*[t.data_ptr() for t in outs]), 5, B, H, W, C, B * R, R, 7, 7,
                                                         0.0, None, 0, stream))


@pytest.mark.parametrize("groups,n,bpi,pf", [(4, 10, 4, 3), (3, 10, 4, 40005), (2, 37, 5, 7), (4, 97, 16, 10002)])
def test_crop_and_resize_multi_ragged_triples_with_warmup(pkg, knobs, groups, n, bpi, pf):
    """Grouped launch whose triples do not end on an image boundary (n % boxes_per_image != 0), warm-up workgroups forced
    (FRCNN_CROP_PREFETCH = lead * 10000 + workgroups per image): an XCD's box range then spans more (partial) images than
    whole ones, and every box must still get its slot.  Bit-exact against the oracle's crop_and_resize."""
    import ctypes

    from tf_faster_rcnn_amd import _lib

    rng = np.random.default_rng(7 * groups + n)
    B, H, W, C = (n + bpi - 1) // bpi, 6, 9, 8
    fms = [rng.normal(0, 1, (B, H, W, C)).astype(np.float32) for _ in range(groups)]
    boxes = [rng.uniform(-0.1, 1.1, (n, 4)).astype(np.float32) for _ in range(groups)]
    ind = (np.arange(n) // bpi).astype(np.int32)
    d_fm, d_boxes = [dev(a) for a in fms], [dev(a) for a in boxes]
    outs = [torch.full((n, 7, 7, C), float("nan"), dtype=torch.float32, device="cuda") for _ in range(groups)]
    arr = ctypes.c_void_p * groups
    knobs.setenv("FRCNN_CROP_PREFETCH", pf)
    _lib.check(_lib.lib().frcnn_crop_and_resize_multi(arr(*[t.data_ptr() for t in d_fm]), arr(*[t.data_ptr() for t in d_boxes]),
                                                     arr(*[t.data_ptr() for t in outs]), groups, B, H, W, C, n, bpi, 7, 7, 0.0,
                                                     None, 0, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    for g in range(groups):
        assert_bits(outs[g], oracle.crop_and_resize(fms[g], boxes[g], ind, (7, 7)), f"ragged crop multi triple {g} of {groups}")


# ------------------------------------------------------------------------------------------------
# full BASELINE sizes: size-independent properties (the oracle is too slow there)
# ------------------------------------------------------------------------------------------------
def _roi_bbox_raw(pkg, deltas, scores, anchors, hp, K, R):
    """frcnn_roi_bbox through the C ABI with scores and valid counts (the layer returns boxes only)."""
    lib = pkg._lib.lib()
    B, N = scores.shape[0], anchors.shape[0]
    out = torch.empty((B, R, 4), dtype=torch.float32, device="cuda")
    sc = torch.empty((B, R), dtype=torch.float32, device="cuda")
    valid = torch.empty((B,), dtype=torch.int32, device="cuda")
    ws = torch.empty(lib.frcnn_roi_bbox_workspace_bytes(B, N, K, R), dtype=torch.uint8, device="cuda")
    pkg._lib.check(lib.frcnn_roi_bbox(deltas.data_ptr(), scores.data_ptr(), anchors.data_ptr(), pkg._lib.float4(hp["variances"]),
                                      B, N, K, R, float(hp["nms_iou_threshold"]), out.data_ptr(), sc.data_ptr(), valid.data_ptr(),
                                      ws.data_ptr(), ws.numel(), None))
    torch.cuda.synchronize()
    return out, sc, valid


def _check_nms_properties(boxes, sc, valid, thr):
    """Rows are score-ordered and clipped, padding is zero, and no two kept boxes overlap above the threshold."""
    B, R, _ = boxes.shape
    idx = torch.arange(R, device=boxes.device)[None, :]
    live = idx < valid[:, None]
    assert (boxes[~live] == 0).all() and (sc[~live] == 0).all()
    assert ((boxes >= 0) & (boxes <= 1)).all()
    d = sc[:, 1:] - sc[:, :-1]
    assert (d[live[:, 1:]] <= 0).all(), "scores must be non-increasing over the valid rows"
    y1, x1, y2, x2 = boxes.unbind(-1)
    area = (y2 - y1) * (x2 - x1)
    ih = (torch.minimum(y2[:, :, None], y2[:, None, :]) - torch.maximum(y1[:, :, None], y1[:, None, :])).clamp(min=0)
    iw = (torch.minimum(x2[:, :, None], x2[:, None, :]) - torch.maximum(x1[:, :, None], x1[:, None, :])).clamp(min=0)
    inter = ih * iw
    iou = inter / (area[:, :, None] + area[:, None, :] - inter)
    pair = live[:, :, None] & live[:, None, :] & (idx[:, :, None] < idx[:, None, :]) & (area[:, :, None] > 0) & (area[:, None, :] > 0)
    # the kernel tests the UNCLIPPED boxes; clipping can only change the IoU of boxes that left the image, so allow a margin
    inside = ((boxes > 0) & (boxes < 1)).all(-1)
    strict = pair & inside[:, :, None] & inside[:, None, :]
    assert not (iou[strict] > thr * (1 + 1e-5)).any(), "two kept boxes overlap above the NMS threshold"


def test_full_size_config3_shard_properties(pkg):
    """BASELINE config 3 per-GPU shard: vgg16 grid (8649 anchors, 512 channels), B=16, train mode, post_nms 2000."""
    hp = pkg.train_utils.get_hyper_params("vgg16", train_nms_topn=2000)
    anchors = pkg.bbox_utils.generate_anchors(hp)
    deltas, scores = synth.rpn_outputs(31, 16, 31)
    boxes, sc, valid = _roi_bbox_raw(pkg, dev(deltas).reshape(16, -1, 4), dev(scores).reshape(16, -1), anchors, hp, 6000, 2000)
    assert (valid > 0).all()
    _check_nms_properties(boxes, sc, valid, hp["nms_iou_threshold"])
    layer_out = pkg.faster_rcnn.RoIBBox(anchors, "training", hp)([dev(deltas), dev(scores)])
    assert torch.equal(layer_out, boxes)
    # crop_and_resize is linear in the feature map: scaling by 2 is exact, and channels are independent
    fm = dev(synth.feature_map(31, 16, 31, 512))
    pool = pkg.faster_rcnn.RoIPooling(hp)
    pooled = pool([fm, boxes])
    assert pooled.shape == (16, 2000, 7, 7, 512)
    assert torch.equal(pool([fm * 2, boxes]), pooled * 2)
    perm = torch.randperm(512, device="cuda")
    assert torch.equal(pool([fm[..., perm].contiguous(), boxes]), pooled[..., perm])
    # a zero-padded RoI samples pixel (0,0) everywhere; a whole-image RoI returns the corners exactly
    probe = boxes.clone()
    probe[:, 0] = torch.tensor([0.0, 0.0, 1.0, 1.0], device="cuda")
    probe[:, 1] = 0.0
    p2 = pool([fm, probe])
    assert torch.equal(p2[:, 0, 0, 0], fm[:, 0, 0]) and torch.equal(p2[:, 0, 6, 6], fm[:, 30, 30])
    assert torch.equal(p2[:, 1], fm[:, 0, 0][:, None, None, :].expand(-1, 7, 7, -1))


def test_full_size_config4_properties(pkg):
    """BASELINE config 4: RPN targets, B=64, 9216 anchors x 100 gt boxes."""
    hp = pkg.train_utils.get_hyper_params("mobilenet_v2")
    anchors = pkg.bbox_utils.generate_anchors(hp)
    N = anchors.shape[0]
    gt, labels = synth.gt_boxes(41, 64, 100)
    gt_t, lab_t = dev(gt), dev(labels)
    iou = pkg.bbox_utils.generate_iou_map(anchors, gt_t)
    # the formula is symmetric in its two boxes: swapping the roles gives the transposed map exactly
    # (the second operand is staged in LDS: at most 3072 boxes, so the swap uses the first 3072 anchors)
    swapped = pkg.bbox_utils.generate_iou_map(gt_t, anchors[None, :3072].expand(64, -1, -1).contiguous())
    assert torch.equal(torch.nan_to_num(iou[:, :3072], nan=-1.0), torch.nan_to_num(swapped.transpose(1, 2), nan=-1.0))
    assert ((iou >= 0) & (iou <= 1)).all()
    assert (iou[dev(labels < 0)[:, None, :].expand(-1, N, -1)] == 0).all()  # zero-padded gts never overlap
    rp, rn = synth.sampling_ints(41, 64, N)
    deltas, lab = pkg.train_utils.calculate_rpn_actual_outputs(anchors, gt_t, lab_t, hp, random_masks=(dev(rp), dev(rn)))
    lab = lab.reshape(64, N)
    pos, neg = (lab == 1), (lab == 0)
    assert (pos.sum(1) <= 128).all() and ((pos.sum(1) + neg.sum(1)) == 256).all()
    miou = iou.max(2).values
    assert (miou[neg] < 0.3).all()
    # a positive either exceeds 0.7 or attains the column maximum of a valid gt (clipped anchors coincide, so the
    # maximum can be attained by several anchors; the kernel forces the first of them)
    col_best = ((iou == iou.max(1, keepdim=True).values) & dev(labels >= 0)[:, None, :]).any(2)
    assert (miou[pos] > 0.7).logical_or(col_best[pos]).all()
    assert (deltas[~pos] == 0).all() and torch.isfinite(deltas).all()
    # the fused reductions agree with reductions of the materialised map
    m2, a2, s2 = pkg.eval_utils.match_predictions(anchors[None].expand(64, -1, -1).contiguous(), gt_t)
    ar = torch.arange(100, device="cuda")[None, None, :]
    first_max = torch.where(iou == miou[..., None], ar, 1000).min(2).values  # first maximal index, ties included
    assert torch.equal(m2, miou) and torch.equal(a2.long(), first_max)


def test_full_size_config5_shard_properties(pkg):
    """BASELINE config 5 per-GPU shard: 1000x1000 images (63x63 grid, 35721 anchors), pre_nms_topn 12000, B=16."""
    hp = pkg.train_utils.get_hyper_params("mobilenet_v2", img_size=1000, feature_map_shape=63, pre_nms_topn=12000)
    anchors = pkg.bbox_utils.generate_anchors(hp)
    assert anchors.shape[0] == 35721
    deltas, scores = synth.rpn_outputs(51, 16, 63)
    boxes, sc, valid = _roi_bbox_raw(pkg, dev(deltas).reshape(16, -1, 4), dev(scores).reshape(16, -1), anchors, hp, 12000, 300)
    assert (valid == 300).all()
    _check_nms_properties(boxes, sc, valid, hp["nms_iou_threshold"])
    top_v, top_i = pkg.bbox_utils.top_k(dev(scores).reshape(16, -1), 12000)
    assert (top_v[:, 1:] <= top_v[:, :-1]).all() and torch.equal(top_v[:, 0], sc[:, 0])  # the best candidate is always kept
    assert torch.equal(torch.gather(dev(scores).reshape(16, -1), 1, top_i.long()), top_v)
    fm = dev(synth.feature_map(51, 16, 63, 576))  # 9.1 MB per image: the channel-slice crop kernel
    pool = pkg.faster_rcnn.RoIPooling(hp)
    pooled = pool([fm, boxes])
    assert torch.equal(pool([fm * 2, boxes]), pooled * 2)
    assert (pooled >= 0).all() and (pooled.amax((1, 2, 3)) <= fm.amax((1, 2))).all()  # convex combinations of the taps
