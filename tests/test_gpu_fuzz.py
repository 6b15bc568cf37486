"""GPU parity, randomised: many small random shapes / thresholds per entry point against the CPU oracle (bit for bit).

The shapes are drawn from a fixed seed, so the suite is deterministic; it complements tests/test_gpu_parity.py (named
configurations and edge cases) by sweeping the dispatch boundaries of the kernels: chunk and band sizes, vector / scalar
paths, the schedules behind frcnn_roi_bbox, odd depths, K not a multiple of 64, R above and below K.
"""

import numpy as np
import pytest

import oracle
import synth

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu

from tests.test_gpu_parity import assert_bits, dev  # noqa: E402


@pytest.fixture(scope="module")
def pkg():
    import tf_faster_rcnn_amd as p
    return p


def test_fuzz_roi_bbox_all_schedules(pkg, knobs):
    rng = np.random.default_rng(2024)
    for case in range(360):
        G = int(rng.integers(3, 24))
        N = G * G * 9
        K = int(rng.integers(1, N + 1))
        R = int(rng.integers(1, min(2 * K, 2500) + 1))
        thr = float(rng.choice([0.1, 0.3, 0.5, 0.7, 0.9]))
        B = int(rng.integers(1, 4))
        schedule = ("fused", "bitmask", "unfused")[case % 3]
        knobs.delenv("FRCNN_ROI_BBOX_UNFUSED", raising=False)
        knobs.setenv("FRCNN_ROI_BBOX_BITMASK_MIN_R", {"fused": "0", "bitmask": "1", "unfused": "0"}[schedule])
        if schedule == "unfused":
            knobs.setenv("FRCNN_ROI_BBOX_UNFUSED", "1")
        knobs.setenv("FRCNN_PROPOSAL_THREADS", "256" if (case // 3) % 2 else "1024")  # block size of the fused kernel
        knobs.setenv("FRCNN_PROPOSAL_EXACT_SLICES", "1" if (case // 6) % 2 else "0")   # exact-rank slices / sampled cuts
        knobs.setenv("FRCNN_PROPOSAL_CHUNK", ("64", "128", "256")[(case // 12) % 3])    # NMS chunk width
        knobs.setenv("FRCNN_SORT_RADIX_ONLY", "1" if (case // 9) % 2 else "0")         # full-sort kernel: bucket / radix
        knobs.setenv("FRCNN_BITMASK_THREADS", "1024" if (case // 3) % 2 else "256")     # workgroup size of the bitmask scans
        hp = pkg.train_utils.get_hyper_params("mobilenet_v2")
        hp.update(feature_map_shape=G, pre_nms_topn=K, test_nms_topn=R, nms_iou_threshold=thr)
        anchors = oracle.generate_anchors(hp)
        deltas, scores = synth.rpn_outputs(1000 + case, B, G, unique_scores=bool(case % 4))
        if case % 5 == 0:
            deltas = (deltas * 0.1).astype(np.float32)  # heavy overlap: long suppression chains
        got = pkg.faster_rcnn.RoIBBox(dev(anchors), "inference", hp)([dev(deltas), dev(scores)])
        want = oracle.roi_bbox(deltas, scores, anchors, hp["variances"], K, R, thr)
        assert_bits(got, want, f"RoIBBox fuzz case {case}: {schedule} B={B} G={G} K={K} R={R} thr={thr}")


def test_roi_bbox_fast_slices_score_distributions(pkg, knobs):
    """The fused kernel cuts its slices at a sampled key boundary and bucket-sorts them; score distributions that
    stress the cut (heavy 16-bit prefix bins), the buckets (clusters, ties) and the fall-back to the exact selection."""
    rng = np.random.default_rng(31)
    G, K, R = 20, 3000, 300
    N = G * G * 9
    dists = {
        "uniform": lambda: rng.random((2, N)),
        "sigmoid_saturated": lambda: 1.0 / (1.0 + np.exp(-8.0 * rng.normal(0, 1, (2, N)))),
        "constant": lambda: np.full((2, N), 0.25),
        "two_values": lambda: rng.choice([0.125, 0.75], (2, N)),
        "few_values": lambda: rng.choice(rng.random(40), (2, N)),
        "tight_cluster": lambda: 0.5 + 1e-6 * rng.random((2, N)),
        "tiny": lambda: 1e-30 * rng.random((2, N)),
        "signed_wide": lambda: rng.normal(0, 1e4, (2, N)),
        "mixed_ties": lambda: np.where(rng.random((2, N)) < 0.5, 0.9, rng.random((2, N))),
        "exponential": lambda: np.exp(-30.0 * rng.random((2, N))),
    }
    hp = pkg.train_utils.get_hyper_params("mobilenet_v2")
    hp.update(feature_map_shape=G, pre_nms_topn=K, test_nms_topn=R)
    anchors = oracle.generate_anchors(hp)
    layer = pkg.faster_rcnn.RoIBBox(dev(anchors), "inference", hp)
    for name, draw in dists.items():
        for scale in (1.0, 0.05):  # little suppression (one slice) / heavy suppression (several slices)
            deltas, _ = synth.rpn_outputs(700, 2, G)
            deltas = (deltas * scale).astype(np.float32)
            scores = draw().astype(np.float32).reshape(2, G, G, 9)
            want = oracle.roi_bbox(deltas, scores, anchors, hp["variances"], K, R, 0.7)
            for exact in ("0", "1"):
                knobs.setenv("FRCNN_PROPOSAL_EXACT_SLICES", exact)
                got = layer([dev(deltas), dev(scores)])
                assert_bits(got, want, f"RoIBBox scores={name} delta_scale={scale} exact_slices={exact}")


def test_roi_bbox_minus_inf_scores_all_schedules(pkg, knobs):
    """Scores of -inf never pass combined_non_max_suppression's strict `score > -inf`: with more than N - K of them the
    top-K holds dead candidates, which every schedule must skip (and NaN-free ties between them order by index)."""
    rng = np.random.default_rng(99)
    for case, schedule in enumerate(("fused", "bitmask", "unfused") * 3):
        knobs.delenv("FRCNN_ROI_BBOX_UNFUSED", raising=False)
        knobs.setenv("FRCNN_ROI_BBOX_BITMASK_MIN_R", {"fused": "0", "bitmask": "1", "unfused": "0"}[schedule])
        if schedule == "unfused":
            knobs.setenv("FRCNN_ROI_BBOX_UNFUSED", "1")
        G, K, R = 12, 1000, (40, 400, 1200)[case // 3]
        hp = pkg.train_utils.get_hyper_params("mobilenet_v2")
        hp.update(feature_map_shape=G, pre_nms_topn=K, test_nms_topn=R)
        anchors = oracle.generate_anchors(hp)
        deltas, scores = synth.rpn_outputs(500 + case, 3, G)
        scores = scores.reshape(3, -1).copy()
        scores[rng.random(scores.shape) < 0.6] = -np.inf   # 1296 anchors, ~520 finite: fewer than K
        scores[2] = -np.inf                                 # an image with nothing to keep
        got = pkg.faster_rcnn.RoIBBox(dev(anchors), "inference", hp)([dev(deltas), dev(scores)])
        want, _, valid = oracle.roi_bbox(deltas, scores, anchors, hp["variances"], K, R, 0.7, return_all=True)
        assert valid[2] == 0
        assert_bits(got, want, f"RoIBBox -inf scores: {schedule} R={R}")


def test_fuzz_roi_pooling(pkg, knobs):
    rng = np.random.default_rng(7)
    for case in range(240):
        B = int(rng.integers(1, 4))
        H, W = int(rng.integers(1, 20)), int(rng.integers(1, 20))
        C = int(rng.choice([1, 3, 4, 8, 20, 36, 64, 100, 576]))
        R = int(rng.integers(1, 40))
        ph, pw = int(rng.integers(1, 9)), int(rng.integers(1, 9))
        knobs.setenv("FRCNN_CROP_SLICE", str(case % 2))
        knobs.setenv("FRCNN_CROP_ROWS", str((case // 2) % 2))  # (box, output row) workgroups (diagnostic schedule)
        fm = rng.normal(0, 1, (B, H, W, C)).astype(np.float32)
        rois = rng.uniform(-0.2, 1.2, (B, R, 4)).astype(np.float32)  # inverted and out-of-image boxes included
        hp = synth.hyper_params(pooling_size=(ph, pw))
        got = pkg.faster_rcnn.RoIPooling(hp)([dev(fm), dev(rois)])
        want = oracle.roi_pooling(fm, rois, (ph, pw))
        assert_bits(got, want, f"RoIPooling fuzz case {case}: B={B} H={H} W={W} C={C} R={R} crop=({ph},{pw})")


def test_fuzz_roi_pooling_warmup_workgroups(pkg, knobs):
    """The box kernel with warm-up workgroups in its slot sequence (FRCNN_CROP_PREFETCH = n per image; default 48 for large
    feature maps): every box is still written exactly once, whatever the number of images per XCD range, boxes per image
    (fewer or more than the 128-box lead) and warm-up workgroups -- also through the grouped launch."""
    rng = np.random.default_rng(19)
    lib = pkg._lib.lib()
    for case in range(60):
        B = int(rng.integers(1, 20))
        H, W = int(rng.integers(2, 12)), int(rng.integers(2, 12))
        C = int(rng.choice([4, 8, 64]))
        R = int(rng.choice([1, 2, 7, 8, 9, 31, 127, 128, 129, 300]))
        knobs.setenv("FRCNN_CROP_PREFETCH", int(rng.choice([1, 2, 5, 48, 200])))
        fm = rng.normal(0, 1, (B, H, W, C)).astype(np.float32)
        rois = rng.uniform(-0.1, 1.1, (B, R, 4)).astype(np.float32)
        hp = synth.hyper_params(pooling_size=(3, 2))
        want = oracle.roi_pooling(fm, rois, (3, 2))
        got = pkg.faster_rcnn.RoIPooling(hp)([dev(fm), dev(rois)])
        assert_bits(got, want, f"RoIPooling warm-up case {case}: B={B} H={H} W={W} C={C} R={R}")
        if case % 3 == 0:  # grouped launch: the same triple g times, outputs poisoned first
            g = 2 + case % 3
            fms, rs = [dev(fm) for _ in range(g)], [dev(rois) for _ in range(g)]
            outs = [torch.full((B, R, 3, 2, C), float("nan"), device="cuda") for _ in range(g)]
            tab = pkg._lib.ptr_table
            pkg._lib.check(lib.frcnn_crop_and_resize_multi(tab(fms), tab(rs), tab(outs), g, B, H, W, C, B * R, R, 3, 2, 0.0,
                                                           None, 0, torch.cuda.current_stream().cuda_stream))
            for o in outs:
                assert_bits(o, want, f"grouped RoIPooling warm-up case {case}: g={g} B={B} R={R}")


def test_fuzz_roi_pooling_sorted_channel_split(pkg, knobs):
    """frcnn_roi_pooling's schedule for feature maps beyond an XCD's L2, forced on small shapes: channels cut over 2 / 4 / 8
    XCDs, the RoIs of every image visited in spatial order (or not); outputs land in their own slots, bit-exact."""
    rng = np.random.default_rng(23)
    for case in range(72):
        B = int(rng.integers(1, 7))
        H, W = int(rng.integers(1, 14)), int(rng.integers(1, 14))
        C = int(rng.choice([4, 8, 12, 36, 64, 576]))
        R = int(rng.choice([1, 2, 3, 17, 64, 300, 1025]))
        ph, pw = int(rng.integers(1, 5)), int(rng.integers(1, 5))
        knobs.setenv("FRCNN_CROP_SPLIT", [2, 4, 8][case % 3])
        knobs.setenv("FRCNN_CROP_SORT", (case // 3) % 2)
        fm = rng.normal(0, 1, (B, H, W, C)).astype(np.float32)
        rois = rng.uniform(-0.2, 1.2, (B, R, 4)).astype(np.float32)
        if case % 5 == 0:
            rois[0, : R // 2] = rois[0, 0]  # equal keys: ties by index
        if case % 7 == 0:
            rois[-1, -1, 0] = np.nan  # a NaN centre must not break the permutation
        hp = synth.hyper_params(pooling_size=(ph, pw))
        got = pkg.faster_rcnn.RoIPooling(hp)([dev(fm), dev(rois)])
        want = oracle.roi_pooling(fm, rois, (ph, pw))
        assert_bits(got, want, f"RoIPooling split case {case}: B={B} H={H} W={W} C={C} R={R} crop=({ph},{pw})")


def test_fuzz_roi_pooling_staged_rows(pkg, knobs):
    """frcnn_roi_pooling's schedule for feature maps beyond an XCD's L2 (the (RoI, output row) items of every image ordered by
    the source row they sample, a workgroup per source-row pair and channel part with the two rows staged in LDS by
    global_load_lds), forced on small shapes: out-of-range rows (no taps, sorted last), a single source row, rows nobody
    samples, NaN boxes, several parameter chunks per row pair, every number of channel parts, every dealing of the row pairs
    to the XCDs; the grouped entry point with a
    workspace takes the same route.  Bit-exact: every output lands in its own slot."""
    rng = np.random.default_rng(29)
    lib = pkg._lib.lib()
    knobs.setenv("FRCNN_CROP_STAGE", 1)
    for case in range(120):
        B = int(rng.integers(1, 7))
        H, W = int(rng.integers(1, 40)), int(rng.integers(1, 14))
        C = int(rng.choice([4, 8, 12, 36, 64, 576, 1024]))
        R = int(rng.choice([1, 2, 3, 17, 64, 300, 1025]))
        ph, pw = int(rng.integers(1, 9)), int(rng.integers(1, 5))
        q_min = int(rng.choice([0, 0, 2, 4, 8]))  # a minimum number of channel parts (0: the fewest that fit)
        if q_min:
            knobs.setenv("FRCNN_CROP_SPLIT", q_min)
        else:
            knobs.delenv("FRCNN_CROP_SPLIT")
        rg = [None, 0, 1, 2, 3, 4, 8, 64][case % 8]  # row pairs per XCD group (None: the library's choice for this H)
        if rg is None:
            knobs.delenv("FRCNN_CROP_PREFETCH")
        else:
            knobs.setenv("FRCNN_CROP_PREFETCH", rg)
        fm = rng.normal(0, 1, (B, H, W, C)).astype(np.float32)
        lo, hi = [(-0.2, 1.2), (0.0, 1.0), (-2.0, 3.0), (0.4, 0.6)][case % 4]
        rois = rng.uniform(lo, hi, (B, R, 4)).astype(np.float32)
        if case % 5 == 0:
            rois[0, : R // 2] = rois[0, 0]  # many items on one source row: several chunks of parameters
        if case % 7 == 0:
            rois[-1, -1, 0] = np.nan  # in_y = NaN: a row without taps
        if case % 11 == 0:
            rois[:] = 0.0  # zero-padded RoIs only: every item samples row 0
        hp = synth.hyper_params(pooling_size=(ph, pw))
        want = oracle.roi_pooling(fm, rois, (ph, pw))
        got = pkg.faster_rcnn.RoIPooling(hp)([dev(fm), dev(rois)])
        assert_bits(got, want, f"RoIPooling staged-rows case {case}: B={B} H={H} W={W} C={C} R={R} crop=({ph},{pw}) Q>={q_min} rg={rg}")
        if case % 4 == 0 and C % 4 == 0:  # the grouped entry point hands the RoIPooling layout to frcnn_roi_pooling when a knob is forced
            g = 1 + case % 3
            fms, rs = [dev(fm) for _ in range(g)], [dev(rois) for _ in range(g)]
            outs = [torch.full((B, R, ph, pw, C), float("nan"), device="cuda") for _ in range(g)]
            ws = torch.empty(lib.frcnn_roi_pooling_workspace_bytes(B, R), dtype=torch.uint8, device="cuda")
            tab = pkg._lib.ptr_table
            knobs.setenv("FRCNN_CROP_ROWS", 0)  # a forced knob keeps the grouped launch from taking the one-launch form
            pkg._lib.check(lib.frcnn_crop_and_resize_multi(tab(fms), tab(rs), tab(outs), g, B, H, W, C, B * R, R, ph, pw, 0.0,
                                                           ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream))
            knobs.delenv("FRCNN_CROP_ROWS")
            for o in outs:
                assert_bits(o, want, f"grouped RoIPooling staged-rows case {case}: g={g}")


def test_fuzz_decoder(pkg, knobs):
    rng = np.random.default_rng(11)
    for case in range(240):
        B, R, L = int(rng.integers(1, 4)), int(rng.integers(1, 400)), int(rng.integers(2, 30))
        T = int(rng.integers(1, 300))
        thr = float(rng.choice([0.05, 0.3, 0.5, 0.8]))
        knobs.setenv("FRCNN_DECODER_UNFUSED", str(case % 2))
        knobs.setenv("FRCNN_DECODER_THREADS", "256" if (case // 2) % 2 else "1024")  # block size of the fused kernel
        hp = synth.hyper_params()
        rois = np.sort(rng.random((B, R, 2, 2)).astype(np.float32), axis=2).reshape(B, R, 4)
        deltas = rng.normal(0, 0.5, (B, R, L * 4)).astype(np.float32)
        if case % 3 == 0:
            probs = rng.random((B, R, L)).astype(np.float32)  # un-normalised: many classes fire per RoI
        else:
            logits = rng.normal(0, 1, (B, R, L)) + 5 * np.eye(L)[rng.integers(0, L, (B, R))]
            probs = (np.exp(logits) / np.exp(logits).sum(-1, keepdims=True)).astype(np.float32)
        layer = pkg.faster_rcnn.Decoder(hp["variances"], L, max_total_size=T, score_threshold=thr)
        got = layer([dev(rois), dev(deltas), dev(probs)])
        want = oracle.decoder(rois, deltas, probs, hp["variances"], L, T, thr)
        for g, w, name in zip(got, want, ("boxes", "labels", "scores")):
            assert_bits(g, w, f"Decoder fuzz case {case}: B={B} R={R} L={L} T={T} thr={thr} {name}")


def test_fuzz_iou_and_targets(pkg):
    rng = np.random.default_rng(13)
    for case in range(160):
        B, G, M = int(rng.integers(1, 5)), int(rng.integers(2, 20)), int(rng.integers(1, 40))
        hp = pkg.train_utils.get_hyper_params("mobilenet_v2")
        hp.update(feature_map_shape=G, total_pos_bboxes=int(rng.integers(0, 40)), total_neg_bboxes=int(rng.integers(0, 40)))
        anchors = oracle.generate_anchors(hp)
        N = anchors.shape[0]
        gt, labels = synth.gt_boxes(300 + case, B, M)
        assert_bits(pkg.bbox_utils.generate_iou_map(dev(anchors), dev(gt)), oracle.generate_iou_map(anchors, gt),
                    f"iou map fuzz case {case}: B={B} G={G} M={M}")
        rp = rng.integers(1, 500, (B, N), dtype=np.int32)
        rn = rng.integers(1, 500, (B, N), dtype=np.int32)
        got_d, got_l = pkg.train_utils.calculate_rpn_actual_outputs(dev(anchors), dev(gt), dev(labels), hp,
                                                                    random_masks=(dev(rp), dev(rn)))
        want_d, want_l = oracle.calculate_rpn_actual_outputs(anchors, gt, labels, hp, (rp, rn))
        assert_bits(got_l, want_l, f"rpn labels fuzz case {case}")
        assert_bits(got_d, want_d, f"rpn deltas fuzz case {case}")
        # RoIDelta on jittered gts + random boxes
        R, L = int(rng.integers(1, 300)), int(rng.integers(2, 25))
        rois = np.sort(rng.random((B, R, 2, 2)).astype(np.float32), axis=2).reshape(B, R, 4)
        k = min(R, M)
        rois[:, :k] = np.clip(gt[:, :k] + rng.normal(0, 0.03, (B, k, 4)), 0, 1)
        lab = np.where(labels >= 0, labels % (L - 1 if L > 2 else 1) + (1 if L > 2 else 0), labels).astype(np.int32)
        hp_r = dict(hp, total_labels=L)
        rp2 = rng.integers(1, 500, (B, R), dtype=np.int32)
        rn2 = rng.integers(1, 500, (B, R), dtype=np.int32)
        gd, gl = pkg.faster_rcnn.RoIDelta(hp_r)([dev(rois), dev(gt), dev(lab)], random_masks=(dev(rp2), dev(rn2)))
        wd, wl = oracle.roi_delta(rois, gt, lab, hp_r, (rp2, rn2))
        assert_bits(gl, wl, f"RoIDelta labels fuzz case {case}")
        assert_bits(gd, wd, f"RoIDelta deltas fuzz case {case}")


def test_fuzz_top_k_and_combined_nms(pkg):
    rng = np.random.default_rng(17)
    for case in range(160):
        B, N = int(rng.integers(1, 4)), int(rng.integers(1, 5000))
        K = int(rng.integers(1, N + 1))
        scores = rng.random((B, N)).astype(np.float32)
        if case % 3 == 0:
            scores = np.round(scores * 50) / 50  # many ties
        gv, gi = pkg.bbox_utils.top_k(dev(scores), K)
        wv, wi = oracle.top_k(scores, K)
        assert_bits(gi, wi, f"top_k fuzz case {case} indices (N={N}, K={K})")
        assert_bits(gv, wv, f"top_k fuzz case {case} values")
        n, c = int(rng.integers(1, 300)), int(rng.integers(1, 8))
        q = c if case % 2 else 1
        boxes = rng.uniform(-0.1, 1.1, (B, n, q, 4)).astype(np.float32)
        sc = rng.random((B, n, c)).astype(np.float32)
        mpc, mt = int(rng.integers(1, 120)), int(rng.integers(1, 200))
        kw = dict(max_output_size_per_class=mpc, max_total_size=mt, iou_threshold=float(rng.choice([0.2, 0.5, 0.8])),
                  score_threshold=float(rng.choice([-np.inf, 0.3, 0.6])))
        got = pkg.bbox_utils.non_max_suppression(dev(boxes), dev(sc), **kw)
        want = oracle.combined_non_max_suppression(boxes, sc, kw["max_output_size_per_class"], kw["max_total_size"],
                                                   kw["iou_threshold"], kw["score_threshold"])
        for g, w, name in zip(got, want, ("boxes", "scores", "classes", "valid")):
            assert_bits(g, w, f"combined NMS fuzz case {case} {name} (n={n}, q={q}, c={c})")
