"""Randomised parity cases beyond the fixed-seed tests: RoIBBox / Decoder / RPN targets / RoIPooling / top_k / RoIDelta /
combined NMS through the HIP path against the CPU oracle, with shapes, thresholds, score distributions and schedule knobs
drawn from a generator.  Test infrastructure (uses oracle/).  ``run`` is shared by ``tests/test_gpu_stress.py`` (a short,
fixed-seed slice that the driver's ``pytest -m gpu`` runs) and ``tools/stress_parity.py`` (minutes of fresh seeds).
"""
import time

import numpy as np
import torch

import oracle
import synth
import tf_faster_rcnn_amd as pkg

KNOBS = ("FRCNN_CROP_SPLIT", "FRCNN_CROP_SORT", "FRCNN_CROP_PREFETCH", "FRCNN_CROP_SLICE", "FRCNN_CROP_ROWS", "FRCNN_SORT_RADIX_ONLY",
         "FRCNN_PROPOSAL_THREADS", "FRCNN_PROPOSAL_CHUNK", "FRCNN_PROPOSAL_EXACT_SLICES", "FRCNN_ROI_BBOX_BITMASK_MIN_R",
         "FRCNN_DECODER_THREADS", "FRCNN_BITMASK_THREADS", "FRCNN_IOU_REDUCE_SPLIT", "FRCNN_CROP_STAGE")


class Mismatch(AssertionError):
    pass


def dev(a):
    return torch.as_tensor(a).cuda()


def same(got, want, what):
    g = got.detach().cpu().numpy()
    if g.shape != want.shape or not np.array_equal(g.view(np.uint32), np.ascontiguousarray(want).view(np.uint32)):
        raise Mismatch(what)


def run(rng, budget, progress=None, max_cases=None):
    """Cases drawn from ``rng`` for ``budget`` seconds (or ``max_cases``); raises ``Mismatch`` naming the first case that differs
    from the oracle in any bit.  Returns the number of cases per kind.  The schedule knobs are restored afterwards."""
    try:
        return _run(rng, budget, progress, max_cases)
    finally:
        for name in KNOBS:
            pkg._lib.set_schedule(name, None)


def _run(rng, budget, progress, max_cases):
    def scores_like(shape):
        kind = rng.integers(0, 7)
        if kind == 0:
            return rng.random(shape)
        if kind == 1:
            return 1.0 / (1.0 + np.exp(-rng.uniform(1, 10) * rng.normal(0, 1, shape)))
        if kind == 2:
            return rng.choice(rng.random(int(rng.integers(1, 50))), shape)
        if kind == 3:
            return np.exp(-rng.uniform(1, 60) * rng.random(shape))
        if kind == 4:
            return rng.normal(0, 10.0 ** rng.integers(-3, 6), shape)
        if kind == 5:
            return np.where(rng.random(shape) < 0.5, rng.random(), rng.random(shape))
        return 0.5 + 10.0 ** rng.integers(-7, -1) * rng.random(shape)

    lib = pkg._lib
    n = {"roi_bbox": 0, "decoder": 0, "targets": 0, "roi_pooling": 0, "top_k": 0, "roi_delta": 0, "combined_nms": 0}
    t0 = time.time()
    t_print = t0
    while time.time() - t0 < budget:
        if max_cases is not None and sum(n.values()) >= max_cases:
            break
        if progress is not None and time.time() - t_print > 30:  # a sign of life (a silent GPU job is taken to be hung)
            t_print = time.time()
            progress(f"{t_print - t0:.0f} s {n}")
        which = rng.integers(0, 7)
        if which == 6:
            B, nb = int(rng.integers(1, 4)), int(rng.integers(1, 900))
            c = int(rng.integers(1, 6))
            q = int(rng.choice([1, c]))
            mpc, mt = int(rng.integers(1, 300)), int(rng.integers(1, 300))
            iou = float(rng.choice([0.3, 0.5, 0.7, 0.9, 0.05, 0.999, 1.0, 1.5, 1e-5, 0.0, 2e6]))
            sthr = float(rng.choice([float("-inf"), 0.0, 0.3, 0.6]))
            ctr = rng.random((B, nb, q, 2))
            hw = rng.uniform(0.01, float(rng.choice([0.05, 0.25, 0.6])), (B, nb, q, 2))
            boxes = np.concatenate([ctr - hw, ctr + hw], -1).astype(np.float32)
            if rng.random() < 0.3:
                boxes[:, : nb // 3] = boxes[:, nb // 3: 2 * (nb // 3)]  # exact duplicates
            if rng.random() < 0.2:
                boxes[:, ::7, :, 2:] = boxes[:, ::7, :, :2]  # zero-area boxes
            if rng.random() < 0.2:
                boxes[:, ::5] = boxes[:, ::5, :, [2, 3, 0, 1]]  # corners the other way round
            sc = scores_like((B, nb, c)).astype(np.float32)
            got = pkg.bbox_utils.non_max_suppression(dev(boxes), dev(sc), max_output_size_per_class=mpc, max_total_size=mt,
                                                    iou_threshold=iou, score_threshold=sthr)
            want = oracle.combined_non_max_suppression(boxes, sc, mpc, mt, iou, sthr)
            for g, w_, name in zip(got, want, ("boxes", "scores", "classes", "valid")):
                same(g, w_, f"combined NMS B={B} n={nb} q={q} c={c} mpc={mpc} mt={mt} iou={iou} thr={sthr} {name}")
            n["combined_nms"] += 1
            continue
        if which == 3:
            B = int(rng.integers(1, 5))
            H, W = int(rng.integers(1, 40)), int(rng.integers(1, 40))
            C = int(rng.choice([1, 3, 4, 8, 36, 64, 100, 256, 576]))
            R = int(rng.choice([1, 2, 7, 33, 64, 300, 700]))
            ph, pw = int(rng.integers(1, 9)), int(rng.integers(1, 9))
            for name, val in (("FRCNN_CROP_SPLIT", rng.choice([-1, 0, 2, 4, 8])), ("FRCNN_CROP_SORT", rng.integers(0, 2)),
                              ("FRCNN_CROP_PREFETCH", rng.choice([-1, 0, 7, 32, 640032])), ("FRCNN_CROP_SLICE", rng.choice([-1, 0, 1])),
                              ("FRCNN_CROP_ROWS", rng.choice([-1, 0, 1])), ("FRCNN_CROP_STAGE", rng.choice([-1, 0, 1]))):
                lib.set_schedule(name, None if val < 0 else int(val))
            fm = rng.normal(0, 1, (B, H, W, C)).astype(np.float32)
            rois = rng.uniform(-0.2, 1.2, (B, R, 4)).astype(np.float32)
            got = pkg.faster_rcnn.RoIPooling(synth.hyper_params(pooling_size=(ph, pw)))([dev(fm), dev(rois)])
            same(got, oracle.roi_pooling(fm, rois, (ph, pw)), f"RoIPooling B={B} H={H} W={W} C={C} R={R} ({ph},{pw})")
            n["roi_pooling"] += 1
            continue
        if which == 4:
            B, N = int(rng.integers(1, 4)), int(rng.integers(1, 30000))
            K = int(rng.integers(1, min(N, 12000) + 1))
            sc = scores_like((B, N)).astype(np.float32)
            lib.set_schedule("FRCNN_SORT_RADIX_ONLY", int(rng.integers(0, 2)))
            vals, idx = pkg.bbox_utils.top_k(dev(sc), K)
            wv, wi = oracle.top_k(sc, K)
            same(idx, wi, f"top_k indices B={B} N={N} K={K}")
            same(vals, wv, f"top_k values B={B} N={N} K={K}")
            n["top_k"] += 1
            continue
        if which == 5:
            B, R, M, L = int(rng.integers(1, 4)), int(rng.integers(1, 1500)), int(rng.integers(1, 40)), int(rng.integers(2, 30))
            hp = pkg.train_utils.get_hyper_params("vgg16")
            hp.update(total_labels=L, total_pos_bboxes=int(rng.integers(0, 200)), total_neg_bboxes=int(rng.integers(0, 200)))
            gt, labels = synth.gt_boxes(int(rng.integers(1 << 30)), B, M, L)
            rois = np.clip(gt[:, rng.integers(0, M, R)] + rng.normal(0, 0.05, (B, R, 4)), 0, 1).astype(np.float32)
            rois[:, (3 * R) // 4:] = 0
            lo, hi = [(1, 4), (1, 2000), (3900, 4300), (-3, 60), (1, 2 ** 30)][int(rng.integers(0, 5))]
            rp = rng.integers(lo, hi, (B, R)).astype(np.int32)
            rn = rng.integers(lo, hi, (B, R)).astype(np.int32)
            gd, gl = pkg.faster_rcnn.RoIDelta(hp)([dev(rois), dev(gt), dev(labels)], random_masks=(dev(rp), dev(rn)))
            wd, wl = oracle.roi_delta(rois, gt, labels, hp, (rp, rn))
            same(gl, wl, f"RoIDelta labels B={B} R={R} M={M} L={L}")
            same(gd, wd, f"RoIDelta deltas B={B} R={R} M={M} L={L}")
            n["roi_delta"] += 1
            continue
        if which == 0:
            G = int(rng.integers(3, 40))
            N = G * G * 9
            K = int(rng.integers(1, N + 1))
            R = int(rng.integers(1, min(2 * K, 2200) + 1))
            B = int(rng.integers(1, 4))
            thr = float(rng.choice([0.3, 0.5, 0.7, 0.9, 0.05, 0.999, 1.0, 1.5, 1e-5]))
            hp = pkg.train_utils.get_hyper_params("mobilenet_v2")
            hp.update(feature_map_shape=G, pre_nms_topn=K, test_nms_topn=R, nms_iou_threshold=thr)
            anchors = oracle.generate_anchors(hp)
            deltas, _ = synth.rpn_outputs(int(rng.integers(1 << 30)), B, G)
            deltas = (deltas * rng.choice([1.0, 0.3, 0.05])).astype(np.float32)
            scores = scores_like((B, G, G, 9)).astype(np.float32)
            for name, val in (("FRCNN_PROPOSAL_THREADS", rng.choice([256, 1024])), ("FRCNN_PROPOSAL_CHUNK", rng.choice([64, 128, 256])),
                              ("FRCNN_PROPOSAL_EXACT_SLICES", rng.integers(0, 2)), ("FRCNN_SORT_RADIX_ONLY", rng.integers(0, 2)),
                              ("FRCNN_ROI_BBOX_BITMASK_MIN_R", rng.choice([0, 1, 512])), ("FRCNN_BITMASK_THREADS", rng.choice([256, 1024]))):
                lib.set_schedule(name, int(val))
            got = pkg.faster_rcnn.RoIBBox(dev(anchors), "inference", hp)([dev(deltas), dev(scores)])
            want = oracle.roi_bbox(deltas, scores, anchors, hp["variances"], K, R, thr)
            same(got, want, f"RoIBBox G={G} K={K} R={R} B={B} thr={thr}")
            n["roi_bbox"] += 1
        elif which == 1:
            B, R, L = int(rng.integers(1, 4)), int(rng.integers(1, 500)), int(rng.integers(2, 40))
            T = int(rng.integers(1, 300))
            thr = float(rng.choice([0.0, 0.05, 0.3, 0.5, 0.8]))
            rois = np.sort(rng.random((B, R, 2, 2)).astype(np.float32), axis=2).reshape(B, R, 4)
            if rng.random() < 0.3:
                rois[:, R // 2:] = 0  # zero-padded RoIs
            deltas = rng.normal(0, rng.choice([0.1, 0.5]), (B, R, L * 4)).astype(np.float32)
            if rng.random() < 0.4:
                probs = scores_like((B, R, L)).astype(np.float32)
            else:
                logits = rng.normal(0, 1, (B, R, L)) + 5 * np.eye(L)[rng.integers(0, L, (B, R))]
                probs = (np.exp(logits) / np.exp(logits).sum(-1, keepdims=True)).astype(np.float32)
            lib.set_schedule("FRCNN_DECODER_THREADS", int(rng.choice([256, 1024])))
            hp = synth.hyper_params()
            layer = pkg.faster_rcnn.Decoder(hp["variances"], L, max_total_size=T, score_threshold=thr)
            got = layer([dev(rois), dev(deltas), dev(probs)])
            want = oracle.decoder(rois, deltas, probs, hp["variances"], L, T, thr)
            for g, w, name in zip(got, want, ("boxes", "labels", "scores")):
                same(g, w, f"Decoder B={B} R={R} L={L} T={T} thr={thr} {name}")
            n["decoder"] += 1
        else:
            B, G, M = int(rng.integers(1, 5)), int(rng.integers(2, 36)), int(rng.integers(1, 60))
            hp = pkg.train_utils.get_hyper_params("mobilenet_v2")
            hp.update(feature_map_shape=G, total_pos_bboxes=int(rng.integers(0, 300)), total_neg_bboxes=int(rng.integers(0, 300)))
            anchors = oracle.generate_anchors(hp)
            N = anchors.shape[0]
            gt, labels = synth.gt_boxes(int(rng.integers(1 << 30)), B, M)
            if rng.random() < 0.5:
                k = min(M, 5)
                gt[:, :k] = anchors[rng.integers(0, N, k)]
                labels[:, :k] = 1
            lo, hi = [(1, 4), (1, 2560), (3000, 4200), (-3, 60), (1, 2 ** 30)][int(rng.integers(0, 5))]
            rp = rng.integers(lo, hi, (B, N)).astype(np.int32)
            rn = rng.integers(lo, hi, (B, N)).astype(np.int32)
            lib.set_schedule("FRCNN_IOU_REDUCE_SPLIT", int(rng.integers(0, 2)))
            gd, gl = pkg.train_utils.calculate_rpn_actual_outputs(dev(anchors), dev(gt), dev(labels), hp, random_masks=(dev(rp), dev(rn)))
            wd, wl = oracle.calculate_rpn_actual_outputs(anchors, gt, labels, hp, (rp, rn))
            same(gl, wl, f"rpn labels B={B} G={G} M={M} ints=[{lo},{hi})")
            same(gd, wd, f"rpn deltas B={B} G={G} M={M}")
            n["targets"] += 1
    return n
