"""Seeded synthetic inputs for the proposal/RoI hot path (tests and bench; NumPy only).

Shapes and distributions follow SURVEY.md section 8d: the conv backbone and the dense heads are out
of scope, so their outputs are synthesised.  Everything is ``numpy.random.default_rng(seed)`` fp32.
"""

from __future__ import annotations

from typing import Dict, Tuple

import numpy as np


def rpn_outputs(seed: int, B: int, G: int, A: int = 9, unique_scores: bool = True) -> Tuple[np.ndarray, np.ndarray]:
    """RPN head outputs: deltas ``(B,G,G,A*4)`` ~ N(0, 0.5); scores ``(B,G,G,A)`` = sigmoid(N(0,2)),
    made unique per image (a random permutation of a strictly increasing grid mixed in) so tie order is moot."""
    rng = np.random.default_rng(seed)
    deltas = rng.normal(0.0, 0.5, size=(B, G, G, A * 4)).astype(np.float32)
    N = G * G * A
    raw = rng.normal(0.0, 2.0, size=(B, N))
    scores = 1.0 / (1.0 + np.exp(-raw))
    if unique_scores:
        # rank-preserving de-duplication: replace by the sorted unique grid of the same length
        for b in range(B):
            order = np.argsort(scores[b], kind="stable")
            grid = np.linspace(0.001, 0.999, N, dtype=np.float64)
            s = np.empty(N, dtype=np.float64)
            s[order] = grid
            scores[b] = s
        scores32 = scores.astype(np.float32)
        for b in range(B):
            assert np.unique(scores32[b]).size == N
    else:
        scores32 = scores.astype(np.float32)
    return deltas, scores32.reshape(B, G, G, A)


def feature_map(seed: int, B: int, G: int, C: int) -> np.ndarray:
    """Backbone feature map ``(B,G,G,C)`` NHWC, post-ReLU like."""
    rng = np.random.default_rng(seed + 1000)
    return np.maximum(rng.normal(0.0, 1.0, size=(B, G, G, C)), 0.0).astype(np.float32)


def head_outputs(seed: int, B: int, R: int, L: int) -> Tuple[np.ndarray, np.ndarray]:
    """FC-head outputs for the Decoder: deltas ``(B,R,L*4)`` ~ N(0,1); probs ``(B,R,L)`` = softmax of
    N(0,1) logits + 6 on a random class (background with p=0.3), so ~70 % of RoIs pass the 0.5 threshold."""
    rng = np.random.default_rng(seed + 2000)
    deltas = rng.normal(0.0, 1.0, size=(B, R, L * 4)).astype(np.float32)
    logits = rng.normal(0.0, 1.0, size=(B, R, L))
    is_bg = rng.random(size=(B, R)) < 0.3
    cls = np.where(is_bg, 0, rng.integers(1, L, size=(B, R)))
    np.put_along_axis(logits, cls[..., None], np.take_along_axis(logits, cls[..., None], -1) + 6.0, -1)
    logits -= logits.max(-1, keepdims=True)
    e = np.exp(logits)
    probs = (e / e.sum(-1, keepdims=True)).astype(np.float32)
    return deltas, probs


def gt_boxes(seed: int, B: int, M: int, L: int = 21) -> Tuple[np.ndarray, np.ndarray]:
    """Padded ground truth: ``m_valid ~ U{1..M}`` boxes per image from uniform centres and log-uniform
    sizes in [0.05, 0.6], clipped to [0,1]; the rest zero boxes with label -1 (utils/data_utils.py:235-242)."""
    rng = np.random.default_rng(seed + 3000)
    boxes = np.zeros((B, M, 4), dtype=np.float32)
    labels = np.full((B, M), -1, dtype=np.int32)
    for b in range(B):
        mv = int(rng.integers(1, M + 1))
        cy, cx = rng.random(mv), rng.random(mv)
        h = np.exp(rng.uniform(np.log(0.05), np.log(0.6), mv))
        w = np.exp(rng.uniform(np.log(0.05), np.log(0.6), mv))
        bx = np.stack([cy - h / 2, cx - w / 2, cy + h / 2, cx + w / 2], -1)
        boxes[b, :mv] = np.clip(bx, 0.0, 1.0).astype(np.float32)
        labels[b, :mv] = rng.integers(1, L, size=mv)
    return boxes, labels


def sampling_ints(seed: int, B: int, N: int, total_pos: int = 128, total_neg: int = 128) -> Tuple[np.ndarray, np.ndarray]:
    """The random integers of the two randomly_select_xyz_mask calls (utils/train_utils.py:82-83):
    U[1, 10*total_pos) and U[1, 10*(total_pos+total_neg))."""
    rng = np.random.default_rng(seed + 4000)
    rp = rng.integers(1, 10 * total_pos, size=(B, N), dtype=np.int32)
    rn = rng.integers(1, 10 * (total_pos + total_neg), size=(B, N), dtype=np.int32)
    return rp, rn


def hyper_params(backbone: str = "mobilenet_v2", **overrides) -> Dict:
    """The reference's hyper-parameter table (utils/train_utils.py:17-56), restated for tests/bench."""
    table = {
        "vgg16": {"img_size": 500, "feature_map_shape": 31},
        "mobilenet_v2": {"img_size": 500, "feature_map_shape": 32},
    }
    hp = dict(table[backbone])
    hp.update({
        "anchor_ratios": [1., 2., 1. / 2.], "anchor_scales": [128, 256, 512],
        "pre_nms_topn": 6000, "train_nms_topn": 1500, "test_nms_topn": 300, "nms_iou_threshold": 0.7,
        "total_pos_bboxes": 128, "total_neg_bboxes": 128, "pooling_size": (7, 7),
        "variances": [0.1, 0.1, 0.2, 0.2],
    })
    hp.update(overrides)
    hp["anchor_count"] = len(hp["anchor_ratios"]) * len(hp["anchor_scales"])
    return hp
