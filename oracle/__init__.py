"""CPU oracle for the proposal/RoI hot path -- TEST INFRASTRUCTURE ONLY.

A ctypes/NumPy front end over ``libfrcnn_oracle.so`` (plain C, ``frcnn_oracle.c``), the CPU
restatement of the reference's algorithm.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this package, and only as the checker or the
reported baseline -- never the product package.

PARITY STATUS: "parity unpinned" against TensorFlow 2.0 (see ``frcnn_oracle.h``).
"""

from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libfrcnn_oracle.so")

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int32)
_u8p = ctypes.POINTER(ctypes.c_uint8)
_f64p = ctypes.POINTER(ctypes.c_double)


def build(force: bool = False) -> str:
    """Compile the C restatement with the committed Makefile; returns the library path."""
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "frcnn_oracle.c"))
    ):
        subprocess.run(["make", "-C", _HERE, "-B", "libfrcnn_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib: Optional[ctypes.CDLL] = None


def lib() -> ctypes.CDLL:
    """Load (building on first use if needed) the oracle shared library."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.orc_expf.restype = ctypes.c_float
        _lib.orc_expf.argtypes = [ctypes.c_float]
        _lib.orc_logf.restype = ctypes.c_float
        _lib.orc_logf.argtypes = [ctypes.c_float]
        _lib.orc_get_threads.restype = ctypes.c_int
    return _lib


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a: Optional[np.ndarray], t):
    return None if a is None else a.ctypes.data_as(t)


def set_threads(n: int) -> None:
    lib().orc_set_threads(ctypes.c_int(int(n)))


def get_threads() -> int:
    return int(lib().orc_get_threads())


def expf(x: float) -> float:
    return float(lib().orc_expf(ctypes.c_float(x)))


def logf(x: float) -> float:
    return float(lib().orc_logf(ctypes.c_float(x)))


def exp_array(x) -> np.ndarray:
    x = _f32(x)
    return np.array([expf(float(v)) for v in x.ravel()], dtype=np.float32).reshape(x.shape)


def log_array(x) -> np.ndarray:
    x = _f32(x)
    return np.array([logf(float(v)) for v in x.ravel()], dtype=np.float32).reshape(x.shape)


def generate_base_anchors(hyper_params) -> np.ndarray:
    ratios = np.asarray(hyper_params["anchor_ratios"], dtype=np.float64)
    scales = np.asarray(hyper_params["anchor_scales"], dtype=np.float64)
    out = np.empty((len(scales) * len(ratios), 4), dtype=np.float32)
    lib().orc_base_anchors(ctypes.c_double(float(hyper_params["img_size"])), _p(ratios, _f64p),
                           ctypes.c_int(len(ratios)), _p(scales, _f64p), ctypes.c_int(len(scales)),
                           _p(out, _f32p))
    return out


def generate_anchors(hyper_params) -> np.ndarray:
    base = generate_base_anchors(hyper_params)
    G = int(hyper_params["feature_map_shape"])
    A = base.shape[0]
    out = np.empty((G * G * A, 4), dtype=np.float32)
    lib().orc_generate_anchors(ctypes.c_int(G), _p(base, _f32p), ctypes.c_int(A), _p(out, _f32p))
    return out


def get_bboxes_from_deltas(anchors, deltas, variances: Optional[Sequence[float]] = None) -> np.ndarray:
    """anchors (N,4) or same shape as deltas; deltas (..., N, 4)."""
    anchors, deltas = _f32(anchors), _f32(deltas)
    shape = deltas.shape
    N = shape[-2]
    B = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
    if anchors.ndim == 2:
        assert anchors.shape[0] == N
        batched = 0
    else:
        assert anchors.shape == deltas.shape
        batched = 1
    out = np.empty(shape, dtype=np.float32)
    var = None if variances is None else _f32(variances)
    lib().orc_decode(_p(anchors, _f32p), ctypes.c_int(batched), _p(deltas, _f32p), _p(var, _f32p),
                     ctypes.c_int64(B), ctypes.c_int64(N), _p(out, _f32p))
    return out


def get_deltas_from_bboxes(bboxes, gt_boxes, div_variances: Optional[Sequence[float]] = None) -> np.ndarray:
    bboxes, gt = _f32(bboxes), _f32(gt_boxes)
    shape = gt.shape
    N = shape[-2]
    B = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
    batched = int(bboxes.ndim > 2)
    out = np.empty(shape, dtype=np.float32)
    var = None if div_variances is None else _f32(div_variances)
    lib().orc_encode(_p(bboxes, _f32p), ctypes.c_int(batched), _p(gt, _f32p), _p(var, _f32p),
                     ctypes.c_int64(B), ctypes.c_int64(N), _p(out, _f32p))
    return out


def generate_iou_map(bboxes, gt_boxes) -> np.ndarray:
    bboxes, gt = _f32(bboxes), _f32(gt_boxes)
    B, M = gt.shape[0], gt.shape[1]
    batched = int(bboxes.ndim == 3)
    N = bboxes.shape[-2]
    out = np.empty((B, N, M), dtype=np.float32)
    lib().orc_iou_map(_p(bboxes, _f32p), ctypes.c_int(batched), _p(gt, _f32p), ctypes.c_int(B),
                      ctypes.c_int(N), ctypes.c_int(M), _p(out, _f32p))
    return out


def top_k(scores, k: int) -> Tuple[np.ndarray, np.ndarray]:
    scores = _f32(scores)
    B, N = scores.shape
    vals = np.empty((B, k), dtype=np.float32)
    idx = np.empty((B, k), dtype=np.int32)
    rc = lib().orc_top_k(_p(scores, _f32p), ctypes.c_int(B), ctypes.c_int(N), ctypes.c_int(k),
                         _p(vals, _f32p), _p(idx, _i32p))
    if rc != 0:
        raise ValueError("input must have at least k columns")
    return vals, idx


def combined_non_max_suppression(boxes, scores, max_output_size_per_class: int, max_total_size: int,
                                 iou_threshold: float = 0.5, score_threshold: float = float("-inf"),
                                 clip_boxes: bool = True):
    boxes, scores = _f32(boxes), _f32(scores)
    B, n, q, _ = boxes.shape
    c = scores.shape[2]
    T = int(max_total_size)
    ob = np.empty((B, T, 4), dtype=np.float32)
    os_ = np.empty((B, T), dtype=np.float32)
    oc = np.empty((B, T), dtype=np.float32)
    ov = np.empty((B,), dtype=np.int32)
    rc = lib().orc_combined_nms(_p(boxes, _f32p), _p(scores, _f32p), ctypes.c_int(B), ctypes.c_int(n),
                                ctypes.c_int(q), ctypes.c_int(c), ctypes.c_int(int(max_output_size_per_class)),
                                ctypes.c_int(T), ctypes.c_float(iou_threshold), ctypes.c_float(score_threshold),
                                ctypes.c_int(int(clip_boxes)), _p(ob, _f32p), _p(os_, _f32p), _p(oc, _f32p),
                                _p(ov, _i32p))
    if rc != 0:
        raise ValueError("bad combined_non_max_suppression arguments")
    return ob, os_, oc, ov


def crop_and_resize(image, boxes, box_ind, crop_size, extrapolation_value: float = 0.0) -> np.ndarray:
    image, boxes, box_ind = _f32(image), _f32(boxes), _i32(box_ind)
    B, H, W, C = image.shape
    n = boxes.shape[0]
    ch, cw = int(crop_size[0]), int(crop_size[1])
    out = np.empty((n, ch, cw, C), dtype=np.float32)
    rc = lib().orc_crop_and_resize(_p(image, _f32p), ctypes.c_int(B), ctypes.c_int(H), ctypes.c_int(W),
                                   ctypes.c_int(C), _p(boxes, _f32p), _p(box_ind, _i32p), ctypes.c_int(n),
                                   ctypes.c_int(ch), ctypes.c_int(cw), ctypes.c_float(extrapolation_value),
                                   _p(out, _f32p))
    if rc != 0:
        raise ValueError("box index out of range")
    return out


def roi_bbox(deltas, scores, anchors, variances, pre_nms_topn: int, post_nms_topn: int,
             iou_threshold: float, return_all: bool = False):
    """RoIBBox.call.  deltas (B,...,4) reshaped to (B,N,4); scores reshaped to (B,N)."""
    anchors = _f32(anchors)
    N = anchors.shape[0]
    deltas = _f32(deltas)
    B = deltas.shape[0]
    deltas = deltas.reshape(B, N, 4)
    scores = _f32(scores).reshape(B, N)
    var = _f32(variances)
    R = int(post_nms_topn)
    ob = np.empty((B, R, 4), dtype=np.float32)
    os_ = np.empty((B, R), dtype=np.float32)
    ov = np.empty((B,), dtype=np.int32)
    rc = lib().orc_roi_bbox(_p(deltas, _f32p), _p(scores, _f32p), _p(anchors, _f32p), _p(var, _f32p),
                            ctypes.c_int(B), ctypes.c_int(N), ctypes.c_int(int(pre_nms_topn)), ctypes.c_int(R),
                            ctypes.c_float(iou_threshold), _p(ob, _f32p), _p(os_, _f32p), _p(ov, _i32p))
    if rc != 0:
        raise ValueError("input must have at least k columns")
    return (ob, os_, ov) if return_all else ob


def roi_pooling(feature_map, rois, pooling_size) -> np.ndarray:
    fm, rois = _f32(feature_map), _f32(rois)
    B, H, W, C = fm.shape
    R = rois.shape[1]
    ph, pw = int(pooling_size[0]), int(pooling_size[1])
    out = np.empty((B, R, ph, pw, C), dtype=np.float32)
    rc = lib().orc_roi_pooling(_p(fm, _f32p), ctypes.c_int(B), ctypes.c_int(H), ctypes.c_int(W), ctypes.c_int(C),
                               _p(rois, _f32p), ctypes.c_int(R), ctypes.c_int(ph), ctypes.c_int(pw), _p(out, _f32p))
    if rc != 0:
        raise ValueError("box index out of range")
    return out


def decoder(rois, deltas, probs, variances, total_labels: int, max_total_size: int = 200,
            score_threshold: float = 0.5, iou_threshold: float = 0.5, return_valid: bool = False):
    """Decoder.call -> (boxes, labels, scores) in the reference's return order."""
    rois, deltas, probs = _f32(rois), _f32(deltas), _f32(probs)
    B, R = rois.shape[0], rois.shape[1]
    L = int(total_labels)
    var = _f32(variances)
    T = int(max_total_size)
    ob = np.empty((B, T, 4), dtype=np.float32)
    ol = np.empty((B, T), dtype=np.float32)
    os_ = np.empty((B, T), dtype=np.float32)
    ov = np.empty((B,), dtype=np.int32)
    rc = lib().orc_decoder(_p(rois, _f32p), _p(deltas, _f32p), _p(probs, _f32p), _p(var, _f32p), ctypes.c_int(B),
                           ctypes.c_int(R), ctypes.c_int(L), ctypes.c_int(T), ctypes.c_float(score_threshold),
                           ctypes.c_float(iou_threshold), _p(ob, _f32p), _p(ol, _f32p), _p(os_, _f32p), _p(ov, _i32p))
    if rc != 0:
        raise ValueError("bad decoder arguments")
    return (ob, ol, os_, ov) if return_valid else (ob, ol, os_)


def randomly_select_xyz_mask(mask, select_xyz, random_mask) -> np.ndarray:
    mask = np.ascontiguousarray(mask).astype(np.uint8)
    rnd = _i32(random_mask)
    sel = _i32(np.atleast_1d(select_xyz))
    B, N = mask.shape
    out = np.empty((B, N), dtype=np.uint8)
    lib().orc_select_mask(_p(mask, _u8p), _p(rnd, _i32p), _p(sel, _i32p), ctypes.c_int(sel.shape[0]),
                          ctypes.c_int(B), ctypes.c_int(N), _p(out, _u8p))
    return out.astype(bool)


def calculate_rpn_actual_outputs(anchors, gt_boxes, gt_labels, hyper_params, random_masks, debug: bool = False):
    anchors, gt_boxes, gt_labels = _f32(anchors), _f32(gt_boxes), _i32(gt_labels)
    rp, rn = _i32(random_masks[0]), _i32(random_masks[1])
    B, M = gt_boxes.shape[0], gt_boxes.shape[1]
    N = anchors.shape[0]
    G = int(hyper_params["feature_map_shape"])
    A = int(hyper_params["anchor_count"])
    var = _f32(hyper_params["variances"])
    od = np.empty((B, N, 4), dtype=np.float32)
    ol = np.empty((B, N), dtype=np.float32)
    dm = np.empty((B, N), dtype=np.float32) if debug else None
    dr = np.empty((B, N), dtype=np.int32) if debug else None
    dc = np.empty((B, max(M, 1)), dtype=np.int32) if debug else None
    lib().orc_rpn_targets(_p(anchors, _f32p), _p(gt_boxes, _f32p), _p(gt_labels, _i32p), _p(rp, _i32p), _p(rn, _i32p),
                          ctypes.c_int(B), ctypes.c_int(N), ctypes.c_int(M), ctypes.c_int(int(hyper_params["total_pos_bboxes"])),
                          ctypes.c_int(int(hyper_params["total_neg_bboxes"])), _p(var, _f32p), _p(od, _f32p), _p(ol, _f32p),
                          _p(dm, _f32p), _p(dr, _i32p), _p(dc, _i32p))
    labels = ol.reshape(B, G, G, A)
    if debug:
        return od, labels, dm, dr, dc[:, :M]
    return od, labels


def roi_delta(roi_bboxes, gt_boxes, gt_labels, hyper_params, random_masks):
    """models/faster_rcnn.py:206-235 (RoIDelta.call).  ``random_masks = (pos, neg)``: the (B,R) int32 random integers of
    the two ``randomly_select_xyz_mask`` calls.  Returns ``(roi_bbox_deltas (B,R*L,4), roi_bbox_labels (B,R,L))``."""
    rois, gt_boxes, gt_labels = _f32(roi_bboxes), _f32(gt_boxes), _i32(gt_labels)
    rp, rn = _i32(random_masks[0]), _i32(random_masks[1])
    B, R = rois.shape[0], rois.shape[1]
    M = gt_boxes.shape[1]
    if M < 1:
        raise ValueError("RoIDelta needs at least one (possibly padded) gt box per image")
    L = int(hyper_params["total_labels"])
    var = _f32(hyper_params["variances"])
    od = np.empty((B, R * L, 4), dtype=np.float32)
    ol = np.empty((B, R, L), dtype=np.float32)
    lib().orc_roi_delta(_p(rois, _f32p), _p(gt_boxes, _f32p), _p(gt_labels, _i32p), _p(rp, _i32p), _p(rn, _i32p),
                        ctypes.c_int(B), ctypes.c_int(R), ctypes.c_int(M), ctypes.c_int(L),
                        ctypes.c_int(int(hyper_params["total_pos_bboxes"])), ctypes.c_int(int(hyper_params["total_neg_bboxes"])),
                        _p(var, _f32p), _p(od, _f32p), _p(ol, _f32p))
    return od, ol
