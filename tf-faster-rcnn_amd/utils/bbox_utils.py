"""Bounding-box helpers -- drop-in for the reference's ``utils/bbox_utils.py``.

Same function names, argument meaning and return layouts as the reference; tensors are
``torch.Tensor`` on a ROCm device instead of ``tf.Tensor`` and the arithmetic runs in the HIP
kernels of ``libfrcnn_hip.so``.  Boxes are ``[y1, x1, y2, x2]`` normalised to ``[0, 1]``.
"""

from __future__ import annotations

from typing import Any, Dict, NamedTuple, Optional, Sequence, Tuple

import numpy as np
import torch

from .. import _lib

HyperParams = Dict[str, Any]


def generate_base_anchors(hyper_params: HyperParams) -> torch.Tensor:
    """Anchor templates centred at the origin, one per (scale, ratio).

    Mirrors reference ``utils/bbox_utils.py:13-34``: ``scale /= img_size`` and ``scale ** 2 / ratio``
    are Python floats (fp64); ``tf.sqrt`` receives that float as an fp32 tensor; ``w * ratio`` and
    the halvings are fp32.  This is 9 boxes of host arithmetic, done on the host like the reference.

    Returns:
        ``(anchor_count, 4)`` fp32 device tensor.
    """
    img_size = hyper_params["img_size"]
    base_anchors = []
    for scale in hyper_params["anchor_scales"]:
        scale = scale / img_size
        for ratio in hyper_params["anchor_ratios"]:
            w = np.sqrt(np.float32(scale ** 2 / ratio))
            h = np.float32(w * np.float32(ratio))
            base_anchors.append([-h / np.float32(2), -w / np.float32(2), h / np.float32(2), w / np.float32(2)])
    device = _lib.require_gpu()
    return torch.tensor(np.asarray(base_anchors, dtype=np.float32), device=device)


def generate_anchors(hyper_params: HyperParams) -> torch.Tensor:
    """Broadcast the base anchors over the feature-map grid and clip to ``[0, 1]``.

    Mirrors reference ``utils/bbox_utils.py:37-57``.  The grid coordinates
    ``range(G) / G + stride / 2`` are evaluated in fp64 and cast once to fp32 (``:49``) on the host;
    the broadcast add + clip over ``G*G*A`` anchors is ``frcnn_generate_anchors``.

    Returns:
        ``(G*G*anchor_count, 4)`` fp32 device tensor, cell-major (``gy*G+gx``), anchor-minor.
    """
    feature_map_shape = int(hyper_params["feature_map_shape"])
    stride = 1 / feature_map_shape
    grid_coords = (np.arange(0, feature_map_shape) / feature_map_shape + stride / 2).astype(np.float32)
    base_anchors = generate_base_anchors(hyper_params)
    device = base_anchors.device
    coords = torch.tensor(grid_coords, device=device)
    count = base_anchors.shape[0]
    out = torch.empty((feature_map_shape * feature_map_shape * count, 4), dtype=torch.float32, device=device)
    with _lib.on_device(device):
        _lib.check(_lib.lib().frcnn_generate_anchors(_lib.ptr(base_anchors), count, _lib.ptr(coords), feature_map_shape,
                                                     _lib.ptr(out), _lib.stream_ptr(device)))
    return out


class CombinedNMSOutput(NamedTuple):
    """Return value of ``tf.image.combined_non_max_suppression``."""

    nmsed_boxes: torch.Tensor
    nmsed_scores: torch.Tensor
    nmsed_classes: torch.Tensor
    valid_detections: torch.Tensor


def non_max_suppression(pred_bboxes: torch.Tensor, pred_labels: torch.Tensor, **kwargs: Any) -> CombinedNMSOutput:
    """Batched, per-class greedy NMS == ``tf.image.combined_non_max_suppression``.

    Mirrors reference ``utils/bbox_utils.py:60-83`` (a pure pass-through of ``**kwargs``).

    Args:
        pred_bboxes: ``(batch, total_bboxes, q, 4)`` with ``q`` 1 or ``total_labels``.
        pred_labels: ``(batch, total_bboxes, total_labels)`` scores.
        **kwargs: ``max_output_size_per_class``, ``max_total_size``, ``iou_threshold=0.5``,
            ``score_threshold=-inf``, ``pad_per_class=False``, ``clip_boxes=True``, ``name``.

    Returns:
        ``(boxes (B,T,4), scores (B,T), classes (B,T) fp32, valid (B,) int32)``, ``T = max_total_size``.
    """
    allowed = {"max_output_size_per_class", "max_total_size", "iou_threshold", "score_threshold",
               "pad_per_class", "clip_boxes", "name"}
    unknown = set(kwargs) - allowed
    if unknown:
        raise TypeError(f"combined_non_max_suppression() got unexpected keyword arguments {sorted(unknown)}")
    if "max_output_size_per_class" not in kwargs or "max_total_size" not in kwargs:
        raise TypeError("combined_non_max_suppression() missing max_output_size_per_class / max_total_size")
    if kwargs.get("pad_per_class", False):
        raise NotImplementedError("pad_per_class=True is not used by the reference and is not implemented")
    max_per_class = int(kwargs["max_output_size_per_class"])
    max_total = int(kwargs["max_total_size"])
    iou_threshold = float(kwargs.get("iou_threshold", 0.5))
    score_threshold = float(kwargs.get("score_threshold", float("-inf")))
    clip_boxes = bool(kwargs.get("clip_boxes", True))

    boxes = _lib.device_tensor(pred_bboxes, torch.float32)
    device = boxes.device
    scores = _lib.device_tensor(pred_labels, torch.float32, device)
    if boxes.dim() != 4 or boxes.shape[3] != 4:
        raise ValueError(f"boxes must be 4-D [batch_size, num_boxes, q, 4], got {tuple(boxes.shape)}")
    if scores.dim() != 3:
        raise ValueError(f"scores must be 3-D [batch_size, num_boxes, num_classes], got {tuple(scores.shape)}")
    B, n, q, _ = boxes.shape
    if scores.shape[0] != B or scores.shape[1] != n:
        raise ValueError("boxes and scores must have the same batch size and number of boxes")
    c = scores.shape[2]
    out_boxes = torch.empty((B, max_total, 4), dtype=torch.float32, device=device)
    out_scores = torch.empty((B, max_total), dtype=torch.float32, device=device)
    out_classes = torch.empty((B, max_total), dtype=torch.float32, device=device)
    out_valid = torch.empty((B,), dtype=torch.int32, device=device)
    L = _lib.lib()
    nbytes = L.frcnn_combined_nms_workspace_bytes(B, n, q, c, max_per_class, max_total)
    ws = _lib.workspace(nbytes, device)
    with _lib.on_device(device):
        _lib.check(L.frcnn_combined_nms(_lib.ptr(boxes), _lib.ptr(scores), B, n, q, c, max_per_class, max_total,
                                        iou_threshold, score_threshold, int(clip_boxes), _lib.ptr(out_boxes),
                                        _lib.ptr(out_scores), _lib.ptr(out_classes), _lib.ptr(out_valid), _lib.ptr(ws),
                                        ws.numel(), _lib.stream_ptr(device)))
    return CombinedNMSOutput(out_boxes, out_scores, out_classes, out_valid)


def _codec_layout(first: torch.Tensor, second: torch.Tensor) -> Tuple[torch.Tensor, int, int, int]:
    """Resolve the broadcast of ``first`` (anchors/bboxes) against ``second`` (deltas/gt boxes).

    Returns ``(first_contiguous, batched_flag, B, N)`` for tensors viewed as ``(B, N, 4)``.
    """
    if second.dim() < 1 or second.shape[-1] != 4 or first.shape[-1] != 4:
        raise ValueError("box tensors must have a last dimension of 4")
    if first.shape == second.shape:
        total = second.numel() // 4
        return first, 0, 1, total
    if first.dim() == 2 and second.dim() >= 2 and second.shape[-2] == first.shape[0]:
        n = first.shape[0]
        return first, 0, second.numel() // (4 * n) if n else 0, n
    # general broadcasting: materialise the first operand (rare; not on the reference's paths)
    expanded = torch.broadcast_to(first, second.shape).contiguous()
    return expanded, 0, 1, second.numel() // 4


def get_bboxes_from_deltas(anchors: torch.Tensor, deltas: torch.Tensor,
                           variances: Optional[Sequence[float]] = None) -> torch.Tensor:
    """Decode box deltas back into boxes.  Mirrors reference ``utils/bbox_utils.py:86-108``.

    ``variances`` (not in the reference signature) fuses the callers' ``deltas *= variances``
    (``models/faster_rcnn.py:149,79``) into the same kernel; leave it ``None`` for the plain function.
    """
    deltas_t = _lib.device_tensor(deltas, torch.float32)
    device = deltas_t.device
    anchors_t = _lib.device_tensor(anchors, torch.float32, device)
    if anchors_t.dim() > deltas_t.dim():
        deltas_t = torch.broadcast_to(deltas_t, anchors_t.shape).contiguous()
    first, batched, B, N = _codec_layout(anchors_t, deltas_t)
    out = torch.empty_like(deltas_t)
    var = None if variances is None else _lib.float4(variances)
    with _lib.on_device(device):
        _lib.check(_lib.lib().frcnn_decode_boxes(_lib.ptr(first), batched, _lib.ptr(deltas_t), var, B, N, _lib.ptr(out),
                                                 _lib.stream_ptr(device)))
    return out


def get_deltas_from_bboxes(bboxes: torch.Tensor, gt_boxes: torch.Tensor,
                           variances: Optional[Sequence[float]] = None) -> torch.Tensor:
    """Encode ground-truth boxes relative to boxes.  Mirrors reference ``utils/bbox_utils.py:111-135``.

    ``variances`` (not in the reference signature) fuses the callers' ``/ variances``
    (``utils/train_utils.py:200``); leave it ``None`` for the plain function.
    """
    gt_t = _lib.device_tensor(gt_boxes, torch.float32)
    device = gt_t.device
    bboxes_t = _lib.device_tensor(bboxes, torch.float32, device)
    if bboxes_t.dim() > gt_t.dim():
        gt_t = torch.broadcast_to(gt_t, bboxes_t.shape).contiguous()
    first, batched, B, N = _codec_layout(bboxes_t, gt_t)
    out = torch.empty_like(gt_t)
    var = None if variances is None else _lib.float4(variances)
    with _lib.on_device(device):
        _lib.check(_lib.lib().frcnn_encode_boxes(_lib.ptr(first), batched, _lib.ptr(gt_t), var, B, N, _lib.ptr(out),
                                                 _lib.stream_ptr(device)))
    return out


def generate_iou_map(bboxes: torch.Tensor, gt_boxes: torch.Tensor) -> torch.Tensor:
    """Pairwise IoU map.  Mirrors reference ``utils/bbox_utils.py:138-160``.

    Args:
        bboxes: ``(total_bboxes, 4)`` (shared by the batch, as at ``utils/train_utils.py:176``) or
            ``(batch, total_bboxes, 4)``.
        gt_boxes: ``(batch, total_gt_boxes, 4)``.

    Returns:
        ``(batch, total_bboxes, total_gt_boxes)`` fp32.  No epsilon: ``0/0`` is NaN as in the reference.
    """
    gt_t = _lib.device_tensor(gt_boxes, torch.float32)
    device = gt_t.device
    bboxes_t = _lib.device_tensor(bboxes, torch.float32, device)
    if gt_t.dim() != 3 or gt_t.shape[-1] != 4:
        raise ValueError(f"gt_boxes must be (batch, total_gt_boxes, 4), got {tuple(gt_t.shape)}")
    B, M, _ = gt_t.shape
    if bboxes_t.dim() == 2:
        batched = 0
    elif bboxes_t.dim() == 3 and bboxes_t.shape[0] == B:
        batched = 1
    else:
        raise ValueError(f"bboxes must be (N,4) or (batch,N,4), got {tuple(bboxes_t.shape)}")
    N = bboxes_t.shape[-2]
    out = torch.empty((B, N, M), dtype=torch.float32, device=device)
    with _lib.on_device(device):
        _lib.check(_lib.lib().frcnn_iou_map(_lib.ptr(bboxes_t), batched, _lib.ptr(gt_t), B, N, M, _lib.ptr(out),
                                            _lib.stream_ptr(device)))
    return out


def top_k(values: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """``tf.nn.top_k(values, k, sorted=True)`` over the last axis of a 2-D tensor.

    The reference calls it at ``models/faster_rcnn.py:153`` and ``rpn_predictor.py:64``.
    Descending values; equal values by ascending index.  ``k`` larger than the row length raises.
    """
    v = _lib.device_tensor(values, torch.float32)
    if v.dim() != 2:
        raise ValueError(f"top_k expects a 2-D tensor, got {tuple(v.shape)}")
    B, N = v.shape
    k = int(k)
    out_v = torch.empty((B, k), dtype=torch.float32, device=v.device)
    out_i = torch.empty((B, k), dtype=torch.int32, device=v.device)
    with _lib.on_device(v.device):
        _lib.check(_lib.lib().frcnn_top_k(_lib.ptr(v), B, N, k, _lib.ptr(out_v), _lib.ptr(out_i), _lib.stream_ptr(v.device)))
    return out_v, out_i


def crop_and_resize(image: torch.Tensor, boxes: torch.Tensor, box_indices: torch.Tensor,
                    crop_size: Sequence[int], extrapolation_value: float = 0.0) -> torch.Tensor:
    """``tf.image.crop_and_resize(method="bilinear")`` as called at ``models/faster_rcnn.py:283-288``.

    Args:
        image: ``(batch, H, W, C)`` NHWC.
        boxes: ``(n, 4)``.
        box_indices: ``(n,)`` int32 image index of each box; out of range raises like TF.
        crop_size: ``(crop_h, crop_w)``.

    Returns:
        ``(n, crop_h, crop_w, C)`` fp32.
    """
    img = _lib.device_tensor(image, torch.float32)
    device = img.device
    bx = _lib.device_tensor(boxes, torch.float32, device)
    ind = _lib.device_tensor(box_indices, torch.int32, device)
    if img.dim() != 4:
        raise ValueError(f"image must be 4-D, got {tuple(img.shape)}")
    if bx.dim() != 2 or bx.shape[1] != 4:
        raise ValueError(f"boxes must be (n, 4), got {tuple(bx.shape)}")
    n = bx.shape[0]
    if ind.shape != (n,):
        raise ValueError("box_indices must have shape (n,)")
    B, H, W, C = img.shape
    if n and (int(ind.min()) < 0 or int(ind.max()) >= B):
        raise ValueError("box_index has values outside [0, batch_size)")
    ch, cw = int(crop_size[0]), int(crop_size[1])
    out = torch.empty((n, ch, cw, C), dtype=torch.float32, device=device)
    with _lib.on_device(device):
        _lib.check(_lib.lib().frcnn_crop_and_resize(_lib.ptr(img), B, H, W, C, _lib.ptr(bx), _lib.ptr(ind), n, 0, ch, cw,
                                                    float(extrapolation_value), _lib.ptr(out), _lib.stream_ptr(device)))
    return out

