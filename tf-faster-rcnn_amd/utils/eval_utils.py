"""Prediction / ground-truth matching for evaluation -- the device front half of the reference's
``utils/eval_utils.py:update_stats`` (``:69-72``) plus the host-side TP/FP bookkeeping that consumes it.

Only the tensor work (RoI x gt IoU, row max, first argmax, descending argsort) runs on the GPU; the per-detection loop
(``:79-97``) is Python over small host arrays, as in the reference.  mAP integration (``calculate_ap`` /
``calculate_mAP``) is out of scope.
"""

from __future__ import annotations

from typing import Any, Dict, Sequence, Tuple

import torch

from .. import _lib
from . import bbox_utils

ClassStats = Dict[str, Any]
StatsDict = Dict[int, ClassStats]


def init_stats(labels: Sequence[str]) -> StatsDict:
    """One counter record per foreground class (index 0 is background).  Mirrors ``utils/eval_utils.py:17-39``."""
    return {i: {"label": name, "total": 0, "tp": [], "fp": [], "scores": []} for i, name in enumerate(labels) if i != 0}


def match_predictions(pred_bboxes: torch.Tensor, gt_boxes: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """``utils/eval_utils.py:69-72`` fused: no ``(B,R,M)`` map is materialised.

    Args:
        pred_bboxes: ``(batch, total_bboxes, 4)``.
        gt_boxes: ``(batch, total_gt_boxes, 4)``, ``total_gt_boxes >= 1``.

    Returns:
        ``merged_iou_map (B,R)`` fp32 (row max, NaN never wins), ``max_indices_each_gt (B,R)`` int32 (first maximal
        gt), ``sorted_ids (B,R)`` int32 (descending IoU, ties by lower index -- ``tf.argsort`` is ``top_k`` of the row).
    """
    pred = _lib.device_tensor(pred_bboxes, torch.float32)
    device = pred.device
    gt = _lib.device_tensor(gt_boxes, torch.float32, device)
    if pred.dim() != 3 or pred.shape[2] != 4:
        raise ValueError(f"pred_bboxes must be (batch, total_bboxes, 4), got {tuple(pred.shape)}")
    B, R, _ = pred.shape
    if gt.dim() != 3 or gt.shape[0] != B or gt.shape[2] != 4 or gt.shape[1] < 1:
        raise ValueError(f"gt_boxes must be ({B}, total_gt_boxes >= 1, 4), got {tuple(gt.shape)}")
    M = gt.shape[1]
    merged = torch.empty((B, R), dtype=torch.float32, device=device)
    arg = torch.empty((B, R), dtype=torch.int32, device=device)
    with _lib.on_device(device):
        _lib.check(_lib.lib().frcnn_iou_map_reduce(_lib.ptr(pred), 1, _lib.ptr(gt), B, R, M, _lib.ptr(merged), _lib.ptr(arg),
                                                   None, None, 0, _lib.stream_ptr(device)))
    _, sorted_ids = bbox_utils.top_k(merged, R)
    return merged, arg, sorted_ids


def update_stats(pred_bboxes: torch.Tensor, pred_labels: torch.Tensor, pred_scores: torch.Tensor, gt_boxes: torch.Tensor,
                 gt_labels: torch.Tensor, stats: StatsDict) -> StatsDict:
    """Accumulate per-class TP/FP records for one batch.  Mirrors ``utils/eval_utils.py:42-98``: every gt label except
    -1 adds to its class total (``:73-78``); detections are visited in descending best-IoU order, background (label 0)
    skipped; a detection is a TP when its best IoU is >= 0.5, the classes agree and that gt has not been claimed yet in
    this image, else an FP (``:79-97``)."""
    merged, arg, order = (t.cpu().numpy() for t in match_predictions(pred_bboxes, gt_boxes))
    labels = torch.as_tensor(pred_labels).cpu().numpy()
    scores = torch.as_tensor(pred_scores).cpu().numpy()
    gtl = torch.as_tensor(gt_labels).cpu().numpy()
    for value in gtl.reshape(-1).tolist():
        if value != -1:
            stats[int(value)]["total"] += 1
    for b in range(merged.shape[0]):
        claimed = set()
        for r in order[b].tolist():
            cls = int(labels[b, r])
            if cls == 0:
                continue
            gt_id = int(arg[b, r])
            rec = stats[cls]
            rec["scores"].append(scores[b, r])
            hit = merged[b, r] >= 0.5 and cls == int(gtl[b, gt_id]) and gt_id not in claimed
            rec["tp"].append(1 if hit else 0)
            rec["fp"].append(0 if hit else 1)
            if hit:
                claimed.add(gt_id)
    return stats
