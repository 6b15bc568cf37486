"""Training-time helpers on the proposal path -- drop-in for the reference's ``utils/train_utils.py``.

Covers the hyper-parameter table, ``randomly_select_xyz_mask``, ``calculate_rpn_actual_outputs`` and the two batch
feeders built on it (``faster_rcnn_generator``, ``rpn_generator``), which compute the targets of the next batch on a
side stream.  Losses and the data pipeline are out of scope.
"""

from __future__ import annotations

import math
from collections import deque
from typing import Any, Callable, Dict, Iterable, Iterator, Optional, Sequence, Tuple

import torch

from .. import _lib

HyperParams = Dict[str, Any]

# reference utils/train_utils.py:17-30
RPN: Dict[str, HyperParams] = {
    "vgg16": {
        "img_size": 500,
        "feature_map_shape": 31,
        "anchor_ratios": [1., 2., 1. / 2.],
        "anchor_scales": [128, 256, 512],
    },
    "mobilenet_v2": {
        "img_size": 500,
        "feature_map_shape": 32,
        "anchor_ratios": [1., 2., 1. / 2.],
        "anchor_scales": [128, 256, 512],
    }
}


def get_hyper_params(backbone: str, **kwargs: Any) -> HyperParams:
    """Hyper-parameter dictionary for ``backbone``.  Mirrors reference ``utils/train_utils.py:33-56``:
    a copy of the backbone defaults, the shared defaults, then overrides that apply only to keys that
    already exist and whose new value is truthy; unknown backbones raise ``KeyError``.
    """
    hyper_params = dict(RPN[backbone])
    hyper_params["pre_nms_topn"] = 6000
    hyper_params["train_nms_topn"] = 1500
    hyper_params["test_nms_topn"] = 300
    hyper_params["nms_iou_threshold"] = 0.7
    hyper_params["total_pos_bboxes"] = 128
    hyper_params["total_neg_bboxes"] = 128
    hyper_params["pooling_size"] = (7, 7)
    hyper_params["variances"] = [0.1, 0.1, 0.2, 0.2]
    for key, value in kwargs.items():
        if key in hyper_params and value:
            hyper_params[key] = value
    hyper_params["anchor_count"] = len(hyper_params["anchor_ratios"]) * len(hyper_params["anchor_scales"])
    return hyper_params


def get_step_size(total_items: int, batch_size: int) -> int:
    """Steps needed to cover a dataset once (reference ``utils/train_utils.py:59-69``)."""
    return math.ceil(total_items / batch_size)


def randomly_select_xyz_mask(mask: torch.Tensor, select_xyz: Any,
                             random_mask: Optional[torch.Tensor] = None, seed: Optional[int] = None) -> torch.Tensor:
    """Keep up to ``select_xyz`` true elements per batch row.  Mirrors reference
    ``utils/train_utils.py:72-88``.

    The reference multiplies the mask by ``tf.random.uniform([1, 10*max(select)), int32)`` and keeps the
    elements whose rank in the descending argsort is below ``select`` (ties: lower index first).
    TF's Philox stream cannot be reproduced, so the random integers may be passed in as
    ``random_mask`` ``(batch, total_items)`` int32 (extra, optional argument); when omitted they are
    drawn in the kernel with the same range (``seed``: extra, optional; a fresh host-side draw per call
    otherwise).  Either way the call only enqueues work: ``max(select)`` is taken on the device.  Where
    ``tf.random.uniform`` would raise (``10*max(select) <= 1``) nothing is selected.

    Args:
        mask: ``(batch, total_items)`` bool.
        select_xyz: ``(1,)`` or ``(batch,)`` int32 (tensor, list or int).
    """
    m = _lib.device_tensor(mask, torch.bool)
    device = m.device
    if m.dim() != 2:
        raise ValueError(f"mask must be (batch, total_items), got {tuple(m.shape)}")
    B, N = m.shape
    sel = _lib.device_tensor(select_xyz, torch.int32, device).reshape(-1)
    if sel.numel() not in (1, B):
        raise ValueError(f"select_xyz must have 1 or batch_size entries, got {sel.numel()}")
    m8 = m.view(torch.uint8)
    out = torch.empty((B, N), dtype=torch.uint8, device=device)
    with _lib.on_device(device):
        if random_mask is None:
            if seed is None:
                seed = int(torch.randint(0, 2 ** 62, (1,)).item())  # host RNG: no device sync
            _lib.check(_lib.lib().frcnn_select_mask_seeded(_lib.ptr(m8), _lib.ptr(sel), sel.numel(), int(seed) & 0xFFFFFFFFFFFFFFFF,
                                                           B, N, _lib.ptr(out), _lib.stream_ptr(device)))
        else:
            rnd = _lib.device_tensor(random_mask, torch.int32, device)
            if rnd.shape != m.shape:
                raise ValueError("random_mask must have the shape of mask")
            _lib.check(_lib.lib().frcnn_select_mask(_lib.ptr(m8), _lib.ptr(rnd), _lib.ptr(sel), sel.numel(), B, N,
                                                    _lib.ptr(out), _lib.stream_ptr(device)))
    return out.view(torch.bool)


def calculate_rpn_actual_outputs(anchors: torch.Tensor, gt_boxes: torch.Tensor, gt_labels: torch.Tensor,
                                 hyper_params: HyperParams,
                                 random_masks: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
                                 seed: Optional[int] = None,
                                 out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
                                 stream: Optional[Any] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """RPN regression and classification targets for a batch.  Mirrors reference
    ``utils/train_utils.py:149-202``: anchors x gt IoU, row/column argmax, positives (IoU > 0.7 or best
    anchor of a valid gt) sampled to ``total_pos_bboxes``, negatives (IoU < 0.3) sampled to fill
    ``total_pos + total_neg``, labels in {+1, 0, -1}, deltas of the positives divided by ``variances``.

    Args:
        anchors: ``(total_anchors, 4)``.
        gt_boxes: ``(batch, padded_gt_boxes, 4)`` (zero padded).
        gt_labels: ``(batch, padded_gt_boxes)`` int32 (``-1`` padded).
        hyper_params: needs ``feature_map_shape``, ``anchor_count``, ``total_pos_bboxes``,
            ``total_neg_bboxes``, ``variances``.
        random_masks: optional ``(pos, neg)`` int32 tensors ``(batch, total_anchors)``: the random
            integers of the two sampling calls (extra argument, for reproducibility/parity).
        seed: seed of the in-kernel draw used when ``random_masks`` is omitted.
        out: optional preallocated ``(bbox_deltas (batch, total_anchors, 4), bbox_labels (batch, total_anchors))`` fp32
            device tensors to write into (extra argument: a feeder loop that recycles its buffers allocates nothing).
        stream: optional ``torch.cuda.Stream`` (or raw HIP stream handle) to enqueue on instead of the current stream
            (extra argument; inputs, outputs and the workspace must be safe to use on it).

    Returns:
        ``bbox_deltas (batch, total_anchors, 4)`` and
        ``bbox_labels (batch, feature_map_shape, feature_map_shape, anchor_count)``, both fp32.
    """
    anchors_t = _lib.device_tensor(anchors, torch.float32)
    device = anchors_t.device
    gt_t = _lib.device_tensor(gt_boxes, torch.float32, device)
    lab_t = _lib.device_tensor(gt_labels, torch.int32, device)
    feature_map_shape = int(hyper_params["feature_map_shape"])
    anchor_count = int(hyper_params["anchor_count"])
    total_pos = int(hyper_params["total_pos_bboxes"])
    total_neg = int(hyper_params["total_neg_bboxes"])
    variances = _lib.float4(hyper_params["variances"])
    if anchors_t.dim() != 2 or anchors_t.shape[1] != 4:
        raise ValueError(f"anchors must be (total_anchors, 4), got {tuple(anchors_t.shape)}")
    if gt_t.dim() != 3 or gt_t.shape[2] != 4:
        raise ValueError(f"gt_boxes must be (batch, padded_gt_boxes, 4), got {tuple(gt_t.shape)}")
    B, M, _ = gt_t.shape
    N = anchors_t.shape[0]
    if lab_t.shape != (B, M):
        raise ValueError("gt_labels must be (batch, padded_gt_boxes)")
    if feature_map_shape * feature_map_shape * anchor_count != N:
        raise ValueError("anchors do not match feature_map_shape**2 * anchor_count")  # the reference's reshape fails
    rp = rn = None
    if random_masks is not None:
        rp = _lib.device_tensor(random_masks[0], torch.int32, device)
        rn = _lib.device_tensor(random_masks[1], torch.int32, device)
        if rp.shape != (B, N) or rn.shape != (B, N):
            raise ValueError("random_masks must be two (batch, total_anchors) tensors")
    if seed is None:
        # a fresh draw per call, like the reference's unseeded tf.random.uniform (host RNG, no device sync)
        seed = 0 if random_masks is not None else int(torch.randint(0, 2 ** 62, (1,)).item())
    if out is None:
        deltas = torch.empty((B, N, 4), dtype=torch.float32, device=device)
        labels = torch.empty((B, N), dtype=torch.float32, device=device)
    else:
        deltas, labels = out
        for t, shape in ((deltas, (B, N, 4)), (labels, (B, N))):
            if tuple(t.shape) != shape or t.dtype != torch.float32 or t.device != device or not t.is_contiguous():
                raise ValueError(f"out must be contiguous fp32 tensors of shape ({B}, {N}, 4) and ({B}, {N}) on {device}")
    q = _lib.stream_ptr(device) if stream is None else int(getattr(stream, "cuda_stream", stream))
    L = _lib.lib()
    nbytes = L.frcnn_rpn_targets_workspace_bytes(B, N, M)
    ws = _lib.workspace(nbytes, device, q)
    with _lib.on_device(device):
        _lib.check(L.frcnn_rpn_targets(_lib.ptr(anchors_t), _lib.ptr(gt_t), _lib.ptr(lab_t), _lib.ptr(rp), _lib.ptr(rn),
                                       int(seed) & 0xFFFFFFFFFFFFFFFF, B, N, M, total_pos, total_neg, variances,
                                       _lib.ptr(deltas), _lib.ptr(labels), _lib.ptr(ws), ws.numel(), q))
    return deltas, labels.reshape(B, feature_map_shape, feature_map_shape, anchor_count)



RandomMasks = Callable[[int, int, int], Tuple[Any, Any]]


def _rpn_target_batches(dataset: Iterable[Tuple[Any, ...]], anchors: torch.Tensor, hyper_params: HyperParams,
                        prefetch: int, seed: Optional[int],
                        random_masks: Optional[RandomMasks] = None) -> Iterator[Tuple[torch.Tensor, ...]]:
    """Endless ``(img, gt_boxes, gt_labels, bbox_deltas, bbox_labels)`` stream over ``dataset`` (re-iterated forever, as
    the reference's ``while True``).  Target assignment for up to ``prefetch`` batches ahead is enqueued on side HIP
    streams (two, taken in turn, from ``prefetch = 2`` on), off the consumer's stream (the reference computes it eagerly on the Python feeder thread,
    ``utils/train_utils.py:108-112``); the consumer's current stream waits on the batch's event when it is handed over.

    The sampling integers of a batch (the reference draws them from TF's unseeded RNG, ``utils/train_utils.py:82-83``)
    come, in this order of precedence, from the dataset item itself when it is a 5-tuple
    ``(img, gt_boxes, gt_labels, rnd_pos, rnd_neg)``, from ``random_masks(batch_index, batch, total_anchors)``, or from
    the in-kernel draw seeded with ``seed + batch_index``."""
    anchors_t = _lib.device_tensor(anchors, torch.float32)
    device = anchors_t.device
    N = anchors_t.shape[0]
    # with two or more batches in flight the batches alternate between two side streams: the chip-wide IoU reductions of one
    # batch then run beside the one-workgroup-per-image target kernel of the batch before it
    sides = [torch.cuda.Stream(device) for _ in range(2 if prefetch >= 2 else 1)]
    pending: deque = deque()
    counter = 0
    while True:
        produced = False
        for item in dataset:
            produced = True
            if len(item) == 5:
                img, gt_boxes, gt_labels, rnd_pos, rnd_neg = item
                masks: Optional[Tuple[Any, Any]] = (rnd_pos, rnd_neg)
            else:
                img, gt_boxes, gt_labels = item
                masks = None
            gt_t = _lib.device_tensor(gt_boxes, torch.float32, device)
            lab_t = _lib.device_tensor(gt_labels, torch.int32, device)
            if masks is None and random_masks is not None:
                masks = random_masks(counter, gt_t.shape[0], N)
            if masks is not None:  # brought to the device on the caller's stream, like the gt tensors
                masks = (_lib.device_tensor(masks[0], torch.int32, device), _lib.device_tensor(masks[1], torch.int32, device))
            side = sides[counter % len(sides)]
            ready = _lib.HipEvent()
            ready.record(torch.cuda.current_stream(device).cuda_stream)  # the gt tensors come from the caller's stream
            ready.wait(side.cuda_stream)
            with torch.cuda.stream(side):
                deltas, labels = calculate_rpn_actual_outputs(anchors_t, gt_t, lab_t, hyper_params, random_masks=masks,
                                                              seed=None if seed is None else seed + counter)
            # everything the target kernels read was allocated on the caller's stream and is read on the side stream: the
            # caching allocator must not hand those blocks out again before the side stream has passed this point
            for t in (gt_t, lab_t) + (tuple(masks) if masks is not None else ()):
                t.record_stream(side)
            done = _lib.HipEvent()
            done.record(side.cuda_stream)
            counter += 1
            pending.append((img, gt_t, lab_t, deltas, labels, done, masks))
            if len(pending) > prefetch:
                yield _hand_over(pending.popleft(), device)
        if not produced:
            raise ValueError("the dataset yielded no batch")
        # the reference's generators never end; keep `prefetch` batches in flight across passes


def _hand_over(item: Tuple[Any, ...], device: torch.device) -> Tuple[torch.Tensor, ...]:
    img, gt_t, lab_t, deltas, labels, done, masks = item
    cur = torch.cuda.current_stream(device)
    done.wait(cur.cuda_stream)
    deltas.record_stream(cur)  # allocated on the side stream, consumed on the caller's
    labels.record_stream(cur)
    if masks is not None:  # allocated on the caller's stream, read on the side stream: safe to free once `done` has passed
        masks[0].record_stream(cur)
        masks[1].record_stream(cur)
    return img, gt_t, lab_t, deltas, labels


def faster_rcnn_generator(dataset: Iterable[Tuple[Any, ...]], anchors: torch.Tensor, hyper_params: HyperParams,
                          prefetch: int = 1, seed: Optional[int] = None,
                          random_masks: Optional[RandomMasks] = None) -> Iterator[Tuple[Tuple[torch.Tensor, ...], Tuple[()]]]:
    """Yield ``((img, gt_boxes, gt_labels, bbox_deltas, bbox_labels), ())`` forever.  Mirrors reference
    ``utils/train_utils.py:91-112``; ``dataset`` is any re-iterable of padded ``(img, gt_boxes, gt_labels)`` batches.
    Extra arguments: ``prefetch`` batches of targets kept in flight on a side stream; ``seed`` makes the in-kernel sampling
    reproducible (batch ``i`` uses ``seed + i``); ``random_masks(i, batch, total_anchors) -> (pos, neg)`` supplies the
    sampling integers of batch ``i`` instead (int32 ``(batch, total_anchors)`` each), as do dataset items that carry them
    as a 4th and 5th element."""
    for batch in _rpn_target_batches(dataset, anchors, hyper_params, max(0, int(prefetch)), seed, random_masks):
        yield batch, ()


def rpn_generator(dataset: Iterable[Tuple[Any, ...]], anchors: torch.Tensor, hyper_params: HyperParams,
                  prefetch: int = 1, seed: Optional[int] = None,
                  random_masks: Optional[RandomMasks] = None) -> Iterator[Tuple[Any, Tuple[torch.Tensor, torch.Tensor]]]:
    """Yield ``(img, (bbox_deltas, bbox_labels))`` forever.  Mirrors reference ``utils/train_utils.py:115-136``; extra
    arguments as ``faster_rcnn_generator``."""
    for img, _, _, deltas, labels in _rpn_target_batches(dataset, anchors, hyper_params, max(0, int(prefetch)), seed,
                                                         random_masks):
        yield img, (deltas, labels)
