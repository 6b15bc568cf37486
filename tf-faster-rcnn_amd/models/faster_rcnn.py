"""Proposal / RoI layers -- drop-in for ``RoIBBox``, ``RoIDelta``, ``RoIPooling`` and ``Decoder`` of the
reference's ``models/faster_rcnn.py``.

Same constructors, ``call(inputs)`` contracts and ``get_config()`` keys as the Keras layers they
replace; each ``call`` is one fused C-ABI entry point of ``libfrcnn_hip.so``.  The conv backbone,
the RPN head and the dense heads are out of scope: their outputs enter as tensors.
"""

from __future__ import annotations

from typing import Any, Dict, Optional, Sequence, Tuple

import torch

from .. import _lib

HyperParams = Dict[str, Any]
DecodedPredictions = Tuple[torch.Tensor, torch.Tensor, torch.Tensor]


class Layer:
    """Minimal stand-in for ``tf.keras.layers.Layer``: a name, ``__call__ -> call`` and ``get_config``."""

    _counters: Dict[str, int] = {}

    def __init__(self, name: str = None, trainable: bool = True, dtype: str = "float32", **kwargs: Any) -> None:
        if kwargs:
            raise TypeError(f"Keyword argument not understood: {sorted(kwargs)[0]}")
        if name is None:
            base = self.__class__.__name__.lower()
            idx = Layer._counters.get(base, 0)
            Layer._counters[base] = idx + 1
            name = base if idx == 0 else f"{base}_{idx}"
        self.name = name
        self.trainable = trainable
        self.dtype = dtype

    def get_config(self) -> Dict[str, Any]:
        return {"name": self.name, "trainable": self.trainable, "dtype": self.dtype}

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "Layer":
        return cls(**config)

    def call(self, inputs: Sequence[torch.Tensor]) -> Any:  # pragma: no cover - abstract
        raise NotImplementedError

    def __call__(self, inputs: Sequence[torch.Tensor], **kwargs: Any) -> Any:
        return self.call(inputs, **kwargs)


class Decoder(Layer):
    """Decode Faster R-CNN class probabilities and deltas into final detections.

    Mirrors reference ``models/faster_rcnn.py:19-91``.
    """

    def __init__(self, variances: Sequence[float], total_labels: int, max_total_size: int = 200,
                 score_threshold: float = 0.5, **kwargs: Any) -> None:
        super().__init__(**kwargs)
        self.variances = variances
        self.total_labels = total_labels
        self.max_total_size = max_total_size
        self.score_threshold = score_threshold

    def get_config(self) -> Dict[str, Any]:
        config = super().get_config()
        config.update({
            "variances": self.variances,
            "total_labels": self.total_labels,
            "max_total_size": self.max_total_size,
            "score_threshold": self.score_threshold,
        })
        return config

    def call(self, inputs: Sequence[torch.Tensor]) -> DecodedPredictions:
        """``inputs = [roi_bboxes (B,R,4), pred_deltas (B,R,L*4), pred_label_probs (B,R,L)]``.

        Per class: ``deltas * variances`` (``:78-79``), decode against the RoI (``:80-81``); rows whose
        argmax is background are zeroed (``:82-83``); combined NMS with 200 per class / 200 total,
        ``score_threshold``, TF's default ``iou_threshold=0.5`` and ``clip_boxes=True`` (``:84-90``).

        Returns:
            ``(final_bboxes (B,T,4), final_labels (B,T) fp32, final_scores (B,T))`` -- the reference's
            return order (``:91``).
        """
        rois = _lib.device_tensor(inputs[0], torch.float32)
        device = rois.device
        deltas = _lib.device_tensor(inputs[1], torch.float32, device)
        probs = _lib.device_tensor(inputs[2], torch.float32, device)
        L = int(self.total_labels)
        if rois.dim() != 3 or rois.shape[2] != 4:
            raise ValueError(f"roi_bboxes must be (batch, total_bboxes, 4), got {tuple(rois.shape)}")
        B, R, _ = rois.shape
        if deltas.numel() != B * R * L * 4:
            raise ValueError(f"pred_deltas {tuple(deltas.shape)} cannot be reshaped to ({B}, {R}, {L}, 4)")
        if probs.shape != (B, R, L):
            raise ValueError(f"pred_label_probs must be ({B}, {R}, {L}), got {tuple(probs.shape)}")
        T = int(self.max_total_size)
        boxes = torch.empty((B, T, 4), dtype=torch.float32, device=device)
        labels = torch.empty((B, T), dtype=torch.float32, device=device)
        scores = torch.empty((B, T), dtype=torch.float32, device=device)
        lib = _lib.lib()
        ws = _lib.workspace(lib.frcnn_decoder_workspace_bytes(B, R, L, T), device)
        with _lib.on_device(device):
            _lib.check(lib.frcnn_decoder(_lib.ptr(rois), _lib.ptr(deltas), _lib.ptr(probs), _lib.float4(self.variances),
                                         B, R, L, T, float(self.score_threshold), 0.5, _lib.ptr(boxes), _lib.ptr(labels),
                                         _lib.ptr(scores), None, _lib.ptr(ws), ws.numel(), _lib.stream_ptr(device)))
        return boxes, labels, scores


class RoIBBox(Layer):
    """Convert RPN outputs into a fixed number of region proposals.

    Mirrors reference ``models/faster_rcnn.py:94-166``.
    """

    def __init__(self, anchors: torch.Tensor, mode: str, hyper_params: HyperParams, **kwargs: Any) -> None:
        super().__init__(**kwargs)
        self.hyper_params = hyper_params
        self.mode = mode
        self.anchors = _lib.device_tensor(anchors, torch.float32)

    def get_config(self) -> Dict[str, Any]:
        config = super().get_config()
        config.update({"hyper_params": self.hyper_params, "anchors": self.anchors.cpu().numpy(), "mode": self.mode})
        return config

    def call(self, inputs: Sequence[torch.Tensor]) -> torch.Tensor:
        """``inputs = [rpn_bbox_deltas (B,G,G,A*4), rpn_labels (B,G,G,A)]`` (any shape with the right
        element count; they are viewed as ``(B,N,4)`` / ``(B,N)`` like ``:147-148``).

        ``deltas * variances`` and decode (``:149-150``), ``top_k(pre_nms_topn)`` (``:153``), gather
        (``:154-155``), single-class NMS with ``post_nms_topn`` outputs at ``nms_iou_threshold``
        (``:158-164``).  Output rows are score-ordered, clipped to ``[0,1]`` and zero padded.

        Returns:
            ``(B, post_nms_topn, 4)`` fp32 (no gradient flows, like ``tf.stop_gradient`` at ``:166``).
        """
        hp = self.hyper_params
        pre_nms_topn = int(hp["pre_nms_topn"])
        post_nms_topn = int(hp["train_nms_topn"] if self.mode == "training" else hp["test_nms_topn"])
        nms_iou_threshold = float(hp["nms_iou_threshold"])
        device = self.anchors.device
        deltas = _lib.device_tensor(inputs[0], torch.float32, device)
        scores = _lib.device_tensor(inputs[1], torch.float32, device)
        total_anchors = self.anchors.shape[0]
        B = deltas.shape[0]
        if deltas.numel() != B * total_anchors * 4 or scores.numel() != B * total_anchors:
            raise ValueError(
                f"rpn outputs {tuple(deltas.shape)} / {tuple(scores.shape)} do not match {total_anchors} anchors")
        out = torch.empty((B, post_nms_topn, 4), dtype=torch.float32, device=device)
        lib = _lib.lib()
        ws = _lib.workspace(lib.frcnn_roi_bbox_workspace_bytes(B, total_anchors, pre_nms_topn, post_nms_topn), device)
        with _lib.on_device(device):
            _lib.check(lib.frcnn_roi_bbox(_lib.ptr(deltas), _lib.ptr(scores), _lib.ptr(self.anchors),
                                          _lib.float4(hp["variances"]), B, total_anchors, pre_nms_topn, post_nms_topn,
                                          nms_iou_threshold, _lib.ptr(out), None, None, _lib.ptr(ws), ws.numel(),
                                          _lib.stream_ptr(device)))
        return out


class RoIDelta(Layer):
    """Build Faster R-CNN regression and classification targets for sampled RoIs.

    Mirrors reference ``models/faster_rcnn.py:169-235``.
    """

    def __init__(self, hyper_params: HyperParams, **kwargs: Any) -> None:
        super().__init__(**kwargs)
        self.hyper_params = hyper_params

    def get_config(self) -> Dict[str, Any]:
        config = super().get_config()
        config.update({"hyper_params": self.hyper_params})
        return config

    def call(self, inputs: Sequence[torch.Tensor], random_masks: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
             seed: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """``inputs = [roi_bboxes (B,R,4), gt_boxes (B,M,4), gt_labels (B,M) int32]``.

        RoI x gt IoU (``:215``), per-RoI best gt and max IoU (``:216-217``), positives ``> 0.5`` sampled to
        ``total_pos_bboxes`` (``:218-219``), negatives in ``(0.1, 0.5)`` sampled to ``total_neg_bboxes`` (``:220-221``),
        gathered gt boxes / labels (``:222-229``; positives keep their gt label, negatives get label 0, the rest -1),
        ``get_deltas_from_bboxes / variances`` (``:230``), ``one_hot`` labels and the per-class delta scatter
        (``:231-234``).

        Extra arguments (the reference draws from TF's unseeded RNG): ``random_masks = (pos, neg)``, two int32
        ``(B,R)`` tensors holding the random integers of the two sampling calls; ``seed`` for the in-kernel draw
        otherwise.

        Returns:
            ``(roi_bbox_deltas (B, R*total_labels, 4), roi_bbox_labels (B, R, total_labels))``, fp32, no gradient
            (``:235``).
        """
        rois = _lib.device_tensor(inputs[0], torch.float32)
        device = rois.device
        gt = _lib.device_tensor(inputs[1], torch.float32, device)
        lab = _lib.device_tensor(inputs[2], torch.int32, device)
        hp = self.hyper_params
        L = int(hp["total_labels"])
        total_pos, total_neg = int(hp["total_pos_bboxes"]), int(hp["total_neg_bboxes"])
        if rois.dim() != 3 or rois.shape[2] != 4:
            raise ValueError(f"roi_bboxes must be (batch, total_bboxes, 4), got {tuple(rois.shape)}")
        B, R, _ = rois.shape
        if gt.dim() != 3 or gt.shape[0] != B or gt.shape[2] != 4:
            raise ValueError(f"gt_boxes must be ({B}, padded_gt_boxes, 4), got {tuple(gt.shape)}")
        M = gt.shape[1]
        if M < 1:
            raise ValueError("gt_boxes needs at least one (possibly padded) box per image")  # TF: gather from an empty axis
        if tuple(lab.shape) != (B, M):
            raise ValueError(f"gt_labels must be ({B}, {M}), got {tuple(lab.shape)}")
        rp = rn = None
        if random_masks is not None:
            rp = _lib.device_tensor(random_masks[0], torch.int32, device)
            rn = _lib.device_tensor(random_masks[1], torch.int32, device)
            if tuple(rp.shape) != (B, R) or tuple(rn.shape) != (B, R):
                raise ValueError("random_masks must be two (batch, total_bboxes) tensors")
        if seed is None:
            seed = 0 if random_masks is not None else int(torch.randint(0, 2 ** 62, (1,)).item())
        deltas = torch.empty((B, R * L, 4), dtype=torch.float32, device=device)
        labels = torch.empty((B, R, L), dtype=torch.float32, device=device)
        lib = _lib.lib()
        ws = _lib.workspace(lib.frcnn_roi_delta_workspace_bytes(B, R, M), device)
        with _lib.on_device(device):
            _lib.check(lib.frcnn_roi_delta(_lib.ptr(rois), _lib.ptr(gt), _lib.ptr(lab), _lib.ptr(rp), _lib.ptr(rn),
                                           int(seed) & 0xFFFFFFFFFFFFFFFF, B, R, M, L, total_pos, total_neg,
                                           _lib.float4(hp["variances"]), _lib.ptr(deltas), _lib.ptr(labels), _lib.ptr(ws),
                                           ws.numel(), _lib.stream_ptr(device)))
        return deltas, labels


class RoIPooling(Layer):
    """Crop and resize feature-map regions into a fixed pooling shape.

    Mirrors reference ``models/faster_rcnn.py:238-293``.
    """

    def __init__(self, hyper_params: HyperParams, **kwargs: Any) -> None:
        super().__init__(**kwargs)
        self.hyper_params = hyper_params

    def get_config(self) -> Dict[str, Any]:
        config = super().get_config()
        config.update({"hyper_params": self.hyper_params})
        return config

    def call(self, inputs: Sequence[torch.Tensor]) -> torch.Tensor:
        """``inputs = [feature_map (B,H,W,C) NHWC, roi_bboxes (B,R,4)]``.

        Box ``b*R + r`` samples image ``b`` (``:280-282``); bilinear ``crop_and_resize`` to
        ``pooling_size`` with extrapolation value 0 (``:283-288``).

        Returns:
            ``(B, R, pooled_h, pooled_w, C)`` fp32, ``[y][x][c]`` innermost (``:289-292``).
        """
        fm = _lib.device_tensor(inputs[0], torch.float32)
        device = fm.device
        rois = _lib.device_tensor(inputs[1], torch.float32, device)
        ph, pw = (int(v) for v in self.hyper_params["pooling_size"])
        if fm.dim() != 4:
            raise ValueError(f"feature_map must be (batch, H, W, C), got {tuple(fm.shape)}")
        if rois.dim() != 3 or rois.shape[2] != 4 or rois.shape[0] != fm.shape[0]:
            raise ValueError(f"roi_bboxes must be (batch, total_bboxes, 4), got {tuple(rois.shape)}")
        B, H, W, C = fm.shape
        R = rois.shape[1]
        out = torch.empty((B, R, ph, pw, C), dtype=torch.float32, device=device)
        if B * R:
            lib = _lib.lib()
            ws = _lib.workspace(lib.frcnn_roi_pooling_workspace_bytes(B, R), device)
            with _lib.on_device(device):
                _lib.check(lib.frcnn_roi_pooling(_lib.ptr(fm), B, H, W, C, _lib.ptr(rois), R, ph, pw, _lib.ptr(out), _lib.ptr(ws),
                                                 ws.numel(), _lib.stream_ptr(device)))
        return out
