"""Stage pipeline over HIP streams for the RoI path: RoIBBox | RoIPooling (+ heads) | Decoder.

The proposal and Decoder kernels are latency bound (one workgroup per image: 16 of 256 CUs at batch
16) while crop_and_resize is HBM-write bound and fills the chip.  Run one after the other on one
stream they leave the GPU mostly idle for half of every step.  This runner keeps ``depth`` batches
in flight: each stage owns a HIP stream, events carry the per-batch dependency
``RoIBBox -> RoIPooling -> heads -> Decoder`` of the reference model (models/faster_rcnn.py:318-319,
361-365), and the stages of consecutive batches overlap (batch i+1's proposals and batch i-1's
Decoder run under batch i's crop).  Per batch the arithmetic, the kernels and the results are exactly
those of the three layers called in sequence (tests/test_gpu_parity.py::test_pipeline_*).

Measured alternatives (BASELINE config 2, MI355X; ablations with FRCNN_PIPE_SKIP): the crop chain alone sustains 125 us
per batch on one stream and 105 us when two streams alternate batches, but with the proposal and Decoder kernels present
two crop lanes give no gain (136 us per batch either way: the single-workgroup kernels slow down under two overlapping
crops and become the critical chain); confining them to reserved CUs with hipExtStreamCreateWithCUMask makes every
cross-stream event wait cost ~200 us; a persistent crop kernel of 1024-thread workgroups that owns a fixed number of CUs
(so that the small kernels always find free ones and two crop launches can run side by side) is 10 % slower than the
box-per-workgroup kernel on the whole chip and its two-lane form reached only 180 us per batch.  Hence one pooling stream
and the plain kernel.

Host cost per batch is three C-ABI calls and a handful of event operations: outputs live in a ring of
``depth`` preallocated slots, so nothing is allocated after the first batch of a given shape.

Train mode (``roi_delta=`` instead of ``decoder=``) runs the reference's training forward, models/faster_rcnn.py:327-335:
RoIBBox (``train_nms_topn`` RoIs, the whole-chip bitmask schedule from 512 on) -> RoIPooling and, beside it, RoIDelta,
which needs the RoIs and the ground truth but not the pooled features.  The proposal stream and the third stream form two
lanes that take the batches in turn, each running proposals -> RoIDelta of its batch in stream order, while the pooling
stream crops the batch before: the bitmask schedule's chain of eight launches takes longer beside a running crop than the
crop itself, so every chain gets two crop periods.
"""

from __future__ import annotations

import ctypes
import os
import time
from typing import Any, Callable, Dict, List, Optional, Sequence, Tuple, Union

import torch

from . import _lib
from .models.faster_rcnn import Decoder, RoIBBox, RoIDelta, RoIPooling

Heads = Union[Callable[[torch.Tensor, torch.Tensor], Tuple[torch.Tensor, torch.Tensor]],
              Tuple[torch.Tensor, torch.Tensor]]


def _stage_streams(device: torch.device, want: int = 3) -> List[torch.cuda.Stream]:
    """``want`` streams that share a hardware queue neither with each other nor with the caller's stream.

    A process has four hardware queues (GPU_MAX_HW_QUEUES) and HIP deals streams onto them; which queue a new stream gets
    depends on the streams other code (torch's pool, RCCL) made before.  Two stage streams on one queue serialise behind
    each other's event waits: measured 7 % (two of the small stages) to 14 % (the pooling stage involved) of the pipeline's
    throughput, depending only on how many streams existed before the pipeline was built.  So the streams are probed: a
    stream spins for a moment, and every candidate whose marker completes meanwhile sits on a different queue.  Falls back
    to the first candidates when the probe is inconclusive.
    """
    cands = [torch.cuda.Stream(device) for _ in range(8)]
    cur = torch.cuda.current_stream(device)
    for s in cands:  # first use creates the queue: keep that out of the probe
        torch.cuda.Event().record(s)
    torch.cuda.synchronize(device)

    def independent_of(busy: List[torch.cuda.Stream], pool: List[torch.cuda.Stream]) -> Optional[List[torch.cuda.Stream]]:
        cycles = 2_000_000
        for _ in range(3):
            ends = []
            for b in busy:
                with torch.cuda.stream(b):
                    torch.cuda._sleep(cycles)
                    e = torch.cuda.Event()
                    e.record(b)
                    ends.append(e)
            marks = []
            for s in pool:
                e = torch.cuda.Event()
                e.record(s)
                marks.append(e)
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 300e-6:
                pass
            free = [m.query() for m in marks]
            still_spinning = not any(e.query() for e in ends)
            torch.cuda.synchronize(device)
            if still_spinning:
                return [s for s, f in zip(pool, free) if f]
            cycles *= 8  # the spin was over before the markers were looked at: spin longer
        return None

    chosen: List[torch.cuda.Stream] = []
    pool = list(cands)
    try:
        while len(chosen) < want:
            pool = independent_of([cur] + chosen if not chosen else [chosen[-1]], pool)
            if not pool:
                break
            chosen.append(pool.pop(0))
    except Exception:  # the probe must never keep the pipeline from being built
        chosen = []
    if os.environ.get("FRCNN_PIPE_DEBUG"):
        print(f"[pipeline] stage-stream probe: chose {[cands.index(s) for s in chosen]} of {len(cands)} candidates", flush=True)
    if len(chosen) < want:
        return cands[:want]
    return chosen


class _TrainInputs:
    """What RoIDelta needs for one batch (train mode), plus the optional dense-heads callable."""

    __slots__ = ("gt_boxes", "gt_labels", "rnd_pos", "rnd_neg", "seed", "heads")

    def __init__(self, gt_boxes, gt_labels, rnd_pos, rnd_neg, seed, heads) -> None:
        self.gt_boxes, self.gt_labels, self.rnd_pos, self.rnd_neg, self.seed, self.heads = gt_boxes, gt_labels, rnd_pos, rnd_neg, seed, heads


class _Done:
    """Completion of a slot's batch for the host: one event, or two recorded on different streams (train mode: the crop on the
    pooling stream, RoIDelta on its proposal lane -- joining them on a stream would put a wait packet between two crops)."""

    __slots__ = ("a", "b", "two")

    def __init__(self) -> None:
        self.a = _lib.HipEvent(_lib.hipEventDisableTiming)  # host-visible: keeps the system-scope release
        self.b: Optional[_lib.HipEvent] = None
        self.two = False

    def record(self, stream: int, second: bool = False) -> None:
        if second:
            if self.b is None:
                self.b = _lib.HipEvent(_lib.hipEventDisableTiming)
            self.b.record(stream)
            self.two = True
        else:
            self.a.record(stream)

    def synchronize(self) -> None:
        self.a.synchronize()
        if self.two:
            self.b.synchronize()

    def wait(self, stream: int) -> None:
        self.a.wait(stream)
        if self.two:
            self.b.wait(stream)


class _Slot:
    __slots__ = ("rois", "pooled", "boxes", "labels", "scores", "done", "inputs", "busy", "pending", "wait_in", "small", "h_small", "host",
                 "failed", "ticket", "rd_deltas", "rd_labels", "tag")

    def __init__(self) -> None:
        self.rois = self.pooled = self.boxes = self.labels = self.scores = None
        self.rd_deltas = self.rd_labels = None  # train mode: RoIDelta's (B,R*L,4) deltas and (B,R,L) one-hot labels
        self.done = _Done()
        self.inputs: Any = None
        self.busy = False
        self.pending: Any = None  # (fm, heads, timing) while the batch waits for its group's crop launch
        self.wait_in = False      # the proposal stage must wait for an event recorded on the caller's stream
        self.small = self.h_small = None  # the small outputs packed in one buffer (device) and its pinned host copy
        self.host: Any = None             # host views of rois / boxes / labels / scores (host_outputs=True)
        self.failed: Optional[BaseException] = None  # the error that kept this slot's batch from being enqueued
        self.ticket = -1
        self.tag: Any = None  # what the gather hook is called with (the caller's step index; the ticket by default)


class RoIPathPipeline:
    """``submit`` batches, ``result`` them in order (or out of order) within ``depth`` submissions.

    Args:
        roi_bbox, roi_pooling, decoder: the layers (same objects the sequential path uses).
        depth: batches in flight = ring slots.  ``submit`` blocks the host once ``depth`` batches are
            pending, so a slot's tensors stay valid until ``depth - 1`` further submissions.
        crop_group: batches that share ONE launch per stage (``frcnn_roi_bbox_multi``, ``frcnn_crop_and_resize_multi``,
            ``frcnn_decoder_multi``).  With ``crop_group > 1`` a batch is enqueued when its group is complete (or at
            ``flush`` / ``result`` / ``synchronize``): one launch tail and one inter-launch gap of the crop kernel per
            group instead of per batch, and the one-workgroup-per-image kernels of the whole group become resident in
            the same crop tail -- at the price of up to ``crop_group - 1`` submissions of extra latency.
        host_outputs: also copy the small per-image outputs of every batch (RoIs, detections: 154 KB at config 2) to
            pinned host memory on the Decoder stream, one asynchronous copy per batch -- the "final host gather" of the
            sharded deployment; ``result`` then carries them under ``"host"``.  The pooled features stay on
            the GPU for the dense heads.
        roi_delta: the ``RoIDelta`` layer: train mode (``decoder`` must then be None).  ``submit`` takes ``gt=(gt_boxes,
            gt_labels)`` (and optionally ``random_masks`` / ``seed`` for the two sampling calls); ``result`` carries
            ``roi_bbox_deltas`` and ``roi_bbox_labels`` instead of the detections.
        gather: ``gather(tag, outputs, stream)`` is called once per batch when its last stage has been enqueued, with the
            batch's small outputs (``roi_bboxes`` and, in inference, ``bboxes`` / ``labels`` / ``scores``) and the stream
            that orders after them -- the hook for a gather that is not per-rank pinned memory
            (``sharding.SharedHostGather.write_step``).
    """

    def __init__(self, roi_bbox: RoIBBox, roi_pooling: RoIPooling, decoder: Optional[Decoder] = None, depth: int = 4,
                 crop_group: int = 1, host_outputs: bool = False, roi_delta: Optional[RoIDelta] = None,
                 gather: Optional[Callable[[int, Dict[str, torch.Tensor], int], None]] = None) -> None:
        self.host_outputs = host_outputs
        if (decoder is None) == (roi_delta is None):
            raise ValueError("give either decoder (inference) or roi_delta (training)")
        self.train = roi_delta is not None
        self.roi_delta = roi_delta
        self.gather = gather
        if depth < 1:
            raise ValueError("depth must be >= 1")
        if not 1 <= crop_group <= 4 or crop_group > depth:
            raise ValueError("crop_group must be 1..4 and at most depth")
        self.crop_group = crop_group
        self._pending: List[int] = []  # slots of submitted batches that wait for their group to be enqueued (flush)
        self.roi_bbox, self.roi_pooling, self.decoder = roi_bbox, roi_pooling, decoder
        self.device = roi_bbox.anchors.device
        if self.device.type != "cuda":
            raise RuntimeError("RoIPathPipeline needs a ROCm GPU; there is no CPU fallback.")
        self.depth = depth
        with torch.cuda.device(self.device):
            self.s_prop, self.s_pool, self.s_dec = _stage_streams(self.device)
            self.slots: List[_Slot] = [_Slot() for _ in range(depth)]
            # stream-to-stream events: device-scope release (no L2 write-back behind the kernel they follow)
            self._ev_in = [_lib.HipEvent() for _ in range(depth)]
            self._ev_rois = [_lib.HipEvent() for _ in range(depth)]
            self._ev_heads = [_lib.HipEvent() for _ in range(depth)]
        self._count = 0
        self._lane = 0  # train mode: the proposal lane of the next group
        self._lanes = 1 if os.environ.get("FRCNN_PIPE_LANES") == "1" else 2  # (diagnostic: one proposal lane)
        # diagnostic ablation only (results are then wrong): FRCNN_PIPE_SKIP=prop,crop,dec leaves those launches out
        self._skip = set(filter(None, os.environ.get("FRCNN_PIPE_SKIP", "").split(",")))
        self._idle = True  # no batch of this pipeline on the GPU: set by synchronize(), cleared by the next enqueue
        self.wait_seconds = 0.0  # host time spent blocked on back-pressure (diagnostic)
        self._shape_key: Optional[Tuple[int, ...]] = None
        self._ws_prop: Optional[torch.Tensor] = None
        self._ws_dec: Optional[torch.Tensor] = None
        hp = roi_bbox.hyper_params
        self._K = int(hp["pre_nms_topn"])
        self._R = int(hp["train_nms_topn"] if roi_bbox.mode == "training" else hp["test_nms_topn"])
        self._thr = float(hp["nms_iou_threshold"])
        self._var_prop = _lib.float4(hp["variances"])
        if self.train:
            hd = roi_delta.hyper_params
            self._var_dec = _lib.float4(hd["variances"])
            self._L = int(hd["total_labels"])
            self._tot = (int(hd["total_pos_bboxes"]), int(hd["total_neg_bboxes"]))
        else:
            self._var_dec = _lib.float4(decoder.variances)
            self._L = int(decoder.total_labels)
        self._pool = tuple(int(v) for v in roi_pooling.hyper_params["pooling_size"])

    # -- shape-dependent set-up, once per input shape ------------------------------------------------
    def _prepare(self, deltas: torch.Tensor, scores: torch.Tensor, fm: torch.Tensor) -> None:
        key = (deltas.shape[0], deltas.numel(), scores.numel()) + tuple(fm.shape)
        if key == self._shape_key:
            return
        self.synchronize()
        N = self.roi_bbox.anchors.shape[0]
        B = deltas.shape[0]
        if deltas.numel() != B * N * 4 or scores.numel() != B * N:
            raise ValueError(f"rpn outputs {tuple(deltas.shape)} / {tuple(scores.shape)} do not match {N} anchors")
        if fm.dim() != 4 or fm.shape[0] != B:
            raise ValueError(f"feature_map must be ({B}, H, W, C), got {tuple(fm.shape)}")
        lib = _lib.lib()
        R, L = self._R, self._L
        T = 0 if self.train else int(self.decoder.max_total_size)
        ph, pw = self._pool
        dev, f32 = self.device, torch.float32
        if self.train:
            names, sizes, shapes = ("roi_bboxes",), [B * R * 4], [(B, R, 4)]
        else:
            names = ("roi_bboxes", "bboxes", "labels", "scores")
            sizes = [B * R * 4, B * T * 4, B * T, B * T]
            shapes = [(B, R, 4), (B, T, 4), (B, T), (B, T)]
        offs = [0]
        for n_el in sizes:
            offs.append(offs[-1] + (n_el + 63) // 64 * 64)  # every section 256-byte aligned

        def views(buf: torch.Tensor) -> List[torch.Tensor]:
            return [buf[o:o + n_el].view(shape) for o, n_el, shape in zip(offs, sizes, shapes)]

        for s in self.slots:
            s.small = torch.empty(offs[-1], dtype=f32, device=dev)  # rois | boxes | labels | scores: one copy moves all four
            if self.train:
                (s.rois,) = views(s.small)
                s.rd_deltas = torch.empty((B, R * L, 4), dtype=f32, device=dev)
                s.rd_labels = torch.empty((B, R, L), dtype=f32, device=dev)
            else:
                s.rois, s.boxes, s.labels, s.scores = views(s.small)
            s.pooled = torch.empty((B, R, ph, pw, fm.shape[3]), dtype=f32, device=dev)
            if self.host_outputs:
                s.h_small = torch.empty(offs[-1], dtype=f32, pin_memory=True)
                s.host = dict(zip(names, views(s.h_small)))
        self._ws_prop = torch.empty(max(1, lib.frcnn_roi_bbox_workspace_bytes(B, N, self._K, R)), dtype=torch.uint8, device=dev)
        self._ws_dec = None if self.train else torch.empty(max(1, lib.frcnn_decoder_workspace_bytes(B, R, L, T)), dtype=torch.uint8, device=dev)
        self._ws_pool = torch.empty(max(1, lib.frcnn_roi_pooling_workspace_bytes(B, R)), dtype=torch.uint8, device=dev)
        # RoIPooling's pre-pass (source-row order of feature maps beyond L2) runs on the proposal stream, a batch ahead of the
        # crop that consumes it: one workspace per slot then (allocated when the library first reports that schedule)
        self._ws_stage: Dict[int, torch.Tensor] = {}
        self._prepared = ctypes.c_int(0)
        self._ws_prop2 = torch.empty_like(self._ws_prop) if self.train else None  # the second proposal lane's workspace
        self._ws_delta: Dict[Tuple[int, int], torch.Tensor] = {}  # train mode: RoIDelta's workspace per (gt boxes per image, lane)
        torch.cuda.synchronize(dev)
        self._slot_tables = {}  # pointer tables of slot-owned tensors per group of slots (they only change with the shape)
        self._shape_key = key

    # -- one batch ------------------------------------------------------------------------------------
    def submit(self, rpn_bbox_deltas: torch.Tensor, rpn_labels: torch.Tensor, feature_map: torch.Tensor,
               heads: Optional[Heads] = None, timing: Optional[Tuple[Any, Any]] = None,
               gt: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
               random_masks: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, seed: Optional[int] = None,
               tag: Any = None) -> int:
        """Enqueue one batch; returns its ticket.  ``tag`` (default: the ticket) is what the ``gather`` hook is called with.

        Inference: ``heads`` is either ``(pred_deltas (B,R,L*4), pred_label_probs (B,R,L))`` already on the device, or
        a callable ``heads(pooled, rois) -> (pred_deltas, pred_label_probs)`` that is run on the pooling
        stage's stream right after the crop (the dense heads of the model).  Training: ``gt = (gt_boxes (B,M,4),
        gt_labels (B,M) int32)`` feeds RoIDelta (``random_masks``: the two ``(B,R)`` int32 tensors of its sampling calls,
        ``seed``: the in-kernel draw otherwise); ``heads`` may be a callable ``heads(pooled, rois)`` whose work (the dense
        heads and the loss) is enqueued on the pooling stream and whose return value is ignored.  ``timing`` = two events
        (``_lib.HipEvent`` created with timing enabled; ``record(stream_handle)``) recorded immediately before and after
        the crop kernel on its stream (with ``crop_group > 1``: around the group's launch, taken from the first batch
        of the group that carries them).
        """
        dev = self.device
        deltas = _lib.device_tensor(rpn_bbox_deltas, torch.float32, dev)
        scores = _lib.device_tensor(rpn_labels, torch.float32, dev)
        fm = _lib.device_tensor(feature_map, torch.float32, dev)
        self._prepare(deltas, scores, fm)
        converted = any(a is not b for a, b in ((deltas, rpn_bbox_deltas), (scores, rpn_labels), (fm, feature_map)))
        if self.train:
            if gt is None:
                raise ValueError("train mode: submit needs gt=(gt_boxes, gt_labels)")
            if heads is not None and not callable(heads):
                raise ValueError("train mode: heads must be a callable or None")
            heads, conv = self._check_gt(gt, random_masks, seed, deltas.shape[0], heads)
            converted = converted or conv
        elif heads is None:
            raise ValueError("inference mode: submit needs the head outputs (or a callable producing them)")
        elif not callable(heads):
            # head outputs handed over as tensors: normalise them HERE, on the caller's stream and before the input event
            # is recorded, so that the Decoder stage is ordered after any dtype / layout / host-to-device copy they need
            heads, conv = self._check_heads(heads, deltas.shape[0])
            converted = converted or conv
        k = self._count % self.depth
        slot = self.slots[k]
        if slot.busy:
            if slot.pending is not None:
                self.flush()
            t_wait = time.perf_counter()
            slot.done.synchronize()  # back-pressure; also makes dropping the old inputs safe
            self.wait_seconds += time.perf_counter() - t_wait
        # The inputs were produced on the caller's stream: the proposal stage waits for an event recorded there -- unless
        # that stream is idle, in which case they are complete and the stream is left alone.  (A process has four hardware
        # queues, GPU_MAX_HW_QUEUES; when the caller's stream shares one with the pooling or Decoder stream, which depends on
        # how many streams other libraries created first, every marker recorded on it queues behind that stage's kernels:
        # measured 7-14 % of the throughput.)
        cur = torch.cuda.current_stream(dev)
        slot.wait_in = converted or not cur.query()  # a conversion was just enqueued on `cur`: never skip the wait then
        if slot.wait_in:
            self._ev_in[k].record(cur.cuda_stream)
        slot.pending = (fm, heads, timing)
        slot.inputs = (deltas, scores, fm)  # keep alive until the slot comes round again
        slot.busy = True
        slot.failed = None
        slot.ticket = self._count
        slot.tag = self._count if tag is None else tag
        self._pending.append(k)
        ticket = self._count
        self._count += 1
        if len(self._pending) >= self.crop_group:
            self.flush()
        return ticket

    def _check_gt(self, gt: Any, random_masks: Any, seed: Optional[int], B: int, heads: Any) -> Tuple[Any, bool]:
        """Train mode: the ground truth (and sampling integers) as contiguous device tensors of the right shape; packed
        with the optional heads callable into the slot's pending entry.  Conversions run on the CURRENT stream."""
        R = self._R
        try:
            gt_boxes, gt_labels = gt
        except (TypeError, ValueError):
            raise ValueError("gt must be (gt_boxes, gt_labels)") from None
        g = _lib.device_tensor(gt_boxes, torch.float32, self.device)
        lab = _lib.device_tensor(gt_labels, torch.int32, self.device)
        if g.dim() != 3 or g.shape[0] != B or g.shape[2] != 4 or g.shape[1] < 1 or tuple(lab.shape) != (B, g.shape[1]):
            raise ValueError(f"gt must be ({B}, M, 4) boxes and ({B}, M) labels with M >= 1, got {tuple(g.shape)} / {tuple(lab.shape)}")
        conv = (g is not gt_boxes) or (lab is not gt_labels)
        rp = rn = None
        if random_masks is not None:
            rp = _lib.device_tensor(random_masks[0], torch.int32, self.device)
            rn = _lib.device_tensor(random_masks[1], torch.int32, self.device)
            if tuple(rp.shape) != (B, R) or tuple(rn.shape) != (B, R):
                raise ValueError(f"random_masks must be two ({B}, {R}) tensors")
            conv = conv or (rp is not random_masks[0]) or (rn is not random_masks[1])
        if seed is None:
            seed = 0 if random_masks is not None else int(torch.randint(0, 2 ** 62, (1,)).item())
        return _TrainInputs(g, lab, rp, rn, int(seed) & 0xFFFFFFFFFFFFFFFF, heads), conv

    def _check_heads(self, heads: Any, B: int) -> Tuple[Tuple[torch.Tensor, torch.Tensor], bool]:
        """Head outputs as contiguous fp32 device tensors of the right shape (ValueError otherwise), and whether that
        took a conversion.  Any conversion is enqueued on the CURRENT stream: the caller decides which stream that is."""
        R, L = self._R, int(self.decoder.total_labels)
        try:
            pred_deltas, pred_probs = heads
        except (TypeError, ValueError):
            raise ValueError("heads must be (pred_deltas, pred_label_probs) or a callable returning them") from None
        d = _lib.device_tensor(pred_deltas, torch.float32, self.device)
        p = _lib.device_tensor(pred_probs, torch.float32, self.device)
        if d.numel() != B * R * L * 4 or tuple(p.shape) != (B, R, L):
            raise ValueError(f"head outputs {tuple(d.shape)} / {tuple(p.shape)} do not match "
                             f"({B}, {R}, {L}*4) / ({B}, {R}, {L})")
        return (d, p), (d is not pred_deltas) or (p is not pred_probs)

    def flush(self) -> None:
        """Enqueue the three stages of every submitted batch that is still waiting for its group: ONE proposal launch,
        ONE crop launch and ONE Decoder launch for all of them (the `*_multi` entry points).  If that fails (a head
        callable raising or returning the wrong shape, an error code from the library) the group's batches are marked
        failed -- ``result`` raises for them -- and the error propagates."""
        if not self._pending:
            return
        ks, self._pending = self._pending, []
        slots = [self.slots[k] for k in ks]
        try:
            with _lib.on_device(self.device):
                self._enqueue(ks, slots)
        except BaseException as exc:
            try:  # whatever was enqueued for the group on any stage completes before its slots are reused
                q_prop, q_pool, q_dec = self.s_prop.cuda_stream, self.s_pool.cuda_stream, self.s_dec.cuda_stream
                if self.train:  # both proposal lanes -> pooling stream, where `done` is recorded
                    for q in (q_prop, q_dec):
                        self._ev_rois[ks[0]].record(q)
                        self._ev_rois[ks[0]].wait(q_pool)
                    q_dec = q_pool
                else:
                    self._ev_rois[ks[0]].record(q_prop)
                    self._ev_rois[ks[0]].wait(q_pool)
                    self._ev_heads[ks[0]].record(q_pool)
                    self._ev_heads[ks[0]].wait(q_dec)
            finally:
                for s in slots:
                    s.failed = exc
                    s.pending = None
                    s.done.two = False  # (everything was joined on one stream above)
                    s.done.record(q_dec)
            raise

    def _enqueue(self, ks: List[int], slots: List[_Slot]) -> None:
        lib = _lib.lib()
        n = len(slots)
        N = self.roi_bbox.anchors.shape[0]
        R, L = self._R, self._L
        q_prop, q_pool, q_dec = self.s_prop.cuda_stream, self.s_pool.cuda_stream, self.s_dec.cuda_stream
        fm0 = slots[0].pending[0]
        B, H, W, C = fm0.shape
        ph, pw = self._pool
        tab = _lib.ptr_table
        own = self._slot_tables.get(tuple(ks))
        if own is None:
            fields = ("rois", "pooled") if self.train else ("rois", "pooled", "boxes", "labels", "scores")
            own = {name: tab([getattr(s, name) for s in slots]) for name in fields}
            if self.host_outputs:
                own["copies"] = []  # arguments of frcnn_gather_copy_multi, at most 8 segments a launch
                for i in range(0, n, 8):
                    grp = slots[i:i + 8]
                    own["copies"].append((tab([s.small for s in grp]), tab([s.h_small for s in grp]),
                                          (ctypes.c_size_t * len(grp))(*[s.small.numel() * 4 for s in grp]), len(grp)))
            self._slot_tables[tuple(ks)] = own

        if self.train:
            # Train mode has TWO proposal lanes (the proposal stream and the third stream) that take the groups in turn: the
            # whole-chip bitmask schedule is a chain of eight launches that, beside a running crop, takes longer than the crop
            # itself; on one stream it would be the pipeline's critical chain, on two every chain has two crop periods.
            # A lane runs proposals -> host gather of the RoIs -> RoIDelta (which needs the RoIs and the ground truth, not the
            # pooled features) in stream order, no event in between.
            lane = self._lane
            self._lane = (self._lane + 1) % self._lanes
            q_prop = (q_prop, q_dec)[lane]
            ws_prop = self._ws_prop if lane == 0 else self._ws_prop2
        else:
            ws_prop = self._ws_prop

        for k in ks:
            if self.slots[k].wait_in:
                self._ev_in[k].wait(q_prop)
        if "prop" in self._skip:
            pass
        elif self.train and self._idle:
            # nothing of this pipeline is running on the GPU (first group after construction / synchronize()): the proposals
            # have the chip to themselves and take the single-batch entry point, whose one-workgroup-per-image kernels
            # use 1024 threads (the grouped one picks the 256-thread forms that fit beside a running crop, slower alone)
            for s in slots:
                _lib.check(lib.frcnn_roi_bbox(s.inputs[0].data_ptr(), s.inputs[1].data_ptr(), self.roi_bbox.anchors.data_ptr(),
                                              self._var_prop, B, N, self._K, R, self._thr, s.rois.data_ptr(), None, None,
                                              ws_prop.data_ptr(), ws_prop.numel(), q_prop))
        else:
            _lib.check(lib.frcnn_roi_bbox_multi(tab([s.inputs[0] for s in slots]), tab([s.inputs[1] for s in slots]),
                                                self.roi_bbox.anchors.data_ptr(), self._var_prop, n, B, N, self._K, R,
                                                self._thr, own["rois"], ws_prop.data_ptr(), ws_prop.numel(), q_prop))
        self._idle = False
        # the crop's pre-pass, where its schedule has one: it needs the RoIs only, so it rides on the proposal stream and
        # the pooling stream carries the crop kernel alone (a group is either prepared as a whole or not at all)
        prepared = "crop" not in self._skip and "prop" not in self._skip
        if prepared:
            for k, s in zip(ks, slots):
                ws = self._ws_stage.get(k)
                _lib.check(lib.frcnn_roi_pooling_prepare(s.pending[0].data_ptr(), B, H, W, C, s.rois.data_ptr(), R, ph, pw,
                                                         s.pooled.data_ptr(), ws.data_ptr() if ws is not None else self._ws_pool.data_ptr(),
                                                         self._ws_pool.numel(), ctypes.byref(self._prepared), q_prop))
                if not self._prepared.value:
                    prepared = False  # (decided by shape and knobs: the same for every batch of the group)
                    break
                if ws is None:
                    # first time for this slot: the pre-pass just went into the shared workspace, which the next batch would
                    # overwrite -- give the slot its own and run the (cheap) pre-pass again into it
                    ws = self._ws_stage[k] = torch.empty_like(self._ws_pool)
                    _lib.check(lib.frcnn_roi_pooling_prepare(s.pending[0].data_ptr(), B, H, W, C, s.rois.data_ptr(), R, ph, pw,
                                                             s.pooled.data_ptr(), ws.data_ptr(), ws.numel(),
                                                             ctypes.byref(self._prepared), q_prop))
        ev_rois = self._ev_rois[ks[0]]
        ev_rois.record(q_prop)

        if self.train:
            if self.host_outputs:
                for srcs, dsts, sizes, cnt in own["copies"]:
                    _lib.check(lib.frcnn_gather_copy_multi(srcs, dsts, sizes, cnt, q_prop))
            if self.gather is not None:
                for s in slots:
                    self.gather(s.tag, {"roi_bboxes": s.rois}, q_prop)
            if "dec" not in self._skip:
                for s in slots:
                    ti = s.pending[1]
                    M = ti.gt_boxes.shape[1]
                    ws = self._ws_delta.get((M, lane))
                    if ws is None:
                        ws = self._ws_delta[(M, lane)] = torch.empty(max(1, lib.frcnn_roi_delta_workspace_bytes(B, R, M)),
                                                                     dtype=torch.uint8, device=self.device)
                    _lib.check(lib.frcnn_roi_delta(s.rois.data_ptr(), ti.gt_boxes.data_ptr(), ti.gt_labels.data_ptr(),
                                                   _lib.ptr(ti.rnd_pos), _lib.ptr(ti.rnd_neg), ti.seed, B, R, M, L, self._tot[0],
                                                   self._tot[1], self._var_dec, s.rd_deltas.data_ptr(), s.rd_labels.data_ptr(),
                                                   ws.data_ptr(), ws.numel(), q_prop))
            for s in slots:  # the batch's RoIDelta half of `done`, on its lane
                s.done.record(q_prop, second=True)

        ev_rois.wait(q_pool)  # (ev_in is implied: the proposal stage waited for it)
        timing = next((s.pending[2] for s in slots if s.pending[2] is not None), None)
        if timing is not None:
            timing[0].record(q_pool)
        if prepared:
            _lib.check(lib.frcnn_roi_pooling_prepared_multi(tab([s.pending[0] for s in slots]), own["rois"], own["pooled"],
                                                            tab([self._ws_stage[k] for k in ks]), self._ws_pool.numel(), n, B, H, W,
                                                            C, R, ph, pw, q_pool))
        elif "crop" not in self._skip:
            _lib.check(lib.frcnn_crop_and_resize_multi(tab([s.pending[0] for s in slots]), own["rois"], own["pooled"],
                                                       n, B, H, W, C, B * R, R, ph, pw, 0.0, self._ws_pool.data_ptr(),
                                                       self._ws_pool.numel(), q_pool))
        if timing is not None:
            timing[1].record(q_pool)
        outs = []
        for s in slots:
            fm, heads, _ = s.pending
            if self.train:
                if heads.heads is not None:  # the dense heads / losses of the training model, on the pooling stream
                    with torch.cuda.stream(self.s_pool):
                        heads.heads(s.pooled, s.rois)
                outs.append((heads.gt_boxes, heads.gt_labels, heads.rnd_pos, heads.rnd_neg))
            elif callable(heads):
                # the dense heads run on the pooling stream, and so does any conversion of what they return: the event the
                # Decoder stage waits for (ev_heads) is recorded on that stream after both
                with torch.cuda.stream(self.s_pool):
                    raw = heads(s.pooled, s.rois)
                    (pred_deltas, pred_probs), conv = self._check_heads(raw, B)
                if conv:
                    s.inputs = s.inputs + (raw,)  # what the conversion reads stays alive until the slot comes round again
                outs.append((pred_deltas, pred_probs))
            else:
                outs.append(heads)  # normalised and checked in submit()

        if self.train:
            # the batch is done when its crop (+ heads) AND its RoIDelta are: two events, one per stream (_Done)
            q_done = q_pool
        else:
            ev_heads = self._ev_heads[ks[0]]
            if "headsev" in self._skip:  # diagnostic: the Decoder ordered behind the RoIs only (right only for precomputed heads)
                ev_rois.wait(q_dec)
            else:
                ev_heads.record(q_pool)
                ev_heads.wait(q_dec)
            if "dec" not in self._skip:
                T = int(self.decoder.max_total_size)
                _lib.check(lib.frcnn_decoder_multi(own["rois"], tab([o[0] for o in outs]), tab([o[1] for o in outs]),
                                                   self._var_dec, n, B, R, L, T, float(self.decoder.score_threshold), 0.5,
                                                   own["boxes"], own["labels"], own["scores"], self._ws_dec.data_ptr(),
                                                   self._ws_dec.numel(), q_dec))
            if self.host_outputs:
                # ONE copy kernel moves the group's small outputs into their pinned buffers (a copy-engine transfer here costs
                # 18 % of the throughput)
                for srcs, dsts, sizes, cnt in own["copies"]:
                    _lib.check(lib.frcnn_gather_copy_multi(srcs, dsts, sizes, cnt, q_dec))
            if self.gather is not None:
                for s in slots:
                    self.gather(s.tag, {"roi_bboxes": s.rois, "bboxes": s.boxes, "labels": s.labels, "scores": s.scores}, q_dec)
            q_done = q_dec
        for s, o in zip(slots, outs):
            s.done.record(q_done)
            s.inputs = s.inputs + tuple(o)
            s.pending = None

    def result(self, ticket: int, wait: str = "host") -> Dict[str, Any]:
        """Outputs of batch ``ticket``: ``roi_bboxes``, ``roi_pooled`` and ``bboxes``, ``labels``, ``scores`` (inference) or
        ``roi_bbox_deltas``, ``roi_bbox_labels`` (training).

        ``wait="host"`` blocks until the batch is done; ``wait="stream"`` makes the caller's current stream
        wait instead (no host block).  The tensors are ring slots: consume them before ``depth - 1`` further
        ``submit`` calls."""
        if not (self._count - self.depth <= ticket < self._count) or ticket < 0:
            raise ValueError(f"ticket {ticket} is not in flight (last {self.depth} of {self._count} submissions are)")
        slot = self.slots[ticket % self.depth]
        if slot.pending is not None:
            self.flush()
        if slot.failed is not None:
            raise RuntimeError(f"batch {ticket} was not enqueued: {slot.failed}") from slot.failed
        if wait == "host":
            slot.done.synchronize()
        elif wait == "stream":
            slot.done.wait(torch.cuda.current_stream(self.device).cuda_stream)
        else:
            raise ValueError("wait must be 'host' or 'stream'")
        if self.train:
            out = {"roi_bboxes": slot.rois, "roi_pooled": slot.pooled, "roi_bbox_deltas": slot.rd_deltas,
                   "roi_bbox_labels": slot.rd_labels}
        else:
            out = {"roi_bboxes": slot.rois, "roi_pooled": slot.pooled, "bboxes": slot.boxes, "labels": slot.labels,
                   "scores": slot.scores}
        if self.host_outputs:
            out["host"] = slot.host  # valid once the batch is done (wait="host"), pinned memory of the slot
        return out

    def synchronize(self) -> None:
        """Block until every submitted batch is done."""
        self.flush()
        for s in self.slots:
            if s.busy:
                s.done.synchronize()
                s.busy = False
                s.inputs = None
        self._idle = True


def run_sequential(roi_bbox: RoIBBox, roi_pooling: RoIPooling, decoder: Decoder,
                   inputs: Sequence[torch.Tensor]) -> Dict[str, torch.Tensor]:
    """The same batch through the three layers on the current stream (the reference's call order)."""
    deltas, scores, fm, pred_deltas, pred_probs = inputs
    rois = roi_bbox([deltas, scores])
    pooled = roi_pooling([fm, rois])
    boxes, labels, sc = decoder([rois, pred_deltas, pred_probs])
    return {"roi_bboxes": rois, "roi_pooled": pooled, "bboxes": boxes, "labels": labels, "scores": sc}


def run_sequential_train(roi_bbox: RoIBBox, roi_pooling: RoIPooling, roi_delta: RoIDelta, inputs: Sequence[torch.Tensor],
                         random_masks: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
                         seed: Optional[int] = None) -> Dict[str, torch.Tensor]:
    """The training forward (models/faster_rcnn.py:327-335) on the current stream, the reference's call order:
    RoIBBox -> RoIPooling -> RoIDelta."""
    deltas, scores, fm, gt_boxes, gt_labels = inputs
    rois = roi_bbox([deltas, scores])
    pooled = roi_pooling([fm, rois])
    rd, rl = roi_delta([rois, gt_boxes, gt_labels], random_masks=random_masks, seed=seed)
    return {"roi_bboxes": rois, "roi_pooled": pooled, "roi_bbox_deltas": rd, "roi_bbox_labels": rl}
