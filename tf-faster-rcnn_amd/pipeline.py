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
"""

from __future__ import annotations

import ctypes
import os
import time
from typing import Any, Callable, Dict, List, Optional, Sequence, Tuple, Union

import torch

from . import _lib
from .models.faster_rcnn import Decoder, RoIBBox, RoIPooling

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


class _Slot:
    __slots__ = ("rois", "pooled", "boxes", "labels", "scores", "done", "inputs", "busy", "pending", "wait_in", "small", "h_small", "host",
                 "failed", "ticket")

    def __init__(self) -> None:
        self.rois = self.pooled = self.boxes = self.labels = self.scores = None
        self.done = _lib.HipEvent(_lib.hipEventDisableTiming)  # host-visible: keeps the system-scope release
        self.inputs: Any = None
        self.busy = False
        self.pending: Any = None  # (fm, heads, timing) while the batch waits for its group's crop launch
        self.wait_in = False      # the proposal stage must wait for an event recorded on the caller's stream
        self.small = self.h_small = None  # the small outputs packed in one buffer (device) and its pinned host copy
        self.host: Any = None             # host views of rois / boxes / labels / scores (host_outputs=True)
        self.failed: Optional[BaseException] = None  # the error that kept this slot's batch from being enqueued
        self.ticket = -1


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
    """

    def __init__(self, roi_bbox: RoIBBox, roi_pooling: RoIPooling, decoder: Decoder, depth: int = 4,
                 crop_group: int = 1, host_outputs: bool = False) -> None:
        self.host_outputs = host_outputs
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
        # diagnostic ablation only (results are then wrong): FRCNN_PIPE_SKIP=prop,crop,dec leaves those launches out
        self._skip = set(filter(None, os.environ.get("FRCNN_PIPE_SKIP", "").split(",")))
        self.wait_seconds = 0.0  # host time spent blocked on back-pressure (diagnostic)
        self._shape_key: Optional[Tuple[int, ...]] = None
        self._ws_prop: Optional[torch.Tensor] = None
        self._ws_dec: Optional[torch.Tensor] = None
        hp = roi_bbox.hyper_params
        self._K = int(hp["pre_nms_topn"])
        self._R = int(hp["train_nms_topn"] if roi_bbox.mode == "training" else hp["test_nms_topn"])
        self._thr = float(hp["nms_iou_threshold"])
        self._var_prop = _lib.float4(hp["variances"])
        self._var_dec = _lib.float4(decoder.variances)
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
        R, T, L = self._R, int(self.decoder.max_total_size), int(self.decoder.total_labels)
        ph, pw = self._pool
        dev, f32 = self.device, torch.float32
        sizes = [B * R * 4, B * T * 4, B * T, B * T]
        offs = [0]
        for n_el in sizes:
            offs.append(offs[-1] + (n_el + 63) // 64 * 64)  # every section 256-byte aligned
        shapes = [(B, R, 4), (B, T, 4), (B, T), (B, T)]

        def views(buf: torch.Tensor) -> List[torch.Tensor]:
            return [buf[o:o + n_el].view(shape) for o, n_el, shape in zip(offs, sizes, shapes)]

        for s in self.slots:
            s.small = torch.empty(offs[-1], dtype=f32, device=dev)  # rois | boxes | labels | scores: one copy moves all four
            s.rois, s.boxes, s.labels, s.scores = views(s.small)
            s.pooled = torch.empty((B, R, ph, pw, fm.shape[3]), dtype=f32, device=dev)
            if self.host_outputs:
                s.h_small = torch.empty(offs[-1], dtype=f32, pin_memory=True)
                s.host = dict(zip(("roi_bboxes", "bboxes", "labels", "scores"), views(s.h_small)))
        self._ws_prop = torch.empty(max(1, lib.frcnn_roi_bbox_workspace_bytes(B, N, self._K, R)), dtype=torch.uint8, device=dev)
        self._ws_dec = torch.empty(max(1, lib.frcnn_decoder_workspace_bytes(B, R, L, T)), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize(dev)
        self._slot_tables = {}  # pointer tables of slot-owned tensors per group of slots (they only change with the shape)
        self._shape_key = key

    # -- one batch ------------------------------------------------------------------------------------
    def submit(self, rpn_bbox_deltas: torch.Tensor, rpn_labels: torch.Tensor, feature_map: torch.Tensor,
               heads: Heads, timing: Optional[Tuple[Any, Any]] = None) -> int:
        """Enqueue one batch; returns its ticket.

        ``heads`` is either ``(pred_deltas (B,R,L*4), pred_label_probs (B,R,L))`` already on the device, or
        a callable ``heads(pooled, rois) -> (pred_deltas, pred_label_probs)`` that is run on the pooling
        stage's stream right after the crop (the dense heads of the model).  ``timing`` = two events
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
        if not callable(heads):
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
        self._pending.append(k)
        ticket = self._count
        self._count += 1
        if len(self._pending) >= self.crop_group:
            self.flush()
        return ticket

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
                self._ev_rois[ks[0]].record(q_prop)
                self._ev_rois[ks[0]].wait(q_pool)
                self._ev_heads[ks[0]].record(q_pool)
                self._ev_heads[ks[0]].wait(q_dec)
            finally:
                for s in slots:
                    s.failed = exc
                    s.pending = None
                    s.done.record(self.s_dec.cuda_stream)
            raise

    def _enqueue(self, ks: List[int], slots: List[_Slot]) -> None:
        lib = _lib.lib()
        dev = self.device
        n = len(slots)
        N = self.roi_bbox.anchors.shape[0]
        R, T, L = self._R, int(self.decoder.max_total_size), int(self.decoder.total_labels)
        q_prop, q_pool, q_dec = self.s_prop.cuda_stream, self.s_pool.cuda_stream, self.s_dec.cuda_stream
        fm0 = slots[0].pending[0]
        B, H, W, C = fm0.shape
        ph, pw = self._pool
        tab = _lib.ptr_table
        own = self._slot_tables.get(tuple(ks))
        if own is None:
            own = {name: tab([getattr(s, name) for s in slots]) for name in ("rois", "pooled", "boxes", "labels", "scores")}
            if self.host_outputs:
                own["copies"] = []  # arguments of frcnn_gather_copy_multi, at most 8 segments a launch
                for i in range(0, n, 8):
                    grp = slots[i:i + 8]
                    own["copies"].append((tab([s.small for s in grp]), tab([s.h_small for s in grp]),
                                          (ctypes.c_size_t * len(grp))(*[s.small.numel() * 4 for s in grp]), len(grp)))
            self._slot_tables[tuple(ks)] = own

        for k in ks:
            if self.slots[k].wait_in:
                self._ev_in[k].wait(q_prop)
        if "prop" not in self._skip:
            _lib.check(lib.frcnn_roi_bbox_multi(tab([s.inputs[0] for s in slots]), tab([s.inputs[1] for s in slots]),
                                                self.roi_bbox.anchors.data_ptr(), self._var_prop, n, B, N, self._K, R,
                                                self._thr, own["rois"], self._ws_prop.data_ptr(),
                                                self._ws_prop.numel(), q_prop))
        ev_rois = self._ev_rois[ks[0]]
        ev_rois.record(q_prop)

        ev_rois.wait(q_pool)  # (ev_in is implied: the proposal stage waited for it)
        timing = next((s.pending[2] for s in slots if s.pending[2] is not None), None)
        if timing is not None:
            timing[0].record(q_pool)
        if "crop" not in self._skip:
            _lib.check(lib.frcnn_crop_and_resize_multi(tab([s.pending[0] for s in slots]), own["rois"], own["pooled"],
                                                       n, B, H, W, C, B * R, R, ph, pw, 0.0, q_pool))
        if timing is not None:
            timing[1].record(q_pool)
        outs = []
        for s in slots:
            fm, heads, _ = s.pending
            if callable(heads):
                # the dense heads run on the pooling stream, and so does any conversion of what they return: the event the
                # Decoder stage waits for (ev_heads) is recorded on that stream after both
                with torch.cuda.stream(self.s_pool):
                    raw = heads(s.pooled, s.rois)
                    (pred_deltas, pred_probs), conv = self._check_heads(raw, B)
                if conv:
                    s.inputs = s.inputs + (raw,)  # what the conversion reads stays alive until the slot comes round again
            else:
                pred_deltas, pred_probs = heads  # normalised and checked in submit()
            outs.append((pred_deltas, pred_probs))
        ev_heads = self._ev_heads[ks[0]]
        ev_heads.record(q_pool)

        ev_heads.wait(q_dec)
        if "dec" not in self._skip:
            _lib.check(lib.frcnn_decoder_multi(own["rois"], tab([o[0] for o in outs]), tab([o[1] for o in outs]),
                                               self._var_dec, n, B, R, L, T, float(self.decoder.score_threshold), 0.5,
                                               own["boxes"], own["labels"], own["scores"], self._ws_dec.data_ptr(),
                                               self._ws_dec.numel(), q_dec))
        if self.host_outputs:
            # ONE copy kernel moves the group's small outputs into their pinned buffers (a copy-engine transfer here costs
            # 18 % of the throughput)
            for srcs, dsts, sizes, cnt in own["copies"]:
                _lib.check(lib.frcnn_gather_copy_multi(srcs, dsts, sizes, cnt, q_dec))
        for s, (pred_deltas, pred_probs) in zip(slots, outs):
            s.done.record(q_dec)
            s.inputs = s.inputs + (pred_deltas, pred_probs)
            s.pending = None

    def result(self, ticket: int, wait: str = "host") -> Dict[str, Any]:
        """Outputs of batch ``ticket``: ``roi_bboxes``, ``roi_pooled``, ``bboxes``, ``labels``, ``scores``.

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


def run_sequential(roi_bbox: RoIBBox, roi_pooling: RoIPooling, decoder: Decoder,
                   inputs: Sequence[torch.Tensor]) -> Dict[str, torch.Tensor]:
    """The same batch through the three layers on the current stream (the reference's call order)."""
    deltas, scores, fm, pred_deltas, pred_probs = inputs
    rois = roi_bbox([deltas, scores])
    pooled = roi_pooling([fm, rois])
    boxes, labels, sc = decoder([rois, pred_deltas, pred_probs])
    return {"roi_bboxes": rois, "roi_pooled": pooled, "bboxes": boxes, "labels": labels, "scores": sc}
