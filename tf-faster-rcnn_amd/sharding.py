"""Data-parallel sharding of a batch over the GPUs of one node.

The hot path is per-image everywhere, so multi-GPU execution is: one process per GPU,
contiguous image shards, every rank runs the same kernels on its shard, and the small per-image outputs
(RoIs, detections, targets) are gathered to the host at the end.  There is no collective on the data
path; ``gather_to_host`` is the "final host gather" and works on CPU tensors over any backend (gloo in
the CPU tests, the default group on a GPU node).
"""

from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch


def shard_bounds(batch: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous shard ``[lo, hi)`` of ``batch`` images for ``rank``: GPU g of P gets images
    ``[g*B/P, (g+1)*B/P)``; remainders go to the lowest ranks, empty shards are allowed."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError(f"bad rank/world_size {rank}/{world_size}")
    if batch < 0:
        raise ValueError("negative batch")
    q, r = divmod(batch, world_size)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard(tensors: Sequence[torch.Tensor], rank: int, world_size: int) -> List[torch.Tensor]:
    """Slice every tensor's leading (batch) dimension to this rank's shard (views, no copies)."""
    out = []
    for t in tensors:
        lo, hi = shard_bounds(t.shape[0], rank, world_size)
        out.append(t[lo:hi])
    return out


def gather_to_host(local: torch.Tensor, batch: int, group=None, dst: int = 0) -> Optional[torch.Tensor]:
    """Concatenate the per-rank shards (in shard order) on rank ``dst``; other ranks get ``None``.

    ``local`` is this rank's ``(hi-lo, ...)`` result; it is copied to the host first, so the exchange is
    host-side and never touches the GPU interconnect.  Without an initialised process group this is a
    plain device-to-host copy.
    """
    import torch.distributed as dist

    host = local.detach().cpu().contiguous()
    if not (dist.is_available() and dist.is_initialized()):
        if host.shape[0] != batch:
            raise ValueError("single process: the local result must cover the whole batch")
        return host
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_bounds(batch, rank, world)
    if host.shape[0] != hi - lo:
        raise ValueError(f"rank {rank}: expected {hi - lo} rows, got {host.shape[0]}")
    parts: List[Optional[torch.Tensor]] = [None] * world
    dist.gather_object(host, parts if rank == dst else None, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([p for p in parts if p is not None and p.shape[0] > 0] or [host[:0]], dim=0)


class SharedHostGather:
    """The final host gather of a batch that is sharded over the GPUs of one node, with no collective: ONE host buffer in
    POSIX shared memory holds the outputs of the whole (global) batch in shard order, every rank maps it, and every rank
    puts its own images' rows in their place.  On a GPU the rows travel in two hops, both off the enqueuing thread: a copy
    kernel on the rank's stream writes them into a pinned staging ring, and the worker thread of the native runtime
    (``frcnn_host_gather_*``, csrc/host_gather.hip) moves a staged step into the shared buffer as soon as the step's event
    has completed.  ``write_step`` is therefore ONE C-ABI call per step and blocks only for back-pressure, i.e. when the
    step that used the same staging slot ``slots`` steps ago has still not been published.  (Writing from the GPU straight
    into the shared mapping needs ``hipHostRegister``, which this platform refuses for ordinary users, shared or anonymous
    mappings alike: measured, rc = invalid argument.)

    Consistency: ``view(step, name)`` is the whole batch only after EVERY rank has called ``flush()`` and the ranks have
    passed a barrier, and only for the last ``slots`` steps (step ``s`` lives in slot ``s % slots`` and is overwritten by
    step ``s + slots``).  Between two such points a slot may hold rows of different steps from different ranks.

    The reference is single-device and returns host arrays from ``predict`` (faster_rcnn_predictor.py:56); this is what
    that becomes when the batch is cut over several processes.

    Args:
        sections: ``(name, per_image_shape, dtype)`` of every gathered output, e.g. ``("roi_bboxes", (300, 4), torch.float32)``
            (at most 8).  A row (one image of one section) must be a multiple of 16 bytes.
        batch: images of the global batch.
        slots: ring depth of the staging ring and of the shared buffer.
        rank, world_size: this process's position; default from the initialised process group, else 0 of 1.
        device: the GPU that produces the rows (None = no GPU: rows are host tensors; used by the CPU tests of the layout).
        name: shared-memory name; rank 0 creates it and broadcasts it through the process group when omitted.  With an
            explicit name and no process group the other ranks retry for up to 30 s until rank 0 has created it.
    """

    def __init__(self, sections: Sequence[Tuple[str, Tuple[int, ...], torch.dtype]], batch: int, slots: int = 2,
                 rank: Optional[int] = None, world_size: Optional[int] = None, device: Optional[torch.device] = None,
                 name: Optional[str] = None, group=None) -> None:
        import time
        from multiprocessing import shared_memory

        import numpy as np
        import torch.distributed as dist

        have_pg = dist.is_available() and dist.is_initialized()
        self.rank = (dist.get_rank(group) if have_pg else 0) if rank is None else rank
        self.world = (dist.get_world_size(group) if have_pg else 1) if world_size is None else world_size
        self.batch, self.slots, self.device = int(batch), int(slots), device
        if self.slots < 1:
            raise ValueError("slots must be >= 1")
        if len(sections) > 8:
            raise ValueError("at most 8 sections (one copy launch per step)")
        self.lo, self.hi = shard_bounds(self.batch, self.rank, self.world)
        n_loc = self.hi - self.lo
        self._layout = {}
        off = loc = 0
        for sec_name, shape, dtype in sections:
            row = int(np.prod(shape, dtype=np.int64)) * torch.empty((), dtype=dtype).element_size()
            if row % 16:
                raise ValueError(f"section {sec_name}: one image is {row} bytes, must be a multiple of 16")
            self._layout[sec_name] = (off, tuple(shape), dtype, row, loc)  # offsets in the shared slot / in the staging slot
            off += (row * self.batch + 255) // 256 * 256
            loc += (row * n_loc + 255) // 256 * 256
        self.slot_bytes, self._stage_bytes = off, max(loc, 256)
        total = max(off * self.slots, 256)
        self._shm = None
        self._handle = None
        if self.rank == 0:
            self._shm = shared_memory.SharedMemory(create=True, size=total, name=name)
            names = [self._shm.name]
        else:
            names = [name]
        if name is None and self.world > 1:
            if not have_pg:
                raise ValueError("world_size > 1 needs an initialised process group or an explicit name")
            dist.broadcast_object_list(names, src=0, group=group)  # also orders creation before the other ranks' open
        elif self.world > 1 and have_pg:
            dist.barrier(group)
        if self.rank != 0:
            deadline = time.monotonic() + (0.0 if have_pg else 30.0)  # no process group: nothing orders us after rank 0's create
            while True:
                try:
                    self._shm = shared_memory.SharedMemory(name=names[0])
                    if self._shm.size >= total:
                        break
                    self._shm.close()  # created but not yet sized
                    self._shm = None
                    err: Exception = RuntimeError(f"shared memory {names[0]!r} is smaller than {total} bytes")
                except FileNotFoundError as exc:
                    err = exc
                if time.monotonic() >= deadline:
                    raise RuntimeError(f"rank {self.rank}: shared memory {names[0]!r} of rank 0 is not there: {err}") from err
                time.sleep(0.01)
            try:  # the creator unlinks; keep this process's resource tracker from doing it a second time at exit
                from multiprocessing import resource_tracker
                resource_tracker.unregister(self._shm._name, "shared_memory")  # type: ignore[attr-defined]
            except Exception:
                pass
        self._np = np.frombuffer(self._shm.buf, dtype=np.uint8, count=total)
        self._host = torch.from_numpy(self._np)
        self._stage = None
        self._tables = None
        if device is not None:
            import ctypes

            from . import _lib
            with _lib.on_device(device):
                self._stage = [torch.empty(self._stage_bytes, dtype=torch.uint8, pin_memory=True) for _ in range(self.slots)]
                h = ctypes.c_void_p()
                _lib.check(_lib.lib().frcnn_host_gather_create(self.slots, ctypes.byref(h)))
                self._handle = h
            # per slot: the staging and destination addresses and the sizes of this rank's rows, section by section
            base = self._host.data_ptr()
            ns = len(self._layout)
            self._tables = []
            for k in range(self.slots):
                st = (ctypes.c_void_p * ns)(*[self._stage[k].data_ptr() + loc_ for (_, _, _, _, loc_) in self._layout.values()])
                ds = (ctypes.c_void_p * ns)(*[base + k * self.slot_bytes + off_ + self.lo * row_
                                              for (off_, _, _, row_, _) in self._layout.values()])
                self._tables.append((st, ds))
            self._sizes = (ctypes.c_size_t * ns)(*[n_loc * row_ for (_, _, _, row_, _) in self._layout.values()])
            self._srcs = (ctypes.c_void_p * ns)()
            self._submit = _lib.lib().frcnn_host_gather_submit

    def view(self, step: int, name: str) -> torch.Tensor:
        """Host tensor ``(batch, *per_image_shape)`` of section ``name`` for ``step`` (slot ``step % slots`` of the shared
        buffer, the same memory in every rank).  See the class docstring for when it is consistent."""
        off, shape, dtype, row, _ = self._layout[name]
        start = (step % self.slots) * self.slot_bytes + off
        return self._host[start:start + row * self.batch].view(dtype).view((self.batch,) + shape)

    def write_step(self, step: int, outputs, stream: Optional[int] = None) -> None:
        """Put this rank's rows of every section for ``step`` on their way: ``outputs[name]`` is the local
        ``(hi-lo, *per_image_shape)`` tensor.  On a GPU: one call into the native runtime (copy kernel + event on ``stream``,
        the host copy happens on its worker thread)."""
        n = self.hi - self.lo
        k = step % self.slots
        gpu = self.device is not None
        for i, (name, (off, shape, dtype, row, loc)) in enumerate(self._layout.items()):
            local = outputs[name]
            if local.numel() * local.element_size() != n * row or local.dtype != dtype or not local.is_contiguous():
                raise ValueError(f"{name}: expected a contiguous {(n,) + shape} {dtype} tensor, got {tuple(local.shape)} {local.dtype}")
            if n == 0:
                continue
            if gpu:
                self._srcs[i] = local.data_ptr()
            else:
                start = k * self.slot_bytes + off + self.lo * row
                self._host[start:start + n * row].copy_(local.reshape(-1).view(torch.uint8))
        if gpu and n > 0:
            from . import _lib
            if stream is None:
                stream = _lib.stream_ptr(self.device)
            st, ds = self._tables[k]
            with _lib.on_device(self.device):
                _lib.check(self._submit(self._handle, k, self._srcs, st, ds, self._sizes, len(self._layout), stream))

    def flush(self) -> None:
        """Block until every step written so far is in the shared buffer."""
        if self._handle is not None:
            from . import _lib
            _lib.check(_lib.lib().frcnn_host_gather_flush(self._handle))

    def stats(self):
        """``{"publish_seconds", "submit_wait_seconds", "jobs"}`` of the native runtime (zeros without a GPU)."""
        out = {"publish_seconds": 0.0, "submit_wait_seconds": 0.0, "jobs": 0}
        if self._handle is not None:
            import ctypes

            from . import _lib
            a, b, c = ctypes.c_double(), ctypes.c_double(), ctypes.c_longlong()
            _lib.check(_lib.lib().frcnn_host_gather_stats(self._handle, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
            out = {"publish_seconds": a.value, "submit_wait_seconds": b.value, "jobs": int(c.value)}
        return out

    def close(self) -> None:
        """Publish what is queued, stop the worker, unmap; the creating rank removes the shared memory.  Idempotent."""
        handle, self._handle = getattr(self, "_handle", None), None
        if handle is not None:
            from . import _lib
            _lib.lib().frcnn_host_gather_destroy(handle)  # joins the worker: nothing writes the mapping afterwards
        shm = getattr(self, "_shm", None)
        if shm is None:
            return
        self._host = None
        self._np = None
        self._stage = None
        self._tables = None
        try:
            shm.close()
        except BufferError:
            pass
        if self.rank == 0:
            try:
                shm.unlink()
            except FileNotFoundError:
                pass
        self._shm = None

    def __del__(self) -> None:
        try:
            self.close()
        except Exception:
            pass
