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
