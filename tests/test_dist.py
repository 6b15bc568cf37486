"""Multi-process (world_size 2, gloo, CPU) test of the N>1 path: contiguous image shards, no data-path
collective, final host gather in shard order.  The per-image work is stood in for by the CPU oracle here
(the GPU kernels need a device); what is under test is the sharding/gather plumbing bench.py and a
multi-GPU caller use."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, batch: int, out_path: str) -> None:
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle
        import synth
        from tf_faster_rcnn_amd import sharding

        G, K, R = 8, 64, 16
        hp = synth.hyper_params(feature_map_shape=G, pre_nms_topn=K, test_nms_topn=R)
        anchors = oracle.generate_anchors(hp)
        deltas, scores = synth.rpn_outputs(0, batch, G)  # every rank builds the same global batch ...
        d_loc, s_loc = sharding.shard([torch.from_numpy(deltas), torch.from_numpy(scores)], rank, world)  # ... and keeps its shard
        lo, hi = sharding.shard_bounds(batch, rank, world)
        assert d_loc.shape[0] == hi - lo
        if hi > lo:
            rois = oracle.roi_bbox(d_loc.numpy(), s_loc.numpy(), anchors, hp["variances"], K, R, 0.7)
        else:
            rois = np.zeros((0, R, 4), np.float32)
        full = sharding.gather_to_host(torch.from_numpy(rois), batch)
        # barrier + max-over-ranks timing contract of bench.py
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert t.item() == world
        if rank == 0:
            want = oracle.roi_bbox(deltas, scores, anchors, hp["variances"], K, R, 0.7)
            assert full is not None and tuple(full.shape) == want.shape
            np.save(out_path, np.array([np.array_equal(full.numpy(), want)]))
        else:
            assert full is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("batch", [4, 5, 1])
def test_sharded_path_matches_single_process(tmp_path, batch):
    out = str(tmp_path / "ok.npy")
    mp.spawn(_worker, args=(2, _free_port(), batch, out), nprocs=2, join=True)
    assert np.load(out)[0]


def _gather_worker(rank: int, world: int, port: int, batch: int, out_path: str) -> None:
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sh = None
    try:
        from tf_faster_rcnn_amd import sharding

        sections = [("roi_bboxes", (5, 4), torch.float32), ("valid", (4,), torch.int32)]
        sh = sharding.SharedHostGather(sections, batch, slots=2, device=None)  # no GPU here: rows are copied by the host
        lo, hi = sharding.shard_bounds(batch, rank, world)
        assert (sh.lo, sh.hi) == (lo, hi)
        full_r = torch.arange(batch * 20, dtype=torch.float32).view(batch, 5, 4)
        full_v = torch.arange(batch * 4, dtype=torch.int32).view(batch, 4)
        for step in range(3):  # three steps through a ring of two slots
            sh.write_step(step, {"roi_bboxes": (full_r[lo:hi] + step).contiguous(), "valid": (full_v[lo:hi] * (step + 1)).contiguous()})
        sh.flush()
        dist.barrier()  # every rank's rows are in place
        ok = True
        for step in (1, 2):  # the two steps still in the ring, seen whole by EVERY rank
            ok &= torch.equal(sh.view(step, "roi_bboxes"), full_r + step)
            ok &= torch.equal(sh.view(step, "valid"), full_v * (step + 1))
        dist.barrier()  # (the failing write below may touch this rank's rows before it raises)
        with pytest.raises(ValueError):
            sh.write_step(0, {"roi_bboxes": full_r[lo:hi].contiguous(), "valid": full_v[lo:hi].float()})
        dist.barrier()
        np.save(f"{out_path}.{rank}.npy", np.array([ok]))
    finally:
        if sh is not None:
            sh.close()
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("batch", [6, 5, 2])
def test_shared_host_gather_layout(tmp_path, batch):
    """sharding.SharedHostGather: every rank writes its shard's rows into ONE shared host buffer in shard order; after a
    barrier every rank sees the whole batch.  (On a GPU the rows are written by frcnn_gather_copy: tests/test_gpu_dist.py.)"""
    out = str(tmp_path / "ok")
    mp.spawn(_gather_worker, args=(2, _free_port(), batch, out), nprocs=2, join=True)
    assert np.load(out + ".0.npy")[0] and np.load(out + ".1.npy")[0]
