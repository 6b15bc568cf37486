"""The N > 1 path on a GPU: `python bench.py --gpus 2` from a plain invocation starts two ranks (both on GPU 0 here: the
box has one GPU, so the barrier runs over gloo), each rank runs the HIP kernels on ITS shard of one global batch, and the
GPUs write their rows into the shared host buffer (sharding.SharedHostGather).  The gathered RoIs / detections are
compared bit for bit with the oracle run on the whole batch."""

import json
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle
import synth

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_bench(tmp_path, *extra):
    dump = str(tmp_path / "gather.npz")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--device", "0", "--dist-backend", "gloo",
           "--steps", "3", "--warmup", "1", "--blocks", "2", "--no-cpu-baseline", "--dump-gather", dump] + list(extra)
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line), np.load(dump)


def test_two_ranks_config3_rois_match_oracle(tmp_path):
    """Config 3 (vgg16, train mode, R = 2000: the whole-chip bitmask NMS schedule) with a global batch of 5 cut 3 + 2."""
    B = 5
    res, got = _run_bench(tmp_path, "--config", "3", "--batch", str(B))
    assert res["n_gpus"] == 2 and res["config"]["world_size_observed"] == 2 and res["config"]["devices"] == [0, 0]
    assert res["scaling"] == "strong" and res["config"]["images_per_rank"] == [3, 2] and res["config"]["global_batch"] == B
    assert res["value"] > 0 and res["roofline"]["launches_timed"] >= 2
    seed = int(got["set_seed"])
    assert seed >= 0
    hp = synth.hyper_params("vgg16", train_nms_topn=2000)
    deltas, scores = synth.rpn_outputs(seed, B, 31)
    want = oracle.roi_bbox(deltas, scores, oracle.generate_anchors(hp), hp["variances"], 6000, 2000, 0.7)
    assert got["roi_bboxes"].shape == want.shape
    assert np.array_equal(got["roi_bboxes"].view(np.uint32), want.view(np.uint32)), "gathered RoIs differ from the oracle"
    assert (np.abs(want).sum(-1) > 0).sum() > B * 500  # real proposals from both shards, not padding


def test_two_ranks_config5_detections_match_oracle(tmp_path):
    """Config 5's shape (63x63 grid, pre_nms_topn 12000) with a global batch of 3 cut 2 + 1: RoIs and detections."""
    B, L, R = 3, 21, 300
    res, got = _run_bench(tmp_path, "--config", "5", "--batch", str(B), "--input-sets", "2")
    assert res["n_gpus"] == 2 and res["config"]["images_per_rank"] == [2, 1]
    seed = int(got["set_seed"])
    assert seed >= 0
    hp = synth.hyper_params("mobilenet_v2", img_size=1000, feature_map_shape=63, pre_nms_topn=12000)
    deltas, scores = synth.rpn_outputs(seed, B, 63)
    hd, hpb = synth.head_outputs(seed, B, R, L)
    rois = oracle.roi_bbox(deltas, scores, oracle.generate_anchors(hp), hp["variances"], 12000, R, 0.7)
    boxes, labels, sc = oracle.decoder(rois, hd, hpb, hp["variances"], L)
    for name, want in (("roi_bboxes", rois), ("bboxes", boxes), ("labels", labels), ("scores", sc)):
        assert np.array_equal(got[name].view(np.uint32), want.view(np.uint32)), f"gathered {name} differ from the oracle"


def test_two_ranks_weak_scaling_headline(tmp_path):
    """The default workload (config 2, weak scaling, stage pipeline) under two ranks: one JSON line from rank 0 that says so."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--device", "0", "--dist-backend", "gloo",
           "--steps", "6", "--warmup", "2", "--blocks", "2", "--input-sets", "3"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["scaling"] == "weak" and res["config"]["global_batch"] == 32
    assert res["config"]["devices"] == [0, 0] and "cpu_baseline" not in res and res["value"] > 0


def test_one_rank_over_rccl(tmp_path):
    """The RCCL branch of the N-rank path (process group "nccl" on the rank's own device, barrier, max-over-ranks all-reduce on
    the device, all_gather_object) with the one rank a 1-GPU box can give it: config 4's shard through the shared host gather."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--config", "4", "--batch", "8", "--steps", "5", "--warmup", "2",
           "--blocks", "2", "--no-cpu-baseline", "--spin-up-ms", "0"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    res = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["config"]["dist_backend"] == "nccl" and res["config"]["world_size_observed"] == 1 and res["value"] > 0
    assert res["blocks"]["ms_per_step_each"] == res["blocks"]["ms_per_step_each_fastest_rank"]  # one rank: max == min
