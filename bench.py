#!/usr/bin/env python3
"""bench.py -- images/s through the proposal/RoI hot path (BASELINE.json metric).

Default workload = BASELINE config 2: one step = RoIBBox (decode + top-6000 + NMS -> 300 RoIs) -> RoIPooling (7x7
crop_and_resize) -> Decoder (per-class NMS) over one batch of 16 synthetic images (mobilenet_v2, 500x500 -> 32x32x576
feature map, 9216 anchors, 21 labels), inputs resident in HBM.  The conv backbone and the dense heads are out of scope:
their outputs are synthetic tensors.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python bench.py --gpus N ...            # spawns N ranks itself (torch.distributed.run) when WORLD_SIZE is unset
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

    --config 2   weak scaling: every rank runs its own batch of 16 (default; the headline)
    --config 3   vgg16 train-mode RoI path, ONE global batch of 32 cut over the ranks (strong scaling)
    --config 4   RPN target assignment, ONE global batch of 64 cut over the ranks (strong scaling)
    --config 5   1000x1000 stress, ONE global batch of 128 cut over the ranks (strong scaling)
    --config 1   the reference's own predictor case, batch 1

Multi-GPU: pure data parallel over images, one process per GPU, contiguous image shards (sharding.shard_bounds), no
collective on the data path; torch.distributed is used only for the barrier and the max-over-ranks time.  The small
per-image outputs are gathered on the host inside the timed region: per rank into pinned memory (config 2: every rank is
an independent replica), into one shared host buffer in shard order (configs 3-5: sharding.SharedHostGather).
The K timed steps are repeated `--blocks` times.  Each block is bracketed by barrier + synchronize; a rank's clock stops
when ITS work is complete (streams drained, its rows in the shared host buffer) and BEFORE the closing barrier, so no
collective sits inside the timed bracket; the block's time is the maximum over the ranks of those local times (per-rank
minimum beside it).  `value` is the median block, min / max are reported beside it.  Every step reads a different one of >= depth input sets,
so no step finds its inputs in L2 / Infinity Cache.  Rank 0 prints ONE JSON line.
"""

from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md)

# every n-th launch of the dominant kernel is bracketed with HIP events on its stream (each pair adds two packets to the
# gap between consecutive launches: bracketing all of them costs 2-4 % of the pipeline's throughput)
TIME_EVERY = int(os.environ.get("FRCNN_BENCH_TIME_EVERY", "2"))

# untimed spin-up in front of every other_configs measurement (each starts after hundreds of milliseconds of host-side set-up,
# i.e. on a GPU whose clocks have dropped again); disclosed in the line as other_configs.spin_up_ms_each
OTHER_SPIN_MS = 60.0

METRIC = "images/sec through RPN decode+NMS+RoIPool+class NMS (bs16, 500²); HBM GB/s"

CONFIGS = {
    1: dict(kind="infer", backbone="mobilenet_v2", B=1, G=32, C=576, K=6000, R=300, L=21, scaling="weak",
            title="BASELINE config 1: faster_rcnn_predictor.py mobilenet_v2, 500x500, batch 1 (RoIBBox+RoIPooling+Decoder)"),
    2: dict(kind="infer", backbone="mobilenet_v2", B=16, G=32, C=576, K=6000, R=300, L=21, scaling="weak",
            title="BASELINE config 2: mobilenet_v2 RoI path, batch 16 per GPU, 500x500, 9216 anchors/img, pre_nms_topn 6000, "
                  "post_nms 300, 7x7 crop_and_resize of 32x32x576, 21-label Decoder NMS"),
    3: dict(kind="train", backbone="vgg16", B=32, G=31, C=512, K=6000, R=2000, L=21, M=32, scaling="strong",
            title="BASELINE config 3: vgg16 RoI path (31x31x512 feature map, 8649 anchors/img), global batch 32, train mode: "
                  "RoIBBox post_nms 2000 -> RoIPooling 7x7 -> RoIDelta (32 padded gt boxes/img)"),
    4: dict(kind="targets", backbone="mobilenet_v2", B=64, G=32, M=100, scaling="strong",
            title="BASELINE config 4: RPN training targets, anchors (9216) x 100 gt boxes IoU + calculate_rpn_actual_outputs, "
                  "global batch 64"),
    5: dict(kind="infer", backbone="mobilenet_v2", B=128, G=63, C=576, K=12000, R=300, L=21, scaling="strong", img_size=1000,
            title="BASELINE config 5: 1000x1000 stress, 63x63x576 feature map, 35721 anchors/img, pre_nms_topn 12000, "
                  "post_nms 300, global batch 128"),
}
# per-GPU shards of configs 3-5 as BASELINE.json deploys them (4 / 8 / 8 GPUs), timed on one GPU after the headline
SHARDS = {3: 8, 4: 8, 5: 16}


# ----------------------------------------------------------------------------------------------------------------------
# launching
# ----------------------------------------------------------------------------------------------------------------------
def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` from a plain shell: start N fresh rank processes (one per GPU) with
    torch.distributed.run and return their exit code.  Runs BEFORE this process makes any GPU call -- a process that has
    initialised the GPU must never be replaced or forked into workers."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


# ----------------------------------------------------------------------------------------------------------------------
# algorithmic bytes
# ----------------------------------------------------------------------------------------------------------------------
def crop_bytes_per_image(cfg) -> int:
    """crop_and_resize: write R*49*C*4, read H*W*C*4 + R*20."""
    return cfg["R"] * 49 * cfg["C"] * 4 + cfg["G"] * cfg["G"] * cfg["C"] * 4 + cfg["R"] * 20


def targets_bytes_per_image(cfg) -> int:
    """Fused target assignment (no map written): gt M*16 + labels M*4 in; max/argmax N*8 in and out; deltas N*16 + labels N*4 out."""
    N = cfg["G"] * cfg["G"] * 9
    return cfg["M"] * 20 + N * 8 * 2 + N * 20


def iou_map_bytes_per_image(cfg) -> int:
    N = cfg["G"] * cfg["G"] * 9
    return N * cfg["M"] * 4 + cfg["M"] * 16


def pmc_traffic_bytes(kernel_substr: str):
    """(HBM bytes per launch of one batch, the files they come from) from the committed rocprofv3 --pmc passes
    (profiles/r*_pmc_*): separate FETCH_SIZE and WRITE_SIZE runs, KB units, FETCH_SIZE doubled (gfx950 counts 128-B read
    requests as 64 B, MI355X_MICROARCH.md section HBM).  The most recent round's pass wins.  (None, None) if the profiles
    are not there.  This is a figure measured in an earlier run of the same workload, not in this one: a --pmc pass
    serialises the kernels and cannot ride along with the timed run."""
    import csv
    import glob

    def avg(pattern, counter):
        vals = []
        paths = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)),
                       key=lambda q: tuple((os.path.basename(q).split("_pmc_")[0] + "_").split("_")[:2]))
        for path in paths[-1:]:
            with open(path) as fh:
                for row in csv.DictReader(fh):
                    if row["Counter_Name"] == counter and kernel_substr in row["Kernel_Name"]:
                        vals.append(float(row["Counter_Value"]))
        return (sum(vals) / len(vals) if vals else None), (os.path.relpath(paths[-1], ROOT) if paths else None)

    fetch, f_src = avg("r*_pmc_fetch_size_*.csv", "FETCH_SIZE")
    write, w_src = avg("r*_pmc_write_size_*.csv", "WRITE_SIZE")
    if fetch is None or write is None:
        return None, None
    return int((2.0 * fetch + write) * 1024), [f_src, w_src]


# ----------------------------------------------------------------------------------------------------------------------
# workloads
# ----------------------------------------------------------------------------------------------------------------------
class Workload:
    """One config's step on this rank's shard of `b_local` images, over `n_sets` rotating input sets."""

    def __init__(self, pkg, cfg, b_local, seed, device, n_sets, shard=None):
        """`shard` = (lo, hi, global batch): the inputs are generated for the whole global batch from `seed` and this rank
        keeps images [lo, hi) (strong scaling: ONE batch cut over the ranks); None: `b_local` images of its own."""
        import numpy as np
        import torch

        import synth

        self.pkg, self.cfg, self.B, self.device, self.torch = pkg, cfg, b_local, device, torch
        over = {}
        if cfg.get("img_size"):
            over.update(img_size=cfg["img_size"])
        over.update(feature_map_shape=cfg["G"])
        if "K" in cfg:
            over.update(pre_nms_topn=cfg["K"])
        hp = pkg.train_utils.get_hyper_params(cfg["backbone"], **over)
        kind = cfg["kind"]
        if kind == "train":
            hp["train_nms_topn"] = cfg["R"]
        elif kind == "infer":
            hp["test_nms_topn"] = cfg["R"]
        hp["total_labels"] = cfg.get("L", 21)
        self.hyper = hp
        self.anchors = pkg.bbox_utils.generate_anchors(hp)
        N = self.anchors.shape[0]
        B, G = b_local, cfg["G"]
        gen_b, lo, hi = (shard[2], shard[0], shard[1]) if shard else (B, 0, B)
        up = lambda a: torch.as_tensor(np.ascontiguousarray(a[lo:hi])).to(device)  # noqa: E731
        self.host_set0 = None
        self.sets = []
        # set 0 (and, where generation is cheap, every set) is drawn from numpy's default_rng(seed) as SURVEY.md 8d
        # prescribes; for the large configs the further sets are set 0 with the images rotated (distinct addresses, same
        # statistics): what matters for the measurement is that consecutive steps do not re-read cache-resident inputs
        cheap = gen_b * G * G * cfg.get("C", 0) <= 16 * 32 * 32 * 576
        self.set_seeds = []
        for s in range(n_sets):
            if s == 0 or cheap:
                sd = seed * 1000 + s
                self.set_seeds.append(sd)
                if kind == "targets":
                    gt, lab = synth.gt_boxes(sd, gen_b, cfg["M"])
                    host = (gt, lab)
                else:
                    d, sc = synth.rpn_outputs(sd, gen_b, G)
                    fm = synth.feature_map(sd, gen_b, G, cfg["C"])
                    if kind == "infer":
                        hd, hpb = synth.head_outputs(sd, gen_b, cfg["R"], cfg["L"])
                        host = (d, sc, fm, hd, hpb)
                    else:
                        gt, lab = synth.gt_boxes(sd, gen_b, cfg["M"])
                        host = (d, sc, fm, gt, lab)
                if s == 0:
                    self.host_set0 = host
                self.sets.append(tuple(up(a) for a in host))
            else:
                self.set_seeds.append(None)
                self.sets.append(tuple(torch.roll(t, s % max(B, 1), 0).contiguous() if B > 1 else t.clone() for t in self.sets[0]))
        self.lanes = [torch.cuda.Stream(device) for _ in range(2)] if kind == "targets" else None
        self.last_stream = None  # the stream the last step's outputs were produced on (targets: the step's lane)
        self.target_out = None   # targets: one recycled (deltas, labels) pair per lane
        if kind == "targets":
            self.dominant = "iou_reduce_kernel + rpn_targets_kernel"
            self.dominant_bytes = targets_bytes_per_image(cfg) * B + N * 16
            self.bound = "valu/latency (no map written; HBM bytes reported for completeness)"
        else:
            self.roi_bbox = pkg.faster_rcnn.RoIBBox(self.anchors, "training" if kind == "train" else "inference", hp, name="roi_bboxes")
            self.roi_pool = pkg.faster_rcnn.RoIPooling(hp, name="roi_pooling")
            if kind == "infer":
                self.decoder = pkg.faster_rcnn.Decoder(hp["variances"], cfg["L"], name="faster_rcnn_decoder")
            else:
                self.roi_delta = pkg.faster_rcnn.RoIDelta(hp, name="roi_deltas")
            # (frcnn_roi_pooling's choice: feature maps beyond an XCD's L2, or a launch of at most 1024 boxes)
            staged = cfg["G"] * cfg["G"] * cfg["C"] * 4 > 4 * 1024 * 1024
            self.dominant = ("crop_resize_rowstage_kernel<1024> (sequential schedule: the bracket also holds its pre-pass, "
                             "crop_rows_order_kernel + crop_rows_records_kernel; stage pipeline: the pre-pass runs on the proposal "
                             "stream and the bracket holds one crop launch per batch of the group)" if staged else
                             "crop_resize_slice_kernel" if B * cfg["R"] <= 1024 else "crop_resize_kernel<4>")
            self.dominant_bytes = crop_bytes_per_image(cfg) * B
            self.bound = "hbm"
        self.seed = seed

    def make_pipeline(self, depth, group, host_outputs=False, gather=None):
        """The stage pipeline for this workload: inference (RoIBBox | RoIPooling | Decoder) or training (RoIBBox | RoIPooling,
        RoIDelta beside the crop); None for target assignment."""
        kind = self.cfg["kind"]
        if kind == "targets" or self.B <= 0:
            return None
        depth = max(depth, group)
        if kind == "infer":
            return self.pkg.pipeline.RoIPathPipeline(self.roi_bbox, self.roi_pool, self.decoder, depth=depth, crop_group=group,
                                                     host_outputs=host_outputs, gather=gather)
        return self.pkg.pipeline.RoIPathPipeline(self.roi_bbox, self.roi_pool, depth=depth, crop_group=group, roi_delta=self.roi_delta,
                                                 host_outputs=host_outputs, gather=gather)

    def submit(self, pipe, i, ev=None):
        """One step handed to the stage pipeline (every step is a full pass over its batch; the stages of consecutive steps
        overlap).  `ev` brackets the crop kernel on ITS stream; the step index rides along as the gather's tag."""
        s = self.sets[i % len(self.sets)]
        if self.cfg["kind"] == "infer":
            pipe.submit(s[0], s[1], s[2], (s[3], s[4]), timing=ev, tag=i)
        else:
            pipe.submit(s[0], s[1], s[2], None, timing=ev, gt=(s[3], s[4]), seed=self.seed + i, tag=i)

    def sections(self):
        """(name, per-image shape, dtype) of the small outputs that go to the host every step."""
        t, c = self.torch, self.cfg
        if c["kind"] == "infer":
            T = int(self.decoder.max_total_size)
            return [("roi_bboxes", (c["R"], 4), t.float32), ("bboxes", (T, 4), t.float32), ("labels", (T,), t.float32),
                    ("scores", (T,), t.float32)]
        if c["kind"] == "train":
            return [("roi_bboxes", (c["R"], 4), t.float32)]  # RoIDelta's targets feed the loss on the producing GPU
        N = self.anchors.shape[0]
        return [("bbox_labels", (N,), t.float32)]  # (the deltas, 147 KB/img, feed the loss on the producing GPU)

    def step(self, i, ev=None):
        """One pass over input set i; `ev` = (start, end) events recorded around the dominant kernel on the current stream.
        Returns the small outputs as {section: tensor}."""
        s = self.sets[i % len(self.sets)]
        pkg, kind = self.pkg, self.cfg["kind"]
        cur = self.torch.cuda.current_stream(self.device).cuda_stream
        if kind == "targets":
            # consecutive batches go to two alternating HIP streams (the feeders' prefetch lanes, utils/train_utils.py): the
            # chip-wide IoU reductions of batch i + 1 run beside the one-workgroup-per-image target kernel of batch i
            # (stream= and out= are the wrapper's extra arguments for feeder loops: no stream switch, no allocation; each
            # lane recycles its own pair of output tensors, stream order keeps step i + 2 behind step i's gather)
            gt, lab = s
            li = i % len(self.lanes) if self.lanes else 0
            lane = self.lanes[li] if self.lanes else None
            q = lane.cuda_stream if lane is not None else cur
            if self.target_out is None:
                N = self.anchors.shape[0]
                self.target_out = [(self.torch.empty((self.B, N, 4), device=self.device), self.torch.empty((self.B, N), device=self.device))
                                   for _ in range(max(1, len(self.lanes or [])))]
                self.torch.cuda.synchronize(self.device)
            if ev:
                ev[0].record(q)
            deltas, labels = pkg.train_utils.calculate_rpn_actual_outputs(self.anchors, gt, lab, self.hyper, seed=self.seed + i,
                                                                          out=self.target_out[li], stream=q)
            if ev:
                ev[1].record(q)
            self.last_stream = q
            return {"bbox_labels": labels.view(self.B, -1)}
        rois = self.roi_bbox([s[0], s[1]])
        if ev:
            ev[0].record(cur)
        pooled = self.roi_pool([s[2], rois])  # stays on the GPU for the dense heads
        if ev:
            ev[1].record(cur)
        if kind == "infer":
            boxes, labels, scores = self.decoder([rois, s[3], s[4]])
            return {"roi_bboxes": rois, "bboxes": boxes, "labels": labels, "scores": scores}
        self.roi_delta([rois, s[3], s[4]], seed=self.seed + i)
        return {"roi_bboxes": rois}


class IouMapCase:
    """generate_iou_map (utils/bbox_utils.py:138-160), the map materialised: anchors (N,4) x gt (B,M,4) -> (B,N,M).  SURVEY.md
    8d's algorithmic bytes: N*M*4 + M*16 per image, + N*16 (the shared anchors) once per launch."""

    def __init__(self, pkg, cfg, B, seed, device, n_sets=4):
        import torch

        import synth
        hp = pkg.train_utils.get_hyper_params(cfg["backbone"], feature_map_shape=cfg["G"])
        self.pkg, self.B = pkg, B
        self.anchors = pkg.bbox_utils.generate_anchors(hp)
        N = self.anchors.shape[0]
        self.sets = [torch.as_tensor(synth.gt_boxes(seed * 1000 + s, B, cfg["M"])[0]).to(device) for s in range(n_sets)]
        self.bytes = iou_map_bytes_per_image(cfg) * B + N * 16
        self.cur = torch.cuda.current_stream(device).cuda_stream

    def step(self, i, ev=None):
        if ev:
            ev[0].record(self.cur)
        out = self.pkg.bbox_utils.generate_iou_map(self.anchors, self.sets[i % len(self.sets)])
        if ev:
            ev[1].record(self.cur)
        return out


class _NullCtx:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NULL_CTX = _NullCtx()


def stats(samples):
    import numpy as np
    if not samples:
        return None
    a = np.asarray(samples, dtype=np.float64)
    return {"mean": float(a.mean()), "p10": float(np.percentile(a, 10)), "p50": float(np.percentile(a, 50)),
            "p90": float(np.percentile(a, 90)), "n": int(a.size)}


# ----------------------------------------------------------------------------------------------------------------------
# CPU baseline
# ----------------------------------------------------------------------------------------------------------------------
def host_cores():
    """What this job may use of the host: (os.cpu_count(), CPUs in the affinity mask, cgroup CPU quota or None)."""
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, p = fh.read().split()
            if q != "max":
                quota = float(q) / float(p)
    except Exception:
        pass
    return os.cpu_count() or 1, avail, quota


def cpu_baseline(wl, seconds=12.0):
    """The CPU oracle (reference-semantics restatement, OpenMP over images / boxes) on the host cores, same workload,
    set 0 of the inputs: reported baseline only.  Timed with 1 thread, 16 threads and every core this job may use."""
    import oracle

    cfg, hyper = wl.cfg, wl.hyper
    deltas, scores, fm, hd, hp = wl.host_set0
    anchors = oracle.generate_anchors(hyper)

    def one_batch():
        rois = oracle.roi_bbox(deltas, scores, anchors, hyper["variances"], cfg["K"], cfg["R"], hyper["nms_iou_threshold"])
        oracle.roi_pooling(fm, rois, hyper["pooling_size"])
        oracle.decoder(rois, hd, hp, hyper["variances"], cfg["L"])

    count, avail, quota = host_cores()
    all_threads = max(1, min(avail, int(quota) if quota and quota >= 1 else avail))
    plan = [("1", 1)]
    if all_threads > 16:
        plan.append(("16", 16))
    plan.append(("all", all_threads))
    out = {}
    for label, threads in plan:
        oracle.set_threads(threads)
        one_batch()  # warm
        n, t0 = 0, time.perf_counter()
        while True:
            one_batch()
            n += 1
            dt = time.perf_counter() - t0
            if dt >= seconds / len(plan) or n >= 200:
                break
        out[label] = (threads, n, cfg["B"] * n / dt)
    threads, n, ips = out["all"]
    return {
        "value": round(ips, 2), "unit": "images/s", "cores": threads, "kind": "port",
        "sample": f"{n} batches of {cfg['B']} images (config 2, input set 0), oracle/ C restatement of the reference's TF "
                  f"semantics, OpenMP over images/boxes, {threads} threads",
        "host": {"os_cpu_count": count, "affinity_cpus": avail, "cgroup_cpu_quota": quota,
                 "OMP_NUM_THREADS_env": os.environ.get("OMP_NUM_THREADS")},
        "by_threads": {label: {"threads": t, "images_per_s": round(v, 2), "batches": k} for label, (t, k, v) in out.items()},
    }


# ----------------------------------------------------------------------------------------------------------------------
def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--spin-up-ms", type=float, default=300.0,
                    help="untimed steps of the same workload for this many milliseconds before the W warm-up steps (GPU clocks "
                         "ramp up over tens of milliseconds of load; 0 = none).  One block timed BEFORE it is reported as cold_start")
    ap.add_argument("--blocks", type=int, default=5, help="timed blocks of --steps steps each; value = the median block")
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=None, help="override the config's (global / per-GPU) batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", choices=("pipeline", "sequential"), default=None,
                    help="pipeline (default for configs 2, 3, 5): stages on three HIP streams, --depth batches in flight "
                         "(tf_faster_rcnn_amd.pipeline); sequential: the layers one after the other on one stream")
    ap.add_argument("--depth", type=int, default=None, help="batches in flight in the stage pipeline (default 8; 4 for configs 3 / 5)")
    ap.add_argument("--crop-group", type=int, default=None,
                    help="pipeline mode: consecutive batches that share one launch per stage (the *_multi entry points); "
                         "default 2 for config 2, 1 otherwise")
    ap.add_argument("--input-sets", type=int, default=0, help="distinct input sets rotated through the steps (0 = automatic)")
    ap.add_argument("--no-host-gather", action="store_true",
                    help="leave the small per-image outputs (RoIs, detections) on the GPU instead of copying them to host "
                         "memory every step (the sharded deployment's final host gather, SURVEY.md 8e)")
    ap.add_argument("--dist-backend", default="nccl", help="process-group backend for the barrier / max-over-ranks (nccl = RCCL)")
    ap.add_argument("--device", type=int, default=None,
                    help="GPU index for every rank (default LOCAL_RANK modulo the visible GPUs); lets a 1-GPU box rehearse "
                         "the N-rank launch path")
    ap.add_argument("--no-other-schedule", action="store_true",
                    help="do not time the other schedule afterwards (profiling runs: one schedule per kernel trace)")
    ap.add_argument("--dump-gather", default=None,
                    help="configs 3-5: rank 0 saves the gathered host arrays of the last step (npz, with the seed of its input "
                         "set) -- what tests/test_gpu_dist.py compares with the oracle")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group (and use its barrier / all-reduce) even with one rank: lets a 1-GPU box run "
                         "the RCCL branch of the N-rank path")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="do not time config 1, the per-GPU shards of configs 3-5 and the materialised IoU map after the headline")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus)  # nothing above touched the GPU

    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n_dev = max(torch.cuda.device_count(), 1)
    dev_index = (local_rank % n_dev) if args.device is None else args.device
    dist = None
    backend = args.dist_backend
    backend_note = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist  # noqa: F811
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:  # --force-dist from a plain shell: a one-rank group
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl" and (world > n_dev or args.device is not None):
            backend, backend_note = "gloo", "ranks share a GPU: RCCL needs one device per rank, barrier/max over gloo instead"
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)

    import tf_faster_rcnn_amd as pkg
    from tf_faster_rcnn_amd import sharding

    cfg = dict(CONFIGS[args.config])
    if args.batch:
        cfg["B"] = args.batch
    strong = cfg["scaling"] == "strong"
    shard = None
    if strong:
        if world > cfg["B"]:
            raise SystemExit(f"config {args.config}: {world} ranks for a global batch of {cfg['B']} images")
        lo, hi = sharding.shard_bounds(cfg["B"], rank, world)
        b_local, b_global = hi - lo, cfg["B"]
        shard = (lo, hi, b_global)
    else:
        b_local, b_global = cfg["B"], cfg["B"] * world
    mode = args.mode or ("pipeline" if (args.config in (2, 3, 5) and cfg["kind"] != "targets") else "sequential")
    if mode == "pipeline" and cfg["kind"] == "targets":
        raise SystemExit("--mode pipeline is the RoIBBox | RoIPooling | Decoder / RoIDelta stage pipeline (configs 1, 2, 3, 5)")
    gather = not args.no_host_gather
    depth = args.depth or (8 if args.config == 2 else 4)
    group = args.crop_group or (2 if args.config in (2, 5) else 1)  # (config 3: one 1.6 GB crop per launch is enough; measured)

    def auto_sets(c, b, pipelined):
        if args.input_sets:
            return args.input_sets
        per_set = b * (c["G"] * c["G"] * c.get("C", 0) * 4 + c["G"] * c["G"] * 9 * 20)
        want = int(np.ceil(640e6 / max(per_set, 1)))  # > the 256 MiB Infinity Cache, with margin
        return int(min(max(want, 3, (depth + 2) if pipelined else 3), 16))

    n_sets = auto_sets(cfg, max(b_local, 1), mode == "pipeline")
    wl = Workload(pkg, cfg, b_local, seed=1 if strong else rank + 1, device=device, n_sets=n_sets, shard=shard)
    cur = torch.cuda.current_stream(device).cuda_stream
    timing_flags = pkg._lib.hipEventDisableSystemFence  # timing events without the system-scope fence: a default event makes
    # the kernel it follows write back and invalidate L2, which is exactly what the measurement must not add

    def local_sync(pipe, sh):
        """This rank's work is complete: streams drained and (sharded configs) its rows in the shared host buffer."""
        if pipe is not None:
            pipe.synchronize()
        torch.cuda.synchronize()
        if sh is not None:
            sh.flush()

    def barrier():
        if dist is not None:
            dist.barrier()

    def over_ranks(x):
        """(max, min) of a local figure over the ranks."""
        if dist is None:
            return x, x
        t = torch.tensor([x, -x], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0].item()), -float(t[1].item())

    def run_blocks(step_fn, steps, warmup, blocks, every, pipe=None, sh=None, spin_ms=0.0, cold_block=False):
        """`blocks` timed blocks of `steps` steps.  Returns a dict: per-block seconds (max over ranks) and their per-rank minima,
        the event pairs' milliseconds, host enqueue seconds per step, the index of the last step, and -- with cold_block -- the
        same figures of ONE block timed before the spin-up."""
        n_ev = (blocks + (1 if cold_block else 0)) * ((steps + every - 1) // every) if every else 0
        evs = [(pkg._lib.HipEvent(timing_flags), pkg._lib.HipEvent(timing_flags)) for _ in range(n_ev)]
        state = {"g": 0, "ev": 0}

        def one_block():
            local_sync(pipe, sh)
            barrier()
            used = []
            wait0 = pipe.wait_seconds if pipe is not None else 0.0
            t0 = time.perf_counter()
            for i in range(steps):
                ev = None
                if every and i % every == 0:
                    ev = evs[state["ev"]]
                    state["ev"] += 1
                    used.append(ev)
                step_fn(state["g"], ev)
                state["g"] += 1
            host = time.perf_counter() - t0 - ((pipe.wait_seconds - wait0) if pipe is not None else 0.0)
            local_sync(pipe, sh)
            local = time.perf_counter() - t0  # the clock stops BEFORE the closing barrier: no collective inside the bracket
            barrier()
            t_max, t_min = over_ranks(local)
            return t_max, t_min, host, [a.elapsed_time(b) for a, b in used]

        out = {}
        for _ in range(warmup):
            step_fn(state["g"], None)
            state["g"] += 1
        if cold_block:
            # One block timed BEFORE the spin-up: what the driver's `--warmup W` alone would have measured on a GPU that was
            # idle a moment ago (clocks still ramping).  Reported as cold_start, never part of `value`.
            t_max, _, _, smp = one_block()
            out["cold"] = (t_max, smp)
        # Device spin-up (untimed, disclosed as "spin_up_ms" in the JSON line): the same steps for a fixed wall time.  A fresh
        # process finds the GPU in its idle power state and the clocks keep rising for some tens of milliseconds of load: with
        # --steps 20 --warmup 5 the timed blocks of one run read 0.154, 0.164, 0.153, 0.149, 0.145, ... 0.136 ms per step, i.e.
        # the first blocks measure the power management, not the kernels.
        t_spin = time.perf_counter()
        n_spin = 0  # (its own counter: the number of spin-up steps differs from rank to rank, the step index must not)
        while (time.perf_counter() - t_spin) * 1e3 < spin_ms:
            for _ in range(16):
                step_fn(n_spin, None)
                n_spin += 1
            local_sync(pipe, sh)
        if spin_ms > 0:
            for _ in range(warmup):  # the contract's W warm-up steps, directly in front of the timed blocks
                step_fn(state["g"], None)
                state["g"] += 1
        times, mins, host, samples = [], [], [], []
        for _ in range(blocks):
            t_max, t_min, h, smp = one_block()
            times.append(t_max)
            mins.append(t_min)
            host.append(h)
            samples += smp
        out.update(times=times, rank_min=mins, samples=samples, host_s=float(np.median(host)) / max(steps, 1), last_step=state["g"] - 1)
        return out

    # ---- the host gather of the small outputs ------------------------------------------------------------------------
    def make_gather(w, b_glob, rk, wd, pipelined):
        """(shared, pinned): strong-scaling configs gather into ONE shared host buffer in shard order
        (sharding.SharedHostGather), config 2's replicas into per-rank pinned memory."""
        if not gather:
            return None, None
        if w.cfg["scaling"] == "strong":
            return sharding.SharedHostGather(w.sections(), b_glob, slots=max(4, depth + 1) if pipelined else 4, rank=rk,
                                             world_size=wd, device=device), None
        if pipelined:
            return None, None  # the pipeline's own host_outputs
        return None, {n: torch.empty((w.B,) + tuple(shape), dtype=dt, pin_memory=True) for n, shape, dt in w.sections()}

    def sequential_step(w, sh, pin):
        def step(i, ev):
            outs = w.step(i, ev)
            q = w.last_stream or cur
            if sh is not None:
                sh.write_step(i, outs, q)
            elif pin:  # the host gather of the small outputs: one copy kernel for all of them
                pkg._lib.gather_copy_multi([(outs[name].data_ptr(), dst.data_ptr(), outs[name].numel() * 4) for name, dst in pin.items()], q)
            return outs
        return step

    def pipeline_for(w, sh, d, g):
        hook = (lambda tag, outs, q: sh.write_step(tag, outs, q)) if sh is not None else None
        return w.make_pipeline(d, g, host_outputs=gather and sh is None, gather=hook)

    shared, pinned = make_gather(wl, b_global, rank, world, mode == "pipeline")
    pipe = pipeline_for(wl, shared, depth, group) if (mode == "pipeline" or (world == 1 and not args.no_other_schedule)) and cfg["kind"] != "targets" else None
    step_pipeline = (lambda i, ev: wl.submit(pipe, i, ev)) if pipe is not None else None

    if mode == "pipeline":
        step, every, run_pipe = step_pipeline, TIME_EVERY * group, pipe
    else:
        step, every, run_pipe = sequential_step(wl, shared, pinned), TIME_EVERY, None
    main_run = run_blocks(step, args.steps, args.warmup, args.blocks, every, run_pipe, shared, spin_ms=args.spin_up_ms,
                          cold_block=args.spin_up_ms > 0)
    times, samples, host_s = main_run["times"], main_run["samples"], main_run["host_s"]
    gather_stats = shared.stats() if shared is not None else None

    launch_batches = group if mode == "pipeline" else 1  # batches one launch of the dominant kernel covers
    dom = stats(samples)

    def frac_of(w, launch_ms, batches=1):
        return w.dominant_bytes * batches / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS

    other = None
    if world == 1 and not args.no_other_schedule and cfg["kind"] != "targets":
        # the other schedule, same K steps, reported beside the headline (not part of `value`)
        if mode == "pipeline":
            o_sh, o_pin = make_gather(wl, b_global, rank, world, False)
            o = run_blocks(sequential_step(wl, o_sh, o_pin), args.steps, min(args.warmup, 10), 3, 0, None, o_sh)
            if o_sh is not None:
                o_sh.close()
        else:
            o = run_blocks(step_pipeline, args.steps, min(args.warmup, 10), 3, 0, pipe, shared)
        o_med = float(np.median(o["times"]))
        other = {"mode": "sequential" if mode == "pipeline" else "pipeline", "value": round(b_global * args.steps / o_med, 1),
                 "ms_per_step": round(o_med / args.steps * 1e3, 4)}

    # The same schedule on two other input regimes, reported beside the headline (never part of `value`):
    #  warm inputs       -- ONE input set re-read every step, i.e. feature maps served by the Infinity Cache: round 1's
    #                       measurement condition, and what a backbone that has just written them would leave behind;
    #  clustered proposals -- RPN deltas shrunk by 20x, so the decoded boxes sit on their anchors and overlap heavily: the
    #                       lazy proposal kernel consumes far more candidates before 300 boxes are kept (its slow regime).
    other_inputs = None
    if world == 1 and args.config == 2 and not args.no_other_configs and mode == "pipeline":
        other_inputs = {}
        all_sets = wl.sets
        for name in ("warm_inputs", "clustered_proposals"):
            if name == "warm_inputs":
                wl.sets = all_sets[:1]
            else:
                wl.sets = [(s[0] * 0.05,) + tuple(s[1:]) for s in all_sets]
            o = run_blocks(step_pipeline, args.steps, min(args.warmup, 10), 3, every, pipe)
            med = float(np.median(o["times"]))
            d = stats(o["samples"])
            other_inputs[name] = {"value": round(b_global * args.steps / med, 1), "ms_per_step": round(med / args.steps * 1e3, 4),
                                  "crop_launch_ms": round(d["mean"], 5),
                                  "crop_frac_of_hbm_peak": round(frac_of(wl, d["mean"], launch_batches), 4),
                                  "input_sets": len(wl.sets)}
        wl.sets = all_sets

    def kernel_report(name, bound, d, nbytes):
        ach = nbytes / (d["mean"] * 1e-3) / 1e9
        return {"kernel": name, "bound": bound, "avg_launch_ms": round(d["mean"], 5),
                "launch_ms_p10_p50_p90": [round(d[q], 5) for q in ("p10", "p50", "p90")], "launches_timed": d["n"],
                "bytes_per_launch": nbytes, "achieved_GBps": round(ach, 1), "frac_of_hbm_peak": round(ach / HBM_PEAK_GBS, 4)}

    others = None
    if world == 1 and args.config == 2 and not args.no_other_configs:
        # Config 1 and ONE GPU's shard of configs 3 / 4 / 5 as BASELINE.json deploys them, each through the host path of that
        # deployment: the shard's small outputs go into a SharedHostGather (world 1 here: same calls, same host work per step).
        others = {"spin_up_ms_each": OTHER_SPIN_MS}
        k = min(args.steps, 30)
        for c_id in (1, 3, 4, 5):
            c = dict(CONFIGS[c_id])
            c["B"] = SHARDS.get(c_id, c["B"])
            piped = c["kind"] != "targets"  # (config 1 too: consecutive single images of a predict() loop in flight)
            w = Workload(pkg, c, c["B"], seed=7 + c_id, device=device, n_sets=auto_sets(c, c["B"], piped))
            sh, pin = make_gather(w, c["B"], 0, 1, False)
            o = run_blocks(sequential_step(w, sh, pin), k, 5, 3, TIME_EVERY, None, sh, spin_ms=OTHER_SPIN_MS)
            med = float(np.median(o["times"]))
            d = stats(o["samples"])
            key = f"config{c_id}" + ("_shard" if c_id in SHARDS else "")
            entry = {"workload": c["title"] + (f" -- one GPU's shard: {c['B']} images" if c_id in SHARDS else ""),
                     "schedule": ("consecutive batches on two alternating HIP streams" if c["kind"] == "targets" else
                                  "sequential on one stream"), "images_per_step": c["B"], "steps": k, "blocks": 3,
                     "value": round(c["B"] * k / med, 1), "unit": "images/s", "ms_per_step": round(med / k * 1e3, 4),
                     "host_enqueue_ms_per_step": round(o["host_s"] * 1e3, 4),
                     "host_gather": ("shared host buffer (sharding.SharedHostGather, native worker thread)" if sh is not None else
                                     "pinned host memory" if pin else "none"),
                     "dominant_kernel": kernel_report(w.dominant, w.bound, d, w.dominant_bytes)}
            if sh is not None:
                st = sh.stats()
                entry["host_gather_publish_ms_per_step"] = round(st["publish_seconds"] / max(st["jobs"], 1) * 1e3, 4)
                entry["host_gather_submit_wait_ms_per_step"] = round(st["submit_wait_seconds"] / max(st["jobs"], 1) * 1e3, 4)
                sh.close()
            if piped:
                # the same shard through the stage pipeline (4 batches in flight): the deployed schedule of configs 1, 3 and 5;
                # its figures become the entry's headline, the sequential ones (one batch at a time: the latency view)
                # stay beside them
                sh2, _ = make_gather(w, c["B"], 0, 1, True)
                # (config 1: single images, four to a launch of each stage -- the reference's predictor feeds batches of 4,
                # faster_rcnn_predictor.py:19 -- and eight in flight)
                p_depth, p_group = (8, 4) if c_id == 1 else ((4, 2) if c_id == 5 else (4, 1))
                p2 = pipeline_for(w, sh2, p_depth, p_group)
                o2 = run_blocks(lambda i, ev, w=w, p2=p2: w.submit(p2, i, ev), k, 5, 3, p_group, p2, sh2, spin_ms=OTHER_SPIN_MS)
                med2 = float(np.median(o2["times"]))
                d2 = stats(o2["samples"])
                piped_entry = dict(schedule=f"stage pipeline on 3 HIP streams, {p_depth} batches in flight, {p_group} per launch", value=round(c["B"] * k / med2, 1),
                                   ms_per_step=round(med2 / k * 1e3, 4), host_enqueue_ms_per_step=round(o2["host_s"] * 1e3, 4))
                if med2 >= med:  # (batch 1: the pipeline's host work per batch exceeds the GPU time it hides) keep the sequential headline
                    entry["stage_pipeline"] = piped_entry
                    if sh2 is not None:
                        sh2.close()
                    del p2
                    others[key] = entry
                    del w, pin
                    torch.cuda.empty_cache()
                    continue
                entry["sequential"] = {kk: entry[kk] for kk in ("value", "ms_per_step", "host_enqueue_ms_per_step", "dominant_kernel")}
                entry.update(schedule=piped_entry["schedule"] +
                             (" (train mode: RoIDelta beside the crop)" if c["kind"] == "train" else ""),
                             value=round(c["B"] * k / med2, 1), ms_per_step=round(med2 / k * 1e3, 4),
                             host_enqueue_ms_per_step=round(o2["host_s"] * 1e3, 4),
                             dominant_kernel=dict(kernel_report(w.dominant, w.bound, d2, w.dominant_bytes * p_group),
                                                  batches_per_timed_bracket=p_group))
                if sh2 is not None:
                    st = sh2.stats()
                    entry["host_gather_publish_ms_per_step"] = round(st["publish_seconds"] / max(st["jobs"], 1) * 1e3, 4)
                    entry["host_gather_submit_wait_ms_per_step"] = round(st["submit_wait_seconds"] / max(st["jobs"], 1) * 1e3, 4)
                    sh2.close()
                del p2
            others[key] = entry
            del w, pin
            torch.cuda.empty_cache()
        # generate_iou_map with the map materialised (north_star's second HBM-roofline kernel): config 4's global batch on one
        # GPU and one GPU's shard of it.  Off the sharded path -- target assignment never writes the map -- and timed for the
        # kernel's roofline only: every launch bracketed with HIP events.
        for label, b_map in (("iou_map_cfg4", CONFIGS[4]["B"]), ("iou_map_cfg4_shard", SHARDS[4])):
            case = IouMapCase(pkg, CONFIGS[4], b_map, seed=11, device=device)
            o = run_blocks(case.step, k, 5, 3, 1, spin_ms=OTHER_SPIN_MS)
            med = float(np.median(o["times"]))
            others[label] = {"workload": f"generate_iou_map (utils/bbox_utils.py:138-160), map materialised: anchors (9216,4) x gt ({b_map},100,4) "
                                         f"-> ({b_map},9216,100) fp32", "images_per_step": b_map, "steps": k, "blocks": 3,
                             "value": round(b_map * k / med, 1), "unit": "images/s", "ms_per_step": round(med / k * 1e3, 4),
                             "dominant_kernel": kernel_report("iou_map_rows_kernel", "hbm (write) / valu (one correctly rounded divide per pair)",
                                                              stats(o["samples"]), case.bytes)}
            del case
            torch.cuda.empty_cache()

    devices = [dev_index]
    if dist is not None:
        gathered = [None] * world
        dist.all_gather_object(gathered, (rank, dev_index, torch.cuda.get_device_name(device)))
        devices = [d for _, d, _ in sorted(gathered)]

    if rank == 0:
        med = float(np.median(times))
        value = b_global * args.steps / med
        dom_bytes = wl.dominant_bytes * launch_batches
        achieved = dom_bytes / (dom["mean"] * 1e-3) / 1e9 if dom else float("nan")
        traffic = traffic_source = None
        if cfg["kind"] != "targets" and args.config == 2:
            t, traffic_source = pmc_traffic_bytes("crop_resize_kernel")
            traffic = None if t is None else t * launch_batches
        result = {
            "metric": METRIC,
            "value": round(value, 1),
            "unit": "images/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "spin_up_ms": args.spin_up_ms,
            "ms_per_step": round(med / args.steps * 1e3, 4),
            "blocks": {"n": args.blocks, "value_median": round(value, 1),
                       "value_min": round(b_global * args.steps / max(times), 1),
                       "value_max": round(b_global * args.steps / min(times), 1),
                       "ms_per_step_each": [round(t / args.steps * 1e3, 4) for t in times],
                       "ms_per_step_each_fastest_rank": [round(t / args.steps * 1e3, 4) for t in main_run["rank_min"]],
                       "clock": "per rank: stops when its streams are drained and its rows are in the host buffer, before the "
                                "closing barrier; block time = max over ranks"},
            "host_enqueue_ms_per_step": round(host_s * 1e3, 4),
            "higher_is_better": True,
            "scaling": cfg["scaling"],
            "vs_baseline": None,
            "dtype": "f32",
            "data": f"synthetic (SURVEY.md 8d distributions), {len(wl.sets)} input sets rotated through the steps",
            "config": {"workload": cfg["title"], "config_id": args.config,
                       "global_batch": b_global, "images_per_rank": b_local if not strong else
                       [sharding.shard_bounds(b_global, r, world)[1] - sharding.shard_bounds(b_global, r, world)[0] for r in range(world)],
                       "parallelism": f"dp{world} over images, contiguous shards, no collectives",
                       "ranks": world, "devices": devices, "world_size_observed": dist.get_world_size() if dist else 1,
                       "dist_backend": (backend if dist else None), "dist_backend_note": backend_note,
                       "schedule": (f"stage pipeline on 3 HIP streams, {depth} batches in flight, "
                                    f"{group} consecutive batch(es) per launch of each stage" +
                                    (" (train mode: RoIDelta beside the crop)" if cfg["kind"] == "train" else "")
                                    if mode == "pipeline" else
                                    "consecutive batches on two alternating HIP streams" if cfg["kind"] == "targets" else
                                    "sequential on one stream"),
                       "host_gather": ("none" if not gather else
                                       "every step, inside the timed region: " +
                                       ("each rank's GPU copies its images' small outputs to pinned staging (copy kernel), the native "
                                        "worker thread moves them into ONE shared host buffer in shard order (sharding.SharedHostGather)"
                                        if shared is not None else
                                        "RoIs and detections of every batch copied to pinned host memory (copy kernel)") +
                                       "; pooled features stay on the GPU")},
            "roofline": {"kernel": wl.dominant, "bound": "hbm", "bound_note": wl.bound,
                         "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": (None if traffic is None else
                                            {"files": traffic_source, "note": "rocprofv3 --pmc passes of an earlier run of this workload "
                                             "(WRITE_SIZE + 2 x FETCH_SIZE, KB), not collected in this run"}),
                         "bytes_per_launch": dom_bytes, "avg_launch_ms": round(dom["mean"], 5) if dom else None,
                         "launch_ms_p10_p50_p90": [round(dom[q], 5) for q in ("p10", "p50", "p90")] if dom else None,
                         "launches_timed": dom["n"] if dom else 0},
        }
        if "cold" in main_run:
            c_t, c_smp = main_run["cold"]
            c_d = stats(c_smp)
            result["cold_start"] = {"note": f"one block of {args.steps} steps timed after {args.warmup} warm-up steps and BEFORE the spin-up "
                                            "(GPU clocks still ramping): what --warmup alone measures in a fresh process",
                                    "value": round(b_global * args.steps / c_t, 1), "ms_per_step": round(c_t / args.steps * 1e3, 4),
                                    "crop_frac": (round(dom_bytes / (c_d["mean"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                                                  if c_d and cfg["kind"] != "targets" else None),
                                    "dominant_launch_ms": round(c_d["mean"], 5) if c_d else None}
        if gather_stats is not None and gather_stats["jobs"]:
            result["host_gather_publish_ms_per_step"] = round(gather_stats["publish_seconds"] / gather_stats["jobs"] * 1e3, 4)
            result["host_gather_submit_wait_ms_per_step"] = round(gather_stats["submit_wait_seconds"] / gather_stats["jobs"] * 1e3, 4)
        if other is not None:
            result["other_schedule"] = other
        if other_inputs:
            result["other_inputs"] = other_inputs
        if others:
            result["other_configs"] = others
        if world == 1 and not args.no_cpu_baseline and args.config == 2:
            result["cpu_baseline"] = cpu_baseline(wl)
        print(json.dumps(result), flush=True)
    if args.dump_gather and shared is not None:
        local_sync(pipe, shared)
        barrier()
        if rank == 0:
            last = main_run["last_step"]
            np.savez(args.dump_gather, set_seed=np.int64(-1 if wl.set_seeds[last % len(wl.sets)] is None else wl.set_seeds[last % len(wl.sets)]),
                     global_batch=np.int64(b_global), world=np.int64(world),
                     **{name: shared.view(last, name).numpy().copy() for name, _, _ in wl.sections()})
    if dist is not None:
        dist.barrier()
    del pipe
    if shared is not None:
        shared.close()
    if dist is not None:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
