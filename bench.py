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
The K timed steps are repeated `--blocks` times (each block bracketed by barrier + synchronize, max over ranks);
`value` is the median block, min / max are reported beside it.  Every step reads a different one of >= depth input sets,
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
    """HBM bytes per launch of one batch from the committed rocprofv3 --pmc passes (profiles/r*_pmc_*): separate FETCH_SIZE
    and WRITE_SIZE runs, KB units, FETCH_SIZE doubled (gfx950 counts 128-B read requests as 64 B, MI355X_MICROARCH.md
    section HBM).  The most recent round's pass wins.  None if the profiles are not there."""
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
        return sum(vals) / len(vals) if vals else None

    fetch = avg("r*_pmc_fetch_size_*.csv", "FETCH_SIZE")
    write = avg("r*_pmc_write_size_*.csv", "WRITE_SIZE")
    if fetch is None or write is None:
        return None
    return int((2.0 * fetch + write) * 1024)


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
            slice_kernel = cfg["G"] * cfg["G"] * cfg["C"] * 4 > 4 * 1024 * 1024 or B * cfg["R"] <= 1024
            self.dominant = "crop_resize_slice_kernel" if slice_kernel else "crop_resize_kernel<4>"
            self.dominant_bytes = crop_bytes_per_image(cfg) * B
            self.bound = "hbm"
        self.seed = seed

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
            gt, lab = s
            if ev:
                ev[0].record(cur)
            deltas, labels = pkg.train_utils.calculate_rpn_actual_outputs(self.anchors, gt, lab, self.hyper, seed=self.seed + i)
            if ev:
                ev[1].record(cur)
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
                         "ramp up over tens of milliseconds of load; 0 = none)")
    ap.add_argument("--blocks", type=int, default=5, help="timed blocks of --steps steps each; value = the median block")
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=None, help="override the config's (global / per-GPU) batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", choices=("pipeline", "sequential"), default=None,
                    help="pipeline (default for config 2): stages on three HIP streams, --depth batches in flight "
                         "(tf_faster_rcnn_amd.pipeline); sequential: the layers one after the other on one stream")
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--crop-group", type=int, default=2,
                    help="pipeline mode: consecutive batches that share one launch per stage (the *_multi entry points)")
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
    ap.add_argument("--no-other-configs", action="store_true",
                    help="do not time config 1 and the per-GPU shards of configs 3-5 after the headline")
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
    if world > 1:
        import torch.distributed as dist  # noqa: F811
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
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
    mode = args.mode or ("pipeline" if (args.config == 2 and cfg["kind"] == "infer") else "sequential")
    if mode == "pipeline" and cfg["kind"] != "infer":
        raise SystemExit("--mode pipeline is the RoIBBox -> RoIPooling -> Decoder stage pipeline (configs 1, 2, 5)")
    gather = not args.no_host_gather

    def auto_sets(c, b):
        if args.input_sets:
            return args.input_sets
        per_set = b * (c["G"] * c["G"] * c.get("C", 0) * 4 + c["G"] * c["G"] * 9 * 20)
        want = int(np.ceil(640e6 / max(per_set, 1)))  # > the 256 MiB Infinity Cache, with margin
        return int(min(max(want, 3, (args.depth + 2) if mode == "pipeline" else 3), 16))

    n_sets = auto_sets(cfg, max(b_local, 1))
    wl = Workload(pkg, cfg, b_local, seed=1 if strong else rank + 1, device=device, n_sets=n_sets, shard=shard)
    cur = torch.cuda.current_stream(device).cuda_stream
    timing_flags = pkg._lib.hipEventDisableSystemFence  # timing events without the system-scope fence: a default event makes
    # the kernel it follows write back and invalidate L2, which is exactly what the measurement must not add

    def barrier(pipe=None):
        if pipe is not None:
            pipe.synchronize()
        torch.cuda.synchronize()
        if shared is not None:
            shared.flush()  # the last steps' rows reach the shared host buffer inside the timed region
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    shared = None

    def max_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def run_blocks(step_fn, steps, warmup, blocks, every, pipe=None, spin_ms=0.0):
        """`blocks` timed blocks of `steps` steps.  Returns (per-block seconds [max over ranks], event pairs, host enqueue s/step)."""
        n_ev = blocks * ((steps + every - 1) // every) if every else 0
        evs = [(pkg._lib.HipEvent(timing_flags), pkg._lib.HipEvent(timing_flags)) for _ in range(n_ev)]
        used = []
        # Device spin-up (untimed, BEFORE the W warm-up steps, disclosed as "spin_up_ms" in the JSON line): the same steps for
        # a fixed wall time.  A fresh process finds the GPU in its idle power state and the clocks keep rising for some tens
        # of milliseconds of load: with --steps 20 --warmup 5 the timed blocks of one run read 0.154, 0.164, 0.153, 0.149,
        # 0.145, ... 0.136 ms per step, i.e. the first blocks measure the power management, not the kernels.
        t_spin = time.perf_counter()
        n_spin = 0
        while (time.perf_counter() - t_spin) * 1e3 < spin_ms:
            for _ in range(16):
                step_fn(n_spin, None)
                n_spin += 1
            if pipe is not None:
                pipe.synchronize()
            torch.cuda.synchronize()
        for i in range(warmup):
            step_fn(i, None)
        times, host = [], []
        g = warmup
        for _ in range(blocks):
            barrier(pipe)
            wait0 = pipe.wait_seconds if pipe is not None else 0.0
            t0 = time.perf_counter()
            for i in range(steps):
                ev = None
                if every and i % every == 0:
                    ev = evs[len(used)]
                    used.append(ev)
                step_fn(g, ev)
                g += 1
            host.append(time.perf_counter() - t0 - ((pipe.wait_seconds - wait0) if pipe is not None else 0.0))
            barrier(pipe)
            times.append(max_over_ranks(time.perf_counter() - t0))
        samples = [a.elapsed_time(b) for a, b in used]
        return times, samples, float(np.median(host)) / max(steps, 1)

    # ---- the host gather of the small outputs ------------------------------------------------------------------------
    pinned = None
    if gather and mode == "sequential":
        if strong:
            shared = sharding.SharedHostGather(wl.sections(), b_global, slots=4, rank=rank, world_size=world, device=device)
        else:
            pinned = {n: torch.empty((b_local,) + tuple(shape), dtype=dt, pin_memory=True) for n, shape, dt in wl.sections()}

    def step_sequential(i, ev, w=None, sh=None, pin=None):
        w = w or wl
        outs = w.step(i, ev)
        sh = shared if sh is None else sh
        pin = pinned if pin is None else pin
        if sh:
            sh.write_step(i, outs, cur)
        elif pin:  # the host gather of the small outputs: one copy kernel for all of them
            pkg._lib.gather_copy_multi([(outs[name].data_ptr(), dst.data_ptr(), outs[name].numel() * 4) for name, dst in pin.items()], cur)
        return outs

    pipe = None
    group = args.crop_group
    if cfg["kind"] == "infer" and b_local > 0:
        pipe = pkg.pipeline.RoIPathPipeline(wl.roi_bbox, wl.roi_pool, wl.decoder, depth=args.depth, crop_group=group,
                                            host_outputs=gather)

    def step_pipeline(i, ev):
        # every step is a full pass (RoIBBox -> RoIPooling -> Decoder) over its batch; the stages of up to --depth
        # consecutive steps overlap on three streams.  The events bracket the crop kernel on ITS stream (with
        # --crop-group g the first batch of a group carries the events of the group's one launch).
        s = wl.sets[i % len(wl.sets)]
        pipe.submit(s[0], s[1], s[2], (s[3], s[4]), timing=ev)

    if mode == "pipeline":
        step, every = step_pipeline, TIME_EVERY * group
    else:
        step, every = step_sequential, TIME_EVERY
    times, samples, host_s = run_blocks(step, args.steps, args.warmup, args.blocks, every, pipe if mode == "pipeline" else None,
                                        spin_ms=args.spin_up_ms)

    launch_batches = group if mode == "pipeline" else 1  # batches one launch of the dominant kernel covers
    dom = stats(samples)

    other = None
    if world == 1 and not args.no_other_schedule and cfg["kind"] == "infer":
        # the other schedule, same K steps, reported beside the headline (not part of `value`)
        if mode == "pipeline":
            pinned = {n: torch.empty((b_local,) + tuple(shape), dtype=dt, pin_memory=True) for n, shape, dt in wl.sections()} if gather else None
            o_times, _, _ = run_blocks(step_sequential, args.steps, min(args.warmup, 10), 3, 0)
        else:
            o_times, _, _ = run_blocks(step_pipeline, args.steps, min(args.warmup, 10), 3, 0, pipe)
        o_med = float(np.median(o_times))
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
            t, smp, _ = run_blocks(step_pipeline, args.steps, min(args.warmup, 10), 3, every, pipe)
            med = float(np.median(t))
            d = stats(smp)
            ach = wl.dominant_bytes * launch_batches / (d["mean"] * 1e-3) / 1e9
            other_inputs[name] = {"value": round(b_global * args.steps / med, 1), "ms_per_step": round(med / args.steps * 1e3, 4),
                                  "crop_launch_ms": round(d["mean"], 5), "crop_frac_of_hbm_peak": round(ach / HBM_PEAK_GBS, 4),
                                  "input_sets": len(wl.sets)}
        wl.sets = all_sets

    others = None
    if world == 1 and args.config == 2 and not args.no_other_configs:
        others = {}
        for c_id in (1, 3, 4, 5):
            c = dict(CONFIGS[c_id])
            c["B"] = SHARDS.get(c_id, c["B"])
            w = Workload(pkg, c, c["B"], seed=7 + c_id, device=device, n_sets=auto_sets(c, c["B"]))
            pin = {n: torch.empty((c["B"],) + tuple(shape), dtype=dt, pin_memory=True) for n, shape, dt in w.sections()} if gather else None
            st = lambda i, ev, w=w, pin=pin: step_sequential(i, ev, w=w, sh=False, pin=pin)  # noqa: E731
            k = min(args.steps, 30)
            t, smp, _ = run_blocks(st, k, 5, 3, 1)
            med = float(np.median(t))
            d = stats(smp)
            ach = w.dominant_bytes / (d["mean"] * 1e-3) / 1e9
            others[f"config{c_id}" + ("_shard" if c_id in SHARDS else "")] = {
                "workload": c["title"] + (f" -- one GPU's shard: {c['B']} images" if c_id in SHARDS else ""),
                "schedule": "sequential on one stream", "images_per_step": c["B"], "steps": k, "blocks": 3,
                "value": round(c["B"] * k / med, 1), "unit": "images/s", "ms_per_step": round(med / k * 1e3, 4),
                "dominant_kernel": {"kernel": w.dominant, "bound": w.bound, "avg_launch_ms": round(d["mean"], 5),
                                    "launch_ms_p10_p50_p90": [round(d[q], 5) for q in ("p10", "p50", "p90")],
                                    "launches_timed": d["n"], "bytes_per_launch": w.dominant_bytes,
                                    "achieved_GBps": round(ach, 1), "frac_of_hbm_peak": round(ach / HBM_PEAK_GBS, 4)}}
            if c["kind"] == "infer" and c["B"] > 1:  # the same shard through the stage pipeline (4 batches in flight)
                p2 = pkg.pipeline.RoIPathPipeline(w.roi_bbox, w.roi_pool, w.decoder, depth=4, crop_group=1, host_outputs=gather)
                sp = lambda i, ev, w=w, p2=p2: p2.submit(*w.sets[i % len(w.sets)][:3], tuple(w.sets[i % len(w.sets)][3:5]))  # noqa: E731
                t2, _, _ = run_blocks(sp, k, 5, 3, 0, p2)
                others[f"config{c_id}_shard"]["stage_pipeline_value"] = round(c["B"] * k / float(np.median(t2)), 1)
                del p2, sp
            del w, pin
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
        traffic = None
        if cfg["kind"] != "targets" and args.config == 2:
            t = pmc_traffic_bytes("crop_resize_kernel")
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
                       "ms_per_step_each": [round(t / args.steps * 1e3, 4) for t in times]},
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
                       "schedule": (f"stage pipeline on 3 HIP streams, {args.depth} batches in flight, "
                                    f"{group} consecutive batch(es) per launch of each stage" if mode == "pipeline"
                                    else "sequential on one stream"),
                       "host_gather": ("none" if not gather else
                                       "every step, inside the timed region: " +
                                       ("each rank's GPU copies its images' small outputs to pinned staging (copy kernel), the host "
                                        "moves them into ONE shared host buffer in shard order (sharding.SharedHostGather)" if shared else
                                        "RoIs and detections of every batch copied to pinned host memory (copy kernel)") +
                                       "; pooled features stay on the GPU")},
            "roofline": {"kernel": wl.dominant, "bound": "hbm", "bound_note": wl.bound,
                         "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "bytes_per_launch": dom_bytes, "avg_launch_ms": round(dom["mean"], 5) if dom else None,
                         "launch_ms_p10_p50_p90": [round(dom[q], 5) for q in ("p10", "p50", "p90")] if dom else None,
                         "launches_timed": dom["n"] if dom else 0},
        }
        if other is not None:
            result["other_schedule"] = other
        if other_inputs:
            result["other_inputs"] = other_inputs
        if others:
            result["other_configs"] = others
        if world == 1 and not args.no_cpu_baseline and args.config == 2:
            result["cpu_baseline"] = cpu_baseline(wl)
        print(json.dumps(result), flush=True)
    if args.dump_gather and shared is not None and rank == 0:
        last = args.warmup + args.blocks * args.steps - 1
        np.savez(args.dump_gather, set_seed=np.int64(-1 if wl.set_seeds[last % len(wl.sets)] is None else wl.set_seeds[last % len(wl.sets)]),
                 global_batch=np.int64(b_global), world=np.int64(world),
                 **{name: shared.view(last, name).numpy().copy() for name, _, _ in wl.sections()})
    if dist is not None:
        dist.barrier()
    if shared is not None:
        shared.close()
    if dist is not None:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
