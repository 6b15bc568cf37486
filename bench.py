#!/usr/bin/env python3
"""bench.py -- images/s through the proposal/RoI hot path (BASELINE.json metric).

One step = RoIBBox (decode + top-6000 + NMS -> 300 RoIs) -> RoIPooling (7x7 crop_and_resize)
-> Decoder (per-class NMS) over one batch of 16 synthetic images (BASELINE config 2:
mobilenet_v2, 500x500 -> 32x32x576 feature map, 9216 anchors, 21 labels), inputs resident in HBM.
The conv backbone and the dense heads are out of scope: their outputs are synthetic tensors.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Multi-GPU: pure data parallel over images, one process per GPU, every rank runs its own batch of 16
(weak scaling), no collective on the data path; torch.distributed is used only for the barrier and
the max-over-ranks time.  Rank 0 prints ONE JSON line.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import synth  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md)

# pipeline mode: every n-th crop launch of the timed region is bracketed with HIP events on its stream (each pair adds two
# packets to the critical gap between consecutive crop launches: bracketing all of them costs 2-4 % of the throughput)
TIME_EVERY = int(os.environ.get("FRCNN_BENCH_TIME_EVERY", "4"))

# BASELINE config 2
CFG = dict(backbone="mobilenet_v2", B=16, G=32, C=576, K=6000, R=300, L=21, pool=7)


def crop_bytes_per_image(cfg) -> int:
    """Algorithmic bytes of crop_and_resize per image: write R*49*C*4, read H*W*C*4 + R*20."""
    return cfg["R"] * cfg["pool"] * cfg["pool"] * cfg["C"] * 4 + cfg["G"] * cfg["G"] * cfg["C"] * 4 + cfg["R"] * 20


def pmc_traffic_bytes():
    """HBM bytes per crop launch from the committed rocprofv3 --pmc passes (profiles/r01_pmc_*): separate
    FETCH_SIZE and WRITE_SIZE runs, KB units, FETCH_SIZE doubled (gfx950 counts 128-B read requests as 64 B,
    MI355X_MICROARCH.md section HBM).  None if the profiles are not there."""
    import csv
    import glob

    def avg(pattern, counter):
        vals = []
        # the most recent pass only: per-round files are named r<round>_<letter>_pmc_*, the first pass r01_pmc_*
        paths = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)),
                       key=lambda q: tuple((os.path.basename(q).split("_pmc_")[0] + "_").split("_")[:2]))  # (round, letter or "")
        for path in paths[-1:]:
            with open(path) as fh:
                for row in csv.DictReader(fh):
                    if row["Counter_Name"] == counter and "crop_resize_kernel" in row["Kernel_Name"]:
                        vals.append(float(row["Counter_Value"]))
        return sum(vals) / len(vals) if vals else None

    fetch = avg("r*_pmc_fetch_size_*.csv", "FETCH_SIZE")
    write = avg("r*_pmc_write_size_*.csv", "WRITE_SIZE")
    if fetch is None or write is None:
        return None
    return int((2.0 * fetch + write) * 1024)


def make_inputs(cfg, seed):
    deltas, scores = synth.rpn_outputs(seed, cfg["B"], cfg["G"])
    fm = synth.feature_map(seed, cfg["B"], cfg["G"], cfg["C"])
    hd, hp = synth.head_outputs(seed, cfg["B"], cfg["R"], cfg["L"])
    return deltas, scores, fm, hd, hp


def cpu_baseline(cfg, inputs, hyper, min_seconds=10.0):
    """The CPU oracle (reference-semantics restatement) on the host cores: reported baseline only."""
    import oracle

    deltas, scores, fm, hd, hp = inputs
    anchors = oracle.generate_anchors(hyper)

    def one_batch():
        rois = oracle.roi_bbox(deltas, scores, anchors, hyper["variances"], cfg["K"], cfg["R"], hyper["nms_iou_threshold"])
        oracle.roi_pooling(fm, rois, hyper["pooling_size"])
        oracle.decoder(rois, hd, hp, hyper["variances"], cfg["L"])

    out = {}
    for label, threads in (("1", 1), ("all", max(1, min(int(os.environ.get("OMP_NUM_THREADS", "16")), os.cpu_count() or 1)))):
        oracle.set_threads(threads)
        one_batch()  # warm
        n, t0 = 0, time.perf_counter()
        while True:
            one_batch()
            n += 1
            dt = time.perf_counter() - t0
            if dt >= min_seconds / 2 or n >= 200:
                break
        out[label] = (threads, n, cfg["B"] * n / dt)
    threads, n, ips = out["all"]
    return {
        "value": round(ips, 2), "unit": "images/s", "cores": threads, "kind": "port",
        "sample": f"{n} batches of {cfg['B']} images (config 2), oracle/ C restatement, OpenMP over images/boxes",
        "value_1thread": round(out["1"][2], 2),
    }


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", choices=("pipeline", "sequential"), default="pipeline",
                    help="pipeline: stages on three HIP streams, --depth batches in flight (tf_faster_rcnn_amd.pipeline); "
                         "sequential: the three layers one after the other on one stream")
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--crop-group", type=int, default=2,
                    help="pipeline mode: consecutive batches that share one launch per stage (the *_multi entry points)")
    ap.add_argument("--no-host-gather", action="store_true",
                    help="leave the small per-image outputs (RoIs, detections) on the GPU instead of copying them to pinned "
                         "host memory every step (the sharded deployment's final host gather, SURVEY.md 8e)")
    ap.add_argument("--dist-backend", default="nccl", help="process-group backend for the barrier / max-over-ranks (nccl = RCCL)")
    ap.add_argument("--device", type=int, default=None,
                    help="GPU index for this rank (default LOCAL_RANK); lets a 1-GPU box rehearse the N-rank launch path")
    ap.add_argument("--no-other-schedule", action="store_true",
                    help="do not time the other schedule afterwards (profiling runs: one schedule per kernel trace)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    dev_index = local_rank if args.device is None else args.device
    if world > 1:
        import torch.distributed as dist  # noqa: F811
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(dev_index)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(args.dist_backend)
    else:
        torch.cuda.set_device(0 if args.device is None else args.device)
    device = torch.device("cuda", torch.cuda.current_device())

    import tf_faster_rcnn_amd as pkg

    cfg = dict(CFG)
    hyper = pkg.train_utils.get_hyper_params(cfg["backbone"])
    hyper["total_labels"] = cfg["L"]
    inputs = make_inputs(cfg, seed=rank)
    deltas, scores, fm, hd, hp = (torch.as_tensor(a).to(device) for a in inputs)
    anchors = pkg.bbox_utils.generate_anchors(hyper)
    roi_bbox = pkg.faster_rcnn.RoIBBox(anchors, "inference", hyper, name="roi_bboxes")
    roi_pool = pkg.faster_rcnn.RoIPooling(hyper, name="roi_pooling")
    decoder = pkg.faster_rcnn.Decoder(hyper["variances"], cfg["L"], name="faster_rcnn_decoder")

    # timing events without the system-scope fence (hipEventDisableSystemFence): a default event makes the kernel it follows
    # write back and invalidate L2, which is exactly what the measurement must not add
    ev0 = [pkg._lib.HipEvent(pkg._lib.hipEventDisableSystemFence) for _ in range(args.steps)]
    ev1 = [pkg._lib.HipEvent(pkg._lib.hipEventDisableSystemFence) for _ in range(args.steps)]

    cur = torch.cuda.current_stream(device).cuda_stream

    gather = not args.no_host_gather
    T = int(decoder.max_total_size)
    host_out = [torch.empty(shape, dtype=torch.float32, pin_memory=True)
                for shape in ((cfg["B"], cfg["R"], 4), (cfg["B"], T, 4), (cfg["B"], T), (cfg["B"], T))] if gather else []

    def step_sequential(i=None):
        rois = roi_bbox([deltas, scores])
        if i is not None:
            ev0[i].record(cur)  # on the stream the kernel is launched on (torch's current stream)
        pooled = roi_pool([fm, rois])
        if i is not None:
            ev1[i].record(cur)
        dets = decoder([rois, hd, hp])
        for dst, src in zip(host_out, (rois,) + tuple(dets)):  # the host gather of the small outputs (copy kernels)
            pkg._lib.check(pkg._lib.lib().frcnn_gather_copy(src.data_ptr(), dst.data_ptr(), src.numel() * 4, cur))
        return pooled, dets

    group = args.crop_group
    pipe = pkg.pipeline.RoIPathPipeline(roi_bbox, roi_pool, decoder, depth=args.depth, crop_group=group, host_outputs=gather)

    def step_pipeline(i=None):
        # every step is a full pass (RoIBBox -> RoIPooling -> Decoder) over its batch; the stages of up to
        # --depth consecutive steps overlap on three streams.  The events bracket the crop kernel on ITS stream.
        # (with --crop-group g the first batch of every g carries the events of the group's one launch)
        pipe.submit(deltas, scores, fm, (hd, hp), timing=None if (i is None or i % (TIME_EVERY * group)) else (ev0[i], ev1[i]))

    def barrier():
        pipe.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    host_s = [0.0]

    def timed(step_fn, record):
        for _ in range(args.warmup):
            step_fn()
        barrier()
        wait0 = pipe.wait_seconds
        t_start = time.perf_counter()
        for i in range(args.steps):
            step_fn(i if record else None)
        host_s[0] = time.perf_counter() - t_start - (pipe.wait_seconds - wait0)  # host time to enqueue K steps (not blocked)
        barrier()
        return time.perf_counter() - t_start

    step = step_pipeline if args.mode == "pipeline" else step_sequential
    elapsed = timed(step, True)
    host_ms = host_s[0] / max(args.steps, 1) * 1e3
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    crop_samples = [a.elapsed_time(b) for i, (a, b) in enumerate(zip(ev0, ev1))
                    if args.mode == "sequential" or i % (TIME_EVERY * group) == 0]
    crop_ms = float(np.mean(crop_samples)) if crop_samples else float("nan")
    crop_pct = [round(float(v), 5) for v in np.percentile(crop_samples, [10, 50, 90])] if crop_samples else None
    launch_batches = group if args.mode == "pipeline" else 1  # batches one crop launch covers
    other = None
    if world == 1 and not args.no_other_schedule:  # the other schedule, same K steps, reported beside the headline (not part of `value`)
        other_elapsed = timed(step_sequential if args.mode == "pipeline" else step_pipeline, False)
        other = {"mode": "sequential" if args.mode == "pipeline" else "pipeline",
                 "value": round(cfg["B"] * args.steps / other_elapsed, 1),
                 "ms_per_step": round(other_elapsed / args.steps * 1e3, 4)}

    if rank == 0:
        images = cfg["B"] * world * args.steps
        value = images / elapsed
        crop_bytes = crop_bytes_per_image(cfg) * cfg["B"] * launch_batches
        achieved = crop_bytes / (crop_ms * 1e-3) / 1e9
        result = {
            "metric": "images/sec through RPN decode+NMS+RoIPool+class NMS (bs16, 500\u00b2); HBM GB/s",
            "value": round(value, 1),
            "unit": "images/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "host_enqueue_ms_per_step": round(host_ms, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "BASELINE config 2: mobilenet_v2 RoI path, batch 16 per GPU, 500x500, 9216 anchors/img, "
                                   "pre_nms_topn 6000, post_nms 300, 7x7 crop_and_resize of 32x32x576, 21-label Decoder NMS",
                       "global_batch": cfg["B"] * world, "parallelism": f"dp{world} over images, no collectives",
                       "schedule": (f"stage pipeline on 3 HIP streams, {args.depth} batches in flight, "
                                    f"{group} consecutive batch(es) per launch of each stage" if args.mode == "pipeline"
                                    else "sequential on one stream"),
                       "host_gather": ("RoIs and detections of every batch copied to pinned host memory inside the timed "
                                       "region; pooled features stay on the GPU") if gather else "none"},
            "roofline": {"kernel": "crop_resize_kernel<4>", "bound": "hbm", "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": (lambda t: None if t is None else t * launch_batches)(pmc_traffic_bytes()), "bytes_per_launch": crop_bytes, "avg_launch_ms": round(crop_ms, 5),
                         "launch_ms_p10_p50_p90": crop_pct, "launches_timed": len(crop_samples)},
        }
        if other is not None:
            result["other_schedule"] = other
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(cfg, inputs, hyper)
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
