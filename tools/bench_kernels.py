"""Per-kernel timings against the HBM roofline (algorithmic bytes from SURVEY.md 8d / DESIGN.md).

    python tools/bench_kernels.py [--only crop_cfg2,iou_map_cfg4,...] [--iters 50]

Prints one JSON line per case: average launch time (HIP events on the launch stream), algorithmic GB/s
and the fraction of the 8 TB/s HBM peak.  bench.py remains the contract benchmark; this is the per-kernel
evidence (also the program profiled by the rocprofv3 --pmc passes, see profiles/).
"""

from __future__ import annotations

import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import synth
import tf_faster_rcnn_amd as pkg

PEAK = 8000.0


def timeit(fn, iters, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters  # ms


def timeit_raw(call, iters, warmup=5):
    """`call()` enqueues one launch through the C ABI with every argument prepared beforehand (no allocation, no Python
    layer): the per-call host cost (~5 us) stays below the kernels' run time, so back-to-back launches measure the GPU.
    Returns (ms per launch by HIP events, host ms per enqueue)."""
    import time
    for _ in range(warmup):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    t0 = time.perf_counter()
    for _ in range(iters):
        call()
    host = (time.perf_counter() - t0) / iters * 1e3
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters, host


SETS = 1  # --sets: input sets rotated through the iterations (1 = the same tensors every time, i.e. cache-warm inputs)


def rotating(make, n=None):
    """`make(k)` builds input set k; returns a function that hands out the sets in turn."""
    sets = [make(k) for k in range(n or SETS)]
    state = {"i": 0}

    def nxt():
        state["i"] += 1
        return sets[state["i"] % len(sets)]
    return nxt


def rois_for(B, G, K, R, seed=0, mode="inference", delta_scale=1.0):
    hp = pkg.train_utils.get_hyper_params("mobilenet_v2", feature_map_shape=G, pre_nms_topn=K)
    hp["train_nms_topn" if mode == "training" else "test_nms_topn"] = R
    anchors = pkg.bbox_utils.generate_anchors(hp)
    d, s = synth.rpn_outputs(seed, B, G)
    layer = pkg.faster_rcnn.RoIBBox(anchors, mode, hp)
    return hp, anchors, torch.as_tensor(d * np.float32(delta_scale)).cuda(), torch.as_tensor(s).cuda(), layer


def case_crop(name, B, G, C, R, K, iters):
    hp, anchors, d, s, layer = rois_for(B, G, K, R, mode="training")
    rois = layer([d, s])
    fm0 = torch.as_tensor(synth.feature_map(0, B, G, C)).cuda()
    nxt = rotating(lambda k: fm0 if k == 0 else torch.roll(fm0, k, 0).contiguous())
    pool = pkg.faster_rcnn.RoIPooling(hp)
    ms = timeit(lambda: pool([nxt(), rois]), iters)
    nbytes = B * (R * 49 * C * 4 + G * G * C * 4 + R * 20)
    return dict(case=name, kernel="crop_resize_*", ms=ms, bytes=nbytes, B=B, R=R, C=C, G=G, input_sets=SETS)


def case_iou_map(name, B, G, M, iters):
    hp = pkg.train_utils.get_hyper_params("mobilenet_v2", feature_map_shape=G)
    anchors = pkg.bbox_utils.generate_anchors(hp)
    gt, _ = synth.gt_boxes(0, B, M)
    gt = torch.as_tensor(gt).cuda()
    ms = timeit(lambda: pkg.bbox_utils.generate_iou_map(anchors, gt), iters)
    N = anchors.shape[0]
    nbytes = B * (N * M * 4 + M * 16) + N * 16
    return dict(case=name, kernel="iou_map_kernel<true>", ms=ms, bytes=nbytes, B=B, N=N, M=M)


def case_targets(name, B, G, M, iters):
    hp = pkg.train_utils.get_hyper_params("mobilenet_v2", feature_map_shape=G)
    anchors = pkg.bbox_utils.generate_anchors(hp)
    N = anchors.shape[0]
    gt, lab = synth.gt_boxes(0, B, M)
    rp, rn = synth.sampling_ints(0, B, N)
    gt, lab, rp, rn = (torch.as_tensor(a).cuda() for a in (gt, lab, rp, rn))
    ms = timeit(lambda: pkg.train_utils.calculate_rpn_actual_outputs(anchors, gt, lab, hp, random_masks=(rp, rn)), iters)
    nbytes = B * (M * 16 + M * 4 + N * 8 + N * 16 + N * 4) + N * 16  # fused: no map written (ALU bound, reported only)
    return dict(case=name, kernel="iou_reduce_kernel + rpn_targets_kernel (fused path, no map)", ms=ms, bytes=nbytes, B=B, N=N, M=M,
                images_per_s=B / (ms * 1e-3))


def case_roi_bbox(name, B, G, K, R, iters, mode="inference", delta_scale=1.0):
    """delta_scale < 1: clustered proposals (deltas shrunk, so the decoded boxes stay close to their anchors and overlap
    heavily): NMS consumes far more candidates before R boxes are kept -- the lazy kernel's slow regime."""
    hp, anchors, d, s, layer = rois_for(B, G, K, R, mode=mode, delta_scale=delta_scale)
    ms_layer = timeit(lambda: layer([d, s]), iters)
    N = anchors.shape[0]
    _lib = pkg._lib
    lib = _lib.lib()
    out = torch.empty((B, R, 4), device=d.device)
    ws = torch.empty(int(lib.frcnn_roi_bbox_workspace_bytes(B, N, K, R)) or 16, dtype=torch.uint8, device=d.device)
    args = (_lib.ptr(d), _lib.ptr(s), _lib.ptr(layer.anchors), _lib.float4(hp["variances"]), B, N, K, R,
            float(hp["nms_iou_threshold"]), _lib.ptr(out), None, None, _lib.ptr(ws), ws.numel(), _lib.stream_ptr(d.device))
    ms, host = timeit_raw(lambda: lib.frcnn_roi_bbox(*args), iters)
    assert torch.equal(out, layer([d, s]))
    return dict(case=name, kernel="proposal_kernel", ms=ms, ms_python_layer=ms_layer, host_ms_per_call=host,
                bytes=B * (N * 4 + K * 16 + R * 16), B=B, N=N, K=K, R=R, delta_scale=delta_scale, images_per_s=B / (ms * 1e-3))


def case_roi_delta(name, B, R, M, L, iters):
    hp = pkg.train_utils.get_hyper_params("vgg16")
    hp["total_labels"] = L
    rng = np.random.default_rng(3)
    rois = np.sort(rng.random((B, R, 2, 2)).astype(np.float32), axis=2).reshape(B, R, 4)
    gt, lab = synth.gt_boxes(0, B, M)
    rois, gt, lab = (torch.as_tensor(a).cuda() for a in (rois, gt, lab))
    layer = pkg.faster_rcnn.RoIDelta(hp)
    ms = timeit(lambda: layer([rois, gt, lab], seed=1), iters)
    return dict(case=name, kernel="iou_reduce_kernel + roi_delta_kernel", ms=ms, bytes=B * (R * 16 + M * 20 + R * L * 20), B=B, R=R,
                M=M, L=L, images_per_s=B / (ms * 1e-3))


def case_decoder(name, B, R, L, iters):
    hp = pkg.train_utils.get_hyper_params("mobilenet_v2")
    rng = np.random.default_rng(0)
    rois = np.sort(rng.random((B, R, 2, 2)).astype(np.float32), axis=2).reshape(B, R, 4)
    hd, hpb = synth.head_outputs(0, B, R, L)
    rois, hd, hpb = (torch.as_tensor(a).cuda() for a in (rois, hd, hpb))
    layer = pkg.faster_rcnn.Decoder(hp["variances"], L)
    ms_layer = timeit(lambda: layer([rois, hd, hpb]), iters)
    _lib = pkg._lib
    lib = _lib.lib()
    T = layer.max_total_size
    ob, ol, osc = (torch.empty(sh, device=rois.device) for sh in ((B, T, 4), (B, T), (B, T)))
    ws = torch.empty(int(lib.frcnn_decoder_workspace_bytes(B, R, L, T)) or 16, dtype=torch.uint8, device=rois.device)
    args = (_lib.ptr(rois), _lib.ptr(hd), _lib.ptr(hpb), _lib.float4(layer.variances), B, R, L, T, float(layer.score_threshold), 0.5,
            _lib.ptr(ob), _lib.ptr(ol), _lib.ptr(osc), None, _lib.ptr(ws), ws.numel(), _lib.stream_ptr(rois.device))
    ms, host = timeit_raw(lambda: lib.frcnn_decoder(*args), iters)
    want = layer([rois, hd, hpb])
    assert torch.equal(ob, want[0]) and torch.equal(osc, want[2])
    return dict(case=name, kernel="decoder (sort + nms + merge)", ms=ms, ms_python_layer=ms_layer, host_ms_per_call=host, bytes=B * (R * 16 + R * L * 20 + 200 * 24), B=B, R=R, L=L,
                images_per_s=B / (ms * 1e-3))


CASES = {
    "crop_cfg1": lambda it: case_crop("crop_cfg1", 1, 32, 576, 300, 6000, it),  # one image: the small-launch four-slice schedule
    "crop_cfg2": lambda it: case_crop("crop_cfg2", 16, 32, 576, 300, 6000, it),
    "crop_cfg3": lambda it: case_crop("crop_cfg3", 16, 31, 512, 2000, 6000, it),  # one GPU's shard of B=32 over 2 GPUs
    "crop_cfg5": lambda it: case_crop("crop_cfg5", 16, 63, 576, 300, 12000, it),  # one GPU's shard of B=128 over 8
    "iou_map_cfg4": lambda it: case_iou_map("iou_map_cfg4", 64, 32, 100, it),
    "iou_map_cfg4_shard": lambda it: case_iou_map("iou_map_cfg4_shard", 8, 32, 100, it),
    "targets_cfg4": lambda it: case_targets("targets_cfg4", 64, 32, 100, it),
    "targets_cfg4_shard": lambda it: case_targets("targets_cfg4_shard", 8, 32, 100, it),
    "roi_bbox_cfg2": lambda it: case_roi_bbox("roi_bbox_cfg2", 16, 32, 6000, 300, it),
    "roi_bbox_cfg2_clustered": lambda it: case_roi_bbox("roi_bbox_cfg2_clustered", 16, 32, 6000, 300, it, delta_scale=0.05),
    "roi_bbox_cfg3_shard8": lambda it: case_roi_bbox("roi_bbox_cfg3_shard8", 8, 31, 6000, 2000, it, mode="training"),
    "roi_delta_cfg3_shard8": lambda it: case_roi_delta("roi_delta_cfg3_shard8", 8, 2000, 32, 21, it),
    "crop_cfg3_shard8": lambda it: case_crop("crop_cfg3_shard8", 8, 31, 512, 2000, 6000, it),
    "roi_bbox_cfg3": lambda it: case_roi_bbox("roi_bbox_cfg3", 16, 31, 6000, 2000, it, mode="training"),
    "roi_bbox_cfg5": lambda it: case_roi_bbox("roi_bbox_cfg5", 16, 63, 12000, 300, it),
    "decoder_cfg2": lambda it: case_decoder("decoder_cfg2", 16, 300, 21, it),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--sets", type=int, default=1, help="input sets rotated through the iterations of the crop cases")
    args = ap.parse_args()
    global SETS
    SETS = max(1, args.sets)
    names = [n for n in args.only.split(",") if n] or list(CASES)
    for n in names:
        r = CASES[n](args.iters)
        gbs = r["bytes"] / (r["ms"] * 1e-3) / 1e9
        r.update(ms=round(r["ms"], 5), GBps=round(gbs, 1), frac_of_8TBps=round(gbs / PEAK, 4))
        if "images_per_s" in r:
            r["images_per_s"] = round(r["images_per_s"], 1)
        print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
