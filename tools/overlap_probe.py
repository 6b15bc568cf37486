#!/usr/bin/env python3
"""overlap_probe.py -- how much of a step is host enqueue, and what do in-flight batches on separate
HIP streams (eager or as captured graphs) buy?  Prints one JSON line per variant.

    python tools/overlap_probe.py [--steps 200]
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402
import synth  # noqa: E402,F401


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    args = ap.parse_args()
    import tf_faster_rcnn_amd as pkg

    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    cfg = dict(bench.CFG)
    hyper = pkg.train_utils.get_hyper_params(cfg["backbone"])
    hyper["total_labels"] = cfg["L"]
    deltas, scores, fm, hd, hp = (torch.as_tensor(a).to(device) for a in bench.make_inputs(cfg, 0))
    anchors = pkg.bbox_utils.generate_anchors(hyper)
    roi_bbox = pkg.faster_rcnn.RoIBBox(anchors, "inference", hyper)
    roi_pool = pkg.faster_rcnn.RoIPooling(hyper)
    decoder = pkg.faster_rcnn.Decoder(hyper["variances"], cfg["L"])

    def step():
        rois = roi_bbox([deltas, scores])
        pooled = roi_pool([fm, rois])
        dets = decoder([rois, hd, hp])
        return pooled, dets

    def report(name, elapsed, host):
        print(json.dumps({"variant": name, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
                          "host_ms_per_step": round(host / args.steps * 1e3, 4),
                          "images_per_s": round(cfg["B"] * args.steps / elapsed, 1)}), flush=True)

    for nstreams in (1, 2, 3, 4):
        streams = [torch.cuda.Stream(device) for _ in range(nstreams)]
        for s in streams:
            with torch.cuda.stream(s):
                for _ in range(5):
                    step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            with torch.cuda.stream(streams[i % nstreams]):
                step()
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        report(f"eager_streams{nstreams}", time.perf_counter() - t0, host)

    # stage pipeline: one stream per stage, events carry the roi -> crop -> decoder dependency
    for dec_after in ("crop", "rois"):
        sa, sb, sc = (torch.cuda.Stream(device) for _ in range(3))

        def staged():
            with torch.cuda.stream(sa):
                rois = roi_bbox([deltas, scores])
                ea = torch.cuda.Event()
                ea.record(sa)
            rois.record_stream(sb)
            rois.record_stream(sc)
            with torch.cuda.stream(sb):
                sb.wait_event(ea)
                pooled = roi_pool([fm, rois])
                eb = torch.cuda.Event()
                eb.record(sb)
            with torch.cuda.stream(sc):
                sc.wait_event(eb if dec_after == "crop" else ea)
                dets = decoder([rois, hd, hp])
            return pooled, dets

        for _ in range(10):
            staged()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            staged()
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        report(f"stage_pipeline_decoder_after_{dec_after}", time.perf_counter() - t0, host)

    for nstreams in (1, 2):
        streams = [torch.cuda.Stream(device) for _ in range(nstreams)]
        graphs, keep = [], []
        for s in streams:
            with torch.cuda.stream(s):
                for _ in range(3):
                    step()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                keep.append(step())
            graphs.append(g)
        torch.cuda.synchronize()
        for g, s in zip(graphs, streams):
            with torch.cuda.stream(s):
                g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            with torch.cuda.stream(streams[i % nstreams]):
                graphs[i % nstreams].replay()
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        report(f"graph_streams{nstreams}", time.perf_counter() - t0, host)


if __name__ == "__main__":
    main()
