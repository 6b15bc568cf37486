"""Experiment: RoIPooling at small batches (config 2 shapes, B images of 300 RoIs) under the crop schedule knobs.
    [FRCNN_CROP_SPLIT=4 FRCNN_CROP_PREFETCH=0] python tools/exp_small_crop.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import synth
import tf_faster_rcnn_amd as pkg

hp = pkg.train_utils.get_hyper_params("mobilenet_v2")
pool = pkg.faster_rcnn.RoIPooling(hp)
rng = np.random.default_rng(0)
out = []
for B in (1, 2, 3, 4, 6, 8):
    fm = torch.as_tensor(synth.feature_map(0, B, 32, 576)).cuda()
    ctr = rng.random((B, 300, 2))
    hw = rng.uniform(0.05, 0.3, (B, 300, 2))
    rois = torch.as_tensor(np.clip(np.concatenate([ctr - hw, ctr + hw], -1), 0, 1).astype(np.float32)).cuda()
    for _ in range(20):
        pool([fm, rois])
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(200):
        pool([fm, rois])
    b.record()
    torch.cuda.synchronize()
    out.append(f"B={B}: {a.elapsed_time(b) / 200 * 1e3:.1f} us")
print(os.environ.get("FRCNN_CROP_SPLIT", "-"), os.environ.get("FRCNN_CROP_PREFETCH", "-"), " ".join(out))
