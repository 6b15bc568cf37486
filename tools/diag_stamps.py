"""Diagnostic only: phase shares of the fused proposal kernel from in-kernel wall_clock64 stamps.

Needs a library built with -DFRCNN_STAMPS ("Diagnostics"); run as
FRCNN_HIP_LIB=gpurun_out/diag/libfrcnn_hip_stamps.so python tools/diag_stamps.py
The stamped build's run time is not a benchmark number: read the shares only.
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import synth
import tf_faster_rcnn_amd as pkg

lib = pkg._lib.lib()
lib.frcnn_debug_stamps.restype = ctypes.c_int
lib.frcnn_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
if len(sys.argv) >= 2 and sys.argv[1] == "decoder":  # python tools/diag_stamps.py decoder  -> the Decoder kernel's stamps (cfg2)
    hp = pkg.train_utils.get_hyper_params("mobilenet_v2")
    rois_np = np.sort(np.random.default_rng(0).random((16, 300, 2, 2)).astype(np.float32), axis=2).reshape(16, 300, 4)
    dl, pr = synth.head_outputs(0, 16, 300, 21)
    dec = pkg.faster_rcnn.Decoder(hp["variances"], 21)
    args = [torch.as_tensor(rois_np).cuda(), torch.as_tensor(dl).cuda(), torch.as_tensor(pr).cuda()]
    buf = np.zeros(256, dtype=np.int64)
    for it in range(3):
        dec(args)
        torch.cuda.synchronize()
        n = lib.frcnn_debug_stamps(buf.ctypes.data, 1)
    tags, t = buf[0:2 * n:2] >> 48, buf[1:2 * n:2]
    for i in range(n):
        print(f"tag {tags[i]:3d}  t={(t[i]-t[0])*10/1000:8.2f} us  d={(t[i]-t[i-1])*10/1000 if i else 0:8.2f} us")
    sys.exit(0)
if len(sys.argv) >= 2 and sys.argv[1] == "crop5":  # python tools/diag_stamps.py crop5  -> one mid-image workgroup of the staged-row crop (cfg5 shard)
    B_, G_, C_, R_ = 16, 63, 576, 300
    hp = pkg.train_utils.get_hyper_params("mobilenet_v2", feature_map_shape=G_, pre_nms_topn=12000, img_size=1000)
    anchors = pkg.bbox_utils.generate_anchors(hp)
    d, s = synth.rpn_outputs(0, B_, G_)
    rois = pkg.faster_rcnn.RoIBBox(anchors, "inference", hp)([torch.as_tensor(d).cuda(), torch.as_tensor(s).cuda()])
    fms = [torch.as_tensor(synth.feature_map(k, B_, G_, C_)).cuda() for k in range(3)]
    pool = pkg.faster_rcnn.RoIPooling(hp)
    buf = np.zeros(256, dtype=np.int64)
    for it in range(4):
        torch.cuda.synchronize()
        lib.frcnn_debug_stamps(buf.ctypes.data, 1)
        pool([fms[it % 3], rois])
        torch.cuda.synchronize()
        n = lib.frcnn_debug_stamps(buf.ctypes.data, 1)
    tags, t = buf[0:2 * n:2] >> 48, buf[1:2 * n:2]
    for i in range(n):
        print(f"tag {tags[i]:3d}  t={(t[i]-t[0])*10/1000:8.2f} us  d={(t[i]-t[i-1])*10/1000 if i else 0:8.2f} us")
    sys.exit(0)
if len(sys.argv) >= 2 and sys.argv[1] == "targets":  # python tools/diag_stamps.py targets [B]  -> rpn_targets_kernel's stamps (cfg4)
    B_ = int(sys.argv[2]) if len(sys.argv) >= 3 else 8
    hp = pkg.train_utils.get_hyper_params("mobilenet_v2")
    anchors = pkg.bbox_utils.generate_anchors(hp)
    gt, lab = synth.gt_boxes(0, B_, 100)
    gt, lab = torch.as_tensor(gt).cuda(), torch.as_tensor(lab).cuda()
    buf = np.zeros(256, dtype=np.int64)
    for it in range(3):
        pkg.train_utils.calculate_rpn_actual_outputs(anchors, gt, lab, hp, seed=it)
        torch.cuda.synchronize()
        n = lib.frcnn_debug_stamps(buf.ctypes.data, 1)
    tags, t = buf[0:2 * n:2] >> 48, buf[1:2 * n:2]
    for i in range(n):
        print(f"tag {tags[i]:3d}  t={(t[i]-t[0])*10/1000:8.2f} us  d={(t[i]-t[i-1])*10/1000 if i else 0:8.2f} us")
    sys.exit(0)
# optional: G K R on the command line (default BASELINE config 2: 32 6000 300; config 5: 63 12000 300)
G_, K_, R_ = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (32, 6000, 300)
hp = pkg.train_utils.get_hyper_params("mobilenet_v2", feature_map_shape=G_, pre_nms_topn=K_, test_nms_topn=R_)
anchors = pkg.bbox_utils.generate_anchors(hp)
d, s = synth.rpn_outputs(0, 16, G_)
d, s = torch.as_tensor(d).cuda(), torch.as_tensor(s).cuda()
layer = pkg.faster_rcnn.RoIBBox(anchors, "inference", hp)
buf = np.zeros(256, dtype=np.int64)
for it in range(3):
    layer([d, s])
    torch.cuda.synchronize()
    n = lib.frcnn_debug_stamps(buf.ctypes.data, 1)
    if it == 2:
        tags, t = buf[0:2 * n:2] >> 48, buf[1:2 * n:2]
        cyc = buf[0:2 * n:2] & 0xFFFFFFFFFFFF
        print("shader clock MHz over the kernel:", (cyc[n - 1] - cyc[0]) / ((t[n - 1] - t[0]) * 0.01))
        print("n stamps", n)
        for i in range(n):
            print(f"tag {tags[i]:3d}  t={(t[i]-t[0])*10/1000:8.2f} us  d={(t[i]-t[i-1])*10/1000 if i else 0:8.2f} us")


if len(sys.argv) >= 5 and sys.argv[4] == "cold":
    # the same kernel on inputs that are not in any cache: 40 input sets (> 256 MiB Infinity Cache with the filler) in turn
    sets = []
    for k in range(24):
        dk, sk = synth.rpn_outputs(100 + k, 16, G_)
        sets.append((torch.as_tensor(dk).cuda(), torch.as_tensor(sk).cuda()))
    filler = torch.empty(512 * 1024 * 1024 // 4, device="cuda")
    for k in range(24):
        filler.add_(1.0)  # sweeps 512 MiB through the caches
        torch.cuda.synchronize()
        lib.frcnn_debug_stamps(buf.ctypes.data, 1)
        layer(list(sets[k]))
        torch.cuda.synchronize()
    n = lib.frcnn_debug_stamps(buf.ctypes.data, 1)
    tags, t = buf[0:2 * n:2] >> 48, buf[1:2 * n:2]
    print("---- cold inputs ----")
    for i in range(n):
        print(f"tag {tags[i]:3d}  t={(t[i]-t[0])*10/1000:8.2f} us  d={(t[i]-t[i-1])*10/1000 if i else 0:8.2f} us")
if len(sys.argv) >= 4:
    sys.exit(0)

# ---- the same kernel while the crop kernel fills the chip on another stream (stage pipeline conditions) ----------
# Workgroup 0's phases run at their stand-alone speed (82.6 us against 80.0): what the pipeline loses is RESIDENCY, not
# speed -- a 1024-thread workgroup with ~100 VGPRs needs a CU empty of crop waves and only finds one in a crop launch's tail.
fm = torch.as_tensor(synth.feature_map(0, 16, 32, 576)).cuda()
pool = pkg.faster_rcnn.RoIPooling(hp)
rois = layer([d, s])
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()
lib.frcnn_debug_stamps(buf.ctypes.data, 1)
with torch.cuda.stream(sa):
    for _ in range(4):
        pool([fm, rois])
with torch.cuda.stream(sb):
    layer([d, s])
torch.cuda.synchronize()
n = lib.frcnn_debug_stamps(buf.ctypes.data, 1)
tags, t = buf[0:2 * n:2] >> 48, buf[1:2 * n:2]
print("---- beside 4 crop launches on another stream ----")
for i in range(n):
    print(f"tag {tags[i]:3d}  t={(t[i]-t[0])*10/1000:8.2f} us  d={(t[i]-t[i-1])*10/1000 if i else 0:8.2f} us")


# ---- Decoder kernel phases (tags 20..26: start, argmax/count, compaction, order, per-class NMS, prefix, merge+write) ----
hd, hpb = synth.head_outputs(0, 16, 300, 21)
dec = pkg.faster_rcnn.Decoder(hp["variances"], 21)
hd, hpb = torch.as_tensor(hd).cuda(), torch.as_tensor(hpb).cuda()
for it in range(3):
    torch.cuda.synchronize()
    lib.frcnn_debug_stamps(buf.ctypes.data, 1)
    dec([rois, hd, hpb])
    torch.cuda.synchronize()
n = lib.frcnn_debug_stamps(buf.ctypes.data, 1)
tags, t = buf[0:2 * n:2] >> 48, buf[1:2 * n:2]
print("---- decoder kernel ----")
for i in range(n):
    print(f"tag {tags[i]:3d}  t={(t[i]-t[0])*10/1000:8.2f} us  d={(t[i]-t[i-1])*10/1000 if i else 0:8.2f} us")


# ---- rpn_targets_kernel phases (tags 30..34: start, forced map, positive selection, negative selection, outputs) ----
hp4 = pkg.train_utils.get_hyper_params("mobilenet_v2")
gt, lab = synth.gt_boxes(0, 8, 100)
rp_, rn_ = synth.sampling_ints(0, 8, anchors.shape[0])
args4 = [torch.as_tensor(a).cuda() for a in (gt, lab, rp_, rn_)]
for it in range(3):
    torch.cuda.synchronize()
    lib.frcnn_debug_stamps(buf.ctypes.data, 1)
    pkg.train_utils.calculate_rpn_actual_outputs(anchors, args4[0], args4[1], hp4, random_masks=(args4[2], args4[3]))
    torch.cuda.synchronize()
n = lib.frcnn_debug_stamps(buf.ctypes.data, 1)
tags, t = buf[0:2 * n:2] >> 48, buf[1:2 * n:2]
print("---- rpn_targets kernel ----")
for i in range(n):
    print(f"tag {tags[i]:3d}  t={(t[i]-t[0])*10/1000:8.2f} us  d={(t[i]-t[i-1])*10/1000 if i else 0:8.2f} us")
