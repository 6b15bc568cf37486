"""Experiment: config 5's crop with cold feature maps (4 sets of 146 MB) -- does a linear read of the batch's feature maps
right before the launch (Infinity Cache warm) bring back the warm-input time?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import synth
import tf_faster_rcnn_amd as pkg

B, G, C, R, K = 16, 63, 576, 300, 12000
hp = pkg.train_utils.get_hyper_params("mobilenet_v2", img_size=1000, feature_map_shape=G, pre_nms_topn=K)
anchors = pkg.bbox_utils.generate_anchors(hp)
d, s = synth.rpn_outputs(0, B, G)
rois = pkg.faster_rcnn.RoIBBox(anchors, "inference", hp)([torch.as_tensor(d).cuda(), torch.as_tensor(s).cuda()])
fms = [torch.as_tensor(synth.feature_map(0, B, G, C)).cuda()]
for k in range(1, 4):
    fms.append(torch.roll(fms[0], k, 0).contiguous())
pool = pkg.faster_rcnn.RoIPooling(hp)


def timeit(fn, iters=30):
    for i in range(5):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


print("crop, 1 set           : %.1f us" % timeit(lambda i: pool([fms[0], rois])))
print("crop, 4 sets (cold)   : %.1f us" % timeit(lambda i: pool([fms[i % 4], rois])))
print("linear read only      : %.1f us" % timeit(lambda i: fms[i % 4].sum()))
print("linear read + crop    : %.1f us" % timeit(lambda i: (fms[i % 4].sum(), pool([fms[i % 4], rois]))))
