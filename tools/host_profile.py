"""Where the host time of one pipeline step goes (cProfile over RoIPathPipeline.submit at BASELINE config 2)."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import synth
import tf_faster_rcnn_amd as pkg

hp = pkg.train_utils.get_hyper_params("mobilenet_v2")
hp["total_labels"] = 21
anchors = pkg.bbox_utils.generate_anchors(hp)
layers = (pkg.faster_rcnn.RoIBBox(anchors, "inference", hp), pkg.faster_rcnn.RoIPooling(hp), pkg.faster_rcnn.Decoder(hp["variances"], 21))
pipe = pkg.pipeline.RoIPathPipeline(*layers, depth=8, crop_group=2, host_outputs=True)
d, s = synth.rpn_outputs(0, 16, 32)
fm = synth.feature_map(0, 16, 32, 576)
hd, hpb = synth.head_outputs(0, 16, 300, 21)
d, s, fm, hd, hpb = (torch.as_tensor(a).cuda() for a in (d, s, fm, hd, hpb))
for _ in range(50):
    pipe.submit(d, s, fm, (hd, hpb))
pipe.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
pr = cProfile.Profile()
w0 = pipe.wait_seconds
pr.enable()
for _ in range(n):
    pipe.submit(d, s, fm, (hd, hpb))
pr.disable()
pipe.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
print("back-pressure wait per step (us):", (pipe.wait_seconds - w0) / n * 1e6)
