#!/bin/bash
# Diagnostic build with in-kernel wall-clock stamps (-DFRCNN_STAMPS) -> tools/diag/libfrcnn_hip_stamps.so
# Never shipped, never benchmarked: read the phase SHARES only (tools/diag_stamps.py).
set -e
cd "$(dirname "$0")/../tf-faster-rcnn_amd/csrc"
mkdir -p ../../tools/diag /tmp/frcnn_diag
for f in api codec crop_resize iou_map sort_select nms nms_bitmask proposal decoder rpn_targets host_gather; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -DFRCNN_STAMPS -fgpu-rdc -c $f.hip -o /tmp/frcnn_diag/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fgpu-rdc --hip-link -shared -fPIC -o ../../tools/diag/libfrcnn_hip_stamps.so /tmp/frcnn_diag/*.o
