#!/bin/bash
# rocprofv3 kernel-trace + FETCH/WRITE/TCC passes for every kernel family at the shard sizes of BASELINE configs 2-5.
# usage (on the GPU box): tools/pmc_all.sh <prefix>   -> gpurun_out/<prefix>_<case>.txt summaries
set -e
prefix=$1
for c in roi_bbox_cfg2 roi_bbox_cfg2_clustered decoder_cfg2 crop_cfg2 roi_bbox_cfg3_shard8 crop_cfg3_shard8 roi_delta_cfg3_shard8 \
         iou_map_cfg4 iou_map_cfg4_shard targets_cfg4 targets_cfg4_shard roi_bbox_cfg5 crop_cfg5; do
  extra=""
  case $c in crop_*) extra="--sets 4";; esac
  tools/pmc_case.sh ${prefix}_$c $c $extra > gpurun_out/${prefix}_$c.txt 2>&1 || { echo "FAILED $c"; tail -5 gpurun_out/${prefix}_$c.txt; exit 1; }
  echo "== $c"; head -6 gpurun_out/${prefix}_$c.txt
done
