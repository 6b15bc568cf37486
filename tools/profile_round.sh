#!/bin/bash
# The round's evidence in one GPU call (gpurun_out/<tag>_*; copy what is to be judged into profiles/):
#   1. tools/profile_bench.sh   : the driver's bench line, a kernel trace per schedule, FETCH_SIZE / WRITE_SIZE passes of the crop
#   2. tools/profile_shards.sh  : configs 3 / 4 / 5 at their per-GPU shard sizes through the shared host gather (line, kernel
#                                 stats, timeline of the last block)
#   3. tools/pmc_case.sh        : kernel trace + PMC passes of single kernels (config 5's staged-row crop, config 1's small-launch crop,
#                                 the materialised IoU map, the train-mode proposals)
#   4. two-rank runs on one GPU : strong-scaling configs with a timing record (both ranks share the GPU: a record of the path, not
#                                 a scaling figure)
# usage: tools/profile_round.sh <tag>
set -e
tag=$1
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out
cd $root
tools/profile_bench.sh ${tag}_a
tools/profile_shards.sh ${tag}_b
for c in crop_cfg5 crop_cfg1 crop_cfg2 crop_cfg3_shard8 iou_map_cfg4 iou_map_cfg4_shard roi_bbox_cfg3_shard8 targets_cfg4_shard; do
  extra=""
  case $c in crop_*) extra="--sets 5";; esac
  tools/pmc_case.sh ${tag}_c_pmc_$c $c $extra > $out/${tag}_c_pmc_$c.txt 2>&1 || { echo "FAILED $c"; tail -5 $out/${tag}_c_pmc_$c.txt; exit 1; }
  rm -rf $out/${tag}_c_pmc_$c
done
for cfg in 3 4 5; do
  b=16; [ $cfg = 5 ] && b=32
  timeout -k 10 300 python3 -u bench.py --gpus 2 --device 0 --dist-backend gloo --config $cfg --batch $b --steps 20 --warmup 5 --no-cpu-baseline \
    > $out/${tag}_d_two_ranks_one_gpu_cfg$cfg.json.log 2> $out/${tag}_d_two_ranks_cfg$cfg.err
done
timeout -k 10 300 python3 -u bench.py --force-dist --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $out/${tag}_d_one_rank_rccl_cfg2.json.log 2> $out/${tag}_d_rccl.err
echo profile_round done
