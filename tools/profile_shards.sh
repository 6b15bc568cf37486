#!/bin/bash
# The strong-scaling configs at their deployed per-GPU shard sizes on ONE GPU (world 1: same host path, SharedHostGather
# included): the bench line of each, and a rocprofv3 kernel trace of the same command with the timeline of its last block.
# usage: tools/profile_shards.sh <tag> [configs, default "3 4 5"]      -> gpurun_out/<tag>_*
set -e
tag=$1
cfgs=${2:-"3 4 5"}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in $cfgs; do
  b=8; [ $c = 5 ] && b=16
  args="--config $c --batch $b --steps 20 --warmup 5 --no-cpu-baseline"
  timeout -k 10 200 python3 -u $root/bench.py $args > $out/${tag}_cfg${c}_shard.json.log 2> $out/${tag}_cfg${c}_shard.err
  rm -rf $out/${tag}_trace_cfg$c
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_trace_cfg$c -o t -- \
    python3 $root/bench.py $args --no-other-schedule --blocks 2 > $out/${tag}_cfg${c}_shard_traced.json.log 2> $out/${tag}_trace_cfg$c.err
  cp $(find $out/${tag}_trace_cfg$c -name '*kernel_stats.csv' | head -1) $out/${tag}_cfg${c}_shard_kernel_stats.csv
  python3 $root/tools/block_timeline.py $(find $out/${tag}_trace_cfg$c -name '*kernel_trace.csv' | head -1) 120 > $out/${tag}_cfg${c}_shard_timeline.txt
  rm -rf $out/${tag}_trace_cfg$c
done
echo profile_shards done
