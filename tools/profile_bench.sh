#!/bin/bash
# The round's evidence for bench.py, on the GPU box: the default bench line, one rocprofv3 kernel trace per schedule of the
# SAME workload (cold rotating inputs), and separate FETCH_SIZE / WRITE_SIZE passes over the crop kernel in that state.
# The bench runs with the driver's arguments (--steps 20 --warmup 5), so the files can be laid beside BENCH_rNN.json.
# usage: tools/profile_bench.sh <tag>      -> gpurun_out/<tag>_*
set -e
tag=$1
drv="--steps 20 --warmup 5"
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out
timeout -k 10 600 python3 -u $root/bench.py $drv > $out/${tag}_bench.json.log 2> $out/${tag}_bench.err
cd /tmp && export TMPDIR=/tmp
for mode in pipeline sequential; do
  rm -rf $out/${tag}_trace_$mode
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_trace_$mode -o t -- \
    python3 $root/bench.py $drv --no-cpu-baseline --no-other-configs --no-other-schedule --mode $mode > $out/${tag}_bench_$mode.json.log 2> $out/${tag}_trace_$mode.err
  cp $(find $out/${tag}_trace_$mode -name '*kernel_stats.csv' | head -1) $out/${tag}_bench_${mode}_kernel_stats.csv
  python3 $root/tools/block_timeline.py $(find $out/${tag}_trace_$mode -name '*kernel_trace.csv' | head -1) 90 > $out/${tag}_bench_${mode}_timeline.txt
  rm -rf $out/${tag}_trace_$mode
done
for c in FETCH_SIZE WRITE_SIZE; do
  lc=$(echo $c | tr A-Z a-z)
  rm -rf $out/${tag}_pmc_$lc
  timeout -s KILL 300 rocprofv3 --pmc $c --output-format csv -d $out/${tag}_pmc_$lc -o p -- \
    python3 $root/bench.py --no-cpu-baseline --no-other-configs --no-other-schedule --mode sequential --steps 10 --warmup 2 --blocks 1 > /dev/null 2> $out/${tag}_pmc_$lc.err
  f=$(find $out/${tag}_pmc_$lc -name '*counter_collection.csv' | head -1)
  head -1 $f > $out/${tag}_pmc_${lc}_crop.csv
  grep crop_resize $f >> $out/${tag}_pmc_${lc}_crop.csv
  rm -rf $out/${tag}_pmc_$lc
done
echo profile_bench done
