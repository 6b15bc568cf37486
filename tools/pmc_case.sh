#!/bin/bash
# Kernel-trace stats + PMC passes (one counter group per run, as the guide prescribes) over one case of tools/bench_kernels.py.
# usage: tools/pmc_case.sh <name-under-gpurun_out> <case> [extra bench_kernels.py args]   (env knobs are inherited)
set -e
name=$1; shift
kase=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$name
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o trace -- python3 $GRAFT_REPO_ROOT/tools/bench_kernels.py --only $kase --iters 20 "$@" > $out/trace.log 2>&1
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $grp --output-format csv -d $out/p$i -o p$i -- python3 $GRAFT_REPO_ROOT/tools/bench_kernels.py --only $kase --iters 5 "$@" > $out/p$i.log 2>&1
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $out
