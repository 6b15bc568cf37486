#!/bin/bash
# A/B of two libraries on the SEQUENTIAL schedule with cold inputs: per-kernel averages from a rocprofv3 kernel trace.
# usage (on the GPU box): tools/ab_seq_trace.sh libA.so libB.so
cd /tmp && export TMPDIR=/tmp
for l in "$@"; do
  rm -rf /tmp/abt
  FRCNN_HIP_LIB=$GRAFT_REPO_ROOT/$l timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abt -o t -- \
    python3 $GRAFT_REPO_ROOT/bench.py --mode sequential --no-cpu-baseline --no-other-configs --no-other-schedule > /tmp/abt.log 2>&1 || exit 1
  echo "== $l $(grep '^{' /tmp/abt.log | tail -1 | python3 -c 'import json,sys; print(json.loads(sys.stdin.read())["value"])')"
  python3 - <<PY
import csv,glob
f=glob.glob('/tmp/abt/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r["Name"] for k in ("proposal_kernel","decoder_kernel","crop_resize")): print("  ", r["Name"][:40], r["Calls"], round(float(r["AverageNs"])/1e3,2))
PY
done
