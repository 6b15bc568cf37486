#!/bin/bash
# A/B runs of bench.py: every line on stdin is one run, "[VAR=value ...] :: bench.py arguments"; prints a short summary per run.
# usage: printf '%s\n' ':: --input-sets 1' 'FRCNN_CROP_PREFETCH=32 :: --input-sets 16' | tools/bench_ab.sh
while IFS= read -r line; do
  [ -z "$line" ] && continue
  envs="${line%%::*}"
  args="${line#*::}"
  out=$(env $envs timeout -k 10 300 python -u bench.py --no-cpu-baseline --no-other-configs $args 2>/dev/null | tail -1) || exit 1
  python - "$line" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
r = d["roofline"]
print(sys.argv[1], "| value", d["value"], "min/max", d["blocks"]["value_min"], d["blocks"]["value_max"], "| dom ms", r["avg_launch_ms"], "frac", r["frac"],
      "n", r["launches_timed"], "| other", (d.get("other_schedule") or {}).get("value"), "| host", d["host_enqueue_ms_per_step"], flush=True)
PY
done
