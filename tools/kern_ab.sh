#!/bin/bash
# A/B of tools/bench_kernels.py cases: every stdin line is "[VAR=value ...] :: bench_kernels.py arguments".
while IFS= read -r line; do
  [ -z "$line" ] && continue
  envs="${line%%::*}"
  args="${line#*::}"
  echo "## $line"
  env $envs timeout -k 10 300 python -u tools/bench_kernels.py $args 2>/dev/null | cut -c1-330 || exit 1
done
