#!/bin/bash
# Copy the judged summaries of tools/profile_round.sh <tag> from gpurun_out/ (scratch) into profiles/ (tracked).
# usage: tools/collect_profiles.sh <tag>
set -e
tag=$1
cd "$(dirname "$0")/.."
for f in ${tag}_a_bench.json.log ${tag}_a_bench_pipeline.json.log ${tag}_a_bench_sequential.json.log \
         ${tag}_a_bench_pipeline_kernel_stats.csv ${tag}_a_bench_sequential_kernel_stats.csv \
         ${tag}_a_pmc_fetch_size_crop.csv ${tag}_a_pmc_write_size_crop.csv; do
  cp gpurun_out/$f profiles/
done
for c in 3 4 5; do
  for s in .json.log _traced.json.log _kernel_stats.csv _timeline.txt; do cp gpurun_out/${tag}_b_cfg${c}_shard$s profiles/; done
  cp gpurun_out/${tag}_d_two_ranks_one_gpu_cfg$c.json.log profiles/
done
cp gpurun_out/${tag}_c_pmc_*.txt gpurun_out/${tag}_d_one_rank_rccl_cfg2.json.log profiles/
[ -f gpurun_out/${tag}_a_gpu_tests.log ] && tail -3 gpurun_out/${tag}_a_gpu_tests.log > profiles/${tag}_a_gpu_tests_tail.log
ls profiles/${tag}_* | wc -l
