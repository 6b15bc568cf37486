set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests -x -q -m gpu --timeout 120 --timeout-method thread > gpurun_out/prep_tests.log 2>&1
timeout -k 10 200 python3 -u bench.py --config 5 --batch 16 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > gpurun_out/prep_cfg5.json.log 2>gpurun_out/prep.err
timeout -k 10 200 python3 -u bench.py --config 5 --batch 16 --steps 100 --warmup 5 --no-cpu-baseline --no-other-configs > gpurun_out/prep_cfg5_k100.json.log 2>gpurun_out/prep.err
bash tools/_nt.sh
