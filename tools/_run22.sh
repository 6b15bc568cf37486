set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 120 python -u -m pytest tests/test_gpu_fuzz.py -m gpu -x -q --timeout 60 --timeout-method thread -k "row_ownership" > gpurun_out/r03_y_tests.log 2>&1
run() { # cfg name mode env...
  cfg=$1; name=$2; mode=$3; shift; shift; shift
  extra=""; [ $cfg = 5 ] && extra="--batch 16"
  env "$@" timeout -k 10 120 python -u bench.py --config $cfg $extra --steps 30 --warmup 5 --no-cpu-baseline --no-other-schedule --no-other-configs --mode $mode > gpurun_out/r03_y_cfg${cfg}_$name.json.log 2>gpurun_out/r03_y_err.log
}
run 5 stream_seq sequential FRCNN_CROP_ROWBAND=3
run 5 stream_pipe pipeline FRCNN_CROP_ROWBAND=3
run 5 stream_ns8 sequential FRCNN_CROP_ROWBAND=3 FRCNN_CROP_PREFETCH=8
run 5 stage_seq sequential FRCNN_CROP_ROWBAND=2
