set -e
cd $GRAFT_REPO_ROOT
run() { # name lib env...
  name=$1; lib=$2; shift; shift
  L=""; [ -n "$lib" ] && L="FRCNN_HIP_LIB=tools/diag/lib$lib.so"
  env $L "$@" timeout -k 10 200 python3 -u bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > gpurun_out/nt_$name.json.log 2>gpurun_out/nt.err
}
run base "" A=1
run nt512u4 nt512u4 A=1
run nt1024u4 nt1024u4 A=1
run nt1024u4_l64 nt1024u4 FRCNN_CROP_PREFETCH=640048
run nt1024u2 nt1024u2 A=1
run nt1024u2_l64 nt1024u2 FRCNN_CROP_PREFETCH=640048
run nt1024u2w8 nt1024u2w8 A=1
run nt1024u2w8_l128 nt1024u2w8 FRCNN_CROP_PREFETCH=1280048
run base2 "" A=1
