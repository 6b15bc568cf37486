#!/bin/bash
# PMC passes over the crop kernel (tools/diag/crop_bench quick mode); one counter group per run.
# usage: tools/pmc_crop.sh <outdir-under-gpurun_out> [FRCNN_CROP_CHUNK]
set -e
out=$GRAFT_REPO_ROOT/gpurun_out/$1
export FRCNN_CROP_CHUNK=${2:-0}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum" \
           "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TC_STALL_sum TD_TD_BUSY_sum SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout -s KILL 60 rocprofv3 --pmc $grp --output-format csv -d $out/p$i -o p$i -- $GRAFT_REPO_ROOT/tools/diag/crop_bench quick > $out/p$i.log 2>&1
done
echo done
