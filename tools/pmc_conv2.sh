#!/bin/bash
# Memory-side + issue-side counters of one jatts_conv1d shape / variant (tools/bench_conv.py --only IDX --variant V), one small
# counter set per pass (kernel-trace only).  usage: tools/pmc_conv2.sh IDX VARIANT TAG
IDX=${1:-6}; VAR=${2:-0}; TAG=${3:-pmc_conv2}
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" \
           "TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN2_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TCC_BUSY_avr TCC_TAG_STALL_sum FETCH_SIZE"; do
  name=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set -d $OUT/$name -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_conv.py --only $IDX --variant $VAR --iters 3 > $OUT/$name.log 2>&1
done
python3 $GRAFT_REPO_ROOT/tools/pmc_table.py $OUT conv1d
