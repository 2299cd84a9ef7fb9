#!/bin/bash
# Memory-side + issue-side counters of one jatts_conv1d shape / variant (tools/bench_conv.py --only IDX --variant V), ONE small counter
# set per pass (kernel-trace only), every pass under its own `timeout`: a set the hardware cannot schedule ("Request exceeds the
# capabilities of the hardware to collect", e.g. four TA_* counters at once) aborts rocprofv3 and then HANGS until killed -- that cost
# round 3 twenty-five GPU-minutes.  usage: tools/pmc_conv2.sh IDX VARIANT TAG
IDX=${1:-6}; VAR=${2:-0}; TAG=${3:-pmc_conv2}
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" \
           "TA_TA_BUSY_sum TA_BUSY_avr" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" \
           "TCC_HIT_sum TCC_MISS_sum" \
           "TCC_EA0_RDREQ_sum TCC_REQ_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $set | cut -d' ' -f1)
  timeout -k 5 120 rocprofv3 --kernel-trace --pmc $set -d $OUT/$name -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_conv.py --only $IDX --variant $VAR --iters 3 > $OUT/$name.log 2>&1 || echo "pass $name failed or timed out"
done
python3 $GRAFT_REPO_ROOT/tools/pmc_table.py $OUT conv1d
