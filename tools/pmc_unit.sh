#!/bin/bash
# PMC passes for the unit kernel (counters in their own runs, kernel-trace only).  Usage: tools/pmc_unit.sh C k dil tag
C=${1:-128}; K=${2:-11}; D=${3:-1}; TAG=${4:-pmc}
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  name=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set -d $OUT/$name -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_unit.py --C $C --k $K --dil $D --iters 3 > $OUT.$name.log 2>&1
done
ls -R $OUT | head -30
