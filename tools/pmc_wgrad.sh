#!/bin/bash
# Issue-side counters of the MFMA wgrad kernel over tools/bench_wgrad.py's shapes, one small counter set per pass, each under its own
# timeout (see tools/pmc_conv2.sh).  usage: tools/pmc_wgrad.sh [TAG]
TAG=${1:-r03_pmc_wgrad}
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SALU" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  name=$(echo $set | cut -d' ' -f1)
  timeout -k 5 150 rocprofv3 --kernel-trace --pmc $set -d $OUT/$name -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_wgrad.py > $OUT/$name.log 2>&1 || echo "pass $name failed or timed out"
done
python3 $GRAFT_REPO_ROOT/tools/pmc_table.py $OUT conv_wgrad_mfma_kernel | cut -c1-120
