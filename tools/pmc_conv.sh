#!/bin/bash
# SQ / GRBM counters of one jatts_conv1d shape (tools/bench_conv.py --only IDX), counters in their own passes, kernel-trace only.
IDX=${1:-6}; DT=${2:-f32}; TAG=${3:-pmc_conv}
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  name=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set -d $OUT/$name -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_conv.py --only $IDX --dtype $DT --iters 3 > $OUT.$name.log 2>&1
done
python3 $GRAFT_REPO_ROOT/tools/pmc_table.py $OUT conv1d
