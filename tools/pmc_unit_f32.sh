#!/bin/bash
# SQ / GRBM counters of one f32 fused-unit shape (tools/bench_unit.py --dtype f32), counters in their own passes, kernel-trace only.
C=${1:-128}; K=${2:-7}; D=${3:-1}; TAG=${4:-pmc_unit_f32}
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  name=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set -d $OUT/$name -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_unit.py --C $C --k $K --dil $D --dtype f32 --iters 3 > $OUT.$name.log 2>&1
done
python3 $GRAFT_REPO_ROOT/tools/pmc_table.py $OUT resunit
