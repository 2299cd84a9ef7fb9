#!/bin/bash
# Matrix-pipe utilisation of the f32 attention kernel inside one bench step, 64-key tiles (JATTS_ATTN_KB32=0) against 32-key tiles: two small
# counter sets per setting, each pass under its own timeout (see tools/pmc_conv2.sh).  usage: tools/pmc_attn.sh [TAG]
TAG=${1:-r03_pmc_attn}
cd /tmp; export TMPDIR=/tmp
for kb in 0 1; do
  export JATTS_ATTN_KB32=$kb
  OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG/kb32_$kb
  mkdir -p $OUT
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
    name=$(echo $set | cut -d' ' -f1)
    timeout -k 5 240 rocprofv3 --kernel-trace --pmc $set -d $OUT/$name -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-pmc --no-configs --no-train --no-fast-mode --steps 1 --warmup 1 > $OUT/$name.log 2>&1 || echo "pass $name failed or timed out"
  done
  echo "== JATTS_ATTN_KB32=$kb"
  python3 $GRAFT_REPO_ROOT/tools/pmc_table.py $OUT relattn | cut -c1-110
done
