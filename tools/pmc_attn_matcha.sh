#!/bin/bash
# What the f32 d_k 256 attention kernel waits on inside one Matcha config-3 batch: small counter sets, each pass its own process and timeout.
# usage: tools/pmc_attn_matcha.sh [TAG]   -> gpurun_out/TAG/<first counter>/..., table on stdout
TAG=${1:-r04_pmc_attn}
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  name=$(echo $set | cut -d' ' -f1)
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/$name -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --profile-config matcha --precision fp32 --steps 1 --warmup 1 --no-pmc > $OUT/$name.log 2>&1 || echo "pass $name failed or timed out"
done
python3 $GRAFT_REPO_ROOT/tools/pmc_table.py $OUT "relattn_kernel<float, 256" | cut -c1-120
