#!/bin/bash
# rocprofv3 --kernel-trace --stats of BASELINE configs[2] (Matcha-TTS MAS + HiFi-GAN, 64 utterances, 10 Euler steps) and configs[4]'s per-GPU share
# (mel-VITS, 192-d speaker embedding, 32 utterances): the SAME legs bench.py times for its `configs` block (bench.py --profile-config).
# usage: tools/profile_models.sh [TAG=r04] [precision=fp32]   -> gpurun_out/<TAG>_infer_{matcha,vits}/ ; copy the *kernel_stats.csv to profiles/
TAG=${1:-r04}; PREC=${2:-fp32}
cd /tmp; export TMPDIR=/tmp
for kind in matcha vits; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_infer_${kind}
  mkdir -p $OUT
  timeout 900 rocprofv3 --kernel-trace --stats -d $OUT -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --profile-config $kind --precision $PREC --steps 3 --warmup 1 --no-pmc > $OUT.log 2>&1
  find $OUT -name "*kernel_stats.csv" | head -1
done
