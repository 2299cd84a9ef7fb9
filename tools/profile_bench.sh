#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench's f32 headline leg (no CPU baseline, no PMC child passes, no extra configs):
# the per-kernel summary the roofline block is cross-checked against.  usage: tools/profile_bench.sh TAG [bench args]
# (--precision fp32_bf16x3 ... for the other arithmetics; --only-ragged for the ragged leg alone)
TAG=${1:-r03_bench_prof}; shift
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-pmc --no-configs --no-train --no-fast-mode --no-ragged --no-24k --no-b1 --no-detail --steps 7 --warmup 2 "$@" > $OUT.log 2>&1
find $OUT -name "*kernel_stats.csv" | head -3
