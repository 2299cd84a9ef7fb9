#!/bin/bash
# HBM traffic of the bench's kernels from PMC counters: separate passes (FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2),
# kernel-trace only (MI355X_MICROARCH.md "rocprofv3 PMC slots").  Output: gpurun_out/<tag>/{FETCH_SIZE,WRITE_SIZE}/...
TAG=${1:-pmc_bench}
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $OUT/$c -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT.$c.log 2>&1
done
ls -R $OUT | head
