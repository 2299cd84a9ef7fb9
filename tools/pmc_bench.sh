#!/bin/bash
# HBM traffic of the bench's kernels from PMC counters: separate passes (FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2),
# kernel-trace only (MI355X_MICROARCH.md "rocprofv3 PMC slots").  Output: gpurun_out/<tag>/{FETCH_SIZE,WRITE_SIZE}/...
# The profiled program is bench.py's own --pmc-child leg (one f32 and one f16 step) -- the same passes bench.py spawns
# by itself for roofline.traffic; this script keeps the raw CSVs for profiles/.
TAG=${1:-pmc_bench}
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 600 rocprofv3 --kernel-trace --pmc $c -d $OUT/$c -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --pmc-child --pmc-all > $OUT.$c.log 2>&1
done
ls -R $OUT | head
