#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, own passes) of one unit shape.  Usage: tools/pmc_unit_hbm.sh C k dil tag [dtype]
C=${1:-128}; K=${2:-11}; D=${3:-1}; TAG=${4:-pmch}; DT=${5:-f16}
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
for set in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $set -d $OUT/$set -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_unit.py --C $C --k $K --dil $D --iters 3 --dtype $DT > $OUT.$set.log 2>&1
done
