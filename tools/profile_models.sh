#!/bin/bash
# rocprofv3 --kernel-trace --stats of tools/bench_models.py (configs 3 and 5).  usage: tools/profile_models.sh [bench_models args]
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_models
mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_models.py "$@" > $OUT.log 2>&1
tail -3 $OUT.log | cut -c1-400
