#!/bin/bash
# rocprofv3 --kernel-trace --stats of one trainer's bench leg.  usage: tools/profile_train.sh KIND [steps]
KIND=${1:-vits}; STEPS=${2:-6}
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG:-r04}_train_$KIND
mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/train_line.py $KIND $STEPS > $OUT.log 2>&1
tail -1 $OUT.log | cut -c1-600
find $OUT -name "*kernel_stats.csv" | head -3
