#!/bin/bash
# rocprofv3 --kernel-trace --stats of the B = 1 drop-in path (bench.py --b1-child: model.inference(x) + vocoder.decode(mel) on one 128-phoneme utterance,
# hipGraph replay): the per-kernel table behind b1_latency.kernel_ms.  usage: tools/profile_b1.sh TAG [bench args]
TAG=${1:-r06_b1_prof}; shift
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --b1-child --b1-iters 30 "$@" > $OUT.log 2>&1
find $OUT -name "*kernel_stats.csv" | head -3
