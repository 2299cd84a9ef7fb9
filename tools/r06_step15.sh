#!/bin/bash
# DIAG probes of the 16 x 16 x 32 emulated conv (tools/diag_conv_libs.sh): what is the k = 1 tile waiting for?
O=gpurun_out
(for N in 0 1 2 64 4 8 16 24 32 127; do
  if [ $N = 0 ]; then export JATTS_HIP_LIB=$PWD/jatts_amd/lib/libjatts_hip.so; else export JATTS_HIP_LIB=$PWD/jatts_amd/lib_diag$N/libjatts_hip.so; fi
  for V in 0 2; do
    echo "== DIAG $N variant $V (0: k = 1 -> 128 n x 64 t, two workgroups per CU; 2: 128 x 128, eight waves)"
    JATTS_CONV_EMUL16_VARIANT=$V python tools/bench_conv.py --dtype emul --iters 20 --shapes 0,5,7,2,3 2>&1 | grep "emul v"
  done
done) 2>&1 | tee $O/r06_conv16_diag.txt
