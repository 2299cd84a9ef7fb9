#!/bin/bash
# plain commit + staging plan + anti-phase staging of the eight-wave emulated conv tile: parity, then A/B (lib_diagA = the same build without the anti-phase)
O=gpurun_out
python -m pytest tests/test_emul_gpu.py tests/test_kernels_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee $O/r06_step17_tests.txt
(for L in lib_diagA lib; do
  export JATTS_HIP_LIB=$PWD/jatts_amd/$L/libjatts_hip.so
  for V in 0 2; do
    echo "== $L variant $V"
    JATTS_CONV_EMUL16_VARIANT=$V python tools/bench_conv.py --dtype emul --iters 20 2>&1 | grep "emul v"
  done
done) 2>&1 | tee $O/r06_conv16_antiphase.txt
