#!/bin/bash
O=gpurun_out
JATTS_UNIT16_C32_SMALL=1 python -m pytest tests/test_emul_gpu.py -x -q -m gpu -k "resunit or unit" 2>&1 | tail -2 | tee $O/r06_step39_tests.txt
(for V in base small base small; do
  if [ $V = small ]; then export JATTS_UNIT16_C32_SMALL=1; else unset JATTS_UNIT16_C32_SMALL; fi
  echo "== $V"; python tools/bench_unit.py --all --dtype emul --layout 1 2>&1 | grep "C=  32"; done) 2>&1 | tee $O/r06_unit16_c32_small.txt
