#!/bin/bash
# A/B: C = 64 one-workgroup-per-CU unit at smaller k (JATTS_UNIT16_C64_WIDE_K), C = 32 with 512-column windows (JATTS_UNIT16_C32_WIDE)
O=gpurun_out
JATTS_UNIT16_C64_WIDE_K=3 JATTS_UNIT16_C32_WIDE=1 python -m pytest tests/test_emul_gpu.py -x -q -m gpu -k "resunit or unit" 2>&1 | tail -2 | tee $O/r06_step38_tests.txt
(for V in base wide base wide; do
  if [ $V = wide ]; then export JATTS_UNIT16_C64_WIDE_K=3 JATTS_UNIT16_C32_WIDE=1; else unset JATTS_UNIT16_C64_WIDE_K JATTS_UNIT16_C32_WIDE; fi
  echo "== $V"; python tools/bench_unit.py --all --dtype emul --layout 1 2>&1 | grep "C=  64\|C=  32"; done) 2>&1 | tee $O/r06_unit16_c64_c32_wide.txt
