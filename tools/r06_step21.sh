#!/bin/bash
O=gpurun_out
python -m pytest tests/test_emul_gpu.py -x -q -m gpu -k "unit or resunit" 2>&1 | tail -2 | tee $O/r06_step21_tests.txt
(for P in 0 1 0 1; do
  echo "== JATTS_UNIT16_PERSIST=$P"
  JATTS_UNIT16_PERSIST=$P python tools/bench_unit.py --all --dtype emul --layout 1 2>&1 | grep "C= 256\|C= 128\|sum over"
done) 2>&1 | tee $O/r06_units_persist_ab2.txt
