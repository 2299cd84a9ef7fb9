#!/bin/bash
# persistent C = 128 / 256 units (resunit_emul16p_kernel): parity, then A/B by JATTS_UNIT16_PERSIST on one box
O=gpurun_out
python -m pytest tests/test_emul_gpu.py tests/test_hifigan_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee $O/r06_step20_tests.txt
(for P in 0 1 0 1; do
  echo "== JATTS_UNIT16_PERSIST=$P"
  JATTS_UNIT16_PERSIST=$P python tools/bench_unit.py --all --dtype emul --layout 1 2>&1 | grep "C= 256\|C= 128\|sum over"
done) 2>&1 | tee $O/r06_units_persist_ab.txt
(for C in 128 256; do for K in 3 7 11; do python tools/trace_unit.py --C $C --k $K --dil 3 --dtype emul --layout 1 2>&1 | grep -v amdgpu.ids | head -10; done; done) 2>&1 | tee $O/r06_trace_units_persist.txt
