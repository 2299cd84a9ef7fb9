#!/bin/bash
O=gpurun_out
JATTS_UNIT16_VARIANT=1 python -m pytest tests/test_emul_gpu.py -x -q -m gpu -k "resunit or unit" 2>&1 | tail -2 | tee $O/r06_step34_tests.txt
(for V in 0 1 0 1; do echo "== JATTS_UNIT16_VARIANT=$V"; JATTS_UNIT16_VARIANT=$V python tools/bench_unit.py --all --dtype emul --layout 1 2>&1 | grep "C= 128"; done) 2>&1 | tee $O/r06_unit16_wn4.txt
