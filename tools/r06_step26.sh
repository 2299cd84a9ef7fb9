#!/bin/bash
O=gpurun_out
python -m pytest tests/test_emul_gpu.py tests/test_kernels_gpu.py tests/test_fs2_gpu.py -x -q -m gpu 2>&1 | tail -2 | tee $O/r06_step26_tests.txt
(for V in 3 7 3 7; do echo "== variant $V"; JATTS_CONV_EMUL16_VARIANT=$V python tools/bench_conv.py --dtype emul --iters 20 --shapes 4,6,7,8,11,16,17 2>&1 | grep "emul v"; done
 echo "== product rule, all shapes"; python tools/bench_conv.py --dtype emul --iters 20 2>&1 | grep "emul v") 2>&1 | tee $O/r06_conv16_rule2.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-ragged --no-fast-mode --no-train --no-pmc --no-24k --no-detail 2>/dev/null | tee $O/r06_bench_wide2.json | head -c 300
