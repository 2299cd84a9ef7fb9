#!/bin/bash
O=gpurun_out
python -m pytest tests/test_emul_gpu.py tests/test_kernels_gpu.py tests/test_fs2_gpu.py tests/test_benchsize_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee $O/r06_step33_tests.txt
python tools/bench_conv.py --dtype emul --iters 20 2>&1 | grep "emul v" | tee $O/r06_conv_final_rule.txt | tail -3
