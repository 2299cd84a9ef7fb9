#!/bin/bash
O=gpurun_out
python -m pytest tests/test_kernels_gpu.py tests/test_benchsize_gpu.py tests/test_matcha_gpu.py tests/test_vits_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee $O/r06_step31_tests.txt
