#!/bin/bash
O=gpurun_out
python -m pytest tests/test_emul_gpu.py tests/test_kernels_gpu.py tests/test_fs2_gpu.py tests/test_benchsize_gpu.py tests/test_hifigan_gpu.py tests/test_graph_gpu.py -x -q -m gpu 2>&1 | tail -2 | tee $O/r06_step36_tests.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-ragged --no-fast-mode --no-train --no-pmc --no-24k --no-detail 2>/dev/null | tee $O/r06_bench_direct.json | head -c 200
