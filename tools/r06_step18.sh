#!/bin/bash
# paired bf16 x 3 split (bf3_split2: 4.5 VALU per element): parity, units / convs by shape, a quick headline step
O=gpurun_out
python -m pytest tests/test_emul_gpu.py tests/test_kernels_gpu.py tests/test_hifigan_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee $O/r06_step18_tests.txt
python tools/bench_unit.py --all --dtype emul --layout 1 2>&1 | grep -v amdgpu.ids | tee $O/r06_units_split2.txt
python tools/bench_conv.py --dtype emul --iters 20 2>&1 | grep "emul v" | tee $O/r06_conv_split2.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-ragged --no-fast-mode --no-train --no-configs --no-pmc --no-24k --no-b1 --no-detail 2>/dev/null | tee $O/r06_bench_split2.json | head -c 600
