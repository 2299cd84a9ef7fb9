#!/bin/bash
O=gpurun_out
(python -m pytest tests/test_benchsize_gpu.py -x -q -m gpu -k "test_matcha_bench_utterance_matches_the_reference" 2>&1 | tail -40
echo "=== variant 2 (eight-wave 128 x 128 tile everywhere)"
JATTS_CONV_EMUL16_VARIANT=2 python -m pytest tests/test_benchsize_gpu.py -x -q -m gpu -k "test_matcha_bench_utterance_matches_the_reference" 2>&1 | tail -15
python tools/mfma_forms.py 2>&1 | grep -v amdgpu) 2>&1 | tee $O/r06_step30.txt
