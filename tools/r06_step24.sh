#!/bin/bash
O=gpurun_out
JATTS_CONV_EMUL16_VARIANT=6 python -m pytest tests/test_emul_gpu.py -x -q -m gpu -k "conv1d" 2>&1 | tail -2 | tee $O/r06_step24_tests.txt
(for V in 0 6 0 6; do echo "== variant $V"; JATTS_CONV_EMUL16_VARIANT=$V python tools/bench_conv.py --dtype emul --iters 20 --shapes 0,1,2,3,5,13,14,15,18,20,21,22,23 2>&1 | grep "emul v"; done) 2>&1 | tee $O/r06_conv16_384_tile.txt
