#!/bin/bash
O=gpurun_out
(for V in 0 10 0 10; do echo "== variant $V"; JATTS_CONV_EMUL16_VARIANT=$V python tools/bench_conv.py --dtype emul --iters 30 --shapes 0,3,12,13,14,15,18,19,20,21,22,23 2>&1 | grep "emul v"; done) 2>&1 | tee $O/r06_conv16_192_tile.txt
