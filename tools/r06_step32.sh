#!/bin/bash
O=gpurun_out
(for V in 0 11 0 11; do echo "== variant $V"; JATTS_CONV_EMUL16_VARIANT=$V python tools/bench_conv.py --dtype emul --iters 30 --shapes 4,5,6,7,8,9,10,11 2>&1 | grep "emul v"; done) 2>&1 | tee $O/r06_conv16_512_tile.txt
