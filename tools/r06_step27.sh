#!/bin/bash
# one-utterance launches (the B = 1 drop-in path): smaller emulated conv tiles
O=gpurun_out
(for V in 0 8 9 0 8 9; do echo "== variant $V"; JATTS_CONV_EMUL16_VARIANT=$V python tools/bench_conv.py --dtype emul --iters 50 --batch 1 --shapes 0,1,2,3 2>&1 | grep "emul v"; done) 2>&1 | tee $O/r06_conv16_b1_tiles.txt
