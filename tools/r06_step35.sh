#!/bin/bash
O=gpurun_out
(for V in 0 1 0 1; do echo "== JATTS_CONV_EMUL16_DIRECT_EPI=$V"; JATTS_CONV_EMUL16_DIRECT_EPI=$V python tools/bench_conv.py --dtype emul --iters 20 2>&1 | grep "emul v"; done) 2>&1 | tee $O/r06_conv16_direct_epilogue.txt
