#!/bin/bash
# A/B: wider emulated conv tiles in the 16 x 16 x 32 form (JATTS_CONV_EMUL16_VARIANT 3 / 4 / 5) against the product choice (0)
O=gpurun_out
(for V in 0 3 4 5; do
  echo "== variant $V"
  JATTS_CONV_EMUL16_VARIANT=$V python tools/bench_conv.py --dtype emul --iters 20 2>&1 | grep "emul v"
done) 2>&1 | tee $O/r06_conv16_wide_tiles.txt
