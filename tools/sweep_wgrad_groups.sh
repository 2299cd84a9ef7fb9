#!/bin/bash
# Which split-K factor (sequence groups) is fastest for jatts_conv1d_wgrad's MFMA kernel?  JATTS_WGRAD_GROUPS forces it; 0 = the library's rule.
for G in 0 2 4 8 11 16 24 32; do echo "== groups $G"; JATTS_WGRAD_GROUPS=$G python tools/bench_wgrad.py 2>&1 | tail -8; done
