#!/bin/bash
O=gpurun_out
(for V in 0 x 0 x; do if [ $V = x ]; then unset JATTS_CONV_EMUL16_DIRECT_EPI; echo "== product rule (direct when no residual)"; else export JATTS_CONV_EMUL16_DIRECT_EPI=0; echo "== LDS tile always"; fi
  python tools/bench_models.py --model vits --precision fp32_bf16x3 --steps 5 --no-vocoder 2>&1 | grep -v amdgpu | tail -4; done) 2>&1 | tee $O/r06_vits_direct_epi.txt
