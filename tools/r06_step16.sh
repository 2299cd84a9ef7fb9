#!/bin/bash
# phase clocks of the 16 x 16 x 32 emulated conv (tools/trace_conv16.py; DIAG build jatts_amd/lib_diagT)
O=gpurun_out
export JATTS_HIP_LIB=$PWD/jatts_amd/lib_diagT/libjatts_hip.so
(for S in 5 0 7 2; do for V in 0 2; do JATTS_CONV_EMUL16_VARIANT=$V python tools/trace_conv16.py --only $S 2>&1 | grep -v amdgpu.ids; done; done) 2>&1 | tee $O/r06_conv16_trace.txt
