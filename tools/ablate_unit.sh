#!/bin/bash
# DIAGNOSIS builds of libjatts_hip.so -- never the shipped library (jatts_amd/lib/libjatts_hip.so is built without JATTS_DIAG, where
# JATTS_ABLATE is forced to 0 and conv variant 8 does not exist):
#   tools/ablate_unit.sh           -> jatts_amd/lib/diag_ab<n>/libjatts_hip.so for n in ABLATE_SET (profiling-only ablations of the fused unit)
#   ABLATE_SET=0 tools/ablate_unit.sh -> jatts_amd/lib/diag_ab0/ (DIAG only: jatts_conv1d variant 8, "nothing streamed", wrong results by design)
# Point a tool at one of them with JATTS_HIP_LIB=<path> (jatts_amd/_abi.py).
set -e
cd "$(dirname "$0")/../jatts_amd/csrc"
for n in ${ABLATE_SET:-1 2 3 4 5 6 7 8 9}; do
  make -j8 OUT=../lib/diag_ab$n CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable -Wno-pass-failed -DJATTS_DIAG -DJATTS_ABLATE=$n"
done
ls -la ../lib/diag_ab*/libjatts_hip.so
