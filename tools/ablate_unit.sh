#!/bin/bash
# Build profiling-only ablation variants of the fused unit kernel (JATTS_ABLATE=n) into jatts_amd/lib/ablate/.
set -e
cd "$(dirname "$0")/../jatts_amd/csrc"
mkdir -p ../lib/ablate
for n in ${ABLATE_SET:-1 2 3 4 5 6 7 8 9}; do
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-pass-failed -Wno-unused-variable -DJATTS_ABLATE=$n -c conv_mfma.hip -o ../lib/ablate/conv_$n.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/ablate/libjatts_hip_ab$n.so ../lib/ablate/conv_$n.o ../lib/api.o ../lib/attention.o ../lib/rowwise.o ) &
  if (( n % 3 == 0 )); then wait; fi
done
wait
ls -la ../lib/ablate/*.so
