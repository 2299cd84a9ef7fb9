#!/bin/bash
# DIAG-only builds of the emulated conv (JATTS_CEMUL_DIAG bit sets; wrong results, timing probes): jatts_amd/lib_diag<N>/libjatts_hip.so, only
# conv1d_emul.o differs from jatts_amd/lib.  tools/diag_conv_libs.sh 1 2 4 ...   (git-ignored; delete the directories after the measurement)
set -e
cd "$(dirname "$0")/../jatts_amd/csrc"
make -s
build() {
  N=$1; D=../lib_diag$N; mkdir -p $D
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-pass-failed -DJATTS_CEMUL_DIAG=$N -c conv1d_emul.hip -o $D/conv1d_emul.o
  OBJS=$(ls ../lib/*.o | grep -v conv1d_emul.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libjatts_hip.so $OBJS $D/conv1d_emul.o
  rm $D/conv1d_emul.o
  echo built $D
}
for N in "$@"; do build $N & 
  while [ $(jobs -r | wc -l) -ge 4 ]; do sleep 1; done
done
wait
