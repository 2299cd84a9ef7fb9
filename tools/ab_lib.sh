#!/bin/bash
# A/B of two builds of the library on one box: tools/ab_lib.sh <base.so> [<new.so> = jatts_amd/lib/libjatts_hip.so]
# (JATTS_HIP_LIB selects the build; split conv shapes and the 36 split dilation units.  The base build is a copy made before the change,
#  e.g. under lib_ab/ -- git-ignored, travels with the gpurun snapshot.)
BASE=${1:?base library}; NEW=${2:-jatts_amd/lib/libjatts_hip.so}
for L in "$BASE" "$NEW"; do
  echo "=== $L"
  JATTS_HIP_LIB=$PWD/$L python tools/bench_conv.py --dtype split 2>&1 | tail -14
  JATTS_HIP_LIB=$PWD/$L python tools/bench_unit.py --dtype split --all 2>&1 | tail -3
done
