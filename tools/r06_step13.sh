O=gpurun_out
(for d in 0 1 2 3; do
  if [ $d = 0 ]; then unset JATTS_HIP_LIB; else export JATTS_HIP_LIB=$PWD/jatts_amd/lib_diag$d/libjatts_hip.so; fi
  echo "== DIAG $d (1: no B refills, 2: no A refills, 3: neither)"
  python tools/trace_unit.py --C 128 --k 7 --dtype emul --layout 1 2>&1 | grep "lifetime\|conv1\|conv2\|ticks"
  python tools/trace_unit.py --C 256 --k 7 --dtype emul --layout 1 2>&1 | grep "lifetime\|conv1\|ticks"
done) 2>&1 | tee $O/r06_unit16_diag.txt
