O=gpurun_out
(for d in 1 2; do
  export JATTS_HIP_LIB=$PWD/jatts_amd/lib_diag$d/libjatts_hip.so
  echo "== build $d (1: rows order, 2: pairs order)"
  python tools/trace_unit.py --C 128 --k 7 --dtype emul --layout 1 2>&1 | grep "launch\|lifetime\|conv1\|conv2\|ticks"
  python tools/trace_unit.py --C 256 --k 7 --dtype emul --layout 1 2>&1 | grep "launch\|lifetime\|conv1\|ticks"
  python tools/bench_unit.py --all --dtype emul --layout 1 2>&1 | grep "C= 128 k=11\|C= 256 k=11\|C= 128 k= 7\|C=  64 k=11"
done) 2>&1 | tee $O/r06_unit16_order.txt
