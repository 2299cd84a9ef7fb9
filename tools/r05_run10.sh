set -x
O=gpurun_out
for a in "128 7" "128 11" "64 7" "256 7"; do set -- $a; echo "== emul C=$1 k=$2"; python tools/trace_unit.py --C $1 --k $2 --dtype emul; done > $O/r05_trace_emul.txt 2>&1
python -m pytest tests/test_kernels_gpu.py -q -k "ragged_1d" 2>&1 | tail -3 > $O/r05_t_ragged1d.txt
python tools/bench_unit.py --all --dtype emul > $O/r05_units_emul5.txt 2>&1
tail -3 $O/r05_t_ragged1d.txt; tail -2 $O/r05_units_emul5.txt
