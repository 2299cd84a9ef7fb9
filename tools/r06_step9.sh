O=gpurun_out
timeout 1500 python -m pytest tests/test_emul_gpu.py -m gpu -q -x -k "resunit or unit_weight" 2>&1 | tail -4 > $O/r06_t_emul16.txt
tail -n 4 $O/r06_t_emul16.txt
(echo "=== layout 1"; python tools/bench_unit.py --all --dtype emul --layout 1 2>&1 | grep "C=") > $O/r06_units_mfma16.txt 2>&1
cat $O/r06_units_mfma16.txt
(for a in "128 7" "256 7" "64 7"; do set -- $a; echo "== emul C=$1 k=$2 layout 1"; python tools/trace_unit.py --C $1 --k $2 --dtype emul --layout 1 2>&1 | grep -v amdgpu.ids | head -10; done) 2>&1 | tee $O/r06_trace_emul16.txt
