O=gpurun_out
(for a in "128 7" "256 7" "64 7" "32 7"; do set -- $a; for l in 0 1; do echo "== emul C=$1 k=$2 layout $l"; python tools/trace_unit.py --C $1 --k $2 --dtype emul --layout $l 2>&1 | grep -v amdgpu.ids | head -10; done; done) > $O/r06_trace_emul_forms.txt 2>&1
cat $O/r06_trace_emul_forms.txt
