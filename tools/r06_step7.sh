set -x
O=gpurun_out
timeout 1500 python -m pytest tests/test_emul_gpu.py -m gpu -q -x -k "resunit or unit_weight" 2>&1 | tail -8 > $O/r06_t_emul16.txt
tail -n 8 $O/r06_t_emul16.txt
(for l in 0 1; do echo "=== layout $l"; python tools/bench_unit.py --all --dtype emul --layout $l 2>&1 | grep "C="; done) > $O/r06_units_mfma_forms.txt 2>&1
cat $O/r06_units_mfma_forms.txt
