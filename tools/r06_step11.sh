set -x
O=gpurun_out
timeout 900 python -m pytest tests/test_emul_gpu.py -m gpu -q -x -k "conv1d" 2>&1 | tail -4 > $O/r06_t_conv16.txt
tail -n 4 $O/r06_t_conv16.txt
(for l in 0 1; do echo "=== layout $l"; python tools/bench_conv.py --dtype emul --iters 20 --layout $l 2>&1 | grep "emul"; done) > $O/r06_conv_mfma_forms.txt 2>&1
cat $O/r06_conv_mfma_forms.txt
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/r06_t_all_mid2.txt
tail -n 8 $O/r06_t_all_mid2.txt
python bench.py --no-cpu-baseline --no-train --no-pmc --no-ragged --no-fast-mode --no-b1 --no-24k --steps 10 --warmup 3 > $O/r06_bench_quick.json 2>$O/r06_bench_quick.err
tail -c 1300 $O/r06_bench_quick.json
