set -x
O=gpurun_out
python -m pytest tests/test_emul_gpu.py -q 2>&1 | tail -15 > $O/r05_t_emul.txt
python tools/emul_sweep.py --products 7 --units 400 --convs 700 --out $O/r05_emul_sweep.json > $O/r05_emul_sweep.txt 2>&1
python tools/emul_sweep.py --products 6 --units 400 --convs 700 --out $O/r05_emul6_sweep.json > $O/r05_emul6_sweep.txt 2>&1
(echo "== emul (7 products)"; python tools/bench_unit.py --all --dtype emul; echo "== emul6"; python tools/bench_unit.py --all --dtype emul6) > $O/r05_units_emul4.txt 2>&1
for v in 0 1 3; do echo "== emul7 variant $v"; JATTS_CONV_EMUL_VARIANT=$v python tools/bench_conv.py --dtype emul; done > $O/r05_conv_emul7.txt 2>&1
python tools/bench_unit.py --resblock --dtype emul >> $O/r05_units_emul4.txt 2>&1
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train --no-configs --no-pmc > $O/r05_bench_quick4.json 2> $O/r05_bench_quick4.err
python -m pytest tests/test_hifigan_gpu.py tests/test_fullsize_gpu.py tests/test_benchsize_gpu.py tests/test_spk_concat_gpu.py -q -k "bf16x3" 2>&1 | tail -5 > $O/r05_t_models.txt
tail -n 4 $O/r05_t_emul.txt $O/r05_t_models.txt
