set -x
O=gpurun_out
python -m pytest tests/test_emul_gpu.py tests/test_spk_concat_gpu.py -q 2>&1 | tail -25 > $O/r05_t_emul.txt
python tools/emul_sweep.py --products 7 --units 400 --convs 700 --out $O/r05_emul_sweep.json > $O/r05_emul_sweep.txt 2>&1
python tools/emul_sweep.py --products 6 --units 400 --convs 700 --out $O/r05_emul6_sweep.json > $O/r05_emul6_sweep.txt 2>&1
(echo "== emul (7 products)"; python tools/bench_unit.py --all --dtype emul; echo "== emul6"; python tools/bench_unit.py --all --dtype emul6) > $O/r05_units_emul3.txt 2>&1
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train --no-configs --no-pmc > $O/r05_bench_quick3.json 2> $O/r05_bench_quick3.err
cp bench_detail.json $O/r05_bench_quick3_detail.json
JATTS_RAGGED_1D=0 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train --no-configs --no-pmc --no-fast-mode --no-detail > $O/r05_bench_quick3_rect.json 2>> $O/r05_bench_quick3.err
python -m pytest tests/test_fullsize_gpu.py tests/test_hifigan_gpu.py -q -k "full_batch or independent" 2>&1 | tail -8 > $O/r05_t_misc.txt
tail -n 4 $O/r05_t_emul.txt $O/r05_t_misc.txt
