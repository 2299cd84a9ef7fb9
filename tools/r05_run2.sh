set -x
O=gpurun_out
python -m pytest tests/test_emul_gpu.py -x -q 2>&1 | tail -25 > $O/r05_t_emul.txt
python -m pytest tests/test_hifigan_gpu.py tests/test_fullsize_gpu.py tests/test_benchsize_gpu.py -q -k "bf16x3" 2>&1 | tail -25 > $O/r05_t_models.txt
for v in 0 1 2; do echo "== emul variant $v"; JATTS_CONV_EMUL_VARIANT=$v python tools/bench_conv.py --dtype emul; done > $O/r05_conv_emul.txt 2>&1
(echo "== f32"; python tools/bench_conv.py --dtype f32; echo "== split"; python tools/bench_conv.py --dtype split) >> $O/r05_conv_emul.txt 2>&1
python tools/bench_unit.py --all --dtype emul > $O/r05_units_emul2.txt 2>&1
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train --no-configs --no-pmc > $O/r05_bench_quick.json 2> $O/r05_bench_quick.err
cp bench_detail.json $O/r05_bench_quick_detail.json
tail -3 $O/r05_t_emul.txt $O/r05_t_models.txt
