set -x
O=gpurun_out
F="--steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-train --no-fast-mode --no-pmc --no-ragged"
python bench.py --precision fp32_bf16x3 $F > $O/r06_base_22k_emul.json 2> $O/r06_base.err; cp bench_detail.json $O/r06_base_22k_emul_detail.json
python bench.py --precision fp32_bf16x3 --vocoder 24k $F > $O/r06_base_24k_emul.json 2>> $O/r06_base.err; cp bench_detail.json $O/r06_base_24k_emul_detail.json
python bench.py --precision fp32 --vocoder 24k $F > $O/r06_base_24k_f32.json 2>> $O/r06_base.err; cp bench_detail.json $O/r06_base_24k_f32_detail.json
tail -c 600 $O/r06_base_22k_emul.json; tail -c 600 $O/r06_base_24k_emul.json; tail -c 600 $O/r06_base_24k_f32.json
