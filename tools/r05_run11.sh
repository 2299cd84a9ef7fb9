set -x
O=gpurun_out
python -m pytest tests/test_kernels_gpu.py tests/test_matcha_gpu.py tests/test_benchsize_gpu.py -q -k "relpos_attention or matcha" 2>&1 | tail -4 > $O/r05_t_attn.txt
(python tools/bench_models.py --model matcha --precision fp32 --no-vocoder; python tools/bench_models.py --model matcha --precision fp32 --no-vocoder --shapes | grep -i -E "relattn|text2mel" | head -6) > $O/r05_attn_matcha.txt 2>&1
( time python bench.py > $O/r05_bench_default.json 2> $O/r05_bench_default.err ) 2> $O/r05_bench_default.time
cat $O/r05_t_attn.txt; tail -3 $O/r05_attn_matcha.txt; cat $O/r05_bench_default.time
