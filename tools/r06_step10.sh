O=gpurun_out
timeout 1500 python -m pytest tests/test_hifigan_gpu.py tests/test_emul_gpu.py tests/test_graph_gpu.py -m gpu -q -x 2>&1 | tail -4 > $O/r06_t_emul16_models.txt
tail -n 4 $O/r06_t_emul16_models.txt
timeout 1500 python -m pytest tests/test_benchsize_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x -k "hifigan or full_batch" 2>&1 | tail -4 > $O/r06_t_emul16_bench.txt
tail -n 4 $O/r06_t_emul16_bench.txt
python bench.py --no-cpu-baseline --no-configs --no-train --no-pmc --no-ragged --no-fast-mode --no-b1 --steps 10 --warmup 3 > $O/r06_bench_quick.json 2>$O/r06_bench_quick.err
tail -c 1500 $O/r06_bench_quick.json; tail -n 3 $O/r06_bench_quick.err
