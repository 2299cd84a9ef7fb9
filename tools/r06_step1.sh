# Round-6 first evidence run: the new tests (graph replay, 24 kHz at bench size), PMC traffic of every arithmetic, the new default bench line.
set -x
O=gpurun_out
timeout 900 python -m pytest tests/test_graph_gpu.py -m gpu -x -q 2>&1 | tail -15 > $O/r06_t_graph.txt
timeout 1500 python -m pytest tests/test_benchsize_gpu.py tests/test_fullsize_gpu.py -m gpu -q -k "hifigan or full_batch" --durations=5 2>&1 | tail -20 > $O/r06_t_24k.txt
bash tools/pmc_bench.sh r06_pmc_bench
python tools/pmc_traffic.py $O/r06_pmc_bench $O/r06_traffic.json > $O/r06_traffic.txt 2>&1
rm -rf $O/r06_pmc_bench
timeout 1200 python bench.py --steps 10 --warmup 3 > $O/r06_bench_first.json 2> $O/r06_bench_first.err
cp bench_detail.json $O/r06_bench_first_detail.json
tail -5 $O/r06_t_graph.txt; tail -5 $O/r06_t_24k.txt; tail -c 3000 $O/r06_bench_first.json; tail -5 $O/r06_bench_first.err
