# mid-round evidence: whole GPU suite, PMC traffic of every arithmetic, the driver's bench command
set -x
O=gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > $O/r06_t_all_mid.txt
bash tools/pmc_bench.sh r06_pmc_bench
python tools/pmc_traffic.py $O/r06_pmc_bench $O/r06_traffic.json > $O/r06_traffic.txt 2>&1
rm -rf $O/r06_pmc_bench
timeout 1500 python bench.py --steps 20 --warmup 5 > $O/r06_bench_mid.json 2> $O/r06_bench_mid.err
cp bench_detail.json $O/r06_bench_mid_detail.json
tail -n 6 $O/r06_t_all_mid.txt; tail -c 3000 $O/r06_bench_mid.json; tail -n 3 $O/r06_bench_mid.err
