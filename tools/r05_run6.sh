set -x
O=gpurun_out
python -m pytest tests -m gpu -q --durations=10 2>&1 | tail -30 > $O/r05_t_all.txt
python tools/emul_sweep.py --products 7 --units 400 --convs 700 --out $O/r05_emul_sweep.json > $O/r05_emul_sweep.txt 2>&1
python tools/emul_sweep.py --products 6 --units 400 --convs 700 --out $O/r05_emul6_sweep.json > $O/r05_emul6_sweep.txt 2>&1
bash tools/pmc_bench.sh r05_pmc_bench
python tools/pmc_traffic.py $O/r05_pmc_bench $O/r05_traffic.json > $O/r05_traffic.txt 2>&1
bash tools/profile_bench.sh r05_prof_f32
bash tools/profile_bench.sh r05_prof_ragged --only-ragged
bash tools/profile_bench.sh r05_prof_emul --precision fp32_bf16x3
bash tools/profile_bench.sh r05_prof_emul6 --precision fp32_bf16x3_6p
bash tools/profile_bench.sh r05_prof_split --precision fp32_split
for t in f32 ragged emul emul6 split; do cp $(find $O/r05_prof_$t -name "*kernel_stats.csv" | head -1) $O/r05_bench_${t}_kernel_stats.csv; done
rm -rf $O/r05_prof_f32 $O/r05_prof_ragged $O/r05_prof_emul $O/r05_prof_emul6 $O/r05_prof_split
find $O/r05_pmc_bench -name "*.db" -delete; du -sh $O/r05_pmc_bench
tail -n 5 $O/r05_t_all.txt
