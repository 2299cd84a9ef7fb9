# Round-5 evidence run (1x MI355X): the whole GPU suite, the two 1 100-case error sweeps, PMC traffic, rocprofv3 kernel stats of every arithmetic's
# bench leg + the ragged leg + configs 3 / 5 in the emulated mode, then the driver's own bench command.  Everything lands in gpurun_out/r05_*.
set -x
O=gpurun_out
python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -20 > $O/r05_t_all.txt
python tools/emul_sweep.py --products 7 --units 400 --convs 700 --out $O/r05_emul_sweep.json > $O/r05_emul_sweep.txt 2>&1
python tools/emul_sweep.py --products 6 --units 400 --convs 700 --out $O/r05_emul6_sweep.json > $O/r05_emul6_sweep.txt 2>&1
bash tools/pmc_bench.sh r05_pmc_bench
python tools/pmc_traffic.py $O/r05_pmc_bench $O/r05_traffic.json > $O/r05_traffic.txt 2>&1
rm -rf $O/r05_pmc_bench
bash tools/profile_bench.sh r05_prof_f32
bash tools/profile_bench.sh r05_prof_ragged --only-ragged
bash tools/profile_bench.sh r05_prof_emul --precision fp32_bf16x3
bash tools/profile_bench.sh r05_prof_emul6 --precision fp32_bf16x3_6p
bash tools/profile_bench.sh r05_prof_split --precision fp32_split
for t in f32 ragged emul emul6 split; do cp $(find $O/r05_prof_$t -name "*kernel_stats.csv" | head -1) $O/r05_bench_${t}_kernel_stats.csv; done
rm -rf $O/r05_prof_f32 $O/r05_prof_ragged $O/r05_prof_emul $O/r05_prof_emul6 $O/r05_prof_split
bash tools/profile_models.sh r05 fp32_bf16x3
for k in matcha vits; do cp $(find $O/r05_infer_$k -name "*kernel_stats.csv" | head -1) $O/r05_infer_${k}_emul_kernel_stats.csv; rm -rf $O/r05_infer_$k; done
(echo "== f32"; python tools/bench_unit.py --all --dtype f32; echo "== emul (7 products)"; python tools/bench_unit.py --all --dtype emul; echo "== emul6"; python tools/bench_unit.py --all --dtype emul6; echo "== split"; python tools/bench_unit.py --all --dtype split) > $O/r05_units_by_shape.txt 2>&1
python bench.py --steps 20 --warmup 5 > $O/r05_bench_n1.json 2> $O/r05_bench_n1.err
cp bench_detail.json $O/r05_bench_detail.json
tail -n 3 $O/r05_t_all.txt; tail -c 400 $O/r05_bench_n1.json
