# Round-6 evidence run (1x MI355X): the whole GPU suite, the 1 100-case error sweeps of the product kernels (16 x 16 x 32 form), PMC traffic, rocprofv3 kernel
# stats of the headline / exact-f32 / 24 kHz / ragged / B = 1 legs and of configs 3 / 5, units by shape in both MFMA forms, the MFMA-form probe, then the
# driver's own bench command.  Everything lands in gpurun_out/r06_*.
set -x
O=gpurun_out
python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -20 > $O/r06_gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/r06_smoke.txt 2>&1
python tools/emul_sweep.py --products 7 --units 400 --convs 700 --out $O/r06_emul_sweep.json > $O/r06_emul_sweep.txt 2>&1
python tools/emul_sweep.py --products 6 --units 400 --convs 700 --out $O/r06_emul6_sweep.json > $O/r06_emul6_sweep.txt 2>&1
python tools/mfma_forms.py > $O/r06_mfma_forms.txt 2>&1
bash tools/pmc_bench.sh r06_pmc_bench
python tools/pmc_traffic.py $O/r06_pmc_bench $O/r06_traffic.json > $O/r06_traffic.txt 2>&1
rm -rf $O/r06_pmc_bench
bash tools/profile_bench.sh r06_prof_emul --precision fp32_bf16x3
bash tools/profile_bench.sh r06_prof_f32 --precision fp32
bash tools/profile_bench.sh r06_prof_ragged --only-ragged --precision fp32_bf16x3
bash tools/profile_bench.sh r06_prof_24k --only-24k --precision fp32_bf16x3
bash tools/profile_bench.sh r06_prof_24k_f32 --only-24k --precision fp32
cp $(find $O/r06_prof_emul -name "*kernel_stats.csv" | head -1) $O/r06_bench_kernel_stats.csv
cp $(find $O/r06_prof_f32 -name "*kernel_stats.csv" | head -1) $O/r06_bench_f32_kernel_stats.csv
cp $(find $O/r06_prof_ragged -name "*kernel_stats.csv" | head -1) $O/r06_bench_ragged_kernel_stats.csv
cp $(find $O/r06_prof_24k -name "*kernel_stats.csv" | head -1) $O/r06_bench_24k_kernel_stats.csv
cp $(find $O/r06_prof_24k_f32 -name "*kernel_stats.csv" | head -1) $O/r06_bench_24k_f32_kernel_stats.csv
rm -rf $O/r06_prof_emul $O/r06_prof_f32 $O/r06_prof_ragged $O/r06_prof_24k $O/r06_prof_24k_f32
bash tools/profile_b1.sh r06_b1_prof --precision fp32_bf16x3
cp $(find $O/r06_b1_prof -name "*kernel_stats.csv" | head -1) $O/r06_b1_kernel_stats.csv; rm -rf $O/r06_b1_prof
bash tools/profile_models.sh r06 fp32_bf16x3
for k in matcha vits; do cp $(find $O/r06_infer_$k -name "*kernel_stats.csv" | head -1) $O/r06_infer_${k}_emul_kernel_stats.csv; rm -rf $O/r06_infer_$k; done
(echo "== emul (7 products), v_mfma_f32_16x16x32_bf16 kernels"; python tools/bench_unit.py --all --dtype emul --layout 1; echo "== emul (7 products), 32x32x16 kernels"; python tools/bench_unit.py --all --dtype emul --layout 0; echo "== f32"; python tools/bench_unit.py --all --dtype f32) > $O/r06_units_by_shape.txt 2>&1
(echo "== 16x16x32"; python tools/bench_conv.py --dtype emul --iters 20 --layout 1; echo "== 32x32x16"; python tools/bench_conv.py --dtype emul --iters 20 --layout 0) > $O/r06_conv_by_shape.txt 2>&1
python bench.py --steps 20 --warmup 5 > $O/r06_bench_n1.json 2> $O/r06_bench_n1.err
cp bench_detail.json $O/r06_bench_detail.json
tail -n 3 $O/r06_gpu_tests.txt; tail -n 3 $O/r06_smoke.txt; tail -c 400 $O/r06_bench_n1.json
