# The headline legs' rocprofv3 kernel stats and the driver's bench command on ONE box (r06_final.sh's first profile pass mixed the 24 kHz / B = 1 legs in)
set -x
O=gpurun_out
bash tools/profile_bench.sh r06_prof_emul --precision fp32_bf16x3
bash tools/profile_bench.sh r06_prof_f32 --precision fp32
cp $(find $O/r06_prof_emul -name "*kernel_stats.csv" | head -1) $O/r06_bench_kernel_stats.csv
cp $(find $O/r06_prof_f32 -name "*kernel_stats.csv" | head -1) $O/r06_bench_f32_kernel_stats.csv
rm -rf $O/r06_prof_emul $O/r06_prof_f32
python bench.py --steps 20 --warmup 5 > $O/r06_bench_n1.json 2> $O/r06_bench_n1.err
cp bench_detail.json $O/r06_bench_detail.json
head -3 $O/r06_bench_kernel_stats.csv | cut -c1-200; tail -c 300 $O/r06_bench_n1.json
