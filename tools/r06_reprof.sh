O=gpurun_out
bash tools/profile_bench.sh r06_prof_emul --precision fp32_bf16x3
bash tools/profile_bench.sh r06_prof_f32 --precision fp32
cp $(find $O/r06_prof_emul -name "*kernel_stats.csv" | head -1) $O/r06_bench_kernel_stats.csv
cp $(find $O/r06_prof_f32 -name "*kernel_stats.csv" | head -1) $O/r06_bench_f32_kernel_stats.csv
tail -c 600 $O/r06_prof_emul.log
rm -rf $O/r06_prof_emul $O/r06_prof_f32
head -8 $O/r06_bench_kernel_stats.csv
