set -x
O=gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "two_outputs" 2>&1 | tail -8 > $O/r06_t_qkv.txt
timeout 900 python -m pytest tests/test_graph_gpu.py tests/test_fs2_gpu.py tests/test_matcha_gpu.py tests/test_vits_gpu.py tests/test_spk_concat_gpu.py -m gpu -q 2>&1 | tail -8 > $O/r06_t_models.txt
(for v in 0 4 8 12 16; do echo "=== stagger $v"; JATTS_CONV_EMUL_STAGGER=$v python tools/bench_conv.py --dtype emul --iters 20 2>&1 | grep "k=1"; done) > $O/r06_conv_emul_stagger.txt 2>&1
bash tools/profile_b1.sh r06_b1_prof
cp $(find $O/r06_b1_prof -name "*kernel_stats.csv" | head -1) $O/r06_b1_kernel_stats_v2.csv; rm -rf $O/r06_b1_prof
python bench.py --no-cpu-baseline --no-configs --no-train --no-pmc --no-24k --no-ragged --no-fast-mode --steps 6 --warmup 2 > $O/r06_bench_quick.json 2>$O/r06_bench_quick.err
for f in $O/r06_t_qkv.txt $O/r06_t_models.txt; do tail -n 4 $f; done; cat $O/r06_conv_emul_stagger.txt; head -12 $O/r06_b1_kernel_stats_v2.csv; tail -c 1500 $O/r06_bench_quick.json
