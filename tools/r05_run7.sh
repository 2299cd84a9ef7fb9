set -x
O=gpurun_out
hipcc --offload-arch=gfx950 -O2 -o /tmp/bf16_acc_probe tools/bf16_acc_probe.hip && /tmp/bf16_acc_probe > $O/r05_bf16_acc_probe.txt 2>&1
python bench.py --steps 20 --warmup 5 > $O/r05_bench_n1_pre.json 2> $O/r05_bench_n1_pre.err
cp bench_detail.json $O/r05_bench_detail_pre.json
bash tools/profile_models.sh r05 fp32_bf16x3
for k in matcha vits; do cp $(find $O/r05_infer_$k -name "*kernel_stats.csv" | head -1) $O/r05_infer_${k}_emul_kernel_stats.csv; rm -rf $O/r05_infer_$k; done
cat $O/r05_bf16_acc_probe.txt
