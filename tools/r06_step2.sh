set -x
O=gpurun_out
bash tools/profile_b1.sh r06_b1_prof
cp $(find $O/r06_b1_prof -name "*kernel_stats.csv" | head -1) $O/r06_b1_kernel_stats.csv; rm -rf $O/r06_b1_prof
(for v in 0 8 9 10 11 12; do echo "=== variant $v"; JATTS_CONV_EMUL_VARIANT=$v python tools/bench_conv.py --dtype emul --iters 20 2>&1 | grep "k=1"; done) > $O/r06_conv_emul_k1_variants.txt 2>&1
(for v in 0 4; do echo "=== resunit variant $v"; JATTS_RESUNIT_EMUL_VARIANT=$v python tools/bench_unit.py --all --dtype emul 2>&1 | grep "C= 128"; done) > $O/r06_resunit_c128_variants.txt 2>&1
head -40 $O/r06_b1_kernel_stats.csv; cat $O/r06_conv_emul_k1_variants.txt; cat $O/r06_resunit_c128_variants.txt
