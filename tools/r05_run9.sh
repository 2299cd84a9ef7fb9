set -x
O=gpurun_out
python -m pytest tests/test_emul_gpu.py -q -k "resunit" 2>&1 | tail -4 > $O/r05_t_ksplit.txt
for v in 0 2; do echo "== C256 variant $v"; for d in 1 5; do JATTS_RESUNIT_EMUL_VARIANT=$v python tools/bench_unit.py --C 256 --k 11 --dil $d --dtype emul; JATTS_RESUNIT_EMUL_VARIANT=$v python tools/bench_unit.py --C 256 --k 11 --dil $d --dtype emul6; done; done > $O/r05_ksplit.txt 2>&1
for v in 0 4 5; do echo "== emul7 variant $v"; JATTS_CONV_EMUL_VARIANT=$v python tools/bench_conv.py --dtype emul --iters 20; done > $O/r05_conv_emul7b.txt 2>&1
(echo "== emul6"; python tools/bench_conv.py --dtype emul6 --iters 20) >> $O/r05_conv_emul7b.txt 2>&1
for v in 4 5; do JATTS_CONV_EMUL_VARIANT=$v python -m pytest tests/test_emul_gpu.py -q -k "conv1d_emul and 7" 2>&1 | tail -3; done > $O/r05_t_conv_variants.txt
cat $O/r05_t_ksplit.txt $O/r05_t_conv_variants.txt; grep -v amdgpu $O/r05_ksplit.txt
