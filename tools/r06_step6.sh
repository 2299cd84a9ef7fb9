set -x
O=gpurun_out
(for v in 0 34 33; do echo "=== variant $v"; JATTS_CONV_EMUL_VARIANT=$v python tools/bench_conv.py --dtype emul --iters 20 2>&1 | grep "emul"; done) > $O/r06_conv_emul_il2.txt 2>&1
JATTS_CONV_EMUL_VARIANT=33 timeout 900 python -m pytest tests/test_emul_gpu.py tests/test_kernels_gpu.py -m gpu -x -q -k "emul or conv1d" 2>&1 | tail -4 > $O/r06_t_il_v33.txt
tail -n 3 $O/r06_t_il_v33.txt; grep -v "^+" $O/r06_conv_emul_il2.txt
