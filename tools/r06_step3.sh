set -x
O=gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "two_outputs" 2>&1 | tail -5 > $O/r06_t_qkv.txt
timeout 900 python -m pytest tests/test_graph_gpu.py tests/test_fs2_gpu.py tests/test_matcha_gpu.py tests/test_vits_gpu.py -m gpu -x -q 2>&1 | tail -5 > $O/r06_t_models.txt
for v in 20 22; do JATTS_CONV_EMUL_VARIANT=$v timeout 900 python -m pytest tests/test_emul_gpu.py tests/test_kernels_gpu.py -m gpu -x -q -k "emul or conv1d" 2>&1 | tail -4 > $O/r06_t_il_v$v.txt; done
(for v in 0 20 22; do echo "=== variant $v"; JATTS_CONV_EMUL_VARIANT=$v python tools/bench_conv.py --dtype emul --iters 20 2>&1 | grep "emul"; done) > $O/r06_conv_emul_il.txt 2>&1
tail -3 $O/r06_t_qkv.txt $O/r06_t_models.txt $O/r06_t_il_v20.txt $O/r06_t_il_v22.txt; cat $O/r06_conv_emul_il.txt
