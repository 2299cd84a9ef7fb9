set -x
O=gpurun_out
python -m pytest tests/test_emul_gpu.py -q 2>&1 | tail -25 > $O/r05_t_emul.txt
for v in 0 1 2; do echo "== resblock emul variant $v"; JATTS_RESBLOCK_EMUL_VARIANT=$v python tools/bench_unit.py --resblock --dtype emul; done > $O/r05_resblock_emul.txt 2>&1
(echo "== emul6 v0"; python tools/bench_unit.py --resblock --dtype emul6) >> $O/r05_resblock_emul.txt 2>&1
python -m pytest tests -m gpu -q -x --durations=15 2>&1 | tail -40 > $O/r05_t_all.txt
tail -n 4 $O/r05_t_emul.txt $O/r05_t_all.txt
