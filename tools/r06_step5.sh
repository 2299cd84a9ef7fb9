set -x
O=gpurun_out
timeout 1200 python -m pytest tests/test_emul_gpu.py tests/test_hifigan_gpu.py tests/test_kernels_gpu.py -m gpu -q -x -k "resunit or hifigan or emul or unit" 2>&1 | tail -6 > $O/r06_t_rreg.txt
(for v in 9 0; do echo "=== resunit variant $v"; JATTS_RESUNIT_EMUL_VARIANT=$v python tools/bench_unit.py --all --dtype emul 2>&1 | grep "C=  64\|C=  32"; done) > $O/r06_resunit_rreg.txt 2>&1
python bench.py --no-cpu-baseline --no-configs --no-train --no-pmc --no-24k --no-ragged --no-fast-mode --no-b1 --steps 8 --warmup 2 > $O/r06_bench_quick.json 2>$O/r06_bench_quick.err
tail -n 5 $O/r06_t_rreg.txt; cat $O/r06_resunit_rreg.txt | grep -v "^+"; tail -c 900 $O/r06_bench_quick.json
