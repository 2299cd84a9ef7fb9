#!/bin/bash
O=gpurun_out
python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -20 > $O/r06_gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/r06_smoke.txt 2>&1
tail -n 3 $O/r06_gpu_tests.txt; tail -n 2 $O/r06_smoke.txt
