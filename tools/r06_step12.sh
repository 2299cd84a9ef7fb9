O=gpurun_out
timeout 900 python -m pytest tests/test_emul_gpu.py -m gpu -q -x -k "resunit" 2>&1 | tail -3
python tools/bench_unit.py --all --dtype emul --layout 1 2>&1 | grep "C= 256\|C= 128"
python tools/trace_unit.py --C 128 --k 7 --dtype emul --layout 1 2>&1 | grep -v amdgpu.ids | head -9
