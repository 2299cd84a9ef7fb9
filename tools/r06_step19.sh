#!/bin/bash
O=gpurun_out
(for C in 32 64 128 256; do for K in 3 7 11; do python tools/trace_unit.py --C $C --k $K --dil 3 --dtype emul --layout 1 2>&1 | grep -v amdgpu.ids; done; done) 2>&1 | tee $O/r06_trace_units_split2.txt
