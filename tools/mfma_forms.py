#!/usr/bin/env python3
"""The two bf16 MFMA forms under the power limit (jatts_mfma_probe): 2 x 2 fragments of 32 x 32 x 16 against 4 x 4 fragments of 16 x 16 x 32 per wave -- the same
operand bytes per flop, half the accumulator elements per flop in the 16 x 16 x 32 form -- fed from LDS and from registers, on N(0, 1) operand bits.
    python tools/mfma_forms.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jatts_amd import hip  # noqa: E402

for rep in range(2):
    for name, code in (("32x32x16 bf16", hip.F32E), ("16x16x32 bf16", 16 + hip.F32E), ("32x32x16 f16", hip.F16), ("16x16x32 f16", 16 + hip.F16), ("32x32x2 f32", hip.F32)):
        for feed in (1, 0):
            r = hip.mfma_ceiling(code, feed, target_ms=80.0)
            print(f"{name:15s} {'LDS-fed  ' if feed else 'registers'}  {r['tflops']:8.1f} TFLOP/s  {r['clock_ghz']:.3f} GHz  ({r['ms']:.1f} ms)")
