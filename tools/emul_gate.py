#!/usr/bin/env python3
"""Gate for the f32-equivalent emulated arithmetic (JATTS_F32E): correctness of the fused unit against fp64 next to
the exact-f32 kernel on the same inputs, including single-non-zero contractions (K_eff = 1), then the C = 128 timing that decides go / no-go.
    python tools/emul_gate.py            (prints one line per case; exit 1 if a ratio exceeds 2)"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jatts_amd import hip  # noqa: E402
from tools.split_sweep import KINDS, draw_x, errs, ref_unit  # noqa: E402


def one(C, k, d, lens, kind, g, dev, single=False):
    x = draw_x(g, sum(lens), C, kind)
    w1 = torch.randn(C, C, k, generator=g) / math.sqrt(C * k) * torch.pow(10.0, torch.rand(C, 1, 1, generator=g) * 2 - 1)
    w2 = torch.randn(C, C, k, generator=g) / math.sqrt(C * k)
    if single:       # one non-zero weight per output channel in both convs: every dot product has ONE term
        for w in (w1, w2):
            m = torch.zeros_like(w).view(C, -1)
            m[torch.arange(C), torch.randint(0, C * k, (C,), generator=g)] = 1
            w.mul_(m.view_as(w))
    sc = float(x.abs().max().clamp_min(1e-30))
    b1, b2 = (torch.zeros(C), torch.zeros(C)) if single else (torch.randn(C, generator=g) * 0.05 * min(sc, 1e3), torch.randn(C, generator=g) * 0.05 * min(sc, 1e3))
    ref = ref_unit(x, w1, b1, w2, b2, lens, k, d, 0.1)
    if single:       # the residual add rounds too: compare the conv branch alone
        ref = ref - x.double()
    rb = hip.RaggedBatch(lens, dev)
    xd = x.to(dev)
    y, y32 = torch.empty_like(xd), torch.empty_like(xd)
    hip.hifigan_resunit(rb, 1, xd, y, hip.pack_conv_weight_bf16x3(w1.to(dev), 32), b1.to(dev), hip.pack_conv_weight_bf16x3(w2.to(dev), 32), b2.to(dev),
                        C, k, d, 0.1, hip.F32E)
    hip.hifigan_resunit(rb, 1, xd, y32, hip.pack_conv_weight(w1.to(dev), hip.F32, 32), b1.to(dev), hip.pack_conv_weight(w2.to(dev), hip.F32, 32), b2.to(dev),
                        C, k, d, 0.1, hip.F32)
    if single:
        y, y32 = y.double() - xd.double(), y32.double() - xd.double()     # exact in fp64 only up to the final f32 rounding of y; both paths share it
    (m, e), (m32, e32) = errs(y, ref), errs(y32, ref)
    return m, m32, e, e32


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    worst = 0.0
    for C in (32, 64, 128, 256):
        for k, d in ((3, 1), (7, 3), (11, 5)):
            for kind in KINDS:
                lens = [int(v) for v in torch.randint(1, 500 if C <= 64 else 260, (2,), generator=g)]
                for single in (False, True):
                    m, m32, e, e32 = one(C, k, d, lens, kind, g, dev, single)
                    r = m / max(m32, 1e-300)
                    worst = max(worst, r)
                    flag = "  <-- > 2" if r > 2 else ""
                    print(f"C{C:3d} k{k:2d} d{d} {kind:8s} {'single' if single else 'dense ':6s} max emul {m:.3e} f32 {m32:.3e} ratio {r:5.2f}   rel emul {e:.2e} f32 {e32:.2e}{flag}")
    print(f"worst max-error ratio emul / exact f32: {worst:.3f}")
    return 1 if worst > 2 else 0


if __name__ == "__main__":
    sys.exit(main())
