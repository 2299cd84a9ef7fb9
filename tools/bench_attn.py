#!/usr/bin/env python3
"""Attention launches alone, at the shapes the models use (HIP events around jatts_relpos_attention; random operands).
    python tools/bench_attn.py [--dtype f32|f16|split] [--reps 20]
Shapes: (label, utterances, heads, d_k, T, rel-pos bias).  TFLOP/s counts 4 * T^2 * d_k per (utterance, head)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jatts_amd import _abi, hip  # noqa: E402

SHAPES = [
    ("fs2 encoder", 64, 2, 192, 128, True),
    ("fs2 decoder", 64, 2, 192, 768, True),
    ("fs2 decoder, no bias", 64, 2, 192, 768, False),
    ("matcha decoder T", 64, 2, 256, 768, False),
    ("matcha decoder T/2", 64, 2, 256, 384, False),
    ("matcha mid T/4", 64, 2, 256, 192, False),
    ("vits encoder", 64, 2, 96, 128, True),
    ("d_k 64", 64, 4, 64, 768, True),
    ("d_k 128", 64, 2, 128, 768, True),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f32", choices=["f32", "f16", "split"])
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--data", default="randn", choices=["randn", "zeros", "small"],
                    help="operand values: N(0, 1); all zero; N(0, 1) rounded to 8 mantissa bits (the MFMA clock follows the operand bits)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    code = {"f32": _abi.F32, "f16": _abi.F16, "split": _abi.F32S}[a.dtype]
    tdt = torch.float16 if a.dtype == "f16" else torch.float32
    g0 = torch.Generator(device="cpu").manual_seed(0)

    def rnd(*shape):
        t = torch.randn(*shape, generator=g0)
        if a.data == "zeros":
            t.zero_()
        elif a.data == "small":
            t = t.to(torch.bfloat16).to(torch.float32)
        return t

    for label, B, H, dk, T, rel in SHAPES:
        rb = hip.RaggedBatch([T] * B, dev)
        D = H * dk
        q = rnd(rb.total, D).to(dev, tdt)
        k = rnd(rb.total, D).to(dev, tdt)
        col0, ldvt = rb.vt_layout()
        vt = rnd(D, ldvt).to(dev, tdt)
        g = ku = None
        ldg = 0
        if rel:
            ldg = hip.round_up(T, 8)
            g = rnd(rb.total * H, ldg).to(dev, tdt)
            ku = rnd(rb.total, H).to(dev, torch.float32)

        def run():
            return hip.relpos_attention(rb, q, D, k, D, vt, ldvt, g, ldg, ku, dk ** -0.5, H, dk, code, vt_col0=col0)

        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.reps
        tf = 4.0 * T * T * dk * B * H / (us * 1e-6) / 1e12
        print(f"{label:20s} B={B} H={H} d_k={dk:3d} T={T:4d} rel={int(rel)}  {us:8.1f} us  {tf:6.1f} TFLOP/s")


if __name__ == "__main__":
    main()
