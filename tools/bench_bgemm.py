#!/usr/bin/env python3
"""jatts_bgemm against torch.matmul (rocBLAS) on the attention products of the FastSpeech2 training step (batch 32 x 2 heads, T = 768, d_k = 192).
    python tools/bench_bgemm.py [--T 770]      (T % 4 != 0: the T x T operand takes the element-load path)"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from jatts_amd import hip  # noqa: E402


def t(fn, it=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it


def main():
    dev = torch.device("cuda:0")
    ap = argparse.ArgumentParser()
    ap.add_argument("--T", type=int, default=768)
    B, H, T, dk = 32, 2, ap.parse_args().T, 192
    g = torch.Generator().manual_seed(0)
    q = torch.randn(B, H, T, dk, generator=g).to(dev)
    k = torch.randn(B, H, T, dk, generator=g).to(dev)
    p = torch.randn(B, H, T, T, generator=g).to(dev)
    cases = [("q k^T   (T x T x d_k, NT)", lambda: hip.bgemm(q, k, trans_b=True), lambda: torch.matmul(q, k.transpose(-1, -2)), 2.0 * B * H * T * T * dk),
             ("P v     (T x d_k x T, NN)", lambda: hip.bgemm(p, k), lambda: torch.matmul(p, k), 2.0 * B * H * T * T * dk),
             ("dS^T q  (T x d_k x T, TN)", lambda: hip.bgemm(p, q, trans_a=True), lambda: torch.matmul(p.transpose(-1, -2), q), 2.0 * B * H * T * T * dk),
             ("dO v^T  (T x T x d_k, NT)", lambda: hip.bgemm(q, k, trans_b=True), lambda: torch.matmul(q, k.transpose(-1, -2)), 2.0 * B * H * T * T * dk)]
    for name, f1, f2, fl in cases:
        a, b = t(f1), t(f2)
        print(f"{name:40s} jatts_bgemm {a * 1e3:7.1f} us {fl / a / 1e9:6.1f} TFLOP/s   torch.matmul {b * 1e3:7.1f} us {fl / b / 1e9:6.1f} TFLOP/s")


if __name__ == "__main__":
    main()
