#!/usr/bin/env python3
"""Micro-benchmark of jatts_conv1d_wgrad (f32 MFMA, split-K over the sequences): python tools/bench_wgrad.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from jatts_amd import hip  # noqa: E402

SHAPES = [(384, 192, 5), (384, 192, 3), (384, 192, 1), (1536, 384, 3), (384, 1536, 3), (384, 384, 1), (512, 512, 3), (2048, 512, 1)]


def main():
    dev = torch.device("cuda:0")
    B, T = 32, 768
    rb = hip.RaggedBatch([T] * B, dev)
    for n, c, k in SHAPES:
        x = torch.randn(B * T, c, device=dev)
        dy = torch.randn(B * T, n, device=dev)
        for _ in range(3):
            hip.conv1d_wgrad(rb, x, dy, c, n, k, 1, (k - 1) // 2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            hip.conv1d_wgrad(rb, x, dy, c, n, k, 1, (k - 1) // 2)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        print(f"wgrad n={n:5d} c={c:5d} k={k}  {dt * 1e6:8.1f} us  {2.0 * B * T * n * c * k / dt / 1e12:6.1f} TFLOP/s")


if __name__ == "__main__":
    main()
