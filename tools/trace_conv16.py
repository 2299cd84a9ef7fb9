#!/usr/bin/env python3
"""Phase clocks of the emulated 16 x 16 x 32 conv (DIAG build with -DJATTS_CEMUL_TRACE=1: jatts_amd/lib_diagT, JATTS_HIP_LIB): per workgroup, the
shader clocks wave 0 spends in each phase of the chunk loop (conv1d_emul16.h: JATTS_PH).
    JATTS_HIP_LIB=$PWD/jatts_amd/lib_diagT/libjatts_hip.so python tools/trace_conv16.py --only 5 [--variant 2]"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from jatts_amd import _abi, hip  # noqa: E402
from tools.bench_conv import SHAPES  # noqa: E402

PH = ["prologue", "issue loads", "K-steps", "wait vmcnt(0)", "split + LDS", "barrier", "epilogue"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", type=int, default=5)
    ap.add_argument("--n", type=int, default=16384)
    ap.add_argument("--batch", type=int, default=64)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    c, n, k, T, res = SHAPES[a.only]
    g = torch.Generator().manual_seed(0)
    rb = hip.RaggedBatch([T] * a.batch, dev)
    rows = rb.total
    x = (torch.randn(rows, c, generator=g) * 0.5).to(dev)
    w = hip.pack_conv_weight_bf16x3_k32((torch.randn(n, c, k, generator=g) / (c * k) ** 0.5).to(dev), 64)
    b = torch.zeros(n, device=dev)
    r = torch.zeros(rows, n, device=dev) if res else None
    run = lambda: hip.conv1d(rb, x, w, c, n, k, dtype=hip.F32E, bias=b, resid=r, out=r, out_f32=True, w_layout=1)  # noqa: E731
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    lib = _abi.load()
    buf = torch.zeros(a.n * 16, dtype=torch.int64, device=dev)
    lib.jatts_debug_trace(C.c_void_p(buf.data_ptr()), a.n)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record()
    torch.cuda.synchronize()
    lib.jatts_debug_trace(None, 0)
    ms = e0.elapsed_time(e1)
    t = buf.cpu().numpy().reshape(a.n, 16)
    t = t[t[:, 9] > 0]
    if not len(t):
        sys.exit("no trace records: is JATTS_HIP_LIB a -DJATTS_CEMUL_TRACE=1 build?")
    hw = t[:, 0]
    xcc, hwid = hw >> 32, hw & 0xFFFFFFFF
    cu = (xcc << 8) | ((hwid >> 8) & 0xF) | (((hwid >> 13) & 0x7) << 4)
    ph = t[:, 1:8].astype(np.float64)
    tot = (t[:, 11] - t[:, 10]).astype(np.float64)
    rt = (t[:, 9] - t[:, 8]).astype(np.float64)
    print(f"{c} -> {n} k={k} rows={rows} variant {os.environ.get('JATTS_CONV_EMUL16_VARIANT', '0')}: launch {ms * 1e3:.1f} us = "
          f"{2.0 * c * n * k * rows / ms / 1e9:.1f} TFLOP/s (traced build), {len(t)} workgroups on {len(np.unique(cu))} CUs")
    print(f"  workgroup lifetime: mean {tot.mean():9.0f} clk, median {np.median(tot):9.0f}; shader clock {np.mean(tot / (rt * 10.0)):.3f} GHz (s_memtime / s_memrealtime)")
    for i, nme in enumerate(PH):
        print(f"  {nme:14s} mean {ph[:, i].mean():9.0f} clk ({100 * ph[:, i].mean() / tot.mean():5.1f} %)   median {np.median(ph[:, i]):9.0f}")
    n_steps = c // 32 * k
    print(f"  K-steps: {n_steps} per workgroup -> {ph[:, 2].mean() / n_steps:.0f} clk per step of the wave")
    # how many workgroups share a CU at a time: sum of lifetimes per CU / the CU's span
    st, en = t[:, 8].astype(np.float64), t[:, 9].astype(np.float64)
    occ = []
    for u in np.unique(cu):
        m = cu == u
        occ.append((en[m] - st[m]).sum() / (en[m].max() - st[m].min()))
    print(f"  workgroups resident per CU (lifetimes / span): mean {np.mean(occ):.2f}")


if __name__ == "__main__":
    main()
