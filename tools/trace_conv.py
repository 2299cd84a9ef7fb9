#!/usr/bin/env python3
"""Workgroup timeline of the register-streamed f32 conv (jatts_debug_trace): where does the launch spend its time?
    python tools/trace_conv.py --only 6 --variant 3 [--n 8192]
Per traced workgroup: start / main loop entered / main loop done / stored (s_memtime) + s_memrealtime at both ends."""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from jatts_amd import _abi, hip  # noqa: E402
from tools.bench_conv import SHAPES  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", type=int, default=6)
    ap.add_argument("--variant", type=int, default=3)
    ap.add_argument("--n", type=int, default=8192)
    ap.add_argument("--batch", type=int, default=64)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    c, n, k, T, res = SHAPES[a.only]
    g = torch.Generator().manual_seed(0)
    rb = hip.RaggedBatch([T] * a.batch, dev)
    rows = rb.total
    x = (torch.randn(rows, c, generator=g) * 0.5).to(dev)
    w = hip.pack_conv_weight((torch.randn(n, c, k, generator=g) / (c * k) ** 0.5).to(dev), hip.F32)
    b = torch.zeros(n, device=dev)
    r = torch.zeros(rows, n, device=dev) if res else None
    run = lambda: hip.conv1d(rb, x, w, c, n, k, dtype=hip.F32, bias=b, resid=r, out=r, out_f32=res, variant=a.variant)  # noqa: E731
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
    t = t[t[:, 4] > 0]
    hw = t[:, 0]
    xcc, hwid = hw >> 32, hw & 0xFFFFFFFF
    cu = (xcc << 8) | ((hwid >> 8) & 0xF) | (((hwid >> 13) & 0x7) << 4)
    st = t[:, 1:5].astype(np.float64)
    rt = t[:, 8:10].astype(np.float64)
    d = np.diff(st, axis=1)
    tot = st[:, -1] - st[:, 0]
    n_steps = c // 16 * k
    mfma = n_steps * 2048.0
    print(f"{c} -> {n} k={k} rows={rows} variant {a.variant}: launch {ms * 1e3:.1f} us = {2.0 * c * n * k * rows / ms / 1e9:.1f} TFLOP/s, traced {len(t)} workgroups on "
          f"{len(np.unique(cu))} CUs; MFMA issue per wave = {mfma:.0f} clk")
    ghz = (tot / ((rt[:, 1] - rt[:, 0]) * 10.0))
    print(f"  shader clock (s_memtime / s_memrealtime over a workgroup): mean {ghz.mean():.3f} GHz  min {ghz.min():.3f}  max {ghz.max():.3f}")
    print(f"  workgroup lifetime: median {np.median(tot):9.0f} clk  mean {tot.mean():9.0f}  min {tot.min():9.0f}  max {tot.max():9.0f}")
    for i, nme in enumerate(["prologue", "main loop", "epilogue"]):
        print(f"  {nme:10s} median {np.median(d[:, i]):9.0f} clk  mean {d[:, i].mean():9.0f}  min {d[:, i].min():9.0f}  max {d[:, i].max():9.0f}  ({100 * d[:, i].mean() / tot.mean():5.1f} %)")
    print(f"  main loop / MFMA issue time: median {np.median(d[:, 1]) / mfma:.2f}x  (1.0 = the pipe to itself, 2.0 = shared evenly with one other wave)")
    span = (rt[:, 1].max() - rt[:, 0].min()) * 10.0
    print(f"  traced span {span / 1e3:.1f} us (realtime)")
    for cu_id in np.unique(cu)[:2]:
        one = cu == cu_id
        ev = sorted([(r[0], 1) for r in rt[one]] + [(r[1], -1) for r in rt[one]])
        cur, last, hist = 0, ev[0][0], {}
        for tm, dl in ev:
            hist[cur] = hist.get(cur, 0.0) + (tm - last)
            cur, last = cur + dl, tm
        tt = sum(hist.values())
        print(f"  CU {int(cu_id):#x}: {int(one.sum())} traced workgroups; resident-workgroup histogram over its span: " +
              ", ".join(f"{kk}: {100 * v / tt:.0f} %" for kk, v in sorted(hist.items())))
        base = rt[one][:, 0].min()
        for r0, r1, s in sorted(zip(rt[one][:, 0], rt[one][:, 1], st[one].tolist()))[:10]:
            print(f"     start {(r0 - base) * 10:9.0f} ns  end {(r1 - base) * 10:9.0f} ns   phases (clk): " + " ".join(f"{v - s[0]:8.0f}" for v in s[1:]))


if __name__ == "__main__":
    main()
