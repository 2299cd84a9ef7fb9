#!/usr/bin/env python3
"""Phase timeline of the fused dilation-unit kernel (jatts_debug_trace): where does a workgroup spend its cycles?
    python tools/trace_unit.py --C 128 --k 11 [--dil 1] [--n 4096]
Prints the median / mean cycles of each phase over the traced workgroups (shader clock, s_memtime), the
per-CU concurrency and a short per-CU timeline."""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from jatts_amd import _abi, hip  # noqa: E402

PH = ["stage x", "conv1 (MFMA)", "h -> LDS", "conv2 (MFMA)", "y -> LDS", "resid + store"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--C", type=int, default=128)
    ap.add_argument("--k", type=int, default=11)
    ap.add_argument("--dil", type=int, default=1)
    ap.add_argument("--n", type=int, default=8192)
    ap.add_argument("--dtype", default="f16", choices=["f16", "f32", "split", "emul", "emul6"])
    ap.add_argument("--layout", type=int, default=1, choices=[0, 1], help="emulated units: 1 = v_mfma_f32_16x16x32_bf16 kernels (product), 0 = 32x32x16")
    a = ap.parse_args()
    rates = {256: 8, 128: 64, 64: 128, 32: 256}
    dev = torch.device("cuda:0")
    B, T, rate = 64, 768, rates[a.C]
    rb = hip.RaggedBatch([T] * B, dev)
    rows = B * T * rate
    g = torch.Generator(device="cpu").manual_seed(0)
    dt = {"f16": hip.F16, "f32": hip.F32, "split": hip.F32S, "emul": hip.F32E, "emul6": hip.F32E6}[a.dtype]
    x = (torch.randn(rows, a.C, generator=g) * 0.5).to(dev).to(hip.torch_dtype(dt))
    y = torch.empty_like(x)
    wf = [(torch.randn(a.C, a.C, a.k, generator=g) / (a.C * a.k) ** 0.5).to(dev) for _ in range(2)]
    b = torch.zeros(a.C, device=dev)
    if dt == hip.F32S:
        (w0, i0), (w1, i1) = (hip.pack_conv_weight_split(v, 32) for v in wf)
        run = lambda: hip.hifigan_resunit(rb, rate, x, y, w0, b, w1, b, a.C, a.k, a.dil, 0.1, dt, ws=(i0, i1))
    elif dt in hip.EMUL:
        w = [hip.pack_unit_weight_bf16x3_k32(v) if a.layout else hip.pack_conv_weight_bf16x3(v, 32) for v in wf]
        run = lambda: hip.hifigan_resunit(rb, rate, x, y, w[0], b, w[1], b, a.C, a.k, a.dil, 0.1, dt, w_layout=a.layout)
    else:
        w = [hip.pack_conv_weight(v, dt, 32) for v in wf]
        run = lambda: hip.hifigan_resunit(rb, rate, x, y, w[0], b, w[1], b, a.C, a.k, a.dil, 0.1, dt)
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
    t = t[t[:, 7] > 0]
    rt = t[:, 8:10].astype(np.float64)
    t16 = t
    t = t[:, :8]
    hw = t[:, 0]
    xcc, hwid = hw >> 32, hw & 0xFFFFFFFF
    cu = (xcc << 8) | ((hwid >> 8) & 0xF) | (((hwid >> 13) & 0x7) << 4)   # (xcc, se, cu) -- gfx9 HW_ID layout
    st = t[:, 1:].astype(np.float64)
    d = np.diff(st, axis=1)
    tot = st[:, -1] - st[:, 0]
    print(f"C={a.C} k={a.k} d={a.dil}: launch {ms:.3f} ms, traced {len(t)} workgroups, {len(np.unique(cu))} distinct CUs")
    print(f"  workgroup lifetime: median {np.median(tot):9.0f} clk  mean {tot.mean():9.0f}")
    for i, nme in enumerate(PH):
        print(f"  {nme:14s} median {np.median(d[:, i]):9.0f} clk  mean {d[:, i].mean():9.0f}  ({100 * d[:, i].mean() / tot.mean():5.1f} %)")
    ex = t16[:, 10:13].astype(np.float64)
    if (ex > 0).all():
        print(f"  inside 'h -> LDS': barrier wait {np.mean(ex[:, 0] - st[:, 2]):8.0f}  zero-fill {np.mean(ex[:, 1] - ex[:, 0]):8.0f}  "
              f"bias+lrelu+LDS writes {np.mean(ex[:, 2] - ex[:, 1]):8.0f}  closing barrier {np.mean(st[:, 3] - ex[:, 2]):8.0f} clk (mean)")
    ghz = (tot / ((rt[:, 1] - rt[:, 0]) * 10.0)).mean()   # s_memrealtime ticks are 10 ns
    print(f"  s_memtime runs at {ghz:.3f} ticks/ns (vs the 100 MHz s_memrealtime)")
    # concurrency per CU: how many traced workgroups overlap in time on one CU
    one = cu == cu[0]
    ev = sorted([(r[0], 1) for r in st[one]] + [(r[-1], -1) for r in st[one]])
    cur = mx = 0
    for _, dlt in ev:
        cur += dlt
        mx = max(mx, cur)
    print(f"  CU {int(cu[0]):#x}: {int(one.sum())} traced workgroups, up to {mx} resident at once")
    base = st[one][:, 0].min()
    for r in sorted(st[one].tolist())[:8]:
        print("   ", " ".join(f"{v - base:9.0f}" for v in r))


if __name__ == "__main__":
    main()
