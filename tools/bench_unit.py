#!/usr/bin/env python3
"""Micro-benchmark of the fused HiFi-GAN dilation-unit kernel at bench-sized shapes.
    python tools/bench_unit.py [--C 128 --k 11 --dil 1 --iters 10] [--all]
Prints avg ms, TFLOP/s and algorithmic GB/s per shape (HIP events on the launch stream)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from jatts_amd import hip  # noqa: E402


LAYOUT = [1]      # emulated units: 1 = the v_mfma_f32_16x16x32_bf16 kernels (the product form), 0 = the 32 x 32 x 16 ones (--layout 0)


def run(C, k, d, rate, iters, B=64, T=768, dtype=hip.F16):
    dev = torch.device("cuda:0")
    rb = hip.RaggedBatch([T] * B, dev)
    rows = B * T * rate
    tdt = hip.torch_dtype(dtype)
    g = torch.Generator(device="cpu").manual_seed(0)
    x = (torch.randn(rows, C, generator=g) * 0.5).to(dev).to(tdt)
    y = torch.empty_like(x)
    wa, wb = (torch.randn(C, C, k, generator=g) / (C * k) ** 0.5).to(dev), (torch.randn(C, C, k, generator=g) / (C * k) ** 0.5).to(dev)
    kw = {}
    if dtype == hip.F32S:
        (w1, i1), (w2, i2) = hip.pack_conv_weight_split(wa, 32), hip.pack_conv_weight_split(wb, 32)
        kw["ws"] = (i1, i2)
    elif dtype in hip.EMUL:
        if LAYOUT[0]:
            w1, w2 = hip.pack_unit_weight_bf16x3_k32(wa), hip.pack_unit_weight_bf16x3_k32(wb)
            kw["w_layout"] = 1
        else:
            w1, w2 = hip.pack_conv_weight_bf16x3(wa, 32), hip.pack_conv_weight_bf16x3(wb, 32)
    else:
        w1, w2 = hip.pack_conv_weight(wa, dtype, 32), hip.pack_conv_weight(wb, dtype, 32)
    b1 = torch.zeros(C, device=dev)
    b2 = torch.zeros(C, device=dev)
    for _ in range(2):
        hip.hifigan_resunit(rb, rate, x, y, w1, b1, w2, b2, C, k, d, 0.1, dtype, **kw)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        hip.hifigan_resunit(rb, rate, x, y, w1, b1, w2, b2, C, k, d, 0.1, dtype, **kw)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / iters
    flops = 4.0 * C * C * k * rows
    byts = 2.0 * rows * C * x.element_size()
    print(f"C={C:4d} k={k:2d} d={d} rows={rows:9d}  {ms:7.3f} ms  {flops / ms / 1e9:7.1f} TFLOP/s  {byts / ms / 1e6:7.1f} GB/s")
    return ms


def run_block(C, k, rate, iters, B=64, T=768, dils=(1, 3, 5), dt=hip.F16):
    """Fused ResBlock launch against the three per-unit launches."""
    dev = torch.device("cuda:0")
    rb = hip.RaggedBatch([T] * B, dev)
    rows = B * T * rate
    g = torch.Generator(device="cpu").manual_seed(0)
    x = (torch.randn(rows, C, generator=g) * 0.5).to(dev).to(hip.torch_dtype(dt))
    bufs = [torch.empty_like(x), torch.empty_like(x)]
    units = []
    invs = []
    for d in dils:
        wa, wb = (torch.randn(C, C, k, generator=g) / (C * k) ** 0.5).to(dev), (torch.randn(C, C, k, generator=g) * 0.3 / (C * k) ** 0.5).to(dev)
        if dt == hip.F32S:
            (w1, i1), (w2, i2) = hip.pack_conv_weight_split(wa, 32), hip.pack_conv_weight_split(wb, 32)
            invs.append((i1, i2))
        elif dt in hip.EMUL:
            w1, w2 = hip.pack_conv_weight_bf16x3(wa, 32), hip.pack_conv_weight_bf16x3(wb, 32)
        else:
            w1, w2 = hip.pack_conv_weight(wa, dt, 32), hip.pack_conv_weight(wb, dt, 32)
        units.append((w1, torch.zeros(C, device=dev), w2, torch.zeros(C, device=dev), d))

    def fused():
        hip.hifigan_resblock(rb, rate, x, bufs[0], units, C, k, 0.1, dt, ws=invs or None)

    def unfused():
        cur = x
        for i, (w1, b1, w2, b2, d) in enumerate(units):
            hip.hifigan_resunit(rb, rate, cur, bufs[i & 1], w1, b1, w2, b2, C, k, d, 0.1, dt, ws=invs[i] if invs else None)
            cur = bufs[i & 1]

    res = []
    for fn in (fused, unfused):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        res.append(a.elapsed_time(b) / iters)
    flops = 4.0 * C * C * k * rows * len(dils)
    byts = 2.0 * rows * C * (2 if dt == hip.F16 else 4)      # (F32S: f32 tensors)
    print(f"ResBlock C={C:4d} k={k:2d} rows={rows:9d}: fused {res[0]:7.3f} ms ({flops / res[0] / 1e9:7.1f} TFLOP/s, {byts / res[0] / 1e6:7.1f} GB/s "
          f"of x-in + y-out)   3 unit launches {res[1]:7.3f} ms   speed-up {res[1] / res[0]:.2f}x")
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--C", type=int, default=128)
    ap.add_argument("--k", type=int, default=11)
    ap.add_argument("--dil", type=int, default=1)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--all", action="store_true")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--dtype", default="f16", choices=["f16", "f32", "split", "emul", "emul6"])
    ap.add_argument("--resblock", action="store_true", help="fused ResBlock launches vs per-unit launches (f16)")
    ap.add_argument("--layout", type=int, default=1, choices=[0, 1], help="emulated units: 1 = v_mfma_f32_16x16x32_bf16 kernels (product), 0 = 32x32x16")
    a = ap.parse_args()
    rates = {256: 8, 128: 64, 64: 128, 32: 256}  # HiFi-GAN v1 22.05 kHz stage rates
    dt = {"f16": hip.F16, "f32": hip.F32, "split": hip.F32S, "emul": hip.F32E, "emul6": hip.F32E6}[a.dtype]
    LAYOUT[0] = a.layout
    if a.resblock:
        for C in ((32, 64, 128) if dt == hip.F16 else (32, 64)):
            for k in ((3, 7) if dt != hip.F32 else (3,)):
                if dt in (hip.F32S,) + hip.EMUL and (C, k) == (64, 7):
                    continue
                run_block(C, k, rates[C], a.iters, dt=dt)
        return
    if a.all:
        tot = 0.0
        for C in (256, 128, 64, 32):
            for k in (3, 7, 11):
                for d in (1, 3, 5):
                    tot += run(C, k, d, rates[C], a.iters, dtype=dt)
        print(f"sum over the 36 units of one generator pass: {tot:.2f} ms")
    else:
        run(a.C, a.k, a.dil, rates[a.C], a.iters, B=a.batch, dtype=dt)


if __name__ == "__main__":
    main()
