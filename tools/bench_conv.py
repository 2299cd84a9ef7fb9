#!/usr/bin/env python3
"""Micro-benchmark of jatts_conv1d at the shapes the acoustic models use (FastSpeech2 / Matcha U-Net), per dtype.
    python tools/bench_conv.py [--dtype f32] [--iters 10]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from jatts_amd import hip  # noqa: E402

SHAPES = [  # (c_in, n_out, k, frames per utterance, resid f32 out?)
    (384, 384, 1, 768, True), (384, 768, 1, 768, False), (384, 1536, 3, 768, False), (1536, 384, 3, 768, True),
    (512, 512, 1, 768, True), (512, 1536, 1, 768, False), (512, 2048, 1, 768, False), (2048, 512, 1, 768, True),
    (512, 512, 3, 768, False), (512, 512, 1, 384, True), (512, 2048, 1, 384, False), (1024, 512, 3, 384, False),
    (192, 768, 1, 768, False), (384, 1536, 3, 128, False), (1536, 384, 3, 128, True), (384, 384, 1, 128, True),
    (256, 256, 5, 768, False), (256, 1024, 3, 6144, False),
    (384, 768, 1, 128, False), (192, 768, 1, 128, False), (768, 384, 1, 64, True), (384, 384, 1, 64, True), (1536, 384, 1, 64, True), (384, 1536, 1, 64, False),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f32", choices=["f16", "f32", "split", "emul", "emul6"])
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--only", type=int, default=-1, help="index into SHAPES")
    ap.add_argument("--variant", type=int, default=0, help="jatts_conv_desc.variant (0 = the product heuristic)")
    ap.add_argument("--pre-lrelu", type=float, default=None, help="LeakyReLU prologue slope (the HiFi-GAN upsampling convs)")
    ap.add_argument("--layout", type=int, default=1, choices=[0, 1], help="emulated convs: 1 = v_mfma_f32_16x16x32_bf16 kernels (product), 0 = 32x32x16")
    ap.add_argument("--shapes", default="", help="comma-separated indices into SHAPES")
    a = ap.parse_args()
    dt = {"f16": hip.F16, "f32": hip.F32, "split": hip.F32S, "emul": hip.F32E, "emul6": hip.F32E6}[a.dtype]
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    for c, n, k, T, res in ([SHAPES[int(i)] for i in a.shapes.split(',')] if a.shapes else SHAPES if a.only < 0 else [SHAPES[a.only]]):
        rb = hip.RaggedBatch([T] * a.batch, dev)
        rows = rb.total
        x = (torch.randn(rows, c, generator=g) * 0.5).to(dev).to(hip.torch_dtype(dt))
        wf = (torch.randn(n, c, k, generator=g) / (c * k) ** 0.5).to(dev)
        lay = a.layout if dt in hip.EMUL else 0
        w, winv = (hip.pack_conv_weight_split(wf, 64) if dt == hip.F32S
                   else ((hip.pack_conv_weight_bf16x3_k32(wf, 64) if lay else hip.pack_conv_weight_bf16x3(wf, 64)), None) if dt in hip.EMUL
                   else (hip.pack_conv_weight(wf, dt), None))
        b = torch.zeros(n, device=dev)
        r = torch.zeros(rows, n, device=dev) if res else None

        def run():
            return hip.conv1d(rb, x, w, c, n, k, dtype=dt, bias=b, resid=r, out=r, out_f32=res or dt in (hip.F32S,) + hip.EMUL, variant=a.variant,
                              pre_lrelu=a.pre_lrelu, w_inv=winv, w_layout=lay)
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        print(f"{a.dtype} v{a.variant} {c:5d} -> {n:5d} k={k} rows={rows:6d} resid={int(res)}  {ms * 1e3:8.1f} us  {2.0 * c * n * k * rows / ms / 1e9:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
