#!/usr/bin/env python3
"""Randomised error sweep of the split arithmetic (JATTS_F32S) against the exact-f32 kernels, both measured against fp64.

    python tools/split_sweep.py [--units 120] [--convs 160] [--attn 40] [--seed 0] [--out profiles/r04_split_sweep.json]

The parity tests hold the split kernels to "maximum error <= 2x the exact-f32 kernel's" on a fixed list of shapes.  This draws the
shapes, lengths, magnitudes and input distributions at random (fixed seed: the table is reproducible) and records, per kernel family,
the distribution of  err_split / err_f32  (maximum absolute error against an fp64 evaluation of the same inputs) and of the relative
L2 errors -- the evidence for calling the split mode "not narrower than the reference's arithmetic".  CPU fp64 references: sizes are
kept to what they finish in a fraction of a second.
"""
import argparse
import json
import math
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jatts_amd import hip  # noqa: E402


def draw_x(g, R, C, kind):
    x = torch.randn(R, C, generator=g)
    if kind == "tiny":
        x = x * 10.0 ** float(torch.empty(1).uniform_(-7, -3, generator=g))
    elif kind == "large":
        x = x * 10.0 ** float(torch.empty(1).uniform_(1, 4, generator=g))
    elif kind == "rows":           # rows up to 8 orders of magnitude apart
        x = x * torch.pow(10.0, torch.rand(R, 1, generator=g) * 8 - 6)
    elif kind == "heavy":          # heavy tails: a Student-t-like ratio, clipped
        x = (x / (torch.randn(R, C, generator=g).abs() + 0.05)).clamp(-1e3, 1e3)
    elif kind == "sparse":         # mostly zeros (post-ReLU-like)
        x = x * (torch.rand(R, C, generator=g) < 0.1)
    elif kind == "channels":       # channel blocks orders of magnitude apart
        x = x * torch.pow(10.0, torch.randint(-3, 4, (1, C), generator=g).float())
    return x.contiguous()


KINDS = ["unit", "tiny", "large", "rows", "heavy", "sparse", "channels"]


def ref_unit(x, w1, b1, w2, b2, lens, k, d, slope):
    outs, o = [], 0
    for L in lens:
        xs = x[o:o + L].t().unsqueeze(0).double()
        t = F.conv1d(F.leaky_relu(xs, slope), w1.double(), b1.double(), padding=(k - 1) // 2 * d, dilation=d)
        t = F.conv1d(F.leaky_relu(t, slope), w2.double(), b2.double(), padding=(k - 1) // 2)
        outs.append((xs + t)[0].t())
        o += L
    return torch.cat(outs)


def ref_conv(x, w, b, lens, dil, pad, k, act):
    outs, o = [], 0
    for L in lens:
        xs = F.pad(x[o:o + L].t().unsqueeze(0).double(), (pad, (k - 1) * dil - pad))
        outs.append(F.conv1d(xs, w.double(), b.double(), dilation=dil)[0].t())
        o += L
    y = torch.cat(outs)
    return torch.relu(y) if act == "relu" else torch.tanh(y) if act == "tanh" else y


def errs(y, ref):
    d = (y.double().cpu() - ref)
    return float(d.abs().max()), float(d.norm() / ref.norm().clamp_min(1e-300))


def sweep_units(n, g, dev):
    rows = []
    for i in range(n):
        C = [32, 64, 128, 256][int(torch.randint(0, 4, (1,), generator=g))]
        k = [3, 7, 11][int(torch.randint(0, 3, (1,), generator=g))]
        d = [1, 3, 5][int(torch.randint(0, 3, (1,), generator=g))]
        lens = [int(v) for v in torch.randint(1, 1200 if C <= 64 else 400, (int(torch.randint(1, 4, (1,), generator=g)),), generator=g)]
        kind = KINDS[i % len(KINDS)]
        x = draw_x(g, sum(lens), C, kind)
        sc = float(x.abs().max().clamp_min(1e-30))
        w1 = torch.randn(C, C, k, generator=g) / math.sqrt(C * k) * torch.pow(10.0, torch.rand(C, 1, 1, generator=g) * 2 - 1)
        w2 = torch.randn(C, C, k, generator=g) / math.sqrt(C * k)
        b1, b2 = torch.randn(C, generator=g) * 0.05 * min(sc, 1e3), torch.randn(C, generator=g) * 0.05 * min(sc, 1e3)
        ref = ref_unit(x, w1, b1, w2, b2, lens, k, d, 0.1)
        rb = hip.RaggedBatch(lens, dev)
        xd = x.to(dev)
        ws1, is1 = hip.pack_conv_weight_split(w1.to(dev), 32)
        ws2, is2 = hip.pack_conv_weight_split(w2.to(dev), 32)
        y, y32 = torch.empty_like(xd), torch.empty_like(xd)
        hip.hifigan_resunit(rb, 1, xd, y, ws1, b1.to(dev), ws2, b2.to(dev), C, k, d, 0.1, hip.F32S, ws=(is1, is2))
        hip.hifigan_resunit(rb, 1, xd, y32, hip.pack_conv_weight(w1.to(dev), hip.F32, 32), b1.to(dev),
                            hip.pack_conv_weight(w2.to(dev), hip.F32, 32), b2.to(dev), C, k, d, 0.1, hip.F32)
        (m, e), (m32, e32) = errs(y, ref), errs(y32, ref)
        rows.append(dict(case=f"C{C} k{k} d{d} {lens} {kind}", max_split=m, max_f32=m32, rel_split=e, rel_f32=e32, finite=bool(torch.isfinite(y).all())))
    return rows


def sweep_convs(n, g, dev):
    rows = []
    for i in range(n):
        c_in = 64 * int(torch.randint(1, 17, (1,), generator=g))
        n_out = 32 * int(torch.randint(1, 49, (1,), generator=g))
        k = [1, 1, 3, 3, 5][int(torch.randint(0, 5, (1,), generator=g))]
        dil = 1 if k == 1 else [1, 2, 4][int(torch.randint(0, 3, (1,), generator=g))]
        act = [None, None, "relu", "tanh"][int(torch.randint(0, 4, (1,), generator=g))]
        lens = [int(v) for v in torch.randint(1, 500, (int(torch.randint(1, 4, (1,), generator=g)),), generator=g)]
        kind = KINDS[i % len(KINDS)]
        x = draw_x(g, sum(lens), c_in, kind)
        w = torch.randn(n_out, c_in, k, generator=g) / math.sqrt(c_in * k) * torch.pow(10.0, torch.rand(n_out, 1, 1, generator=g) * 2 - 1)
        b = torch.randn(n_out, generator=g) * 0.05 * min(float(x.abs().max().clamp_min(1e-30)), 1e3)
        pad = (k - 1) // 2 * dil
        ref = ref_conv(x, w, b, lens, dil, pad, k, act)
        rb = hip.RaggedBatch(lens, dev)
        xd = x.to(dev)
        actc = {"relu": hip.ACT_RELU, "tanh": hip.ACT_TANH, None: hip.ACT_NONE}[act]
        wsp, winv = hip.pack_conv_weight_split(w.to(dev), 64)
        y = hip.conv1d(rb, xd, wsp, c_in, n_out, k, dtype=hip.F32S, w_inv=winv, dil=dil, bias=b.to(dev), act=actc, out_f32=True)
        y32 = hip.conv1d(rb, xd, hip.pack_conv_weight(w.to(dev), hip.F32), c_in, n_out, k, dtype=hip.F32, dil=dil, bias=b.to(dev), act=actc)
        (m, e), (m32, e32) = errs(y, ref), errs(y32, ref)
        rows.append(dict(case=f"{c_in}->{n_out} k{k} d{dil} {act} {lens} {kind}", max_split=m, max_f32=m32, rel_split=e, rel_f32=e32,
                         finite=bool(torch.isfinite(y).all())))
    return rows


def sweep_attn(n, g, dev):
    """Plain softmax attention (no rel-pos bias): q, k, v of random magnitude; d_k 64 / 128 / 192 / 256."""
    rows = []
    for i in range(n):
        dk = [64, 128, 192, 256][int(torch.randint(0, 4, (1,), generator=g))]
        H = int(torch.randint(1, 3, (1,), generator=g))
        lens = [int(v) for v in torch.randint(1, 300, (int(torch.randint(1, 4, (1,), generator=g)),), generator=g)]
        R, A = sum(lens), H * dk
        kind = ["unit", "large", "rows", "heavy", "channels"][i % 5]
        q, k_, v = draw_x(g, R, A, "unit"), draw_x(g, R, A, kind), draw_x(g, R, A, kind)
        if kind in ("large", "heavy"):
            k_ = k_ / k_.abs().max() * 8.0          # keep the logits in a range where softmax is not a one-hot of rounding noise
        rb = hip.RaggedBatch(lens, dev)
        scale = dk ** -0.5
        outs, o = [], 0
        for L in lens:
            qq, kk, vv = (t[o:o + L].double().view(L, H, dk).transpose(0, 1) for t in (q, k_, v))
            p = torch.softmax(qq @ kk.transpose(1, 2) * scale, dim=-1)
            outs.append((p @ vv).transpose(0, 1).reshape(L, A))
            o += L
        ref = torch.cat(outs)
        qk = torch.cat([q, k_], 1).to(dev).contiguous()
        vcol, ldvt = rb.vt_layout()
        vt = torch.zeros(A, ldvt, device=dev)
        o = 0
        cols = vcol.cpu().tolist()
        for L, c0 in zip(lens, cols):
            vt[:, c0:c0 + L] = v[o:o + L].t().to(dev)
            o += L
        res = {}
        for name, dt in (("split", hip.F32S), ("f32", hip.F32)):
            y = hip.relpos_attention(rb, qk, 2 * A, qk, 2 * A, vt, ldvt, None, 0, None, scale, H, dk, dt, q_col0=0, k_col0=A, rel_mode=0, vt_col0=vcol)
            res[name] = errs(y, ref) + (bool(torch.isfinite(y).all()),)
        rows.append(dict(case=f"H{H} dk{dk} {lens} {kind}", max_split=res["split"][0], max_f32=res["f32"][0], rel_split=res["split"][1],
                         rel_f32=res["f32"][1], finite=res["split"][2]))
    return rows


def summary(rows):
    t = torch.tensor([[r["max_split"], r["max_f32"], r["rel_split"], r["rel_f32"]] for r in rows], dtype=torch.float64)
    ratio = t[:, 0] / t[:, 1].clamp_min(1e-300)
    worst = int(ratio.argmax())
    return dict(cases=len(rows), all_finite=all(r["finite"] for r in rows),
                max_err_ratio=dict(max=float(ratio.max()), p99=float(ratio.quantile(0.99)), median=float(ratio.median()), min=float(ratio.min()),
                                   above_1=int((ratio > 1).sum()), above_2=int((ratio > 2).sum()), worst_case=rows[worst]["case"]),
                rel_l2=dict(split_max=float(t[:, 2].max()), f32_max=float(t[:, 3].max()), split_median=float(t[:, 2].median()),
                            f32_median=float(t[:, 3].median())))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--units", type=int, default=120)
    ap.add_argument("--convs", type=int, default=160)
    ap.add_argument("--attn", type=int, default=40)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(a.seed)
    out = {"seed": a.seed, "what": "err_split / err_f32, both = max |y - fp64 reference| on the same random inputs; rel_l2 = ||y - ref|| / ||ref||"}
    for name, fn, n in (("resunit", sweep_units, a.units), ("conv1d", sweep_convs, a.convs), ("attention", sweep_attn, a.attn)):
        rows = fn(n, g, dev)
        out[name] = summary(rows)
        out[name + "_worst5"] = sorted(rows, key=lambda r: -r["max_split"] / max(r["max_f32"], 1e-300))[:5]
        print(name, json.dumps(out[name]))
    if a.out:
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
