#!/usr/bin/env python3
"""Randomised error sweep of the f32-equivalent emulated arithmetic (JATTS_F32E: three exact bf16 terms per operand, SEVEN partial
products per product; --products 6: JATTS_F32E6) against the exact-f32 kernels, both measured against fp64 on the same inputs.

    python tools/emul_sweep.py [--products 7] [--units 400] [--convs 700] [--seed 0] [--out profiles/r05_emul_sweep.json]

VERDICT r4 next #1 asked for >= 1 000 cases including single-non-zero contractions (K_eff = 1: every dot product has ONE term, so the
accumulation error both paths share vanishes and what is left is the representation of the product plus its one rounding into the result),
with  ratio = max |emulated - fp64| / max |exact f32 - fp64|  <= 2.0 in every case.  What the arithmetic guarantees, and what is asserted
(exit code 1 on any violation; tests/test_emul_gpu.py asserts the same on its own draw):
  * per product the dropped partial products are <= 2^-24 |w v| with seven products -- ONE f32 rounding's worth, the accuracy of an unfused
    f32 multiply -- and <= 2^-23 |w v| with six, for EVERY input;
  * the bf16 MFMA's f32 accumulate is correctly rounded when the accumulator is at least as large as the arriving products and TRUNCATES an
    accumulator that is smaller (up to 1 ulp; tools/bf16_acc_probe.hip, profiles/r05_bf16_acc_probe.txt).  The SEVEN-product kernels
    (JATTS_F32E) therefore keep the leading product and the six smaller ones in separate accumulators joined by one correctly rounded
    v_add_f32.  Asserted for them, in EVERY case (dense, sparse, single-non-zero alike):
        max-error ratio <= 2.0 and relative-L2 ratio <= 2.0 against the exact-f32 kernel, and EVERY element of every single-non-zero conv
        within 2^-24 (dropped) + 2^-24 (that add) = 2^-23 |w x| = 2 x an f32 FMA's error bound (the exact-f32 kernel: 1 x 2^-24, checked too).
    Measured over the 1 100 cases: max-error ratio median 0.36-0.39 on dense inputs (never above 1.00: the emulation is the MORE accurate of
    the two whenever terms are summed), 1.00 median / 1.82 maximum on few-term cases; per-element maximum 1.43 x 2^-24;
  * the SIX-product kernels (JATTS_F32E6) keep one accumulator: 2^-23 (dropped) + 1 ulp = 4 x 2^-24 per element at K_eff = 1 (measured
    2.7).  Asserted for them: relative-L2 ratio <= 2.0 on dense inputs and <= 3.0 on few-term cases (single-non-zero rows, 90 %-zero
    inputs), the per-element bound.  Their max-error ratio is RECORDED only (dense: median 0.8, p99 1.3-1.7, maximum 2.3; few-term: up to 5
    -- the exact-f32 kernel's error at a launch's largest output is then one or two roundings anywhere in [0, 2^-24], so the ratio of two
    such maxima is a lottery once the emulation's own error reaches 2-3 x 2^-24).
Shapes, lengths, magnitudes and input distributions are drawn at random from a fixed seed (the table is reproducible); the distributions are
tools/split_sweep.py's seven, every third case with single-non-zero weight rows.  CPU fp64 references: sizes are kept to what they
finish in a fraction of a second."""
import argparse
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jatts_amd import hip  # noqa: E402
from tools.split_sweep import KINDS, draw_x, errs, ref_conv, ref_unit  # noqa: E402


def single_nonzero_(w, g):
    """Keep ONE non-zero weight per output channel (in place): every contraction has one term."""
    n = w.shape[0]
    m = torch.zeros(n, w[0].numel())
    m[torch.arange(n), torch.randint(0, w[0].numel(), (n,), generator=g)] = 1
    return w.mul_(m.view_as(w))


def unit_case(g, dev, C, k, d, lens, kind, single, code=None):
    code = hip.F32E if code is None else code
    x = draw_x(g, sum(lens), C, kind)
    sc = float(x.abs().max().clamp_min(1e-30))
    w1 = torch.randn(C, C, k, generator=g) / math.sqrt(C * k) * torch.pow(10.0, torch.rand(C, 1, 1, generator=g) * 2 - 1)
    w2 = torch.randn(C, C, k, generator=g) / math.sqrt(C * k)
    if single:
        single_nonzero_(w1, g)
        single_nonzero_(w2, g)
        w1.mul_(math.sqrt(C * k))      # keep the branch at the input's magnitude
        w2.mul_(math.sqrt(C * k))
        b1 = b2 = torch.zeros(C)
    else:
        b1, b2 = torch.randn(C, generator=g) * 0.05 * min(sc, 1e3), torch.randn(C, generator=g) * 0.05 * min(sc, 1e3)
    ref = ref_unit(x, w1, b1, w2, b2, lens, k, d, 0.1)
    rb = hip.RaggedBatch(lens, dev)
    xd = x.to(dev)
    y, y32 = torch.empty_like(xd), torch.empty_like(xd)
    pk = hip.pack_unit_weight_bf16x3_k32 if LAYOUT[0] else (lambda t: hip.pack_conv_weight_bf16x3(t, 32))
    hip.hifigan_resunit(rb, 1, xd, y, pk(w1.to(dev)), b1.to(dev), pk(w2.to(dev)), b2.to(dev), C, k, d, 0.1, code, w_layout=LAYOUT[0])
    hip.hifigan_resunit(rb, 1, xd, y32, hip.pack_conv_weight(w1.to(dev), hip.F32, 32), b1.to(dev), hip.pack_conv_weight(w2.to(dev), hip.F32, 32), b2.to(dev),
                        C, k, d, 0.1, hip.F32)
    (m, e), (m32, e32) = errs(y, ref), errs(y32, ref)
    return dict(case=f"C{C} k{k} d{d} {lens} {kind}{' single' if single else ''}", kind=kind, single=single, max_emul=m, max_f32=m32, rel_emul=e,
                rel_f32=e32, finite=bool(torch.isfinite(y).all()))


def conv_case(g, dev, c_in, n_out, k, dil, act, lens, kind, single, code=None):
    code = hip.F32E if code is None else code
    x = draw_x(g, sum(lens), c_in, kind)
    w = torch.randn(n_out, c_in, k, generator=g) / math.sqrt(c_in * k) * torch.pow(10.0, torch.rand(n_out, 1, 1, generator=g) * 2 - 1)
    if single:
        single_nonzero_(w, g)
        b, act = torch.zeros(n_out), None
    else:
        b = torch.randn(n_out, generator=g) * 0.05 * min(float(x.abs().max().clamp_min(1e-30)), 1e3)
    pad = (k - 1) // 2 * dil
    ref = ref_conv(x, w, b, lens, dil, pad, k, act)
    rb = hip.RaggedBatch(lens, dev)
    xd = x.to(dev)
    actc = {"relu": hip.ACT_RELU, "tanh": hip.ACT_TANH, None: hip.ACT_NONE}[act]
    pk = (lambda t: hip.pack_conv_weight_bf16x3_k32(t, 64)) if LAYOUT[0] else (lambda t: hip.pack_conv_weight_bf16x3(t, 64))
    y = hip.conv1d(rb, xd, pk(w.to(dev)), c_in, n_out, k, dtype=code, dil=dil, bias=b.to(dev), act=actc, w_layout=LAYOUT[0])
    y32 = hip.conv1d(rb, xd, hip.pack_conv_weight(w.to(dev), hip.F32), c_in, n_out, k, dtype=hip.F32, dil=dil, bias=b.to(dev), act=actc)
    (m, e), (m32, e32) = errs(y, ref), errs(y32, ref)
    row = dict(case=f"{c_in}->{n_out} k{k} d{dil} {act} {lens} {kind}{' single' if single else ''}", kind=kind, single=single, max_emul=m, max_f32=m32,
               rel_emul=e, rel_f32=e32, finite=bool(torch.isfinite(y).all()))
    if single:     # ref is the exact product w x: element-wise error in units of 2^-24 |w x| (bound: 2 with seven products, 3 with six; 1 exact f32)
        den = ref.abs().clamp_min(1e-300) * 2.0 ** -24
        nz = ref != 0
        row["per_product_emul"] = float(((y.double().cpu() - ref).abs() / den)[nz].max()) if nz.any() else 0.0
        row["per_product_f32"] = float(((y32.double().cpu() - ref).abs() / den)[nz].max()) if nz.any() else 0.0
    return row


def ri(g, lo, hi):
    return int(torch.randint(lo, hi, (1,), generator=g))


def sweep_units(n, g, dev, code=None):
    rows = []
    for i in range(n):
        C, k, d = [32, 64, 128, 256][ri(g, 0, 4)], [3, 7, 11][ri(g, 0, 3)], [1, 3, 5][ri(g, 0, 3)]
        lens = [int(v) for v in torch.randint(1, 900 if C <= 64 else 300, (ri(g, 1, 4),), generator=g)]
        rows.append(unit_case(g, dev, C, k, d, lens, KINDS[i % len(KINDS)], single=(i % 3 == 2), code=code))
    return rows


def sweep_convs(n, g, dev, code=None):
    rows = []
    for i in range(n):
        c_in, n_out = 64 * ri(g, 1, 17), 32 * ri(g, 1, 49)
        k = [1, 1, 3, 3, 5][ri(g, 0, 5)]
        dil = 1 if k == 1 else [1, 2, 4][ri(g, 0, 3)]
        act = [None, None, "relu", "tanh"][ri(g, 0, 4)]
        lens = [int(v) for v in torch.randint(1, 400, (ri(g, 1, 4),), generator=g)]
        rows.append(conv_case(g, dev, c_in, n_out, k, dil, act, lens, KINDS[i % len(KINDS)], single=(i % 3 == 2), code=code))
    return rows


def ratio_of(r):
    if r["max_f32"] == 0.0:
        return 1.0 if r["max_emul"] == 0.0 else float("inf")
    return r["max_emul"] / r["max_f32"]


def l2_ratio_of(r):
    if r["rel_f32"] == 0.0:
        return 1.0 if r["rel_emul"] == 0.0 else float("inf")
    return r["rel_emul"] / r["rel_f32"]


def is_single(r):
    return bool(r["single"])


def few_terms(r):
    """Cases whose outputs have one or two terms: single-non-zero weight rows, or 90 %-zero inputs."""
    return r["single"] or r["kind"] == "sparse"


def uniform_scale(r):
    """Dense inputs of ONE magnitude (unit / tiny / large): thousands of outputs share the launch's largest scale, so the maximum error is a
    statistic of many roundings.  With rows / channel blocks orders of magnitude apart or heavy tails a handful of outputs decide it."""
    return not few_terms(r) and r["kind"] in ("unit", "tiny", "large")


PER_PRODUCT_BOUND = {7: 2.01, 6: 4.01}     # units of 2^-24 |w x|: seven products 1 (dropped) + 1 (one correctly rounded add); six 2 + one MFMA accumulate of <= 1 ulp (2)


def violates(r, products=7):
    if not r["finite"] or r.get("per_product_emul", 0.0) > PER_PRODUCT_BOUND[products] or r.get("per_product_f32", 0.0) > 1.0001:
        return True
    if products == 7:      # the acceptance rule as written: every case, maximum error and relative L2
        return ratio_of(r) > 2.0 or l2_ratio_of(r) > 2.0
    if few_terms(r):
        return l2_ratio_of(r) > 3.0
    return l2_ratio_of(r) > 2.0


def _stats(rows):
    ratio = torch.tensor([ratio_of(r) for r in rows], dtype=torch.float64)
    l2 = torch.tensor([l2_ratio_of(r) for r in rows], dtype=torch.float64)
    t = torch.tensor([[r["rel_emul"], r["rel_f32"]] for r in rows], dtype=torch.float64)
    worst = int(ratio.argmax())
    return dict(cases=len(rows),
                max_err_ratio=dict(max=float(ratio.max()), p99=float(ratio.quantile(0.99)), median=float(ratio.median()), min=float(ratio.min()),
                                   above_1=int((ratio > 1).sum()), above_2=int((ratio > 2).sum()), worst_case=rows[worst]["case"]),
                rel_l2_ratio=dict(max=float(l2.max()), median=float(l2.median()), above_2=int((l2 > 2).sum())),
                rel_l2=dict(emul_max=float(t[:, 0].max()), f32_max=float(t[:, 1].max()), emul_median=float(t[:, 0].median()), f32_median=float(t[:, 1].median())))


def summary(rows, products=7):
    dense, few = [r for r in rows if not few_terms(r)], [r for r in rows if few_terms(r)]
    uni = [r for r in dense if uniform_scale(r)]
    out = dict(cases=len(rows), all_finite=all(r["finite"] for r in rows), violations=sum(violates(r, products) for r in rows))
    if uni:
        out["dense_one_magnitude (unit / tiny / large)"] = dict(**_stats(uni))
    a7 = "max_err_ratio <= 2.0 and rel_l2_ratio <= 2.0"
    if dense:
        out["multi_term_dense"] = dict(asserted=a7 if products == 7 else "rel_l2_ratio <= 2.0 (max_err_ratio recorded)", **_stats(dense))
    if few:
        out["few_terms (single-non-zero rows, 90 %-zero inputs)"] = dict(
            asserted=(a7 if products == 7 else "rel_l2_ratio <= 3.0 (max_err_ratio recorded)") + "; per-element bound on the single-non-zero convs", **_stats(few))
        pp = [r for r in few if "per_product_emul" in r]
        if pp:
            out["single_nonzero_conv_per_product_error_in_units_of_2^-24"] = dict(
                cases=len(pp), emul_max=max(r["per_product_emul"] for r in pp), f32_max=max(r["per_product_f32"] for r in pp),
                emul_bound=PER_PRODUCT_BOUND[products], f32_bound=1.0)
    return out


LAYOUT = [1]      # the MFMA form under test: 1 = the v_mfma_f32_16x16x32_bf16 kernels (the product form since round 6), 0 = 32 x 32 x 16 (--layout 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layout", type=int, default=1, choices=[0, 1], help="1 = v_mfma_f32_16x16x32_bf16 kernels (product), 0 = the 32 x 32 x 16 kernels")
    ap.add_argument("--units", type=int, default=400)
    ap.add_argument("--convs", type=int, default=700)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--products", type=int, default=7, choices=[6, 7], help="partial products per product: 7 = JATTS_F32E, 6 = JATTS_F32E6")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    LAYOUT[0] = a.layout
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(a.seed)
    code = hip.F32E if a.products == 7 else hip.F32E6
    out = {"seed": a.seed, "products": a.products, "mfma_form": "16x16x32" if a.layout else "32x32x16",
           "what": "err_emul / err_f32, both = max |y - fp64 reference| on the same random inputs (JATTS_F32E / JATTS_F32E6 vs JATTS_F32 kernels); "
                   "rel_l2 = ||y - ref|| / ||ref||; every third case has ONE non-zero weight per output channel (K_eff = 1)"}
    bad = 0
    for name, fn, n in (("resunit", sweep_units, a.units), ("conv1d", sweep_convs, a.convs)):
        rows = fn(n, g, dev, code)
        out[name] = summary(rows, a.products)
        out[name + "_worst5"] = sorted(rows, key=lambda r: -ratio_of(r))[:5]
        out[name + "_rows"] = [[r["case"], float(f"{r['max_emul']:.4g}"), float(f"{r['max_f32']:.4g}"), float(f"{r['rel_emul']:.4g}"), float(f"{r['rel_f32']:.4g}")]
                               + ([round(r["per_product_emul"], 3), round(r["per_product_f32"], 3)] if "per_product_emul" in r else []) for r in rows]
        bad += out[name]["violations"]
        print(name, json.dumps(out[name]))
    out["total_cases"] = a.units + a.convs
    out["violations"] = bad
    if a.out:
        json.dump(out, open(a.out, "w"), indent=1)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
