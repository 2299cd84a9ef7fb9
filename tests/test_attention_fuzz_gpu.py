"""Randomised check of jatts_relpos_attention against float64 attention: random ragged lengths (1 .. 330, so every partial-tile and
one-row case comes up), every d_k the kernel has, legacy / new rel-pos bias or none, with and without the u . k term, aligned and
element-aligned V^T (NaN in every slack column), key lengths shorter than the padded length (kv_len: the reference's training-time
forward() on a padded batch, attention.py:80-88), f32 / split / f16.  Complements the fixed cases of test_kernels_gpu.py."""
import math
import random

import pytest
import torch

pytestmark = pytest.mark.gpu


def one_case(rng, g, dev, prec):
    from jatts_amd import hip
    from oracle.fs2_oracle import rel_shift_legacy
    dk = rng.choice([32, 64, 96, 128, 192, 256])
    H = rng.choice([1, 2, 4]) if dk <= 128 else rng.choice([1, 2])
    n = rng.choice([1, 2, 3, 5])
    lens = [rng.choice([1, 2, 7, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 200, 257, 330]) if rng.random() < 0.5 else rng.randint(1, 330) for _ in range(n)]
    rel = rng.choice(["none", "legacy", "new"])
    use_ku = rel != "none" or rng.random() < 0.5
    pad_vt = rng.random() < 0.6
    use_kv = rng.random() < 0.3
    dt = {"fp32": hip.F32, "fp32_split": hip.F32S, "fp16": hip.F16}[prec]
    tdt = hip.torch_dtype(dt)
    rnd = (lambda t: t.half().float()) if prec == "fp16" else (lambda t: t)
    A, R, Tm = H * dk, sum(lens), max(lens)
    q, k, v = (rnd(torch.randn(R, A, generator=g)) for _ in range(3))
    ku = torch.randn(R, H, generator=g) if use_ku else None
    scale = 1.0 / math.sqrt(dk)
    cap = Tm + rng.randint(0, 9)
    if rel == "legacy":
        n_pos, ldg = Tm, hip.round_up(Tm, 32)
    elif rel == "new":
        n_pos, ldg = 2 * cap - 1, hip.round_up(2 * cap - 1, 32)
    else:
        n_pos, ldg = 0, 0
    gm = rnd(torch.randn(R, H, ldg, generator=g)) if rel != "none" else None
    kv = [rng.randint(1, T) for T in lens] if use_kv else None
    outs, o = [], 0
    for b, T in enumerate(lens):
        qs, ks, vs = (t[o:o + T].view(T, H, dk).transpose(0, 1).double() for t in (q, k, v))
        s = qs @ ks.transpose(1, 2)
        if ku is not None:
            s = s + ku[o:o + T].t().double().unsqueeze(1)
        if rel == "legacy":
            s = s + rel_shift_legacy(gm[o:o + T, :, :T].permute(1, 0, 2).double())
        elif rel == "new":          # BD'[i, j] = g[i][center - i + j], center = cap - 1 (attention.py:237-261)
            idx = (cap - 1) - torch.arange(T)[:, None] + torch.arange(T)[None, :]
            s = s + gm[o:o + T].permute(1, 0, 2).double().gather(2, idx.expand(H, T, T))
        s = s * scale
        if kv is not None:
            s[:, :, kv[b]:] = float("-inf")
        outs.append((torch.softmax(s, -1) @ vs).transpose(0, 1).reshape(T, A))
        o += T
    ref = torch.cat(outs)
    rb = hip.RaggedBatch(lens, dev)
    vcol, ldvt = rb.vt_layout() if pad_vt else (None, R)
    vt = torch.full((A, ldvt), float("nan"), dtype=tdt, device=dev)
    o = 0
    for b, T in enumerate(lens):
        c0 = int(vcol[b]) if pad_vt else o
        vt[:, c0:c0 + T] = v[o:o + T].t().to(dev).to(tdt)
        o += T
    out = hip.relpos_attention(rb, q.to(dev).to(tdt), A, k.to(dev).to(tdt), A, vt, ldvt,
                               gm.reshape(R, H * ldg).to(dev).to(tdt) if gm is not None else None, ldg,
                               ku.to(dev) if ku is not None else None, scale, H, dk, dt,
                               rel_mode={"none": 0, "legacy": 1, "new": 2}[rel], rel_center=cap - 1 if rel == "new" else 0, vt_col0=vcol,
                               kv_len=torch.tensor(kv, dtype=torch.int32, device=dev) if kv is not None else None)
    out = out.float().cpu().double()
    ok = bool(torch.isfinite(out).all())
    e = float((out - ref).abs().max() / ref.abs().max().clamp_min(1e-30)) if ok else float("inf")
    return e, dict(dk=dk, H=H, lens=lens, rel=rel, ku=use_ku, pad_vt=pad_vt, kv=kv, prec=prec)


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5])
def test_relpos_attention_random_cases(cuda, lib, seed):
    rng = random.Random(seed)
    g = torch.Generator().manual_seed(seed)
    for i in range(40):
        prec = ["fp32", "fp32", "fp32_split", "fp16"][i % 4]
        e, desc = one_case(rng, g, cuda, prec)
        assert e <= (4e-3 if prec == "fp16" else 5e-5), (e, desc)
