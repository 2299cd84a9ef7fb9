"""GPU parity: every C-ABI kernel against the CPU oracle's arithmetic on seeded inputs.

Tolerances (written here, per the north star):
  fp32 mode (exact-f32 MFMA): relative L2 error <= 2e-5 (summation-order differences only).
  fp16 mode (f16 operands, f32 accumulate): operands are rounded to f16 first, the oracle
  computes in f32 on the SAME rounded operands -> relative L2 error <= 2e-3.
  Integer/index work (length regulator, durations given log-durations): bit-exact.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import maxdiff, relerr

pytestmark = pytest.mark.gpu

TOL = {"fp32": 2e-5, "fp16": 2e-3}


def _dt(prec):
    from jatts_amd import hip
    return hip.F32 if prec == "fp32" else hip.F16


def _round(t, prec):
    return t.half().float() if prec == "fp16" else t


def _ragged(lens, dev):
    from jatts_amd import hip
    return hip.RaggedBatch(lens, dev)


def _ref_conv(x, w, b, lens, dil, pad, k, pre_slope=None, act=None):
    """Per-sequence conv on time-major rows; y[t] = sum W[n,c,tap] x[t + tap*dil - pad]."""
    outs, o = [], 0
    for L in lens:
        xs = x[o:o + L].t().unsqueeze(0).double()
        if pre_slope is not None:
            xs = F.leaky_relu(xs, pre_slope)
        right = (k - 1) * dil - pad
        xs = F.pad(xs, (pad, right))
        y = F.conv1d(xs, w.double(), None if b is None else b.double(), dilation=dil)[0].t()
        outs.append(y)
        o += L
    y = torch.cat(outs)
    if act == "relu":
        y = torch.relu(y)
    elif act == "tanh":
        y = torch.tanh(y)
    return y


CONV_CASES = [
    # c_in, n_out, k, dil, lens, act, resid, transposed, pre, n_in
    (64, 128, 3, 1, [37, 256, 5], "relu", False, False, None, 1),
    (384, 1536, 3, 1, [128, 77], "relu", False, False, None, 1),
    (1536, 384, 3, 1, [128, 300], None, True, False, None, 1),
    (80, 256, 5, 1, [90, 41], "tanh", False, False, None, 1),
    (256, 80, 5, 1, [90, 41], None, True, False, None, 1),
    (384, 80, 1, 1, [100], None, False, False, None, 1),
    (384, 384, 1, 1, [33, 65], None, False, True, None, 1),
    (80, 512, 7, 1, [50, 20], None, False, False, None, 1),
    (32, 32, 11, 5, [400, 17], None, False, False, 0.1, 1),
    (64, 48, 3, 3, [70], None, False, False, 0.1, 3),
    (192, 700, 1, 1, [64, 130], None, False, False, None, 1),
    (32, 1, 3, 1, [19], None, False, False, None, 1),
    (96, 64, 7, 1, [40, 9], None, False, False, None, 1),
]


@pytest.mark.parametrize("prec", ["fp32", "fp16"])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv1d(cuda, lib, prec, case):
    from jatts_amd import hip
    c_in, n_out, k, dil, lens, act, resid, transposed, pre, n_in = case
    g = torch.Generator().manual_seed(hash(case[:4]) & 0xFFFF)
    R = sum(lens)
    xs = [_round(torch.randn(R, c_in, generator=g), prec) for _ in range(n_in)]
    w = _round(torch.randn(n_out, c_in, k, generator=g) / math.sqrt(c_in * k), prec)
    b = torch.randn(n_out, generator=g)
    res = torch.randn(R, n_out, generator=g) if resid else None
    pad = (k - 1) // 2 * dil
    in_scale = 1.0 / n_in
    xsum = sum(xs) * in_scale
    if prec == "fp16" and (n_in > 1 or pre is not None):
        # the kernel sums in f32, applies lrelu, then rounds the staged operand to f16
        xsum = (F.leaky_relu(xsum, pre) if pre is not None else xsum).half().float()
        ref = _ref_conv(xsum, w, b, lens, dil, pad, k, None, act)
    else:
        ref = _ref_conv(xsum, w, b, lens, dil, pad, k, pre, act)
    alpha = 0.5 if resid else 1.0
    ref = ref * alpha + (res.double() if resid else 0)
    dt = _dt(prec)
    rb = _ragged(lens, cuda)
    tdt = hip.torch_dtype(dt)
    wp = hip.pack_conv_weight(w.to(cuda), dt)
    c_pad = hip.round_up(c_in, 64)  # the ABI wants c_in % 64 == 0: zero-pad channels like the product does
    xs = [F.pad(x, (0, c_pad - c_in)) for x in xs]
    y = hip.conv1d(rb, [x.to(cuda).to(tdt).contiguous() for x in xs], wp, c_pad, n_out, k, dtype=dt, dil=dil, bias=b.to(cuda),
                   act={"relu": hip.ACT_RELU, "tanh": hip.ACT_TANH, None: hip.ACT_NONE}[act], alpha=alpha,
                   resid=None if res is None else res.to(cuda), out_f32=True, transposed=transposed,
                   pre_lrelu=pre, in_scale=in_scale)
    torch.cuda.synchronize()
    y = y.t() if transposed else y
    e = relerr(y, ref)
    assert e <= TOL[prec], f"conv1d {case} {prec}: rel err {e:.3e}"


@pytest.mark.parametrize("xkind", ["unit", "wide"])
@pytest.mark.parametrize("case", CONV_CASES + [
    (384, 1536, 3, 1, [1024, 700], "relu", False, False, None, 1),      # the 128 x 128 tile (> 600 workgroups need more rows; forced below too)
    (1536, 384, 3, 1, [640], None, True, False, None, 1),
    (512, 512, 5, 2, [300, 41], "tanh", False, False, None, 1),
    (256, 1024, 3, 1, [90, 200], None, False, False, 0.1, 1),           # HiFi-GAN upsampling conv shape with the LeakyReLU prologue
    (64, 64, 4, 1, [5000], None, False, False, 0.1, 1),                  # n_out <= 64 tile
])
def test_conv1d_split(cuda, lib, case, xkind):
    """jatts_conv1d with JATTS_F32S (round 4): f32 tensors, split f16 hi/lo MFMA operands, per-chunk-tile power-of-two activation scales.
    Same tolerance as the exact-f32 kernels (relative L2 <= 2e-5 vs fp64); maximum error at most twice the exact-f32 kernel's on the same
    inputs; `wide`: input channel blocks differing by up to 8 orders of magnitude (every 64-channel chunk gets its own scale, the
    accumulators are rescaled in between) and rows by 6."""
    from jatts_amd import hip
    c_in, n_out, k, dil, lens, act, resid, transposed, pre, n_in = case
    g = torch.Generator().manual_seed((hash(case[:4]) & 0xFFFF) + 1)
    R = sum(lens)
    xs = [torch.randn(R, c_in, generator=g) for _ in range(n_in)]
    if xkind == "wide":
        blk = torch.pow(10.0, torch.randint(-4, 5, ((c_in + 63) // 64,), generator=g).float()).repeat_interleave(64)[:c_in]
        rowm = torch.pow(10.0, torch.rand(R, 1, generator=g) * 6 - 3)
        xs = [x * blk * rowm for x in xs]
    w = torch.randn(n_out, c_in, k, generator=g) / math.sqrt(c_in * k) * torch.pow(10.0, torch.rand(n_out, 1, 1, generator=g) * 2 - 1)
    b = torch.randn(n_out, generator=g)
    res = torch.randn(R, n_out, generator=g) if resid else None
    pad = (k - 1) // 2 * dil
    in_scale = 1.0 / n_in
    ref = _ref_conv(sum(xs) * in_scale, w, b, lens, dil, pad, k, pre, act)
    alpha = 0.5 if resid else 1.0
    ref = ref * alpha + (res.double() if resid else 0)
    rb = _ragged(lens, cuda)
    c_pad = hip.round_up(c_in, 64)
    xs = [F.pad(x, (0, c_pad - c_in)).to(cuda).contiguous() for x in xs]
    wsp, winv = hip.pack_conv_weight_split(w.to(cuda), 64)
    actc = {"relu": hip.ACT_RELU, "tanh": hip.ACT_TANH, None: hip.ACT_NONE}[act]
    kw = dict(dil=dil, bias=b.to(cuda), act=actc, alpha=alpha, out_f32=True, transposed=transposed, pre_lrelu=pre, in_scale=in_scale)
    if (k - 1) * dil > 32:       # outside the split kernel's tiles: refused loudly through the C ABI; the product wrapper (hip.SplitWeight) takes
        from jatts_amd._abi import JattsHipError          # the exact-f32 kernel on the same weight instead
        with pytest.raises(JattsHipError):
            hip.conv1d(rb, xs, wsp, c_pad, n_out, k, dtype=hip.F32S, w_inv=winv, **kw)
        y = hip.conv1d(rb, xs, hip.SplitWeight(w.to(cuda), 64), c_pad, n_out, k, dtype=hip.F32, **kw)
        assert relerr(y, ref) <= TOL["fp32"]
        return
    y = hip.conv1d(rb, xs, wsp, c_pad, n_out, k, dtype=hip.F32S, w_inv=winv, resid=None if res is None else res.to(cuda), **kw)
    y32 = hip.conv1d(rb, xs, hip.pack_conv_weight(w.to(cuda), hip.F32), c_pad, n_out, k, dtype=hip.F32, resid=None if res is None else res.to(cuda), **kw)
    torch.cuda.synchronize()
    yt, y32 = (y.t(), y32.t()) if transposed else (y, y32)
    assert torch.isfinite(yt).all()
    e, m, m32 = relerr(yt, ref), _maxerr(yt, ref), _maxerr(y32, ref)
    e32 = relerr(y32, ref)
    # (the adversarial `wide` inputs put the exact-f32 kernel itself past 2e-5 on some shapes -- tanh of sums spanning 14 orders of magnitude:
    #  the claim under test is "no worse than twice the exact-f32 kernel", with the f32 tolerance as the floor)
    assert e <= max(TOL["fp32"], 2.0 * e32), f"split conv1d {case} {xkind}: rel err {e:.3e} (exact f32 {e32:.3e})"
    assert m <= 2.0 * m32 + 1e-30, f"split conv1d {case} {xkind}: max err {m:.3e} vs exact f32 {m32:.3e}"
    # an utterance alone == inside the batch, bit for bit (one tile geometry whatever the launch size)
    if len(lens) > 1 and not transposed:
        L0 = lens[0]
        y0 = hip.conv1d(_ragged([L0], cuda), [x[:L0].contiguous() for x in xs], wsp, c_pad, n_out, k, dtype=hip.F32S, w_inv=winv,
                        resid=None if res is None else res[:L0].to(cuda).contiguous(), **kw)
        assert torch.equal(y0, y[:L0])


def _maxerr(y, ref):
    return float((y.double().cpu() - ref).abs().max())


def _ref_unit(x, w1, b1, w2, b2, lens, k, d, slope, round16):
    outs, o = [], 0
    for L in lens:
        xs = x[o:o + L].t().unsqueeze(0).double()
        a = F.leaky_relu(xs, slope)
        if round16:
            a = a.half().double()
        h = F.conv1d(a, w1.double(), b1.double(), padding=(k - 1) // 2 * d, dilation=d)
        h = F.leaky_relu(h, slope)
        if round16:
            h = h.half().double()
        y = F.conv1d(h, w2.double(), b2.double(), padding=(k - 1) // 2) + xs
        outs.append(y[0].t())
        o += L
    return torch.cat(outs)


@pytest.mark.parametrize("prec", ["fp32", "fp16"])
@pytest.mark.parametrize("C,k,d,lens", [
    (32, 3, 1, [700, 3, 250]), (32, 11, 5, [600, 31]), (64, 7, 3, [513]), (128, 3, 5, [300, 40]),
    (128, 11, 1, [129]), (256, 7, 5, [150, 64]), (256, 11, 5, [70]), (512, 3, 3, [45]),
])
def test_hifigan_resunit(cuda, lib, prec, C, k, d, lens):
    from jatts_amd import hip
    if prec == "fp32" and C == 512:   # no f32 tile fits 160 KiB LDS at 512 channels: refused loudly, never a silent detour
        from jatts_amd._abi import JattsHipError
        rb = _ragged(lens, cuda)
        z = torch.zeros(sum(lens), C, device=cuda)
        w = torch.zeros(C * C * k, device=cuda)
        with pytest.raises(JattsHipError):
            hip.hifigan_resunit(rb, 1, z, torch.empty_like(z), w, z[0], w, z[0], C, k, d, 0.1, hip.F32)
        return
    g = torch.Generator().manual_seed(C * 100 + k * 10 + d)
    R = sum(lens)
    x = _round(torch.randn(R, C, generator=g), prec)
    w1 = _round(torch.randn(C, C, k, generator=g) / math.sqrt(C * k), prec)
    w2 = _round(torch.randn(C, C, k, generator=g) / math.sqrt(C * k), prec)
    b1, b2 = torch.randn(C, generator=g) * 0.1, torch.randn(C, generator=g) * 0.1
    ref = _ref_unit(x, w1, b1, w2, b2, lens, k, d, 0.1, prec == "fp16")
    dt = _dt(prec)
    tdt = hip.torch_dtype(dt)
    rb = _ragged(lens, cuda)
    xd = x.to(cuda).to(tdt)
    y = torch.full_like(xd, float("nan"))
    hip.hifigan_resunit(rb, 1, xd, y, hip.pack_conv_weight(w1.to(cuda), dt, 32), b1.to(cuda),
                        hip.pack_conv_weight(w2.to(cuda), dt, 32), b2.to(cuda), C, k, d, 0.1, dt)
    torch.cuda.synchronize()
    assert torch.isfinite(y.float()).all(), "unwritten / non-finite outputs"
    e = relerr(y.float(), ref)
    assert e <= TOL[prec], f"resunit C={C} k={k} d={d} {prec}: rel err {e:.3e}"


@pytest.mark.parametrize("xkind", ["unit", "tiny", "large", "wide"])
@pytest.mark.parametrize("C,k,d,lens", [
    (32, 3, 1, [700, 3, 250]), (32, 11, 5, [600, 31]), (64, 7, 3, [513]), (64, 11, 5, [260, 9]), (128, 3, 5, [300, 40]),
    (128, 11, 1, [129]), (128, 11, 5, [300]), (128, 7, 3, [140, 139]), (256, 7, 5, [150, 64]), (256, 11, 5, [70]), (256, 3, 1, [200]),
])
def test_hifigan_resunit_split(cuda, lib, C, k, d, lens, xkind):
    """JATTS_F32S (round 4): f32 activations, split-precision f16 hi/lo MFMA operands with power-of-two tile / channel scales.
    Held to the SAME tolerance as the exact-f32 kernel (relative L2 <= 2e-5 against fp64), and its maximum error against fp64 must
    not exceed twice the exact-f32 kernel's on the same inputs -- at unit, tiny (1e-6), large (3e3) and mixed (8 orders of magnitude
    between rows) activation magnitudes, i.e. wherever f16's narrow exponent range would bite a naive split."""
    from jatts_amd import hip
    g = torch.Generator().manual_seed(C * 100 + k * 10 + d)
    R = sum(lens)
    x = torch.randn(R, C, generator=g)
    if xkind == "tiny":
        x = x * 1e-6
    elif xkind == "large":
        x = x * 3e3
    elif xkind == "wide":
        x = x * torch.pow(10.0, torch.rand(R, 1, generator=g) * 8 - 6)
    w1 = torch.randn(C, C, k, generator=g) / math.sqrt(C * k) * torch.pow(10.0, torch.rand(C, 1, 1, generator=g) * 2 - 1)   # per-channel spread
    w2 = torch.randn(C, C, k, generator=g) / math.sqrt(C * k)
    sb = {"unit": 0.1, "tiny": 1e-7, "large": 300.0, "wide": 0.1}[xkind]
    b1, b2 = torch.randn(C, generator=g) * sb, torch.randn(C, generator=g) * sb
    ref = _ref_unit(x, w1, b1, w2, b2, lens, k, d, 0.1, False)
    rb = _ragged(lens, cuda)
    xd = x.to(cuda)
    ws1, is1 = hip.pack_conv_weight_split(w1.to(cuda), 32)
    ws2, is2 = hip.pack_conv_weight_split(w2.to(cuda), 32)
    y = torch.full_like(xd, float("nan"))
    hip.hifigan_resunit(rb, 1, xd, y, ws1, b1.to(cuda), ws2, b2.to(cuda), C, k, d, 0.1, hip.F32S, ws=(is1, is2))
    y32 = torch.full_like(xd, float("nan"))
    hip.hifigan_resunit(rb, 1, xd, y32, hip.pack_conv_weight(w1.to(cuda), hip.F32, 32), b1.to(cuda),
                        hip.pack_conv_weight(w2.to(cuda), hip.F32, 32), b2.to(cuda), C, k, d, 0.1, hip.F32)
    torch.cuda.synchronize()
    assert torch.isfinite(y).all(), "unwritten / non-finite outputs"
    e, e32 = relerr(y, ref), relerr(y32, ref)
    assert e <= TOL["fp32"], f"split resunit C={C} k={k} d={d} {xkind}: rel err {e:.3e} (exact f32: {e32:.3e})"
    m, m32 = _maxerr(y, ref), _maxerr(y32, ref)
    assert m <= 2.0 * m32 + 1e-30, f"split resunit C={C} k={k} d={d} {xkind}: max err {m:.3e} vs exact f32 {m32:.3e}"
    # an utterance alone == the same utterance inside the batch, bit for bit (tile scales are per utterance tile)
    if len(lens) > 1:
        L0 = lens[0]
        y0 = torch.empty(L0, C, device=cuda)
        hip.hifigan_resunit(_ragged([L0], cuda), 1, xd[:L0].contiguous(), y0, ws1, b1.to(cuda), ws2, b2.to(cuda), C, k, d, 0.1, hip.F32S, ws=(is1, is2))
        assert torch.equal(y0, y[:L0])


@pytest.mark.parametrize("xkind", ["unit", "wide"])
@pytest.mark.parametrize("C,k,dils,lens,mrf", [
    (32, 3, (1, 3, 5), [1300, 3, 250, 40], False), (32, 7, (1, 3, 5), [900, 31], False), (64, 3, (1, 3, 5), [513, 700], False),
    (32, 3, (1, 3, 5), [700, 90], True), (64, 3, (1, 3), [260, 31], True), (32, 7, (2,), [500], False), (32, 3, (1, 3, 5), [2000], False),
])
def test_hifigan_resblock_split(cuda, lib, C, k, dils, lens, mrf, xkind):
    """jatts_hifigan_resblock with JATTS_F32S (round 4): the whole ResBlock in one launch, residual stream in f32 registers, every conv on split
    operands with per-tile scales -- against the fp64 chain of units (the exact-f32 tolerance), against the chain of per-unit split launches
    (same arithmetic up to the scale blocks: the fused window is wider than a unit's tile), and an utterance alone == inside the batch."""
    from jatts_amd import hip
    g = torch.Generator().manual_seed(C * 100 + k * 10 + len(dils))
    R = sum(lens)
    x = torch.randn(R, C, generator=g)
    if xkind == "wide":
        x = x * torch.pow(10.0, torch.rand(R, 1, generator=g) * 6 - 4)
    ws = [(torch.randn(C, C, k, generator=g) / math.sqrt(C * k), torch.randn(C, generator=g) * 0.1,
           torch.randn(C, C, k, generator=g) * 0.5 / math.sqrt(C * k), torch.randn(C, generator=g) * 0.1) for _ in dils]
    adds = [torch.randn(R, C, generator=g) for _ in range(2)] if mrf else None
    ref = x.double()
    for (w1, b1, w2, b2), d in zip(ws, dils):
        ref = _ref_unit(ref, w1, b1, w2, b2, lens, k, d, 0.1, False)
    if mrf:
        ref = (ref + adds[0].double() + adds[1].double()) / 3.0
    rb = _ragged(lens, cuda)
    xd = x.to(cuda)
    packed, invs = [], []
    for (w1, b1, w2, b2), d in zip(ws, dils):
        (p1, i1), (p2, i2) = hip.pack_conv_weight_split(w1.to(cuda), 32), hip.pack_conv_weight_split(w2.to(cuda), 32)
        packed.append((p1, b1.to(cuda), p2, b2.to(cuda), d))
        invs.append((i1, i2))
    y = torch.full_like(xd, float("nan"))
    addd = [a.to(cuda) for a in adds] if mrf else None
    hip.hifigan_resblock(rb, 1, xd, y, packed, C, k, 0.1, hip.F32S, add=addd, out_scale=1.0 / 3.0 if mrf else 1.0, ws=invs)
    cur, bufs = xd, [torch.empty_like(xd), torch.empty_like(xd)]
    for i, ((w1, b1, w2, b2, d), iv) in enumerate(zip(packed, invs)):
        lastu = i == len(packed) - 1
        hip.hifigan_resunit(rb, 1, cur, bufs[i & 1], w1, b1, w2, b2, C, k, d, 0.1, hip.F32S, add=addd if (mrf and lastu) else None,
                            out_scale=1.0 / 3.0 if (mrf and lastu) else 1.0, ws=iv)
        cur = bufs[i & 1]
    # exact f32 per-unit chain: the error yardstick
    c32, b32 = xd, [torch.empty_like(xd), torch.empty_like(xd)]
    for i, ((w1, b1, w2, b2), d) in enumerate(zip(ws, dils)):
        lastu = i == len(ws) - 1
        hip.hifigan_resunit(rb, 1, c32, b32[i & 1], hip.pack_conv_weight(w1.to(cuda), hip.F32, 32), b1.to(cuda), hip.pack_conv_weight(w2.to(cuda), hip.F32, 32),
                            b2.to(cuda), C, k, d, 0.1, hip.F32, add=addd if (mrf and lastu) else None, out_scale=1.0 / 3.0 if (mrf and lastu) else 1.0)
        c32 = b32[i & 1]
    torch.cuda.synchronize()
    assert torch.isfinite(y).all(), "unwritten / non-finite outputs"
    e, e32 = relerr(y, ref), relerr(c32, ref)
    assert e <= max(TOL["fp32"], 2.0 * e32), f"split resblock C={C} k={k} dils={dils} {xkind}: rel err {e:.3e} (exact f32 units {e32:.3e})"
    assert _maxerr(y, ref) <= 2.0 * _maxerr(c32, ref) + 1e-30
    assert relerr(y, cur.double()) <= 1e-5
    if len(lens) > 1:
        L0 = lens[0]
        y0 = torch.empty(L0, C, device=cuda)
        hip.hifigan_resblock(_ragged([L0], cuda), 1, xd[:L0].contiguous(), y0, packed, C, k, 0.1, hip.F32S,
                             add=[a[:L0].contiguous() for a in addd] if mrf else None, out_scale=1.0 / 3.0 if mrf else 1.0, ws=invs)
        assert torch.equal(y0, y[:L0])


def test_hifigan_resunit_split_mrf_and_errors(cuda, lib):
    """The fused MRF mean of the split unit's output pass; zero input (tile maximum 0) stays exact; missing scales are refused."""
    from jatts_amd import hip
    from jatts_amd._abi import JattsHipError
    g = torch.Generator().manual_seed(12)
    lens, C, k, d = [300, 77], 64, 7, 3
    R = sum(lens)
    x, a0, a1 = (torch.randn(R, C, generator=g) for _ in range(3))
    w1, w2 = (torch.randn(C, C, k, generator=g) / math.sqrt(C * k) for _ in range(2))
    b1, b2 = torch.randn(C, generator=g) * 0.1, torch.randn(C, generator=g) * 0.1
    ref = (_ref_unit(x, w1, b1, w2, b2, lens, k, d, 0.1, False) + a0.double() + a1.double()) / 3.0
    rb = _ragged(lens, cuda)
    ws1, is1 = hip.pack_conv_weight_split(w1.to(cuda), 32)
    ws2, is2 = hip.pack_conv_weight_split(w2.to(cuda), 32)
    y = torch.empty(R, C, device=cuda)
    hip.hifigan_resunit(rb, 1, x.to(cuda), y, ws1, b1.to(cuda), ws2, b2.to(cuda), C, k, d, 0.1, hip.F32S,
                        add=[a0.to(cuda), a1.to(cuda)], out_scale=1.0 / 3.0, ws=(is1, is2))
    assert relerr(y, ref) <= TOL["fp32"]
    z = torch.zeros(R, C, device=cuda)
    hip.hifigan_resunit(rb, 1, z, y, ws1, torch.zeros(C, device=cuda), ws2, b2.to(cuda), C, k, d, 0.1, hip.F32S, ws=(is1, is2))
    assert torch.equal(y, b2.to(cuda).expand(R, C))          # conv1(0) = 0 -> h = 0 -> y = 0 + b2
    with pytest.raises((JattsHipError, ValueError)):
        hip.hifigan_resunit(rb, 1, z, y, ws1, b1.to(cuda), ws2, b2.to(cuda), C, k, d, 0.1, hip.F32S)


@pytest.mark.parametrize("C,k,dils,lens,mrf", [
    (32, 3, (1, 3, 5), [1300, 3, 250, 40], False), (32, 7, (1, 3, 5), [900, 31], False), (64, 3, (1, 3, 5), [513, 700], False),
    (64, 7, (1, 3, 5), [600, 64], False), (128, 3, (1, 3, 5), [300, 40], False), (64, 3, (1, 3), [260], False),
    (32, 3, (1, 3, 5), [700, 90], True), (32, 7, (2,), [500], False),
])
def test_hifigan_resblock_fused(cuda, lib, C, k, dils, lens, mrf):
    """jatts_hifigan_resblock (all dilation units of a ResBlock in one launch, residual stream in registers) against
    (a) the fp64 chain of units with the stream rounded to f16 between units, as the per-unit launches store it, and
    (b) the per-unit launches themselves."""
    from jatts_amd import hip
    g = torch.Generator().manual_seed(C * 100 + k * 10 + len(dils))
    R = sum(lens)
    x = _round(torch.randn(R, C, generator=g), "fp16")
    ws = [(_round(torch.randn(C, C, k, generator=g) / math.sqrt(C * k), "fp16"), torch.randn(C, generator=g) * 0.1,
           _round(torch.randn(C, C, k, generator=g) * 0.5 / math.sqrt(C * k), "fp16"), torch.randn(C, generator=g) * 0.1) for _ in dils]
    adds = [_round(torch.randn(R, C, generator=g), "fp16") for _ in range(2)] if mrf else None
    ref = x
    for (w1, b1, w2, b2), d in zip(ws, dils):
        ref = _ref_unit(ref, w1, b1, w2, b2, lens, k, d, 0.1, True).half().float()
    ref = ref.double()
    if mrf:
        ref = (ref + adds[0].double() + adds[1].double()) / 3.0
    rb = _ragged(lens, cuda)
    xd = x.to(cuda).half()
    packed = [(hip.pack_conv_weight(w1.to(cuda), hip.F16, 32), b1.to(cuda), hip.pack_conv_weight(w2.to(cuda), hip.F16, 32), b2.to(cuda), d)
              for (w1, b1, w2, b2), d in zip(ws, dils)]
    y = torch.full_like(xd, float("nan"))
    addd = [a.to(cuda).half() for a in adds] if mrf else None
    hip.hifigan_resblock(rb, 1, xd, y, packed, C, k, 0.1, hip.F16, add=addd, out_scale=1.0 / 3.0 if mrf else 1.0)
    torch.cuda.synchronize()
    assert torch.isfinite(y.float()).all(), "unwritten / non-finite outputs"
    e = relerr(y.float(), ref)
    assert e <= TOL["fp16"], f"resblock C={C} k={k} dils={dils}: rel err {e:.3e}"
    cur, bufs = xd, [torch.empty_like(xd), torch.empty_like(xd)]
    for i, (w1, b1, w2, b2, d) in enumerate(packed):
        lastu = i == len(packed) - 1
        hip.hifigan_resunit(rb, 1, cur, bufs[i & 1], w1, b1, w2, b2, C, k, d, 0.1, hip.F16,
                            add=addd if (mrf and lastu) else None, out_scale=1.0 / 3.0 if (mrf and lastu) else 1.0)
        cur = bufs[i & 1]
    assert relerr(y.float(), cur.float().double()) <= 2e-3   # same arithmetic up to one f16 rounding of the stream per unit


@pytest.mark.parametrize("C,dils,lens,mrf", [(32, (1, 3, 5), [1300, 3, 250, 40], False), (64, (1, 3, 5), [513, 700], False),
                                              (32, (1, 3, 5), [700, 90], True), (64, (1, 3), [260, 31], True), (32, (2,), [500], False)])
def test_hifigan_resblock_fused_f32(cuda, lib, C, dils, lens, mrf):
    """The f32 whole-ResBlock launch (k = 3, C = 32 / 64; round 3) against the fp64 chain of units and against the per-unit f32 launches."""
    from jatts_amd import hip
    k = 3
    g = torch.Generator().manual_seed(C * 10 + len(dils))
    R = sum(lens)
    x = torch.randn(R, C, generator=g)
    ws = [(torch.randn(C, C, k, generator=g) / math.sqrt(C * k), torch.randn(C, generator=g) * 0.1,
           torch.randn(C, C, k, generator=g) * 0.5 / math.sqrt(C * k), torch.randn(C, generator=g) * 0.1) for _ in dils]
    adds = [torch.randn(R, C, generator=g) for _ in range(2)] if mrf else None
    ref = x.double()
    for (w1, b1, w2, b2), d in zip(ws, dils):
        ref = _ref_unit(ref, w1, b1, w2, b2, lens, k, d, 0.1, False)
    if mrf:
        ref = (ref + adds[0].double() + adds[1].double()) / 3.0
    rb = _ragged(lens, cuda)
    xd = x.to(cuda)
    packed = [(hip.pack_conv_weight(w1.to(cuda), hip.F32, 32), b1.to(cuda), hip.pack_conv_weight(w2.to(cuda), hip.F32, 32), b2.to(cuda), d)
              for (w1, b1, w2, b2), d in zip(ws, dils)]
    y = torch.full_like(xd, float("nan"))
    addd = [a.to(cuda) for a in adds] if mrf else None
    hip.hifigan_resblock(rb, 1, xd, y, packed, C, k, 0.1, hip.F32, add=addd, out_scale=1.0 / 3.0 if mrf else 1.0)
    torch.cuda.synchronize()
    assert torch.isfinite(y).all(), "unwritten / non-finite outputs"
    e = relerr(y, ref)
    assert e <= TOL["fp32"], f"resblock f32 C={C} dils={dils}: rel err {e:.3e}"
    cur, bufs = xd, [torch.empty_like(xd), torch.empty_like(xd)]
    for i, (w1, b1, w2, b2, d) in enumerate(packed):
        lastu = i == len(packed) - 1
        hip.hifigan_resunit(rb, 1, cur, bufs[i & 1], w1, b1, w2, b2, C, k, d, 0.1, hip.F32,
                            add=addd if (mrf and lastu) else None, out_scale=1.0 / 3.0 if (mrf and lastu) else 1.0)
        cur = bufs[i & 1]
    assert relerr(y, cur.double()) <= 1e-5
    with pytest.raises(Exception):      # k = 7 chains are not fused at f32 (halo 36 rows a side): UNSUPPORTED, callers issue units
        w7 = hip.pack_conv_weight(torch.randn(C, C, 7, generator=g).to(cuda), hip.F32, 32)
        hip.hifigan_resblock(rb, 1, xd, torch.empty_like(xd), [(w7, packed[0][1], w7, packed[0][1], d) for d in (1, 3, 5)], C, 7, 0.1, hip.F32)


def test_hifigan_resblock_refuses_wide_receptive_fields(cuda, lib):
    from jatts_amd import hip
    from jatts_amd._abi import JattsHipError
    C, k = 32, 15
    rb = _ragged([400], cuda)
    x = torch.zeros(400, C, device=cuda, dtype=torch.float16)
    w = hip.pack_conv_weight(torch.zeros(C, C, k, device=cuda), hip.F16, 32)
    b = torch.zeros(C, device=cuda)
    with pytest.raises(JattsHipError):
        hip.hifigan_resblock(rb, 1, x, torch.empty_like(x), [(w, b, w, b, d) for d in (1, 3, 5)], C, k, 0.1, hip.F16)


def test_hifigan_resunit_fused_mrf_mean(cuda, lib):
    """y = (unit(x) + add0 + add1) / 3 written by the unit's coalesced output pass."""
    from jatts_amd import hip
    g = torch.Generator().manual_seed(11)
    lens, C, k, d = [300, 77], 64, 7, 3
    R = sum(lens)
    x, a0, a1 = (torch.randn(R, C, generator=g).half().float() for _ in range(3))
    w1 = (torch.randn(C, C, k, generator=g) / math.sqrt(C * k)).half().float()
    w2 = (torch.randn(C, C, k, generator=g) / math.sqrt(C * k)).half().float()
    b1, b2 = torch.randn(C, generator=g) * 0.1, torch.randn(C, generator=g) * 0.1
    unit = _ref_unit(x, w1, b1, w2, b2, lens, k, d, 0.1, True)
    ref = (unit.half().double() + a0.double() + a1.double()) / 3.0
    rb = _ragged(lens, cuda)
    y = torch.empty(R, C, device=cuda, dtype=torch.float16)
    hip.hifigan_resunit(rb, 1, x.to(cuda).half(), y, hip.pack_conv_weight(w1.to(cuda), hip.F16, 32), b1.to(cuda),
                        hip.pack_conv_weight(w2.to(cuda), hip.F16, 32), b2.to(cuda), C, k, d, 0.1, hip.F16,
                        add=[a0.to(cuda).half(), a1.to(cuda).half()], out_scale=1.0 / 3.0)
    assert relerr(y.float(), ref) <= TOL["fp16"]


def test_hifigan_resunit_len_mul(cuda, lib):
    """len_mul scales the ragged geometry (HiFi-GAN stages reuse one cu_rows array)."""
    from jatts_amd import hip
    g = torch.Generator().manual_seed(3)
    lens, mul, C, k, d = [5, 9], 8, 32, 3, 3
    R = sum(lens) * mul
    x = torch.randn(R, C, generator=g)
    w1, w2 = torch.randn(C, C, k, generator=g) * 0.1, torch.randn(C, C, k, generator=g) * 0.1
    b1, b2 = torch.randn(C, generator=g), torch.randn(C, generator=g)
    ref = _ref_unit(x, w1, b1, w2, b2, [n * mul for n in lens], k, d, 0.1, False)
    rb = _ragged(lens, cuda)
    y = torch.empty(R, C, device=cuda)
    hip.hifigan_resunit(rb, mul, x.to(cuda), y, hip.pack_conv_weight(w1.to(cuda), hip.F32, 32), b1.to(cuda),
                        hip.pack_conv_weight(w2.to(cuda), hip.F32, 32), b2.to(cuda), C, k, d, 0.1, hip.F32)
    assert relerr(y, ref) <= TOL["fp32"]


@pytest.mark.parametrize("prec", ["fp32", "fp32_split", "fp16"])     # fp32_split (round 4): f32 tensors, split hi / lo MFMA operands, the f32 tolerance
@pytest.mark.parametrize("pad_vt", [False, True])   # True: RaggedBatch.vt_layout -> aligned 16-byte V^T staging
@pytest.mark.parametrize("H,dk,lens,rel", [(2, 32, [24, 9, 33], True), (2, 192, [130, 64], True),
                                          (2, 96, [65], True), (4, 64, [100, 1, 17], False),
                                          (2, 64, [3, 70, 5, 129], True), (1, 256, [200, 33], True), (2, 128, [70, 31], False),
                                          # d_k 256 without a bias (Matcha's blocks): the half-tile pipeline at f32, 512-thread workgroups in the split mode
                                          (2, 256, [300, 77, 130], False)])
def test_relpos_attention(cuda, lib, prec, pad_vt, H, dk, lens, rel):
    from jatts_amd import hip
    from oracle.fs2_oracle import rel_shift_legacy
    g = torch.Generator().manual_seed(H * dk + len(lens))
    A, R, Tm = H * dk, sum(lens), max(lens)
    ldg = hip.round_up(Tm, 32)
    q = _round(torch.randn(R, A, generator=g), prec)
    k = _round(torch.randn(R, A, generator=g), prec)
    v = _round(torch.randn(R, A, generator=g), prec)
    gm = _round(torch.randn(R, H, ldg, generator=g), prec)  # g[row][h][m]
    ku = torch.randn(R, H, generator=g)
    scale = 1.0 / math.sqrt(dk)
    outs, o = [], 0
    for T in lens:
        qs, ks, vs = (t[o:o + T].view(T, H, dk).transpose(0, 1).double() for t in (q, k, v))
        s = qs @ ks.transpose(1, 2) + ku[o:o + T].t().double().unsqueeze(1)
        if rel:
            s = s + rel_shift_legacy(gm[o:o + T, :, :T].permute(1, 0, 2).double())
        p = torch.softmax(s * scale, -1)
        if prec == "fp16":
            pass  # P is rounded to f16 inside the kernel: covered by the tolerance
        outs.append((p @ vs).transpose(0, 1).reshape(T, A))
        o += T
    ref = torch.cat(outs)
    dt = hip.F32S if prec == "fp32_split" else _dt(prec)
    tdt = hip.torch_dtype(dt)
    rb = _ragged(lens, cuda)
    vcol, ldvt = rb.vt_layout() if pad_vt else (None, R)
    vt = torch.full((A, ldvt), float("nan"), dtype=tdt, device=cuda)   # slack columns must never be used
    o = 0
    for b, T in enumerate(lens):
        c0 = int(vcol[b]) if pad_vt else o
        vt[:, c0:c0 + T] = v[o:o + T].t().to(cuda).to(tdt)
        o += T
    out = hip.relpos_attention(rb, q.to(cuda).to(tdt), A, k.to(cuda).to(tdt), A, vt, ldvt,
                               gm.reshape(R, H * ldg).to(cuda).to(tdt) if rel else None, ldg,
                               ku.to(cuda), scale, H, dk, dt, vt_col0=vcol)
    e = relerr(out.float(), ref)
    assert e <= (3e-3 if prec == "fp16" else 5e-5), f"attention {H}x{dk} {prec}: rel err {e:.3e}"
    if prec == "fp32_split":      # next to the exact-f32 kernel on the same inputs (not narrower: at most twice its error), and batch independence
        o32 = hip.relpos_attention(rb, q.to(cuda), A, k.to(cuda), A, vt, ldvt, gm.reshape(R, H * ldg).to(cuda) if rel else None, ldg,
                                   ku.to(cuda), scale, H, dk, hip.F32, vt_col0=vcol)
        assert e <= max(2.0 * relerr(o32, ref), 2e-6), (e, relerr(o32, ref))
        if len(lens) > 1 and pad_vt:
            T0 = lens[0]
            rb0 = _ragged([T0], cuda)
            vc0, ld0 = rb0.vt_layout()
            vt0 = torch.zeros(A, ld0, device=cuda)
            vt0[:, :T0] = v[:T0].t().to(cuda)
            o0 = hip.relpos_attention(rb0, q[:T0].contiguous().to(cuda), A, k[:T0].contiguous().to(cuda), A, vt0, ld0,
                                      gm[:T0].reshape(T0, H * ldg).contiguous().to(cuda) if rel else None, ldg, ku[:T0].contiguous().to(cuda), scale, H, dk,
                                      hip.F32S, vt_col0=vc0)
            assert torch.equal(o0, out[:T0])


@pytest.mark.parametrize("pad_vt", [False, True])
@pytest.mark.parametrize("lens", [[300, 77, 130], [64, 1, 33, 768], [31]])
def test_relpos_attention_bias_free_dk256(cuda, lib, pad_vt, lens):
    """Matcha's plain attention (no rel-pos bias, no u . k term): the exact-f32 d_k 256 instantiation with the half-tile pipeline, the
    pipelined LDS fragment reads and the store-time V^T mask -- partial last tiles, one-row sequences, both V^T alignments, NaN in every
    slack column, against float64 softmax attention."""
    from jatts_amd import hip
    H, dk = 2, 256
    g = torch.Generator().manual_seed(len(lens) + sum(lens))
    A, R = H * dk, sum(lens)
    q, k, v = (torch.randn(R, A, generator=g) for _ in range(3))
    scale = 1.0 / math.sqrt(dk)
    outs, o = [], 0
    for T in lens:
        qs, ks, vs = (t[o:o + T].view(T, H, dk).transpose(0, 1).double() for t in (q, k, v))
        outs.append((torch.softmax(qs @ ks.transpose(1, 2) * scale, -1) @ vs).transpose(0, 1).reshape(T, A))
        o += T
    ref = torch.cat(outs)
    rb = _ragged(lens, cuda)
    vcol, ldvt = rb.vt_layout() if pad_vt else (None, R)
    vt = torch.full((A, ldvt), float("nan"), device=cuda)
    o = 0
    for b, T in enumerate(lens):
        c0 = int(vcol[b]) if pad_vt else o
        vt[:, c0:c0 + T] = v[o:o + T].t().to(cuda)
        o += T
    out = hip.relpos_attention(rb, q.to(cuda), A, k.to(cuda), A, vt, ldvt, None, 0, None, scale, H, dk, hip.F32, vt_col0=vcol)
    assert torch.isfinite(out).all()
    e = relerr(out, ref)
    assert e <= 5e-6, f"bias-free d_k 256 attention: rel err {e:.3e}"


def test_relpos_attention_refuses_offsets_beyond_32_bits(cuda, lib):
    """The tile loads use 32-bit buffer offsets per (utterance, head): a V^T row stride that would overflow them is refused loudly
    (JATTS_ERR_UNSUPPORTED) before any memory is touched."""
    from jatts_amd import hip
    from jatts_amd._abi import JattsHipError
    rb = _ragged([8], cuda)
    q = torch.zeros(8, 512, device=cuda)
    vt = torch.zeros(256, 8, device=cuda)
    with pytest.raises(JattsHipError, match="4 GiB"):
        hip.relpos_attention(rb, q, 512, q, 512, vt, 1 << 23, None, 0, None, 1.0, 1, 256, hip.F32, q_col0=0, k_col0=256)


@pytest.mark.parametrize("prec", ["fp32", "fp16"])
def test_relpos_attention_new_style(cuda, lib, prec):
    """rel_mode 2 (RelPositionMultiHeadedAttention, attention.py:237-261): BD'[i,j] = g[i][center - i + j]."""
    from jatts_amd import hip
    from oracle.vits_oracle import rel_shift_new
    g = torch.Generator().manual_seed(21)
    H, dk, lens, cap = 2, 32, [40, 17, 64], 128
    A, R = H * dk, sum(lens)
    ncol = 2 * cap - 1
    ldg = hip.round_up(ncol, 32)
    q, k, v = (_round(torch.randn(R, A, generator=g), prec) for _ in range(3))
    gm = _round(torch.randn(R, H, ldg, generator=g), prec)
    ku = torch.randn(R, H, generator=g)
    scale = 1.0 / math.sqrt(dk)
    outs, o = [], 0
    for T in lens:
        qs, ks, vs = (t[o:o + T].view(T, H, dk).transpose(0, 1).double() for t in (q, k, v))
        # columns cap-T .. cap+T-2 of the wide table are this utterance's 2T-1 relative positions
        bd = gm[o:o + T, :, cap - T: cap + T - 1].permute(1, 0, 2).double()
        s = qs @ ks.transpose(1, 2) + ku[o:o + T].t().double().unsqueeze(1) + rel_shift_new(bd)
        outs.append((torch.softmax(s * scale, -1) @ vs).transpose(0, 1).reshape(T, A))
        o += T
    ref = torch.cat(outs)
    dt = _dt(prec)
    tdt = hip.torch_dtype(dt)
    out = hip.relpos_attention(_ragged(lens, cuda), q.to(cuda).to(tdt), A, k.to(cuda).to(tdt), A,
                               v.t().contiguous().to(cuda).to(tdt), R, gm.reshape(R, H * ldg).to(cuda).to(tdt), ldg,
                               ku.to(cuda), scale, H, dk, dt, rel_mode=2, rel_center=cap - 1)
    assert relerr(out.float(), ref) <= (5e-5 if prec == "fp32" else 3e-3)


def test_gated_activation_and_flip(cuda, lib):
    from jatts_amd import hip
    g = torch.Generator().manual_seed(22)
    lens, C = [9, 30], 48
    R = sum(lens)
    x = torch.randn(R, 2 * C, generator=g)
    gs = torch.randn(2, 2 * C, generator=g)
    rb = _ragged(lens, cuda)
    y = hip.gated_tanh_sigmoid(rb, x.to(cuda), gs.to(cuda), C, hip.F32)
    xg = x.clone().double()
    xg[:9] += gs[0].double()
    xg[9:] += gs[1].double()
    assert maxdiff(y, torch.tanh(xg[:, :C]) * torch.sigmoid(xg[:, C:])) <= 1e-5
    y0 = hip.gated_tanh_sigmoid(rb, x.to(cuda).half(), None, C, hip.F16)
    xh = x.half().double()
    assert maxdiff(y0.float(), torch.tanh(xh[:, :C]) * torch.sigmoid(xh[:, C:])) <= 2e-3
    z = torch.randn(R, C, generator=g)
    assert torch.equal(hip.flip_channels(z.to(cuda)).cpu(), torch.flip(z, [1]))


@pytest.mark.parametrize("in16,out16", [(False, False), (False, True), (True, True)])
@pytest.mark.parametrize("dim", [64, 256, 384])
def test_layernorm(cuda, lib, in16, out16, dim):
    from jatts_amd import hip
    g = torch.Generator().manual_seed(dim)
    x = torch.randn(37, dim, generator=g) * 3 + 1
    if in16:
        x = x.half().float()
    gamma, beta = torch.randn(dim, generator=g), torch.randn(dim, generator=g)
    ref = F.layer_norm(x.double(), (dim,), gamma.double(), beta.double(), 1e-12)
    xd = x.to(cuda).half() if in16 else x.to(cuda)
    y = hip.layernorm(xd, gamma.to(cuda), beta.to(cuda), hip.F16 if out16 else hip.F32)
    assert relerr(y.float(), ref) <= (1e-3 if out16 else 1e-5)


@pytest.mark.parametrize("prec", ["fp32", "fp16"])
@pytest.mark.parametrize("C,K,lens", [(64, 7, [24, 9, 70]), (384, 31, [100, 15]), (100, 15, [33])])
def test_glu_dwconv_bn_swish(cuda, lib, prec, C, K, lens):
    from jatts_amd import hip
    g = torch.Generator().manual_seed(C + K)
    R = sum(lens)
    x = _round(torch.randn(R, 2 * C, generator=g), prec)
    w = torch.randn(C, K, generator=g) * 0.3
    s, t = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    outs, o = [], 0
    for L in lens:
        h = F.glu(x[o:o + L].double(), dim=-1).t().unsqueeze(0)
        y = F.conv1d(h, w.double().unsqueeze(1), None, padding=(K - 1) // 2, groups=C)[0].t() * s.double() + t.double()
        outs.append(y * torch.sigmoid(y))
        o += L
    ref = torch.cat(outs)
    dt = _dt(prec)
    y = hip.glu_dwconv_bn_swish(_ragged(lens, cuda), x.to(cuda).to(hip.torch_dtype(dt)), C, K, w.to(cuda),
                                s.to(cuda), t.to(cuda), dt)
    assert relerr(y.float(), ref) <= (1e-5 if prec == "fp32" else 1e-3)


def test_predictor_head_and_durations(cuda, lib):
    from jatts_amd import hip
    g = torch.Generator().manual_seed(9)
    x = torch.randn(500, 256, generator=g)
    w = torch.randn(256, generator=g) * 0.1
    v, d = hip.predictor_head(x.to(cuda), w.to(cuda), 0.7, want_duration=True)
    ref = x.double() @ w.double() + 0.7
    assert maxdiff(v, ref) <= 1e-4
    # duration_predictor.py:87-90 on the kernel's own log-durations: bit-exact except within 1e-4 of a .5 boundary
    lin = torch.exp(v.cpu().double()) - 1.0
    want = torch.clamp(torch.round(lin), min=0).long()
    near = (lin - torch.floor(lin) - 0.5).abs() < 1e-4
    assert torch.equal(d.cpu()[~near], want[~near])
    assert d.dtype == torch.int64 and int(d.min()) >= 0


@pytest.mark.parametrize("alpha", [1.0, 1.5, 0.7, 2.5])
def test_length_regulator_bit_exact(cuda, lib, alpha):
    """Bit-exact against the C oracle (itself pinned on the reference KATs)."""
    from jatts_amd import hip
    from oracle import lr_oracle as LR
    g = torch.Generator().manual_seed(int(alpha * 10))
    lens = [128, 1, 37, 300, 64]
    dim = 24
    d = torch.cat([torch.randint(0, 9, (n,), generator=g) for n in lens])
    d[128] = 2  # the length-1 utterance must produce frames
    x = torch.randn(sum(lens), dim, generator=g)
    rb = _ragged(lens, cuda)
    d_eff, cum, olens, _ = hip.lr_durations(rb, d.to(cuda), alpha, zero_rule=0)
    o = 0
    want_olens = []
    for n in lens:
        de, ol = LR.effective_durations(d[o:o + n].numpy()[None], [n], alpha)
        assert np.array_equal(d_eff[o:o + n].cpu().numpy(), de[0])
        assert np.array_equal(cum[o:o + n].cpu().numpy(), np.cumsum(de[0]))
        want_olens.append(int(ol[0]))
        o += n
    assert olens.tolist() == want_olens
    rbo = _ragged(want_olens, cuda)
    out, fidx = hip.lr_gather(rb, cum, rbo, x.to(cuda), want_index=True)
    o = oo = 0
    for n, m in zip(lens, want_olens):
        idx = LR.frame_index(d_eff[o:o + n].cpu().numpy())
        assert np.array_equal(fidx[oo:oo + m].cpu().numpy(), idx)
        assert torch.equal(out[oo:oo + m].cpu(), x[o:o + n][torch.as_tensor(idx)])  # pure gather: exact
        o, oo = o + n, oo + m


def test_length_regulator_golden_kats(cuda, lib, golden_dir):
    """The reference's own LengthRegulator outputs (tests/golden/lr_kat.npz)."""
    from jatts_amd import hip
    z = np.load(golden_dir + "/lr_kat.npz")
    for n in range(int(z["n_cases"])):
        ds, alpha, xs, ref = z[f"c{n}_ds"], float(z[f"c{n}_alpha"]), z[f"c{n}_xs"], z[f"c{n}_out"]
        B, T = ds.shape
        rb = _ragged([T] * B, cuda)
        d = torch.tensor(ds).reshape(-1).to(cuda)
        d_eff, cum, olens, _ = hip.lr_durations(rb, d, alpha, zero_rule=0)
        if sum(olens.tolist()) == 0:  # the reference's BATCHED call applies the rule only when the whole batch sums to 0 (:85-94)
            d_eff, cum, olens, _ = hip.lr_durations(rb, d, alpha, zero_rule=1)
        ol = olens.tolist()
        assert max(ol) == ref.shape[1], (n, ol, ref.shape)
        x = torch.tensor(xs).reshape(B * T, -1).to(cuda)
        out = hip.lr_gather(rb, cum, _ragged(ol, cuda), x).cpu().numpy()
        o = 0
        for b in range(B):
            assert np.array_equal(out[o:o + ol[b]], ref[b, :ol[b]]), (n, b)
            assert not ref[b, ol[b]:].any()  # pad_list zero padding
            o += ol[b]


@pytest.mark.parametrize("prec", ["fp32", "fp16"])
def test_hifigan_output_conv(cuda, lib, prec):
    from jatts_amd import hip
    g = torch.Generator().manual_seed(1)
    lens, mul, C, K = [7, 3], 16, 32, 7
    R = sum(lens) * mul
    xs = [_round(torch.randn(R, C, generator=g), prec) for _ in range(3)]
    w = torch.randn(K, C, generator=g) * 0.1
    outs, o = [], 0
    for n in lens:
        L = n * mul
        a = F.leaky_relu(sum(x[o:o + L] for x in xs).double() / 3.0, 0.01).t().unsqueeze(0)
        y = torch.tanh(F.conv1d(a, w.t().double().unsqueeze(0), torch.tensor([0.3]).double(), padding=3))
        outs.append(y.reshape(-1))
        o += L
    ref = torch.cat(outs)
    dt = _dt(prec)
    y = hip.hifigan_output(_ragged(lens, cuda), mul, [x.to(cuda).to(hip.torch_dtype(dt)) for x in xs], 1.0 / 3.0,
                           0.01, C, K, w.to(cuda), 0.3, dt)
    assert maxdiff(y, ref) <= 2e-5


def test_small_rowwise_kernels(cuda, lib):
    from jatts_amd import hip
    g = torch.Generator().manual_seed(2)
    # embedding * scale
    table = torch.randn(20, 64, generator=g)
    ids = torch.randint(0, 20, (50,), generator=g)
    y = hip.embed_scale(ids.to(cuda), table.to(cuda), 8.0)
    assert torch.equal(y.cpu(), table[ids] * 8.0)
    # affine cast with zero-filled padding columns
    x = torch.randn(9, 80, generator=g)
    s, t = torch.randn(80, generator=g), torch.randn(80, generator=g)
    y = hip.affine_cast(x.to(cuda), hip.F32, scale=s.to(cuda), shift=t.to(cuda), ldy=96)
    assert maxdiff(y[:, :80], x * s + t) <= 1e-6 and not y[:, 80:].any()
    # variance embedding add, k = 1 and k = 9
    lens = [11, 4]
    rb = _ragged(lens, cuda)
    for kp in (1, 9):
        hs = torch.randn(15, 32, generator=g)
        p, e = torch.randn(15, generator=g), torch.randn(15, generator=g)
        wp, we = torch.randn(32, kp, generator=g), torch.randn(32, kp, generator=g)
        bp, be = torch.randn(32, generator=g), torch.randn(32, generator=g)
        ref, o = hs.clone().double(), 0
        for n in lens:
            for (sig, w_, b_) in ((p, wp, bp), (e, we, be)):
                c = F.conv1d(sig[o:o + n].double().view(1, 1, n), w_.double().unsqueeze(1), b_.double(),
                             padding=(kp - 1) // 2)[0].t()
                ref[o:o + n] += c
            o += n
        out = hip.variance_embed_add(rb, hs.to(cuda), p.to(cuda), wp.to(cuda), bp.to(cuda), e.to(cuda),
                                     we.to(cuda), be.to(cuda))
        assert maxdiff(out, ref) <= 1e-5
    # per-sequence vector add, rowdot
    hs = torch.randn(15, 32, generator=g)
    vec = torch.randn(2, 32, generator=g)
    out = hip.add_seq_vector(rb, hs.to(cuda), vec.to(cuda))
    ref = hs.clone()
    ref[:11] += vec[0]
    ref[11:] += vec[1]
    assert maxdiff(out, ref) <= 1e-6
    x = torch.randn(15, 64, generator=g)
    v = torch.randn(2, 32, generator=g)
    out = hip.rowdot(x.to(cuda), 64, 15, 2, 32, v.to(cuda))
    assert maxdiff(out, torch.einsum("rhd,hd->rh", x.view(15, 2, 32), v)) <= 1e-5


def test_gaussian_upsample(cuda, lib):
    from jatts_amd import hip
    g = torch.Generator().manual_seed(4)
    lens = [12, 5]
    d = torch.randint(1, 6, (17,), generator=g)
    hs = torch.randn(17, 16, generator=g)
    outs, o, olens = [], 0, []
    for n in lens:
        ds = d[o:o + n].double()
        T = int(ds.sum())
        t = torch.arange(T).double()
        c = ds.cumsum(0) - ds / 2
        p = torch.softmax(-0.1 * (t[:, None] - c[None]) ** 2, dim=1)
        outs.append(p @ hs[o:o + n].double())
        olens.append(T)
        o += n
    out = hip.gaussian_upsample(_ragged(lens, cuda), d.to(cuda), _ragged(olens, cuda), hs.to(cuda))
    assert maxdiff(out, torch.cat(outs)) <= 1e-5


@pytest.mark.parametrize("time_split", [False, True])
@pytest.mark.parametrize("in_prec,out_prec", [("fp32", "fp32"), ("fp32", "fp16"), ("fp16", "fp32"), ("fp16", "fp16")])
def test_groupnorm_mish(cuda, lib, time_split, in_prec, out_prec):
    """Block1D / ResnetBlock1D body (modules/matchatts/decoder.py:66-97): GroupNorm(8) over (C/8 x T) of each utterance,
    Mish, + per-utterance vector.  Both launch forms (one workgroup per group; time-split with merged chunk statistics)."""
    from jatts_amd import hip
    g = torch.Generator().manual_seed(31)
    lens, C, G = [130, 1, 64, 257, 65], 64, 8
    R = sum(lens)
    x = _round(torch.randn(R, C, generator=g) * 2.0 + 0.3, in_prec)
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
    add = torch.randn(len(lens), C, generator=g)
    outs, o = [], 0
    for b, T in enumerate(lens):
        xb = x[o:o + T].double().t().unsqueeze(0)                     # (1, C, T)
        yb = torch.nn.functional.group_norm(xb, G, gamma.double(), beta.double(), 1e-5)
        yb = torch.nn.functional.mish(yb) + add[b].double().view(1, C, 1)
        outs.append(yb[0].t())
        o += T
    ref = torch.cat(outs)
    rb = _ragged(lens, cuda)
    y = hip.groupnorm_mish(rb, x.to(cuda).to(hip.torch_dtype(_dt(in_prec))), C, G, gamma.to(cuda), beta.to(cuda), _dt(out_prec),
                           addvec=add.to(cuda), time_split=time_split)
    tol = 2e-5 if out_prec == "fp32" else 2e-3
    assert relerr(y.float(), ref) <= tol, relerr(y.float(), ref)


@pytest.mark.parametrize("prec", ["fp32", "fp16"])
def test_snakebeta(cuda, lib, prec):
    """SnakeBeta (modules/matchatts/transformer.py:84-102): x + sin(alpha x)^2 / beta."""
    from jatts_amd import hip
    g = torch.Generator().manual_seed(32)
    R, C = 203, 48
    x = _round(torch.randn(R, C, generator=g) * 3, prec)
    alpha, inv_beta = torch.rand(C, generator=g) * 2 + 0.1, torch.rand(C, generator=g) + 0.2
    ref = x.double() + inv_beta.double() * torch.sin(x.double() * alpha.double()) ** 2
    y = hip.snakebeta(x.to(cuda).to(hip.torch_dtype(_dt(prec))), alpha.to(cuda), inv_beta.to(cuda))
    assert relerr(y.float(), ref) <= (2e-6 if prec == "fp32" else 1e-3)


def test_snakebeta_argument_range(cuda, lib):
    """The activation's own sin^2 (csrc/common.h sin2_f: Cody-Waite reduction by pi/2 + the Cephes kernels, no libm slow path):
    absolute error <= 3e-7 for arguments up to 1e5 in magnitude, exact zeros / NaN propagation at the edges."""
    from jatts_amd import hip
    g = torch.Generator().manual_seed(33)
    mags = torch.tensor([1e-3, 1.0, 10.0, 1e2, 1e3, 1e4, 1e5])
    x = (torch.randn(4096, 8, generator=g) * mags[torch.randint(0, 7, (4096, 1), generator=g)]).contiguous()
    x[0] = torch.tensor([0.0, -0.0, math.pi / 2, -math.pi / 2, math.pi, 1e5, -1e5, float("nan")])
    one = torch.ones(8)
    y = hip.snakebeta(x.to(cuda), one.to(cuda), one.to(cuda)).cpu()
    ref = x.double() + torch.sin(x.float().double()) ** 2
    ok = torch.isfinite(x)
    assert float((y.double() - ref)[ok].abs().max() - 0) <= 3e-7 + float((ref[ok].abs() * 6e-8).max())
    assert torch.isnan(y[0, 7]) and y[0, 0] == 0 and y[0, 1] == 0


@pytest.mark.parametrize("mode", ["fp32", "fp32_v3", "fp32_v5", "fp32_lds", "fp32_split", "fp16"])
def test_conv1d_snakebeta_epilogue(cuda, lib, mode):
    """JATTS_ACT_SNAKEBETA (round 4): Matcha's feed-forward activation in the epilogue of the conv that feeds it
    (modules/matchatts/transformer.py:84-102 after ff.net.0.proj).  Same arithmetic as conv1d followed by the snakebeta op: bit-identical
    in the f32 / split modes (every kernel family that can carry it), and closer to fp64 than the two-launch f16 path (no f16 rounding in
    between)."""
    from jatts_amd import hip
    g = torch.Generator().manual_seed(34)
    c_in, n_out, lens = 256, 1024, [70, 300, 129]
    R = sum(lens)
    x = torch.randn(R, c_in, generator=g)
    w = torch.randn(n_out, c_in, 1, generator=g) / math.sqrt(c_in)
    b = torch.randn(n_out, generator=g)
    la, lb = torch.randn(n_out, generator=g) * 0.5, torch.randn(n_out, generator=g) * 0.5
    a, ib = torch.exp(la), 1.0 / (torch.exp(lb) + 1e-9)
    u = _ref_conv(_round(x, "fp16" if mode == "fp16" else "fp32"), _round(w, "fp16" if mode == "fp16" else "fp32"), b, lens, 1, 0, 1, None, None)
    ref = u + ib.double() * torch.sin(u * a.double()) ** 2
    rb = _ragged(lens, cuda)
    dt = hip.F16 if mode == "fp16" else hip.F32
    xd = x.to(cuda).to(hip.torch_dtype(dt))
    wp = hip.SplitWeight(w.to(cuda), 64) if mode == "fp32_split" else hip.pack_conv_weight(w.to(cuda), dt)
    kw = dict(dtype=dt, bias=b.to(cuda), variant={"fp32_lds": 1, "fp32_v3": 3, "fp32_v5": 5}.get(mode, 0))   # both register-streamed tiles + the LDS-staged kernel
    ad, ibd = a.to(cuda), ib.to(cuda)
    y = hip.conv1d(rb, xd, wp, c_in, n_out, 1, snake=(ad, ibd), **kw)
    two = hip.snakebeta(hip.conv1d(rb, xd, wp, c_in, n_out, 1, **kw), ad, ibd)
    torch.cuda.synchronize()
    if mode == "fp16":
        assert relerr(y.float(), ref) <= relerr(two.float(), ref) * 1.05 <= TOL["fp16"]
    else:
        assert torch.equal(y, two)
        assert relerr(y, ref) <= TOL["fp32"]
    from jatts_amd._abi import JattsHipError
    with pytest.raises(JattsHipError):       # n_out % 4: refused loudly
        hip.conv1d(rb, xd, wp, c_in, n_out - 2, 1, snake=(ad[:-2].contiguous(), ibd[:-2].contiguous()), dtype=dt, bias=b.to(cuda))
    with pytest.raises(ValueError):
        hip.conv1d(rb, xd, wp, c_in, n_out, 1, snake=(ad, ibd), act=hip.ACT_RELU, **kw)


@pytest.mark.parametrize("n_seq", [3, 70, 150])
def test_ragged_1d_grids_equal_rectangular(cuda, lib, n_seq):
    """Round 5: for a non-uniform batch the MFMA conv / fused-unit kernels launch a 1-D grid over exactly the real tiles (jatts_ragged.host_lens;
    csrc/common.h: ragged_locate) instead of n_seq x tiles-of-the-longest.  Same tile geometry inside every sequence => bit-identical outputs, for more
    than 64 sequences (the lookup scans 64 at a time), with empty sequences, in every arithmetic; uniform batches never take the 1-D form."""
    from jatts_amd import hip
    g = torch.Generator().manual_seed(n_seq)
    lens = [int(v) for v in torch.randint(0, 400, (n_seq,), generator=g)]
    lens[1] = 0
    lens[-1] = 777
    rb = _ragged(lens, cuda)
    assert rb.struct().host_lens and not _ragged([5] * n_seq, cuda).struct().host_lens
    R, C, k, d = sum(lens), 64, 7, 3
    x = torch.randn(R, C, generator=g).to(cuda)
    w1, w2 = (torch.randn(C, C, k, generator=g) / math.sqrt(C * k)).to(cuda), (torch.randn(C, C, k, generator=g) / math.sqrt(C * k)).to(cuda)
    b = torch.zeros(C, device=cuda)
    wc = (torch.randn(192, C, 3, generator=g) / math.sqrt(3 * C)).to(cuda)

    def run():
        outs = []
        for dt, pack in ((hip.F32, lambda w: hip.pack_conv_weight(w, hip.F32, 32)), (hip.F32E, lambda w: hip.pack_conv_weight_bf16x3(w, 32)),
                         (hip.F32E6, lambda w: hip.pack_conv_weight_bf16x3(w, 32))):
            y = torch.full_like(x, float("nan"))
            hip.hifigan_resunit(rb, 1, x, y, pack(w1), b, pack(w2), b, C, k, d, 0.1, dt)
            outs.append(y)
        ws1, is1 = hip.pack_conv_weight_split(w1, 32)
        ws2, is2 = hip.pack_conv_weight_split(w2, 32)
        y = torch.full_like(x, float("nan"))
        hip.hifigan_resunit(rb, 1, x, y, ws1, b, ws2, b, C, k, d, 0.1, hip.F32S, ws=(is1, is2))
        outs.append(y)
        outs.append(hip.conv1d(rb, x, hip.pack_conv_weight(wc, hip.F32), C, 192, 3, dtype=hip.F32, act=hip.ACT_RELU))
        outs.append(hip.conv1d(rb, x, hip.pack_conv_weight_bf16x3(wc, 64), C, 192, 3, dtype=hip.F32E, act=hip.ACT_RELU))
        outs.append(hip.conv1d(rb, x.half(), hip.pack_conv_weight(wc, hip.F16), C, 192, 3, dtype=hip.F16, act=hip.ACT_RELU).float())
        torch.cuda.synchronize()
        return outs
    one_d = run()
    prev, hip._RAGGED_1D = hip._RAGGED_1D, False
    try:
        assert not rb.struct().host_lens
        rect = run()
    finally:
        hip._RAGGED_1D = prev
    for a, c in zip(one_d, rect):
        assert torch.isfinite(a).all() and torch.equal(a, c)


@pytest.mark.parametrize("mode", ["F32", "F16", "F32S", "F32E", "F32E6"])
@pytest.mark.parametrize("geom", [(384, 768, 384, [130, 1, 77, 768]), (512, 1024, 512, [300, 64]), (256, 256, 128, [40, 200, 5])],
                         ids=["fs2-384", "matcha-512", "small-256"])
@pytest.mark.parametrize("variant", [0, 1, 3])
def test_conv1d_two_outputs_equal_two_launches(cuda, lib, mode, geom, variant):
    """jatts_conv_desc.n_split (round 6): the Q | K | V projection as ONE launch -- channels < n_split row-major, the rest transposed into the attention
    kernel's V^T layout -- against the two launches it replaces (reference: three Linear calls, modules/transformer/attention.py:39-61;
    modules/matchatts/transformer.py:222-260): every element is the same contraction, so the outputs are BIT-IDENTICAL in every arithmetic and
    kernel variant; V^T slack columns stay untouched."""
    from jatts_amd import hip
    c_in, n_split, n_v, lens = geom
    dt = getattr(hip, mode)
    g = torch.Generator().manual_seed(c_in + n_v)
    R = sum(lens)
    tdt = hip.torch_dtype(dt)
    x = (torch.randn(R, c_in, generator=g) * 0.7).to(cuda).to(tdt)
    w = (torch.randn(n_split + n_v, c_in, 1, generator=g) / c_in ** 0.5).to(cuda)
    b = torch.randn(n_split + n_v, generator=g).to(cuda)
    kw = {}

    def pack(wt):
        if dt == hip.F32S:
            p, inv = hip.pack_conv_weight_split(wt, 64)
            return p, {"w_inv": inv}
        if dt in hip.EMUL:
            return hip.pack_conv_weight_bf16x3(wt, 64), {}
        return hip.pack_conv_weight(wt, dt), {}
    rb = _ragged(lens, cuda)
    vcol, ldvt = rb.vt_layout()
    f32o = dt != hip.F16
    (wa, ka), (wq, kq), (wv, kv) = pack(w), pack(w[:n_split]), pack(w[n_split:])
    qk1 = hip.conv1d(rb, x, wq, c_in, n_split, 1, dtype=dt, bias=b[:n_split].contiguous(), out_f32=f32o, variant=variant, **kq)
    vt1 = torch.full((n_v, ldvt), 7.0, dtype=qk1.dtype, device=cuda)
    hip.conv1d(rb, x, wv, c_in, n_v, 1, dtype=dt, bias=b[n_split:].contiguous(), transposed=True, out=vt1, out_ld=ldvt, y_seq_col0=vcol, out_f32=f32o,
               variant=variant, **kv)
    qk2, vt2 = hip.conv1d(rb, x, wa, c_in, n_split + n_v, 1, dtype=dt, bias=b, out_f32=f32o, variant=variant, split=(n_split, ldvt, vcol), **ka)
    # (the split-f16 arithmetic scales its activation tiles by the block maximum, and the tile may follow the launch's channel count: same values to
    # within its own rounding there, bit-identical everywhere else)
    same = (lambda a, c: relerr(a.float(), c.float()) <= 2e-6) if dt == hip.F32S else torch.equal
    assert qk2.shape == qk1.shape and same(qk1, qk2), f"{mode}: Q | K of the one-launch form differs"
    o = 0
    for i, T in enumerate(lens):
        c0 = int(vcol[i])
        assert same(vt1[:, c0:c0 + T], vt2[:, c0:c0 + T]), f"{mode}: V^T of sequence {i} differs"
        ref = (x[o:o + T].double() @ w[n_split:, :, 0].double().t() + b[n_split:].double()).t()
        assert relerr(vt2[:, c0:c0 + T].float(), ref) <= (3e-3 if dt == hip.F16 else 1e-5)
        o += T


@pytest.mark.parametrize("mode", ["F32E", "F32E6"])
@pytest.mark.parametrize("geom", [(384, 768, 384, [768] * 8), (512, 1024, 512, [768] * 8), (512, 1024, 512, [768]), (256, 512, 256, [700, 768, 64, 768, 768, 31, 768, 768])],
                         ids=["fs2-384-batch", "matcha-512-batch", "matcha-512-one", "256-ragged"])
def test_conv1d_two_outputs_product_tiles(cuda, lib, mode, geom):
    """The Q | K | V launch on the PRODUCT form of the emulated conv (w_layout = 1, 16 x 16 x 32 kernels) at launch sizes that take its wide tiles (round 6:
    384 n x 64 t / 256 n x 64 t when the launch fills the chip, 128 n x 32 t for one utterance): a workgroup writes ONE of the two outputs, so n_split has to be
    a whole number of its tiles -- 1024 is not a multiple of 384 (config 3's 512 -> 1536: the 384-wide tile must not take it; it did for an hour, and only
    `test_matcha_bench_utterance_matches_the_reference` in a batch of 8 noticed).  Bit-identical to the two launches it replaces."""
    from jatts_amd import hip
    c_in, n_split, n_v, lens = geom
    dt = getattr(hip, mode)
    g = torch.Generator().manual_seed(c_in + n_v + len(lens))
    R = sum(lens)
    x = (torch.randn(R, c_in, generator=g) * 0.7).to(cuda)
    w = (torch.randn(n_split + n_v, c_in, 1, generator=g) / c_in ** 0.5).to(cuda)
    b = torch.randn(n_split + n_v, generator=g).to(cuda)
    pack = lambda wt: hip.pack_conv_weight_bf16x3_k32(wt, 64)  # noqa: E731
    rb = _ragged(lens, cuda)
    vcol, ldvt = rb.vt_layout()
    qk1 = hip.conv1d(rb, x, pack(w[:n_split]), c_in, n_split, 1, dtype=dt, bias=b[:n_split].contiguous(), out_f32=True, w_layout=1)
    vt1 = torch.full((n_v, ldvt), 7.0, dtype=qk1.dtype, device=cuda)
    hip.conv1d(rb, x, pack(w[n_split:]), c_in, n_v, 1, dtype=dt, bias=b[n_split:].contiguous(), transposed=True, out=vt1, out_ld=ldvt, y_seq_col0=vcol,
               out_f32=True, w_layout=1)
    qk2, vt2 = hip.conv1d(rb, x, pack(w), c_in, n_split + n_v, 1, dtype=dt, bias=b, out_f32=True, split=(n_split, ldvt, vcol), w_layout=1)
    assert qk2.shape == qk1.shape and torch.equal(qk1, qk2), f"{mode}: Q | K of the one-launch form differs"
    o = 0
    for i, T in enumerate(lens):
        c0 = int(vcol[i])
        assert torch.equal(vt1[:, c0:c0 + T], vt2[:, c0:c0 + T]), f"{mode}: V^T of sequence {i} differs"
        ref = (x[o:o + T].double() @ w[n_split:, :, 0].double().t() + b[n_split:].double()).t()
        assert relerr(vt2[:, c0:c0 + T].float(), ref) <= 1e-5
        o += T
    ref_qk = x.double() @ w[:n_split, :, 0].double().t() + b[:n_split].double()
    assert relerr(qk2.float(), ref_qk) <= 1e-5


def test_conv1d_two_outputs_argument_checks(cuda, lib):
    from jatts_amd import hip
    rb = _ragged([10], cuda)
    x = torch.zeros(10, 64, device=cuda)
    w = hip.pack_conv_weight(torch.zeros(384, 64, 1, device=cuda), hip.F32)
    with pytest.raises(ValueError):
        hip.conv1d(rb, x, w, 64, 384, 1, dtype=hip.F32, split=(128, 16, None))        # not a multiple of 256
    with pytest.raises(ValueError):
        hip.conv1d(rb, x, w, 64, 384, 1, dtype=hip.F32, split=(512, 16, None))        # beyond n_out
