"""SURVEY §8 f.4, second slice: every HIP forward / backward pair of jatts_amd/autograd.py against torch autograd of the
reference module's own torch ops in fp64 on the CPU (same seeded inputs, padded-batch geometry where the reference pads)."""
import math

import pytest
import torch
import torch.nn.functional as F

from helpers import maxdiff, relerr

pytestmark = pytest.mark.gpu
TOL = 3e-5


def _leaf(t, cuda):
    return t.clone().double().requires_grad_(), t.clone().to(cuda).requires_grad_()


def _check(pairs, tol=TOL):
    for name, got, want in pairs:
        e = relerr(got.detach().cpu().double(), want.detach().double())
        assert e <= tol, (name, e)


@pytest.mark.parametrize("rows,dim", [(37, 384), (5, 64), (130, 256), (9, 80)])
def test_layernorm_backward(cuda, lib, rows, dim):
    from jatts_amd import autograd as A
    g = torch.Generator().manual_seed(rows + dim)
    x, w, b, gy = torch.randn(rows, dim, generator=g) * 2 + 0.3, torch.randn(dim, generator=g), torch.randn(dim, generator=g), torch.randn(rows, dim, generator=g)
    (xr, xd), (wr, wd), (br, bd) = _leaf(x, cuda), _leaf(w, cuda), _leaf(b, cuda)
    yr = F.layer_norm(xr, (dim,), wr, br, 1e-12)
    yr.backward(gy.double())
    y = A.LayerNorm.apply(xd, wd, bd, 1e-12)
    y.backward(gy.to(cuda))
    _check([("y", y, yr), ("dx", xd.grad, xr.grad), ("dg", wd.grad, wr.grad), ("db", bd.grad, br.grad)])


@pytest.mark.parametrize("mode", ["relu", "tanh", "swish"])
def test_activation_backward(cuda, lib, mode):
    from jatts_amd import autograd as A
    g = torch.Generator().manual_seed(3)
    x, gy = torch.randn(77, 50, generator=g) * 2, torch.randn(77, 50, generator=g)
    xr, xd = _leaf(x, cuda)
    fn = {"relu": torch.relu, "tanh": torch.tanh, "swish": lambda t: t * torch.sigmoid(t)}[mode]
    yr = fn(xr)
    yr.backward(gy.double())
    y = A.Act.apply(xd, mode)
    y.backward(gy.to(cuda))
    _check([("y", y, yr), ("dx", xd.grad, xr.grad)])


def test_glu_backward(cuda, lib):
    from jatts_amd import autograd as A
    g = torch.Generator().manual_seed(4)
    x, gy = torch.randn(41, 2 * 48, generator=g), torch.randn(41, 48, generator=g)
    xr, xd = _leaf(x, cuda)
    yr = F.glu(xr, dim=1)
    yr.backward(gy.double())
    y = A.GLU.apply(xd)
    y.backward(gy.to(cuda))
    _check([("y", y, yr), ("dx", xd.grad, xr.grad)])


@pytest.mark.parametrize("dim,k,lens", [(64, 7, [50, 50]), (96, 31, [40, 40, 40]), (32, 3, [9, 70])])
def test_depthwise_conv_backward(cuda, lib, dim, k, lens):
    from jatts_amd import autograd as A, hip
    g = torch.Generator().manual_seed(dim + k)
    R = sum(lens)
    x, w, b, gy = torch.randn(R, dim, generator=g), torch.randn(dim, 1, k, generator=g) / math.sqrt(k), torch.randn(dim, generator=g), torch.randn(R, dim, generator=g)
    (xr, xd), (wr, wd), (br, bd) = _leaf(x, cuda), _leaf(w, cuda), _leaf(b, cuda)
    outs, o = [], 0
    for n in lens:
        outs.append(F.conv1d(xr[o:o + n].t().unsqueeze(0), wr, br, padding=(k - 1) // 2, groups=dim)[0].t())
        o += n
    yr = torch.cat(outs)
    yr.backward(gy.double())
    y = A.DepthwiseConv.apply(xd, wd, bd, hip.RaggedBatch(lens, cuda))
    y.backward(gy.to(cuda))
    _check([("y", y, yr), ("dx", xd.grad, xr.grad), ("dw", wd.grad, wr.grad), ("db", bd.grad, br.grad)])


def test_batchnorm_train_backward_and_running_stats(cuda, lib):
    from jatts_amd import autograd as A
    g = torch.Generator().manual_seed(6)
    B, T, dim = 3, 29, 48
    x, w, b, gy = torch.randn(B * T, dim, generator=g) * 1.5 + 0.7, torch.randn(dim, generator=g), torch.randn(dim, generator=g), torch.randn(B * T, dim, generator=g)
    (xr, xd), (wr, wd), (br, bd) = _leaf(x, cuda), _leaf(w, cuda), _leaf(b, cuda)
    bn = torch.nn.BatchNorm1d(dim).double().train()
    with torch.no_grad():
        bn.weight.copy_(w)
        bn.bias.copy_(b)
    yr = F.batch_norm(xr.view(B, T, dim).transpose(1, 2), bn.running_mean, bn.running_var, wr, br, True, 0.1, 1e-5).transpose(1, 2).reshape(B * T, dim)
    yr.backward(gy.double())
    rm, rv = torch.zeros(dim, device=cuda), torch.ones(dim, device=cuda)
    y = A.BatchNormTrain.apply(xd, wd, bd, rm, rv, 0.1, 1e-5)
    y.backward(gy.to(cuda))
    _check([("y", y, yr), ("dx", xd.grad, xr.grad), ("dg", wd.grad, wr.grad), ("db", bd.grad, br.grad),
            ("running_mean", rm, bn.running_mean), ("running_var", rv, bn.running_var)])


def test_embedding_backward(cuda, lib):
    from jatts_amd import autograd as A
    g = torch.Generator().manual_seed(7)
    ids = torch.randint(0, 20, (90,), generator=g)
    table, gy = torch.randn(20, 64, generator=g), torch.randn(90, 64, generator=g)
    tr, td = _leaf(table, cuda)
    yr = F.embedding(ids, tr, padding_idx=0) * 8.0
    yr.backward(gy.double())
    y = A.Embedding.apply(ids.to(cuda), td, 8.0, 0)
    y.backward(gy.to(cuda))
    _check([("y", y, yr), ("dtable", td.grad, tr.grad)])
    assert float(td.grad[0].abs().max()) == 0.0


def test_length_regulator_backward(cuda, lib):
    from jatts_amd import autograd as A, hip
    g = torch.Generator().manual_seed(8)
    B, Tm, dim = 3, 11, 32
    d = torch.randint(0, 5, (B, Tm), generator=g)
    d[1, 7:] = 0
    hs, To = torch.randn(B * Tm, dim, generator=g), int(d.sum(1).max())
    gy = torch.randn(B * To, dim, generator=g)
    hr, hd = _leaf(hs, cuda)
    outs = [F.pad(torch.repeat_interleave(hr.view(B, Tm, dim)[b], d[b], dim=0), (0, 0, 0, To - int(d[b].sum()))) for b in range(B)]
    yr = torch.cat(outs)
    yr.backward(gy.double())
    rb = hip.RaggedBatch([Tm] * B, cuda)
    _, cum, ol, _ = hip.lr_durations(rb, d.reshape(-1).to(cuda), 1.0, zero_rule=0)
    y = A.LengthRegulate.apply(hd, rb, cum, hip.RaggedBatch([To] * B, cuda))
    y.backward(gy.to(cuda))
    _check([("y", y, yr), ("dhs", hd.grad, hr.grad)])


def _rel_shift_legacy(x):
    """attention.py:142-162 (zero_triu False), restated on a (B, H, T, T) tensor."""
    b, h, t1, t2 = x.shape
    xp = torch.cat([torch.zeros(b, h, t1, 1, dtype=x.dtype), x], dim=-1).view(b, h, t2 + 1, t1)
    return xp[:, :, 1:].reshape(b, h, t1, t2)


@pytest.mark.parametrize("T,lens", [(24, [24, 17]), (65, [65, 3, 40])])
def test_shift_softmax_backward(cuda, lib, T, lens):
    from jatts_amd import autograd as A
    g = torch.Generator().manual_seed(T)
    B, H = len(lens), 2
    ac, bd, gp = (torch.randn(B, H, T, T, generator=g) for _ in range(3))
    (ar, ad), (br, bd_) = _leaf(ac, cuda), _leaf(bd, cuda)
    scale = 1.0 / math.sqrt(24)
    s = (ar + _rel_shift_legacy(br)) * scale
    mask = (torch.arange(T)[None, :] >= torch.tensor(lens)[:, None])[:, None, None, :]      # (B, 1, 1, T): True on padded keys
    pr = torch.softmax(s.masked_fill(mask, torch.finfo(torch.float32).min), dim=-1).masked_fill(mask, 0.0)
    pr.backward(gp.double())
    p = A.ShiftSoftmax.apply(ad, bd_, torch.tensor(lens, dtype=torch.int32, device=cuda), scale)
    p.backward(gp.to(cuda))
    _check([("p", p, pr), ("dac", ad.grad, ar.grad), ("dbd", bd_.grad, br.grad)])


def test_rank1_heads_backward(cuda, lib):
    from jatts_amd import autograd as A
    g = torch.Generator().manual_seed(9)
    x, w, b, gy = torch.randn(53, 256, generator=g), torch.randn(1, 256, generator=g) / 16, torch.randn(1, generator=g), torch.randn(53, generator=g)
    (xr, xd), (wr, wd), (br, bd) = _leaf(x, cuda), _leaf(w, cuda), _leaf(b, cuda)
    yr = F.linear(xr, wr, br).squeeze(-1)
    yr.backward(gy.double())
    y = A.RowDot.apply(xd, wd, bd)
    y.backward(gy.to(cuda))
    _check([("y", y, yr), ("dx", xd.grad, xr.grad), ("dw", wd.grad, wr.grad), ("db", bd.grad, br.grad)])
    v, w2, b2, gy2 = torch.randn(53, generator=g), torch.randn(96, 1, 1, generator=g), torch.randn(96, generator=g), torch.randn(53, 96, generator=g)
    (vr, vd), (w2r, w2d), (b2r, b2d) = _leaf(v, cuda), _leaf(w2, cuda), _leaf(b2, cuda)
    yr = F.conv1d(vr.view(1, 1, -1), w2r, b2r)[0].t()
    yr.backward(gy2.double())
    y = A.OuterRows.apply(vd, w2d, b2d)
    y.backward(gy2.to(cuda))
    _check([("y", y, yr), ("dv", vd.grad, vr.grad), ("dw", w2d.grad, w2r.grad), ("db", b2d.grad, b2r.grad)])


@pytest.mark.parametrize("kind,log_offset", [(0, -1.0), (1, -1.0), (1, 1.0)])
def test_masked_loss_backward(cuda, lib, kind, log_offset):
    from jatts_amd import autograd as A, hip
    g = torch.Generator().manual_seed(10 + kind)
    B, T, dim = 3, 14, 5 if kind == 0 else 1
    lens = [14, 9, 2]
    a, b = torch.randn(B * T, dim, generator=g), torch.rand(B * T, dim, generator=g) * 4
    ar, ad = _leaf(a, cuda)
    m = (torch.arange(T)[None, :] < torch.tensor(lens)[:, None]).reshape(-1)
    tgt = torch.log(b.double() + log_offset) if log_offset >= 0 else b.double()
    sel_a, sel_b = ar[m], tgt[m]
    lr_ = (sel_a - sel_b).abs().mean() if kind == 0 else ((sel_a - sel_b) ** 2).mean()
    (lr_ * 1.7).backward()
    n = float(sum(lens) * dim)
    rb = hip.RaggedBatch([T] * B, cuda)
    loss = A.MaskedLoss.apply(ad, b.to(cuda), rb, torch.tensor(lens, dtype=torch.int32, device=cuda), kind, 1.0 / n, log_offset)
    (loss * 1.7).backward()
    _check([("loss", loss, lr_), ("da", ad.grad, ar.grad)])


def test_mask_rows_and_dropout(cuda, lib):
    from jatts_amd import autograd as A, hip
    g = torch.Generator().manual_seed(11)
    B, T, dim = 2, 10, 8
    x = torch.randn(B * T, dim, generator=g).to(cuda).requires_grad_()
    rb = hip.RaggedBatch([T] * B, cuda)
    valid = torch.tensor([10, 4], dtype=torch.int32, device=cuda)
    y = A.MaskRows.apply(x, rb, valid)
    y.sum().backward()
    keep = (torch.arange(T)[None, :] < torch.tensor([10, 4])[:, None]).reshape(-1, 1).float().to(cuda)
    assert torch.equal(y.detach(), x.detach() * keep) and torch.equal(x.grad, keep.expand(-1, dim))
    z = torch.ones(200000, device=cuda, requires_grad=True)
    d = A.Dropout.apply(z, 0.2, 1234)
    d.sum().backward()
    kept = (d.detach() != 0).float()
    assert abs(float(kept.mean()) - 0.8) < 5e-3
    assert torch.allclose(d.detach(), kept / 0.8) and torch.equal(z.grad, d.detach())
    d2 = A.Dropout.apply(z, 0.2, 1235)
    assert not torch.equal(d2.detach(), d.detach())


def test_adam_step_and_grad_clip_match_torch(cuda, lib):
    from jatts_amd import hip
    g = torch.Generator().manual_seed(12)
    p0, grads = torch.randn(1000, generator=g), [torch.randn(1000, generator=g) * 3 for _ in range(4)]
    pr = p0.clone().requires_grad_()
    opt = torch.optim.Adam([pr], lr=1e-2, betas=(0.9, 0.98), eps=1e-9)
    pd, m, v = p0.clone().to(cuda), torch.zeros(1000, device=cuda), torch.zeros(1000, device=cuda)
    for step, gr in enumerate(grads, 1):
        pr.grad = gr.clone()
        torch.nn.utils.clip_grad_norm_([pr], 1.0)
        opt.step()
        ss = torch.zeros((), dtype=torch.float64, device=cuda)
        gd = gr.to(cuda)
        hip.sumsq(gd, ss)
        hip.adam_step(pd, gd, m, v, 1e-2, 0.9, 0.98, 1e-9, 0.0, step, grad_sumsq=ss, max_norm=1.0)
        assert relerr(pd.cpu(), pr.detach()) <= 1e-5, step


def test_mish_backward(cuda, lib):
    from jatts_amd import autograd as A
    g = torch.Generator().manual_seed(13)
    x, gy = torch.randn(90, 33, generator=g) * 3, torch.randn(90, 33, generator=g)
    x[0, :3] = torch.tensor([25.0, -30.0, 19.5])
    xr, xd = _leaf(x, cuda)
    yr = F.mish(xr)
    yr.backward(gy.double())
    y = A.Act.apply(xd, "mish")
    y.backward(gy.to(cuda))
    _check([("y", y, yr), ("dx", xd.grad, xr.grad)])


@pytest.mark.parametrize("dim,groups,lens", [(64, 8, [30, 30]), (512, 8, [26, 26, 26]), (256, 8, [7, 40])])
def test_groupnorm_backward(cuda, lib, dim, groups, lens):
    from jatts_amd import autograd as A, hip
    g = torch.Generator().manual_seed(dim)
    R = sum(lens)
    x, w, b, gy = torch.randn(R, dim, generator=g) * 1.3 + 0.4, torch.randn(dim, generator=g), torch.randn(dim, generator=g), torch.randn(R, dim, generator=g)
    (xr, xd), (wr, wd), (br, bd) = _leaf(x, cuda), _leaf(w, cuda), _leaf(b, cuda)
    outs, o = [], 0
    for n in lens:
        outs.append(F.group_norm(xr[o:o + n].t().unsqueeze(0), groups, wr, br, 1e-5)[0].t())
        o += n
    yr = torch.cat(outs)
    yr.backward(gy.double())
    y = A.GroupNorm.apply(xd, wd, bd, hip.RaggedBatch(lens, cuda), groups, 1e-5)
    y.backward(gy.to(cuda))
    _check([("y", y, yr), ("dx", xd.grad, xr.grad), ("dg", wd.grad, wr.grad), ("db", bd.grad, br.grad)])


def test_snakebeta_backward(cuda, lib):
    from jatts_amd import autograd as A
    g = torch.Generator().manual_seed(14)
    x, al, be, gy = torch.randn(70, 96, generator=g) * 2, torch.randn(96, generator=g) * 0.5, torch.randn(96, generator=g) * 0.5, torch.randn(70, 96, generator=g)
    (xr, xd), (ar, ad), (br, bd) = _leaf(x, cuda), _leaf(al, cuda), _leaf(be, cuda)
    yr = xr + (1.0 / (torch.exp(br) + 1e-9)) * torch.sin(xr * torch.exp(ar)) ** 2
    yr.backward(gy.double())
    y = A.SnakeBeta.apply(xd, ad, bd)
    y.backward(gy.to(cuda))
    _check([("y", y, yr), ("dx", xd.grad, xr.grad), ("dalpha", ad.grad, ar.grad), ("dbeta", bd.grad, br.grad)])


def test_forward_sum_ctc_matches_torch(cuda, lib):
    """ForwardSumLoss's inner loop (losses/forward_sum_loss.py:58-78) against torch's own F.ctc_loss + autograd on the CPU, on
    un-normalised inputs (log_p_attn + prior), ragged lengths, including an utterance with more tokens than frames (zero_infinity)."""
    from jatts_amd import autograd as A
    g = torch.Generator().manual_seed(15)
    ilens, olens = torch.tensor([9, 5, 12, 6]), torch.tensor([31, 14, 40, 4])
    B, T, N = 4, 40, 12
    lp = torch.log_softmax(torch.randn(B, T, N, generator=g) * 2, dim=-1) + torch.randn(B, T, N, generator=g) * 0.3
    lr_ = lp.clone().double().requires_grad_()
    blank = math.log(math.e ** -1)
    pd = F.pad(lr_, (1, 0, 0, 0, 0, 0), value=blank)
    loss = 0
    for b in range(B):
        loss = loss + F.ctc_loss(pd[b, : olens[b], : ilens[b] + 1].unsqueeze(1), torch.arange(1, int(ilens[b]) + 1).unsqueeze(0),
                                 olens[b:b + 1], ilens[b:b + 1], zero_infinity=True)
    loss = loss / B
    (loss * 1.3).backward()
    ld = lp.clone().to(cuda).requires_grad_()
    out = A.ForwardSum.apply(ld, ilens, olens, blank)
    (out * 1.3).backward()
    assert abs(float(out.detach()) - float(loss.detach())) <= 2e-5 * abs(float(loss.detach()))
    assert relerr(ld.grad, lr_.grad) <= 5e-5, relerr(ld.grad, lr_.grad)


def _rel_shift_new(x):
    """attention.py:236-258 (zero_triu False) on (B, H, T, 2T-1)."""
    b, h, t, w = x.shape
    xp = torch.cat([torch.zeros(b, h, t, 1, dtype=x.dtype), x], dim=-1).view(b, h, w + 1, t)
    return xp[:, :, 1:].reshape(b, h, t, w)[:, :, :, : w // 2 + 1]


def test_shift_softmax_new_style_backward(cuda, lib):
    from jatts_amd import autograd as A
    g = torch.Generator().manual_seed(16)
    B, H, T, lens = 2, 2, 19, [19, 11]
    ac, bd, gp = torch.randn(B, H, T, T, generator=g), torch.randn(B, H, T, 2 * T - 1, generator=g), torch.randn(B, H, T, T, generator=g)
    (ar, ad), (br, bd_) = _leaf(ac, cuda), _leaf(bd, cuda)
    scale = 0.2
    s = (ar + _rel_shift_new(br)) * scale
    mask = (torch.arange(T)[None, :] >= torch.tensor(lens)[:, None])[:, None, None, :]
    pr = torch.softmax(s.masked_fill(mask, torch.finfo(torch.float32).min), dim=-1).masked_fill(mask, 0.0)
    pr.backward(gp.double())
    p = A.ShiftSoftmax.apply(ad, bd_, torch.tensor(lens, dtype=torch.int32, device=cuda), scale, 2)
    p.backward(gp.to(cuda))
    _check([("p", p, pr), ("dac", ad.grad, ar.grad), ("dbd", bd_.grad, br.grad)])


def test_wavenet_gate_backward(cuda, lib):
    from jatts_amd import autograd as A, hip
    g = torch.Generator().manual_seed(17)
    x, gy = torch.randn(45, 2 * 40, generator=g) * 2, torch.randn(45, 40, generator=g)
    xr, xd = _leaf(x, cuda)
    yr = torch.tanh(xr[:, :40]) * torch.sigmoid(xr[:, 40:])
    yr.backward(gy.double())
    y = A.Gate.apply(xd, hip.RaggedBatch([20, 25], cuda))
    y.backward(gy.to(cuda))
    _check([("y", y, yr), ("dx", xd.grad, xr.grad)])


def test_weight_norm_and_wavenet_split_add_backward(cuda, lib):
    """WeightNorm == torch.nn.utils.weight_norm's g * v / ||v|| (dim 0) and SplitAdd == (h + o[:, :C], skip + o[:, C:]) (skip None on
    the first layer), forward and backward vs fp64 torch."""
    from jatts_amd import autograd as A
    g = torch.Generator().manual_seed(23)
    for shape in [(384, 192, 5), (48, 7, 1), (3, 1, 3)]:
        v, gg, gy = torch.randn(shape, generator=g), torch.rand(shape[0], 1, 1, generator=g) + 0.5, torch.randn(shape, generator=g)
        (vr, vd), (gr, gd) = _leaf(v, cuda), _leaf(gg, cuda)
        wr = gr * vr / vr.reshape(shape[0], -1).norm(dim=1).reshape(-1, 1, 1)
        wr.backward(gy.double())
        w = A.WeightNorm.apply(gd, vd)
        w.backward(gy.to(cuda))
        _check([("w", w, wr), ("dv", vd.grad, vr.grad), ("dg", gd.grad, gr.grad)])
    rows, C = 333, 96
    o, h, sk = torch.randn(rows, 2 * C, generator=g), torch.randn(rows, C, generator=g), torch.randn(rows, C, generator=g)
    g1, g2 = torch.randn(rows, C, generator=g), torch.randn(rows, C, generator=g)
    for with_skip in (True, False):
        (orf, od), (hr, hd), (sr, sd) = _leaf(o, cuda), _leaf(h, cuda), _leaf(sk, cuda)
        h2r = hr + orf[:, :C]
        s2r = (sr + orf[:, C:]) if with_skip else orf[:, C:]
        ((h2r * g1.double()).sum() + (s2r * g2.double()).sum() + (h2r * h2r).sum()).backward()
        h2, s2 = A.SplitAdd.apply(od, hd, sd if with_skip else None)
        ((h2 * g1.to(cuda)).sum() + (s2 * g2.to(cuda)).sum() + (h2 * h2).sum()).backward()
        pairs = [("h", h2, h2r), ("s", s2, s2r), ("do", od.grad, orf.grad), ("dh", hd.grad, hr.grad)]
        if with_skip:
            pairs.append(("ds", sd.grad, sr.grad))
        else:
            assert sd.grad is None
        _check(pairs)


def test_add_seq_vector_backward(cuda, lib):
    from jatts_amd import autograd as A, hip
    g = torch.Generator().manual_seed(18)
    lens, dim = [70, 3, 129], 200
    R = sum(lens)
    x, v, gy = torch.randn(R, dim, generator=g), torch.randn(len(lens), dim, generator=g), torch.randn(R, dim, generator=g)
    (xr, xd), (vr, vd) = _leaf(x, cuda), _leaf(v, cuda)
    yr = xr + torch.repeat_interleave(vr, torch.tensor(lens), dim=0)
    yr.backward(gy.double())
    y = A.AddSeqVector.apply(xd, vd, hip.RaggedBatch(lens, cuda))
    y.backward(gy.to(cuda))
    _check([("y", y, yr), ("dx", xd.grad, xr.grad), ("dv", vd.grad, vr.grad)])


def test_residual_drop_add_and_zero_pool(cuda, lib):
    """ResidualDropAdd == x + alpha * Dropout(h) (same counter-based mask, forward and backward, p = 0 included), and the zero pool
    (one fill per training step instead of one per accumulator): reductions taken inside a pooled step equal the unpooled ones, a
    second step never sees the first one's sums, and a pool that runs out falls back to torch.zeros and grows."""
    from jatts_amd import autograd as A
    from jatts_amd import hip
    g = torch.Generator().manual_seed(0)
    x = torch.randn(300, 96, generator=g).to(cuda).requires_grad_(True)
    h = torch.randn(300, 96, generator=g).to(cuda).requires_grad_(True)
    dy = torch.randn(300, 96, generator=g).to(cuda)
    for p, alpha, seed in ((0.0, 0.5, 7), (0.2, 1.0, 11), (0.1, 0.5, 12345678901)):
        y = A.ResidualDropAdd.apply(x, h, alpha, p, seed)
        ref = x.detach() + alpha * (hip.dropout(h.detach().contiguous(), p, seed) if p > 0 else h.detach())
        assert torch.equal(y, ref) or float((y - ref).abs().max()) <= 1e-6
        gx, gh = torch.autograd.grad(y, (x, h), dy)
        assert torch.equal(gx, dy)
        refh = alpha * (hip.dropout(dy.contiguous(), p, seed) if p > 0 else dy)
        assert float((gh - refh).abs().max()) <= 1e-6
    a = torch.randn(1000, 384, generator=g).to(cuda)
    want = a.double().sum(0)
    plain = hip.col_sum(a)
    for step in range(3):
        hip.zero_pool_begin(cuda)
        try:
            outs = [hip.col_sum(a) for _ in range(5)]
            assert all(o.data_ptr() != outs[0].data_ptr() for o in outs[1:])
            for o in outs:
                assert float((o.double() - want).abs().max()) <= 1e-2 and float((o - plain).abs().max()) <= 2e-3
        finally:
            hip.zero_pool_end()
    assert hip._ZPOOL.off > 0 and not hip._ZPOOL.active
    small = hip._ZeroPool(cuda, floats=1024)                 # runs out: falls back, then grows at the next begin()
    hip._ZPOOL, keep = small, hip._ZPOOL
    try:
        hip.zero_pool_begin(cuda)
        outs = [hip.col_sum(a) for _ in range(5)]            # 5 x 384 floats > 1024
        assert all(float((o - plain).abs().max()) <= 2e-3 for o in outs) and hip._ZPOOL.misses > 0
        hip.zero_pool_end()                                  # grows here (never inside begin(): a capture must not allocate)
        assert hip._ZPOOL.buf.numel() > 1024 and hip._ZPOOL.misses == 0
        hip.zero_pool_begin(cuda)
        outs = [hip.col_sum(a) for _ in range(5)]
        assert all(float((o - plain).abs().max()) <= 2e-3 for o in outs)
        hip.zero_pool_end()
    finally:
        hip._ZPOOL = keep


def test_gather_grads_into_the_flat_buffer(cuda, lib):
    """jatts_gather_grads: 150 tensors of odd sizes (three launches of <= 64, chunks of 4 096 elements) copied / accumulated into their
    slots of a flat buffer; untouched slots keep their content."""
    from jatts_amd import hip
    g = torch.Generator().manual_seed(31)
    sizes = [int(v) for v in torch.randint(1, 9000, (150,), generator=g)] + [1, 4096, 4097, 70000]
    offs, o = [], 0
    for i, n in enumerate(sizes):
        offs.append(o)
        o += n + (3 if i % 7 == 0 else 0)          # gaps between some slots
    flat = torch.full((o + 5,), -7.0, device=cuda)
    grads = [torch.randn(n, generator=g).to(cuda) for n in sizes]
    hip.gather_grads(grads, offs, flat)
    want = torch.full((o + 5,), -7.0)
    for t, off in zip(grads, offs):
        want[off:off + t.numel()] = t.cpu()
    assert torch.equal(flat.cpu(), want)
    hip.gather_grads(grads[:70], offs[:70], flat, accumulate=True)
    for t, off in zip(grads[:70], offs[:70]):
        want[off:off + t.numel()] += t.cpu()
    assert torch.equal(flat.cpu(), want)
    hip.gather_grads([], [], flat)
    with pytest.raises(ValueError):
        hip.gather_grads([grads[0].double()], [0], flat)


@pytest.mark.parametrize("mode", ["relu", "swish"])
def test_act_dropout_equals_act_then_dropout(cuda, lib, mode):
    """ActDropout == Dropout(Act(x)) (same counter-based mask; bit for bit for ReLU), forward and backward, odd sizes, seed on the device too."""
    from jatts_amd import autograd as A
    g = torch.Generator().manual_seed(41)
    for n in (4099, 64, 3):
        x, gy = torch.randn(n, 7, generator=g), torch.randn(n, 7, generator=g)
        for seed_dev in (None, torch.tensor([12345], dtype=torch.int64, device=cuda)):
            x1, x2 = x.clone().to(cuda).requires_grad_(), x.clone().to(cuda).requires_grad_()
            y1 = A.Dropout.apply(A.Act.apply(x1, mode), 0.3, 77, seed_dev)
            y2 = A.ActDropout.apply(x2, mode, 0.3, 77, seed_dev)
            y1.backward(gy.to(cuda))
            y2.backward(gy.to(cuda))
            if mode == "relu":                       # (the FFN's case) bit for bit
                assert torch.equal(y1, y2) and torch.equal(x1.grad, x2.grad)
            else:                                    # same mask; the fused expression may contract its multiplies differently
                assert torch.equal(y1 == 0, y2 == 0) and maxdiff(y1, y2) <= 1e-6 and maxdiff(x1.grad, x2.grad) <= 1e-6
            if n > 1000:
                assert 0.2 < float((y2 == 0).float().mean()) < (0.75 if mode == "relu" else 0.4)


def test_qkv_split_backward(cuda, lib):
    """QKVSplit == the reference's view / transpose of linear_q/k/v outputs plus the two position biases (attention.py:81-88,190-195),
    forward and backward (d qkv, d pos_bias_u, d pos_bias_v) vs fp64 torch."""
    from jatts_amd import autograd as A
    g = torch.Generator().manual_seed(43)
    for B, T, H, dk in [(3, 37, 2, 24), (2, 5, 4, 8), (1, 129, 2, 192)]:
        Ad = H * dk
        qkv, u, v = torch.randn(B * T, 3 * Ad, generator=g), torch.randn(H, dk, generator=g), torch.randn(H, dk, generator=g)
        gs = [torch.randn(B, H, T, dk, generator=g) for _ in range(4)]
        (qr, qd), (ur, ud), (vr, vd) = _leaf(qkv, cuda), _leaf(u, cuda), _leaf(v, cuda)
        x = qr.view(B, T, 3, H, dk)
        q_, k_, v_ = (x[:, :, j].permute(0, 2, 1, 3) for j in range(3))
        ref = (q_ + ur[None, :, None, :], q_ + vr[None, :, None, :], k_, v_)
        sum((r * g_.double()).sum() for r, g_ in zip(ref, gs)).backward()
        outs = A.QKVSplit.apply(qd, ud, vd, B, T, H)
        sum((o * g_.to(cuda)).sum() for o, g_ in zip(outs, gs)).backward()
        assert all(o.is_contiguous() for o in outs)
        _check([(f"out{i}", o, r) for i, (o, r) in enumerate(zip(outs, ref))] + [("dqkv", qd.grad, qr.grad), ("du", ud.grad, ur.grad), ("dv", vd.grad, vr.grad)])
        # without position biases (Matcha's transformer blocks): (q, k, v)
        qr, qd = _leaf(qkv, cuda)
        x = qr.view(B, T, 3, H, dk)
        ref = tuple(x[:, :, j].permute(0, 2, 1, 3) for j in range(3))
        sum((r * g_.double()).sum() for r, g_ in zip(ref, gs)).backward()
        outs = A.QKVSplit.apply(qd, None, None, B, T, H)
        sum((o * g_.to(cuda)).sum() for o, g_ in zip(outs, gs)).backward()
        assert len(outs) == 3
        _check([(f"p{i}", o, r) for i, (o, r) in enumerate(zip(outs, ref))] + [("dqkv", qd.grad, qr.grad)])


# (2, 2, 257, 192, 300): the 64 x 192 tile; (2, 3, 130, 190, 66): the same tile on the element-load path (no leading dimension is a multiple of 4);
# (2, 2, 770, 192, 770) / T x T operands of a padded batch length with T % 4 != 0 are timed by tools/bench_bgemm.py --T 770
@pytest.mark.parametrize("shape", [(3, 2, 70, 50, 33), (2, 2, 257, 192, 300), (1, 1, 128, 128, 16), (4, 1, 5, 81, 7), (2, 3, 130, 190, 66)])
def test_bgemm_and_bmm_function(cuda, lib, shape):
    """jatts_bgemm (exact-f32 MFMA batched GEMM: the training step's attention products, rocBLAS before round 4) in all four transpose forms
    against fp64 matmul, with strided batch dims and an operand shared over the outer batch index; autograd.BMM forward and gradients
    against torch autograd of the same expression in fp64."""
    from jatts_amd import autograd as A
    from jatts_amd import hip
    O, I, M, N, K = shape
    g = torch.Generator().manual_seed(sum(shape))
    a = torch.randn(O, I, M, K, generator=g).to(cuda)
    b = torch.randn(O, I, K, N, generator=g).to(cuda)
    ref = a.double() @ b.double()
    assert relerr(hip.bgemm(a, b), ref) <= 2e-6
    assert relerr(hip.bgemm(a, b.transpose(-1, -2).contiguous(), trans_b=True), ref) <= 2e-6
    assert relerr(hip.bgemm(a.transpose(-1, -2).contiguous(), b, trans_a=True), ref) <= 2e-6
    assert relerr(hip.bgemm(a.transpose(-1, -2).contiguous(), b.transpose(-1, -2).contiguous(), trans_a=True, trans_b=True), ref) <= 2e-6
    # batch dims with foreign strides (a (O, I, ..) view of an (I, O, ..) tensor), and b shared over O
    a2 = torch.randn(I, O, M, K, generator=g).to(cuda).permute(1, 0, 2, 3)
    bs = torch.randn(I, N, K, generator=g).to(cuda)
    assert relerr(hip.bgemm(a2, bs, trans_b=True), a2.double() @ bs.double().transpose(-1, -2)[None]) <= 2e-6
    # the autograd Function
    for trans_b, bb in ((True, bs), (False, b), (True, b.transpose(-1, -2).contiguous())):
        x1, y1 = a.clone().requires_grad_(True), bb.clone().requires_grad_(True)
        x2, y2 = a.double().clone().requires_grad_(True), bb.double().clone().requires_grad_(True)
        c1 = A.BMM.apply(x1, y1, trans_b)
        c2 = x2 @ ((y2.transpose(-1, -2) if trans_b else y2) if y2.dim() == 4 else y2.transpose(-1, -2)[None])
        w = torch.randn(c2.shape, generator=g).to(cuda)
        (c1 * w).sum().backward()
        (c2 * w.double()).sum().backward()
        assert relerr(c1, c2) <= 2e-6 and relerr(x1.grad, x2.grad) <= 2e-6 and relerr(y1.grad, y2.grad) <= 3e-6
