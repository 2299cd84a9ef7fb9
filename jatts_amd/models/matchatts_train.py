"""Differentiable train-mode forward of Matcha-TTS on the MI355X path -- SURVEY §8 f.4: the tts1 MatchaTTS (reference
jatts/models/matchatts.py:317-480, ground-truth durations + hard length regulator) and the tts2 MatchaTTS_MAS
(jatts/models/matchatts_mas.py:415-550: alignment module + monotonic alignment search + masked Gaussian upsampling); trainer
jatts/trainers/matchatts.py:23-120.  Same padded-batch arithmetic as
MatchaTTS.forward() (eval), rebuilt from the HIP forward / backward pairs of jatts_amd.autograd: conformer text encoder ->
duration predictor -> hard length regulator -> encoder_proj (mu) -> conditional flow matching: y_t = (1 - (1 - s) t) z + t x1,
u = x1 - (1 - s) z, U-Net estimator (ResnetBlock1D = [Conv1d k3 -> GroupNorm(8) -> Mish] x 2 + time vector + 1x1 residual conv;
BasicTransformerBlock = LayerNorm -> attention with the additive 1/0 frame mask -> LayerNorm -> SnakeBeta feed-forward; strided
down / transposed up convolutions) -> masked MSE.  The strided conv is the stride-1 conv with every second row kept, the
transposed conv a stride-1 conv (flipped taps) over the zero-stuffed input: both reuse the MFMA conv forward / dgrad / wgrad.
"""
import math

import torch

from .. import autograd as A
from .. import hip
from .fastspeech2_train import _conformer, _Ctx, _predictor, spk_integrate

GN_EPS = 1e-5
LOG_2PI = math.log(2.0 * math.pi)


def _mask(x, rb, valid):
    return A.MaskRows.apply(x, rb, valid)


def _resnet(c, p, xs, rb, rbs, tmish, valid):
    """ResnetBlock1D (decoder.py:80-97); xs = list of (already masked) inputs, concatenated on the channel axis."""
    x = xs[0] if len(xs) == 1 else torch.cat(xs, dim=1)
    h = c.conv(x, p + "block1.block.0", rb)
    h = A.Act.apply(A.GroupNorm.apply(h, c.p[p + "block1.block.1.weight"], c.p[p + "block1.block.1.bias"], rb, 8, GN_EPS), "mish")
    tv = c.conv(tmish, p + "mlp.1", rbs)                                  # (B, C): Mish -> Linear on the time embedding
    h = _mask(A.AddSeqVector.apply(h, tv, rb), rb, valid)                  # (block1(x) * mask + tv) * mask
    h = c.conv(h, p + "block2.block.0", rb)
    h = A.Act.apply(A.GroupNorm.apply(h, c.p[p + "block2.block.1.weight"], c.p[p + "block2.block.1.bias"], rb, 8, GN_EPS), "mish")
    return _mask(h, rb, valid) + c.conv(x, p + "res_conv", rb)


def _tblock(c, p, x, rb, heads, key_bias, rate):
    """BasicTransformerBlock (transformer.py:276-364): x + attn(LN x); x + FF(LN x).  key_bias (B, T): the 1/0 frame mask that the
    reference passes as `attention_mask` and diffusers adds to the scaled scores."""
    B, T = rb.n_seq, rb.max_len
    n = A.LayerNorm.apply(x, c.p[p + "norm1.weight"], c.p[p + "norm1.bias"], GN_EPS)
    wqkv = torch.cat([c.p[p + "attn1.to_q.weight"], c.p[p + "attn1.to_k.weight"], c.p[p + "attn1.to_v.weight"]], 0).unsqueeze(-1)
    inner = wqkv.shape[0] // 3
    dh = inner // heads
    qh, kh, vh = A.QKVSplit.apply(A.Conv1dFunction.apply(n, wqkv, None, rb, 1, 0), None, None, B, T, heads)       # each (B, heads, T, dh)
    scale = dh ** -0.5
    sc = A.BMM.apply(qh, kh, True) + (key_bias / scale)[:, None, None, :]                              # jatts_bgemm (exact-f32 MFMA)
    pa = A.ShiftSoftmax.apply(sc, None, None, scale)
    a = A.BMM.apply(pa, vh, False).permute(0, 2, 1, 3).reshape(B * T, inner)
    x = x + c.drop(c.conv(a, p + "attn1.to_out.0", rb), rate)
    n = A.LayerNorm.apply(x, c.p[p + "norm3.weight"], c.p[p + "norm3.bias"], GN_EPS)
    u = A.SnakeBeta.apply(c.conv(n, p + "ff.net.0.proj", rb), c.p[p + "ff.net.0.alpha"], c.p[p + "ff.net.0.beta"])
    return x + c.conv(c.drop(u, rate), p + "ff.net.2", rb)


def sinusoidal_time_embedding(t, od):
    """SinusoidalPosEmb(2 odim) of decoder.py:48-63, on t's device: (B,) -> (B, 2 odim)."""
    freq = torch.exp(torch.arange(od, device=t.device).float() * -(math.log(10000) / (od - 1)))
    emb = 1000.0 * t.reshape(-1, 1).float() * freq.unsqueeze(0)
    return torch.cat((emb.sin(), emb.cos()), dim=-1).contiguous()


def _estimator(c, model, rb, rb2, rbs, y, mu, temb, v1, v2, rate):
    """Decoder.forward (decoder.py:413-487) on a padded batch: -> v(y, mu, t) * mask, (B*Te, odim)."""
    e = "decoder.estimator."
    dev = y.device
    B, Te = rb.n_seq, rb.max_len
    heads, od = model.dec_heads, model.odim
    tm = c.conv(A.Act.apply(c.conv(temb, e + "time_mlp.linear_1", rbs), "swish"), e + "time_mlp.linear_2", rbs)
    tmish = A.Act.apply(tm, "mish")
    kb1 = (torch.arange(Te, device=dev).unsqueeze(0) < v1.unsqueeze(1)).float()
    kb2 = (torch.arange(Te // 2, device=dev).unsqueeze(0) < v2.unsqueeze(1)).float()

    def stage(p, rbx, xs, vl, kb):
        h = _resnet(c, p + "0.", xs, rbx, rbs, tmish, vl)
        j = 0
        while (p + f"1.{j}.norm1.weight") in c.p:
            h = _tblock(c, p + f"1.{j}.", h, rbx, heads, kb, rate)
            j += 1
        return h
    x_in = _mask(torch.cat([y, mu], dim=1), rb, v1)                                         # pack([x, mu]) * mask
    h0 = _mask(stage(e + "down_blocks.0.", rb, [x_in], v1, kb1), rb, v1)                    # skip 0 (its users mask it anyway)
    C0 = h0.shape[1]
    full = c.conv(h0, e + "down_blocks.0.2.conv", rb)                                       # Conv1d(k3, stride 2, pad 1) = every 2nd
    h = full.view(B, Te, C0)[:, ::2].reshape(B * (Te // 2), C0)                             # row of the stride-1 conv
    h1 = _mask(stage(e + "down_blocks.1.", rb2, [_mask(h, rb2, v2)], v2, kb2), rb2, v2)
    h = c.conv(h1, e + "down_blocks.1.2", rb2)
    i = 0
    while (e + f"mid_blocks.{i}.0.mlp.1.weight") in c.p:
        h = stage(e + f"mid_blocks.{i}.", rb2, [_mask(h, rb2, v2)], v2, kb2)
        i += 1
    h = _mask(stage(e + "up_blocks.0.", rb2, [_mask(h, rb2, v2), h1], v2, kb2), rb2, v2)
    C1 = h.shape[1]
    # ConvTranspose1d(C, C, 4, stride 2, padding 1): zero-stuff to the full rate, then a stride-1 conv with flipped, transposed taps
    stuffed = torch.zeros(B, Te, C1, device=dev, dtype=h.dtype)
    stuffed[:, ::2] = h.view(B, Te // 2, C1)
    wt = c.p[e + "up_blocks.0.2.conv.weight"].permute(1, 0, 2).flip(2)
    h = A.Conv1dFunction.apply(stuffed.view(B * Te, C1), wt, c.p[e + "up_blocks.0.2.conv.bias"], rb, 1, 2)
    h = stage(e + "up_blocks.1.", rb, [_mask(h, rb, v1), h0], v1, kb1)
    h = c.conv(_mask(h, rb, v1), e + "up_blocks.1.2", rb)
    h = c.conv(_mask(h, rb, v1), e + "final_block.block.0", rb)
    h = A.Act.apply(A.GroupNorm.apply(h, c.p[e + "final_block.block.1.weight"], c.p[e + "final_block.block.1.bias"], rb, 8, GN_EPS), "mish")
    return _mask(c.conv(_mask(h, rb, v1), e + "final_proj", rb), rb, v1)


def train_forward(model, text, text_lengths, feats, feats_lengths, durations, durations_lengths, spembs=None, sids=None, cfm_t=None,
                  cfm_noise=None, seed=0):
    """-> {d_outs, ys, hs, olens_in, cfm_loss} like the reference's forward(), differentiable (cfm_loss, hs = mu, d_outs)."""
    dev = model.encoder_proj.weight.device
    if dev.type != "cuda":
        raise hip._abi.JattsHipError("jatts_amd Matcha-TTS trains on the GPU only (no CPU fallback); call .to('cuda')")
    if durations is None and not model._MAS:
        raise ValueError("MatchaTTS.forward needs durations")
    hip._abi.load()
    c = _Ctx(model, seed)
    R = model.dropout_rates
    Ad, od = model.adim, model.odim
    ilens = [int(v) for v in text_lengths.tolist()]
    olens = [int(v) for v in feats_lengths.tolist()]
    B, Tm, To = len(ilens), max(ilens), max(olens)
    xs = text[:, :Tm].to(dev)
    ys = feats[:, :To].to(dev).float().contiguous()
    rbt = hip.RaggedBatch([Tm] * B, dev)
    rbs = hip.RaggedBatch([1] * B, dev)
    # every length-derived device tensor is built HERE, before GPU work is queued (a host -> device copy in mid-forward stalls the host
    # behind the whole queue)
    kv = hip.h2d(ilens, torch.int32, dev)
    olens_in = [n - n % 2 for n in olens]
    Te = max(olens_in)
    v1 = hip.h2d(olens_in, torch.int32, dev)
    v2 = hip.h2d([n // 2 for n in olens_in], torch.int32, dev)
    # the two CFM draws (flow_matching.py:104-110: rand / randn_like on the model's device) are generated ON the device: a host randn of
    # B x Te x odim elements plus its upload was ~10 ms of every step
    t_dev = (torch.rand(B, device=dev) if cfm_t is None else cfm_t.reshape(B).float().to(dev)).contiguous()
    temb = sinusoidal_time_embedding(t_dev, od)
    z = (torch.randn(B, Te, od, device=dev) if cfm_noise is None else cfm_noise[:, :Te].float().to(dev)).reshape(B * Te, od).contiguous()
    extra = {}
    if model._MAS:
        from ..alignments import frame_token_indices
        extra["_prior"] = beta_binomial_prior_dev(ilens, olens, dev)                         # ForwardSumLoss's static prior (host scipy)
        kvo = hip.h2d(olens, torch.int32, dev)
        tsel, fsel = frame_token_indices(ilens, olens, Tm, To, dev)
        tm_ = torch.arange(Tm, device=dev).unsqueeze(0) < kv.unsqueeze(1)                   # (B, Tm) valid tokens
        fm = (torch.arange(To, device=dev).unsqueeze(0) < kvo.unsqueeze(1)).float()         # (B, To) valid frames
    x = A.Embedding.apply(xs.reshape(-1).to(torch.int64).contiguous(), c.p["encoder.embed.0.weight"], math.sqrt(Ad), 0)
    x = c.drop(x, R["enc_pos"])
    hs = _conformer(c, "encoder.", x, rbt, kv, model.aheads, dict(pos=R["enc_pos"], layer=R["enc"], ffn=R["enc"], attn=R["enc_attn"]))
    if model.spks is not None:
        sid = A.Embedding.apply(sids.to(dev).view(-1).to(torch.int64).contiguous(), c.p["sid_emb.weight"], 1.0, -1)
        hs = A.AddSeqVector.apply(hs, sid, rbt)
    if model.spk_embed_dim is not None:
        hs = spk_integrate(c, model, hs, spembs, rbt, rbs)
    d_outs = A.MaskRows.apply(_predictor(c, "duration_predictor.", hs, rbt, R["dur"]), rbt, kv)
    rbe = hip.RaggedBatch([Te] * B, dev)
    rb2 = hip.RaggedBatch([Te // 2] * B, dev)
    if model._MAS:
        # alignment module on the padded batch (alignments.py:26-60), monotonic alignment search (no gradient), binarisation loss
        rbf = hip.RaggedBatch([To] * B, dev)
        a = "alignment_module."
        tfe = c.conv(A.Act.apply(c.conv(hs, a + "t_conv1", rbt), "relu"), a + "t_conv2", rbt)
        ffe = A.Act.apply(c.conv(ys.reshape(B * To, od), a + "f_conv1", rbf), "relu")
        ffe = c.conv(A.Act.apply(c.conv(ffe, a + "f_conv2", rbf), "relu"), a + "f_conv3", rbf)
        log_p_attn = A.AlignLogProb.apply(ffe, tfe, B, ilens, tsel, tm_)                   # (B, To, Tm), -inf at padded tokens
        from ..alignments import viterbi_path
        ds, path = viterbi_path(log_p_attn.detach(), ilens, olens, tsel, fsel)             # ds (B, Tm) float, path (B, To) int64
        picked = torch.gather(log_p_attn, 2, path.unsqueeze(-1)).squeeze(-1).masked_fill(fm == 0, 0.0)
        bin_loss = -(picked.sum(1) / kvo.float()).mean()                                   # alignments.py:307-309
        # masked Gaussian upsampling (length_regulator.py:110-154): the weights depend on the (integer) durations only
        tpos = torch.arange(To, device=dev).float().unsqueeze(0) * fm                       # padded frames sit at t = 0
        cen = ds.cumsum(-1) - ds / 2
        energy = -0.1 * (tpos.unsqueeze(-1) - cen.unsqueeze(1)) ** 2
        p_up = torch.softmax(energy.masked_fill(~tm_.unsqueeze(1), float("-inf")), dim=2)
        up = A.BMM.apply(p_up.unsqueeze(1), hs.view(B, 1, Tm, Ad), False).squeeze(1)[:, :Te].reshape(B * Te, Ad)   # jatts_bgemm
        extra.update(bin_loss=bin_loss, log_p_attn=log_p_attn, ds=ds)
    else:
        d_flat = durations[:, : int(durations_lengths.max())].to(dev).reshape(-1).to(torch.int64).contiguous()
        if d_flat.numel() != B * Tm:
            raise ValueError("durations must be padded to the text length")
        _, cum, _, _ = hip.lr_durations(rbt, d_flat, 1.0, zero_rule=0)
        up = A.LengthRegulate.apply(hs, rbt, cum, rbe)
    mu = c.conv(up, "encoder_proj", rbe)                                                    # (B*Te, odim): "hs" of the return dict
    ys_e = ys[:, :Te].contiguous()
    y, u = hip.cfm_mix(rbe, ys_e.view(B * Te, od), z, t_dev, model.sigma_min)
    pred = _estimator(c, model, rbe, rb2, rbs, y, mu, temb, v1, v2, R["decoder"])
    n_sel = float(sum(olens_in)) * od
    # F.mse_loss(pred, u, "sum") / (sum(mask) n_feats): the sum runs over the padded frames too (pred is 0 there)
    cfm_loss = A.MaskedLoss.apply(pred, u, rbe, None, 1, 1.0 / n_sel, -1.0)
    return {"d_outs": d_outs.view(B, Tm), "ys": ys_e, "hs": mu.view(B, Te, od), "olens_in": torch.tensor(olens_in), "cfm_loss": cfm_loss,
            "_rb": (rbt, rbe, kv, v1, n_sel), **extra}


_PRIOR_CACHE = {}


def beta_binomial_prior(ilens, olens, w=1.0):
    """ForwardSumLoss._generate_prior (losses/forward_sum_loss.py:80-118): (B, max olen, max ilen) log beta-binomial alignment prior,
    -inf outside each utterance's (olen, ilen) box.  Host-side scipy, cached per (T, N) like the reference."""
    import numpy as np
    from scipy.stats import betabinom
    B, Tt, Tf = len(ilens), max(ilens), max(olens)
    out = torch.full((B, Tf, Tt), float("-inf"))
    for b, (N, T) in enumerate(zip(ilens, olens)):
        if (T, N) not in _PRIOR_CACHE:
            al = w * np.arange(1, T + 1, dtype=float)
            be = w * np.array([T - t + 1 for t in al])
            _PRIOR_CACHE[(T, N)] = torch.from_numpy(betabinom.logpmf(np.arange(N)[..., None], N, al, be)).transpose(0, 1).float()
        out[b, :T, :N] = _PRIOR_CACHE[(T, N)]
    return out


_PRIOR_DEV = {}


def beta_binomial_prior_dev(ilens, olens, device):
    """beta_binomial_prior on the device, cached per (ilens, olens, device): the (B, To, Tm) tensor is 12.6 MB at the recipes' batch -- assembling
    it on the host and uploading it from pageable memory cost ~10 ms of every MAS-phase step."""
    key = (tuple(ilens), tuple(olens), str(device))
    t = _PRIOR_DEV.get(key)
    if t is None:
        t = beta_binomial_prior(list(ilens), list(olens)).to(device)
        if len(_PRIOR_DEV) >= 16:
            _PRIOR_DEV.pop(next(iter(_PRIOR_DEV)))
        _PRIOR_DEV[key] = t
    return hip.keep(t)        # (a graph being captured pins it: the cache evicts)


def criterion(ret, durations, ilens, duration_loss=True, olens=None, forward_sum=False, bin_loss=False, lambda_align=2.0):
    """The loss block of MatchaTTSTrainer._train_step (trainers/matchatts.py:47-103): CFMLoss + EncoderPriorLoss, the duration loss
    once `steps > dp_train_start_steps` (``duration_loss``); for the MAS model also lambda_align x ForwardSumLoss while
    `steps < dp_train_start_steps` (``forward_sum``) and lambda_align x the binarisation loss once `steps > bin_loss_start_steps`
    (``bin_loss``).  ``durations``: ground truth (tts1) -- the MAS model regresses on its own ret["ds"]."""
    rbt, rbe, kv, v1, n_sel = ret["_rb"]
    B, Te, od = ret["hs"].shape
    Tm = ret["d_outs"].shape[1]
    dev = ret["hs"].device
    prior = 0.5 * A.MaskedLoss.apply(ret["hs"].reshape(B * Te, od), ret["ys"].reshape(B * Te, od).contiguous(), rbe, v1, 1, 1.0 / n_sel, -1.0) \
        + LOG_2PI                                                                            # losses/flow_matching.py:53-60
    out = dict(cfm_loss=ret["cfm_loss"], encoder_prior_loss=prior)
    total = ret["cfm_loss"] + prior
    if forward_sum:
        il, ol = [int(v) for v in ilens.tolist()], [int(v) for v in olens.tolist()]
        lp = ret["log_p_attn"] + (ret["_prior"] if "_prior" in ret else beta_binomial_prior_dev(il, ol, dev))
        out["forward_sum_loss"] = A.ForwardSum.apply(lp, ilens, olens, -1.0)                # blank_prob = e^-1
        total = total + lambda_align * out["forward_sum_loss"]
    if bin_loss:
        out["bin_loss"] = ret["bin_loss"]
        total = total + lambda_align * ret["bin_loss"]
    if duration_loss:
        if "ds" in ret:
            durations = ret["ds"]
        tgt = durations[:, :Tm].to(dev).float().reshape(B * Tm, 1).contiguous()
        out["duration_loss"] = A.MaskedLoss.apply(ret["d_outs"].reshape(B * Tm, 1), tgt, rbt, kv, 1, 1.0 / float(int(ilens.sum())), 1.0)
        total = total + out["duration_loss"]
    out["loss"] = total
    return out
