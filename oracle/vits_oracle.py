"""CPU oracle: mel-VITS inference, restated functionally on the reference state_dict.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Pinned against golden vectors captured from
the real reference (tests/golden/make_golden.py -> vits_small.npz) with the sampling noise
injected.  Reference lines followed (under /root/reference/jatts/):
  models/vits.py:413-560 (_forward, is_inference=True, ys=None), :581-679 (inference)
  modules/vits/text_encoder.py:104-140; modules/positional_encoding.py:238-309 (RelPositionalEncoding)
  modules/transformer/attention.py:209-305 (RelPositionMultiHeadedAttention, new rel_shift)
  modules/vits/residual_coupling.py:83-107,189-227 (inverse coupling), vits/flow.py:17-40 (FlipFlow)
  modules/wavenet/wavenet.py:115-153, wavenet/residual_block.py:112-167
  modules/length_regulator.py:111-154 (GaussianUpsampling), duration_predictor.py:78-97
"""
import math

import torch
import torch.nn.functional as F

from .fs2_oracle import (Sub, conv1d_tc, conv_module, duration_from_log,
                         duration_predictor_log, ffn_conv, layer_norm)


def fold_wn(sd, stem):
    """weight for `stem` given either .weight or .weight_g/.weight_v (torch weight_norm, dim=0)."""
    if stem + ".weight" in sd:
        return sd[stem + ".weight"]
    g, v = sd[stem + ".weight_g"], sd[stem + ".weight_v"]
    norm = v.reshape(v.shape[0], -1).norm(dim=1).reshape(-1, *([1] * (v.dim() - 1)))
    return g * v / norm


def rel_pos_table_new(T, d, dtype=torch.float32):
    """RelPositionalEncoding.forward's pos_emb (2T-1, d): row m encodes relative position T-1-m
    (positional_encoding.py:265-309; independent of the table length)."""
    pos = torch.arange(T - 1, -T, -1.0, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))
    pe = torch.zeros(2 * T - 1, d, dtype=torch.float32)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe.to(dtype)


def rel_shift_new(bd):
    """attention.py:237-261 on (H, T, 2T-1) -> (H, T, T)."""
    H, T, W = bd.shape
    padded = torch.cat([bd.new_zeros(H, T, 1), bd], dim=-1).reshape(H, W + 1, T)
    return padded[:, 1:].reshape(H, T, W)[:, :, : W // 2 + 1]


def rel_mhsa_new(x, pos_emb, p, n_heads):
    T, A = x.shape
    dk = A // n_heads
    q = F.linear(x, p["linear_q.weight"], p["linear_q.bias"]).view(T, n_heads, dk)
    k = F.linear(x, p["linear_k.weight"], p["linear_k.bias"]).view(T, n_heads, dk).transpose(0, 1)
    v = F.linear(x, p["linear_v.weight"], p["linear_v.bias"]).view(T, n_heads, dk).transpose(0, 1)
    pp = F.linear(pos_emb, p["linear_pos.weight"]).view(-1, n_heads, dk).transpose(0, 1)  # (H, 2T-1, dk)
    qu = (q + p["pos_bias_u"]).transpose(0, 1)
    qv = (q + p["pos_bias_v"]).transpose(0, 1)
    ac = qu @ k.transpose(1, 2)
    bd = rel_shift_new(qv @ pp.transpose(1, 2))
    attn = torch.softmax((ac + bd) / math.sqrt(dk), dim=-1)
    ctx = (attn @ v).transpose(0, 1).reshape(T, A)
    return F.linear(ctx, p["linear_out.weight"], p["linear_out.bias"])


def conformer_layer_new(x, pos_emb, p, n_heads):
    has_macaron = p.has("feed_forward_macaron.w_1.weight")
    s = 0.5 if has_macaron else 1.0
    if has_macaron:
        x = x + s * ffn_conv(layer_norm(x, p.sub("norm_ff_macaron.")), p.sub("feed_forward_macaron."))
    x = x + rel_mhsa_new(layer_norm(x, p.sub("norm_mha.")), pos_emb, p.sub("self_attn."), n_heads)
    if p.has("conv_module.pointwise_conv1.weight"):
        x = x + conv_module(layer_norm(x, p.sub("norm_conv.")), p.sub("conv_module."))
    x = x + s * ffn_conv(layer_norm(x, p.sub("norm_ff.")), p.sub("feed_forward."))
    if p.has("norm_final.weight"):
        x = layer_norm(x, p.sub("norm_final."))
    return x


def conformer_stack_new(x, e, n_heads):
    """conformer Encoder (encoder.py:233-289) with RelPositionalEncoding + RelPositionMultiHeadedAttention
    (the non-legacy pair): x * sqrt(A) (positional_encoding.py:300), layers, after_norm."""
    A = x.shape[1]
    x = x * math.sqrt(A)
    pos_emb = rel_pos_table_new(x.shape[0], A, x.dtype)
    for i in range(e.count("encoders")):
        x = conformer_layer_new(x, pos_emb, e.sub(f"encoders.{i}."), n_heads)
    if e.has("after_norm.weight"):
        x = layer_norm(x, e.sub("after_norm."))
    return x


def text_encoder(sd, text, n_heads):
    """text_encoder.py:104-140 for one utterance: returns hs, m_p, logs_p as (T, A)."""
    A = sd["text_encoder.emb.weight"].shape[1]
    x = sd["text_encoder.emb.weight"][text] * math.sqrt(A)
    x = conformer_stack_new(x, Sub(sd, "text_encoder.encoder."), n_heads)
    stats = conv1d_tc(x, sd["text_encoder.proj.weight"], sd["text_encoder.proj.bias"])
    return x, stats[:, :A], stats[:, A:]


def gaussian_upsample(hs, ds, delta=0.1):
    """length_regulator.py:111-154, one unmasked utterance (T_feats = ds.sum())."""
    if ds.sum() == 0:
        ds = torch.ones_like(ds)
    T = int(ds.sum())
    t = torch.arange(0, T).float()
    c = ds.cumsum(dim=-1) - ds / 2
    energy = -1 * delta * (t.unsqueeze(-1) - c.unsqueeze(0)) ** 2
    return torch.softmax(energy, dim=1) @ hs


def wavenet(sd, prefix, h, g):
    """wavenet.py:115-153 (no first/last conv, scale_skip_connect) on (T, C); g (G,) or None."""
    n = 0
    while (prefix + f"conv_layers.{n}.conv.bias") in sd:
        n += 1
    skips = 0.0
    x = h
    for i in range(n):
        q = prefix + f"conv_layers.{i}."
        w = fold_wn(sd, q + "conv")
        k = w.shape[-1]
        dil = 1  # base_dilation ** (layer % layers_per_stack) with base_dilation = 1 (vits.py:97)
        y = F.conv1d(x.t().unsqueeze(0), w, sd[q + "conv.bias"], padding=(k - 1) // 2 * dil, dilation=dil)[0].t()
        C2 = y.shape[1] // 2
        xa, xb = y[:, :C2], y[:, C2:]
        if g is not None and (q + "conv1x1_glo.weight_v" in sd or q + "conv1x1_glo.weight" in sd):
            gg = F.linear(g, fold_wn(sd, q + "conv1x1_glo").squeeze(-1))
            xa, xb = xa + gg[:C2], xb + gg[C2:]
        z = torch.tanh(xa) * torch.sigmoid(xb)
        o = F.linear(z, fold_wn(sd, q + "conv1x1_out").squeeze(-1), sd[q + "conv1x1_out.bias"])
        R = x.shape[1]
        x = o[:, :R] + x
        skips = skips + o[:, R:]
    return skips * math.sqrt(1.0 / n)


def flow_inverse(sd, z, g):
    """ResidualAffineCouplingBlock.forward(inverse=True), use_only_mean (residual_coupling.py:96-107,212-227)."""
    idx = sorted({int(k.split(".")[2]) for k in sd if k.startswith("flow.flows.")})
    x = z
    for i in reversed(idx):
        x = torch.flip(x, [1])  # FlipFlow precedes each coupling layer in the reversed order
        q = f"flow.flows.{i}."
        half = x.shape[1] // 2
        xa, xb = x[:, :half], x[:, half:]
        h = F.linear(xa, sd[q + "input_conv.weight"].squeeze(-1), sd[q + "input_conv.bias"])
        h = wavenet(sd, q + "encoder.", h, g)
        m = F.linear(h, sd[q + "proj.weight"].squeeze(-1), sd[q + "proj.bias"])
        x = torch.cat([xa, xb - m], dim=1)
    return x


def vits_inference(sd, text, te_heads, dec_heads, spembs, noise, noise_scale=0.667, durations=None, taps=None):
    """VITS.inference (vits.py:581-679) with the sampling noise injected.
    noise: (T_feats, A) standard normal (the reference draws torch.randn_like(m_p), vits.py:479)."""
    hs, m_p, logs_p = text_encoder(sd, text, te_heads)
    if taps is not None:
        taps["text_encoder_out"] = hs
    if "projection.weight" in sd and spembs is not None:
        sp = F.normalize(spembs.unsqueeze(0)).squeeze(0)
        hs = hs + F.linear(sp, sd["projection.weight"], sd["projection.bias"])
    logd = duration_predictor_log(hs, Sub(sd, "duration_predictor."))
    d_pred = duration_from_log(logd)
    d_used = d_pred if durations is None else durations
    m_up = gaussian_upsample(m_p, d_used)
    logs_up = gaussian_upsample(logs_p, d_used)
    z_p = m_up + noise * torch.exp(logs_up) * noise_scale
    if taps is not None:
        taps["z_p"] = z_p
    z = flow_inverse(sd, z_p, spembs)
    if taps is not None:
        taps["z"] = z
    # vits.py:311-331 passes "rel_pos"/"rel_selfattn" straight to the conformer Encoder: unlike
    # FastSpeech2 there is NO legacy fallback, so the decoder uses the new rel-pos attention too
    zs = conformer_stack_new(z, Sub(sd, "decoder."), dec_heads)
    out = F.linear(zs, sd["feat_out.weight"], sd["feat_out.bias"])
    return dict(feat_gen=out, duration=d_pred, log_duration=logd)
