"""CPU oracle: FastSpeech2 (conformer enc/dec) inference, restated functionally.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Pure torch-CPU fp32 (or
fp64 when the caller passes double tensors); operates directly on a reference
``state_dict`` (same key schema as jatts.models.FastSpeech2) so golden vectors
made with the real reference pin it.  Parity target is the reference's
per-utterance ``inference()`` (B=1), SURVEY.md §8 note N1.

Reference lines followed (all under /root/reference/jatts/):
  models/fastspeech2.py:566-653 (_forward), :655-735 (inference), :737-761
  modules/conformer/encoder.py:233-289, encoder_layer.py:78-178,
  modules/conformer/convolution.py:56-79, swish.py
  modules/transformer/attention.py:39-93,142-206 (legacy rel-pos MHSA)
  modules/transformer/multi_layer_conv.py:52-63, layer_norm.py:12-42
  modules/positional_encoding.py:36-57,221-235 (legacy rel-pos table)
  modules/duration_predictor.py:78-97, variance_predictor.py:65-85
  modules/length_regulator.py:70-97, pre_postnets.py:173-185
"""
import math

import torch
import torch.nn.functional as F

LN_EPS = 1e-12  # layer_norm.py:23
BN_EPS = 1e-5   # torch.nn.BatchNorm1d default (convolution.py:46, pre_postnets.py:118)
PE_TABLE_LEN = 5000  # positional_encoding.py:26 (max_len default)


class Sub:
    """View of a state_dict under a key prefix."""

    def __init__(self, sd, prefix=""):
        self.sd, self.prefix = sd, prefix

    def __getitem__(self, k):
        return self.sd[self.prefix + k]

    def has(self, k):
        return (self.prefix + k) in self.sd

    def sub(self, p):
        return Sub(self.sd, self.prefix + p)

    def count(self, stem):
        """Number of consecutive integer children ``stem.<i>.`` present."""
        n = 0
        while any(key.startswith(f"{self.prefix}{stem}.{n}.") for key in self.sd):
            n += 1
        return n


def layer_norm(x, p):
    return F.layer_norm(x, (x.shape[-1],), p["weight"], p["bias"], LN_EPS)


def legacy_rel_pos_table(T, d, dtype=torch.float32, table_len=PE_TABLE_LEN):
    """pos_emb[p] for p in [0,T): sin/cos((L-1-p) * w_i), L = max(table_len, T).

    positional_encoding.py:36-57 with reverse=True; the table is built once at
    length 5000 and only regrown when T exceeds it, so values depend on L.
    """
    L = max(table_len, T)
    position = torch.arange(L - 1, -1, -1.0, dtype=torch.float32)[:T].unsqueeze(1)
    div_term = torch.exp(
        torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d)
    )
    pe = torch.zeros(T, d, dtype=torch.float32)
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe.to(dtype)


def rel_shift_legacy(bd):
    """attention.py:142-162 on a (H, T, T) tensor."""
    H, T, _ = bd.shape
    padded = torch.cat([bd.new_zeros(H, T, 1), bd], dim=-1)  # (H, T, T+1)
    padded = padded.reshape(H, T + 1, T)
    return padded[:, 1:].reshape(H, T, T)


def rel_shift_closed_form(bd):
    """Same result as rel_shift_legacy via the index formula in SURVEY §8 A3."""
    H, T, _ = bd.shape
    out = bd.new_zeros(H, T, T)
    for i in range(T):
        for j in range(T):
            if j <= i:
                out[:, i, j] = bd[:, i, T - 1 - i + j]
            elif j == i + 1:
                out[:, i, j] = 0
            else:
                out[:, i, j] = bd[:, i + 1, j - i - 2]
    return out


def legacy_rel_mhsa(x, pos_emb, p, n_heads):
    """x (T, A) -> (T, A).  attention.py:164-206 with mask=None / all-true."""
    T, A = x.shape
    dk = A // n_heads
    q = F.linear(x, p["linear_q.weight"], p["linear_q.bias"]).view(T, n_heads, dk)
    k = F.linear(x, p["linear_k.weight"], p["linear_k.bias"]).view(T, n_heads, dk)
    v = F.linear(x, p["linear_v.weight"], p["linear_v.bias"]).view(T, n_heads, dk)
    pp = F.linear(pos_emb, p["linear_pos.weight"]).view(T, n_heads, dk)
    qu = (q + p["pos_bias_u"]).transpose(0, 1)  # (H, T, dk)
    qv = (q + p["pos_bias_v"]).transpose(0, 1)
    k = k.transpose(0, 1)
    v = v.transpose(0, 1)
    pp = pp.transpose(0, 1)
    ac = torch.matmul(qu, k.transpose(1, 2))
    bd = rel_shift_legacy(torch.matmul(qv, pp.transpose(1, 2)))
    scores = (ac + bd) / math.sqrt(dk)
    attn = torch.softmax(scores, dim=-1)
    ctx = torch.matmul(attn, v).transpose(0, 1).reshape(T, A)
    return F.linear(ctx, p["linear_out.weight"], p["linear_out.bias"])


def conv1d_tc(x, w, b=None, dilation=1):
    """'same' Conv1d on a time-major (T, C) tensor."""
    k = w.shape[-1]
    pad = (k - 1) // 2 * dilation
    y = F.conv1d(x.t().unsqueeze(0), w, b, padding=pad, dilation=dilation)
    return y.squeeze(0).t()


def ffn_conv(x, p):
    """multi_layer_conv.py:52-63."""
    h = torch.relu(conv1d_tc(x, p["w_1.weight"], p["w_1.bias"]))
    return conv1d_tc(h, p["w_2.weight"], p["w_2.bias"])


def conv_module(x, p):
    """convolution.py:56-79 (BatchNorm in eval mode)."""
    h = conv1d_tc(x, p["pointwise_conv1.weight"], p["pointwise_conv1.bias"])
    h = F.glu(h, dim=-1)
    C = h.shape[-1]
    w = p["depthwise_conv.weight"]
    k = w.shape[-1]
    h = F.conv1d(
        h.t().unsqueeze(0), w, p["depthwise_conv.bias"], padding=(k - 1) // 2, groups=C
    )
    h = F.batch_norm(
        h, p["norm.running_mean"], p["norm.running_var"], p["norm.weight"],
        p["norm.bias"], False, 0.0, BN_EPS,
    )
    h = h * torch.sigmoid(h)
    h = h.squeeze(0).t()
    return conv1d_tc(h, p["pointwise_conv2.weight"], p["pointwise_conv2.bias"])


def conformer_layer(x, pos_emb, p, n_heads, taps=None):
    """encoder_layer.py:78-178, normalize_before=True, macaron, cnn module."""
    has_macaron = p.has("feed_forward_macaron.w_1.weight")
    ff_scale = 0.5 if has_macaron else 1.0
    if has_macaron:
        x = x + ff_scale * ffn_conv(layer_norm(x, p.sub("norm_ff_macaron.")), p.sub("feed_forward_macaron."))
        if taps is not None:
            taps["after_macaron"] = x
    x = x + legacy_rel_mhsa(layer_norm(x, p.sub("norm_mha.")), pos_emb, p.sub("self_attn."), n_heads)
    if taps is not None:
        taps["after_mha"] = x
    has_conv = p.has("conv_module.pointwise_conv1.weight")
    if has_conv:
        x = x + conv_module(layer_norm(x, p.sub("norm_conv.")), p.sub("conv_module."))
        if taps is not None:
            taps["after_conv"] = x
    x = x + ff_scale * ffn_conv(layer_norm(x, p.sub("norm_ff.")), p.sub("feed_forward."))
    if has_conv:
        x = layer_norm(x, p.sub("norm_final."))
    return x


def conformer_stack(x, p, n_heads, taps=None):
    """encoder.py:233-289 after the input layer: x*sqrt(d), layers, after_norm."""
    T, A = x.shape
    x = x * math.sqrt(A)
    pos_emb = legacy_rel_pos_table(T, A, x.dtype)
    for i in range(p.count("encoders")):
        lt = {} if (taps is not None and i == 0) else None
        x = conformer_layer(x, pos_emb, p.sub(f"encoders.{i}."), n_heads, lt)
        if taps is not None:
            taps[f"layer{i}"] = x
            if lt:
                taps.update({f"layer0_{k}": v for k, v in lt.items()})
    if p.has("after_norm.weight"):
        x = layer_norm(x, p.sub("after_norm."))
    return x


def predictor_trunk(x, p):
    """Conv1d -> ReLU -> LayerNorm(channels) stack; (T, C) in/out.

    duration_predictor.py:60-76,78-84 / variance_predictor.py:47-63,75-78."""
    for i in range(p.count("conv")):
        x = torch.relu(conv1d_tc(x, p[f"conv.{i}.0.weight"], p[f"conv.{i}.0.bias"]))
        x = F.layer_norm(x, (x.shape[-1],), p[f"conv.{i}.2.weight"], p[f"conv.{i}.2.bias"], LN_EPS)
    return x


def variance_predictor(x, p):
    h = predictor_trunk(x, p)
    return F.linear(h, p["linear.weight"], p["linear.bias"])  # (T, 1)


def duration_predictor_log(x, p):
    h = predictor_trunk(x, p)
    return F.linear(h, p["linear.weight"], p["linear.bias"]).squeeze(-1)  # (T,)


def duration_from_log(logd, offset=1.0):
    """duration_predictor.py:86-90."""
    return torch.clamp(torch.round(logd.exp() - offset), min=0).long()


def length_regulate(x, d, alpha=1.0):
    """length_regulator.py:70-97 for one utterance (B=1)."""
    if alpha != 1.0:
        d = torch.round(d.float() * alpha).long()
    if d.sum() == 0:
        d = torch.ones_like(d)
    return torch.repeat_interleave(x, d, dim=0), d


def postnet(x, p):
    """pre_postnets.py:173-185 on (T, odim); BN eval; tanh on all but last."""
    n = p.count("postnet")
    h = x
    for i in range(n):
        q = p.sub(f"postnet.{i}.")
        h = conv1d_tc(h, q["0.weight"], None)
        if q.has("1.running_mean"):
            h = F.batch_norm(
                h.t().unsqueeze(0), q["1.running_mean"], q["1.running_var"],
                q["1.weight"], q["1.bias"], False, 0.0, BN_EPS,
            ).squeeze(0).t()
        if i != n - 1:
            h = torch.tanh(h)
    return h


def fs2_inference(sd, text, n_heads, spembs=None, sids=None, alpha=1.0,
                  durations=None, taps=None):
    """FastSpeech2.inference (fastspeech2.py:655-735), predicted d/p/e.

    ``durations`` (optional LongTensor) overrides the predicted durations (used
    to decouple the bit-exact length-regulator check from exp/round ulps, H3).
    Returns dict(feat_gen, before, duration, pitch, energy, log_duration).
    """
    p = Sub(sd)
    emb = sd["encoder.embed.0.weight"][text]  # (T, A)
    hs = conformer_stack(emb, p.sub("encoder."), n_heads, taps)
    if taps is not None:
        taps["encoder_out"] = hs
    if sids is not None and "sid_emb.weight" in sd:
        hs = hs + sd["sid_emb.weight"][sids.view(-1)[0]]
    if spembs is not None and "projection.weight" in sd:
        sp = F.normalize(spembs.unsqueeze(0)).squeeze(0)
        W = sd["projection.weight"]
        if W.shape[1] == spembs.shape[0]:  # "add"
            hs = hs + F.linear(sp, W, sd["projection.bias"])
        else:  # "concat"
            hs = F.linear(torch.cat([hs, sp.expand(hs.shape[0], -1)], -1), W, sd["projection.bias"])
    p_outs = variance_predictor(hs, p.sub("pitch_predictor."))
    e_outs = variance_predictor(hs, p.sub("energy_predictor."))
    logd = duration_predictor_log(hs, p.sub("duration_predictor."))
    d_pred = duration_from_log(logd)
    d_used = d_pred if durations is None else durations
    p_emb = conv1d_tc(p_outs, sd["pitch_embed.0.weight"], sd["pitch_embed.0.bias"])
    e_emb = conv1d_tc(e_outs, sd["energy_embed.0.weight"], sd["energy_embed.0.bias"])
    hs = hs + e_emb + p_emb
    if taps is not None:
        taps["variance_out"] = hs
    hs, d_eff = length_regulate(hs, d_used, alpha)
    zs = conformer_stack(hs, p.sub("decoder."), n_heads)
    if taps is not None:
        taps["decoder_out"] = zs
    before = F.linear(zs, sd["feat_out.weight"], sd["feat_out.bias"])
    after = before
    if any(k.startswith("postnet.") for k in sd):
        after = before + postnet(before, p.sub("postnet."))
    return dict(feat_gen=after, before=before, duration=d_pred, pitch=p_outs,
                energy=e_outs, log_duration=logd, duration_used=d_eff)
