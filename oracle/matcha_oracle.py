"""CPU oracle: Matcha-TTS (MAS variant) inference — conformer encoder -> duration predictor ->
Gaussian upsampling -> encoder_proj -> CFM U-Net decoder with fixed-step Euler.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Reference lines followed (under
/root/reference/jatts/): models/matchatts_mas.py:415-550 (_forward, is_inference), :552-642;
modules/matchatts/flow_matching.py:48-95; modules/matchatts/decoder.py:48-156 (SinusoidalPosEmb,
Block1D, ResnetBlock1D, Downsample1D, TimestepEmbedding, Upsample1D), :413-487 (Decoder.forward);
modules/matchatts/transformer.py:84-102 (SnakeBeta), :105-157 (FeedForward), :276-364
(BasicTransformerBlock.forward); modules/length_regulator.py:111-154.

Pinning: everything above is pinned by tests/golden/matcha_small.npz (real reference code, noise
injected) EXCEPT the self-attention inside BasicTransformerBlock: it is
`diffusers.models.attention_processor.Attention` (+ `LoRACompatibleLinear`), third-party, unpinned
(setup.cfg:41) and not installed here.  The golden run substitutes a standard scaled-dot-product
attention for those two classes (to_q/to_k/to_v without bias, to_out.0 with bias, scale
dim_head**-0.5 [recalled]); that piece is therefore PARITY UNPINNED.
"""
import math

import torch
import torch.nn.functional as F

from .fs2_oracle import Sub, conformer_stack, duration_from_log, duration_predictor_log, length_regulate
from .vits_oracle import gaussian_upsample


def sinusoidal_pos_emb(t, dim, scale=1000.0):
    """decoder.py:48-64 for a scalar t -> (dim,)."""
    half = dim // 2
    emb = math.log(10000) / (half - 1)
    emb = torch.exp(torch.arange(half).float() * -emb)
    emb = scale * t * emb
    return torch.cat((emb.sin(), emb.cos()), dim=-1)


def conv_tc(x, w, b, stride=1, padding=0):
    return F.conv1d(x.t().unsqueeze(0), w, b, stride=stride, padding=padding)[0].t()


def block1d(x, p):
    """decoder.py:66-77 (mask == 1): Conv1d k3 -> GroupNorm(8) -> Mish; x (T, C)."""
    h = conv_tc(x, p["block.0.weight"], p["block.0.bias"], padding=1)
    h = F.group_norm(h.t().unsqueeze(0), 8, p["block.1.weight"], p["block.1.bias"], 1e-5)[0].t()
    return F.mish(h)


def resnet_block(x, temb, p):
    """decoder.py:80-97."""
    h = block1d(x, p.sub("block1."))
    h = h + F.linear(F.mish(temb), p["mlp.1.weight"], p["mlp.1.bias"])
    h = block1d(h, p.sub("block2."))
    return h + conv_tc(x, p["res_conv.weight"], p["res_conv.bias"])


def sdpa(x, p, heads):
    """Standard SDPA stand-in for diffusers Attention (self-attention, no mask)."""
    T, C = x.shape
    inner = p["to_q.weight"].shape[0]
    dh = inner // heads
    q = F.linear(x, p["to_q.weight"]).view(T, heads, dh).transpose(0, 1)
    k = F.linear(x, p["to_k.weight"]).view(T, heads, dh).transpose(0, 1)
    v = F.linear(x, p["to_v.weight"]).view(T, heads, dh).transpose(0, 1)
    a = torch.softmax(q @ k.transpose(1, 2) * dh ** -0.5, dim=-1)
    o = (a @ v).transpose(0, 1).reshape(T, inner)
    return F.linear(o, p["to_out.0.weight"], p["to_out.0.bias"])


def transformer_block(x, p, heads):
    """transformer.py:276-364 without ada-norm / cross-attention: LN -> attn -> +x; LN -> FF(SnakeBeta) -> +x."""
    h = F.layer_norm(x, (x.shape[1],), p["norm1.weight"], p["norm1.bias"], 1e-5)
    x = sdpa(h, p.sub("attn1."), heads) + x
    h = F.layer_norm(x, (x.shape[1],), p["norm3.weight"], p["norm3.bias"], 1e-5)
    u = F.linear(h, p["ff.net.0.proj.weight"], p["ff.net.0.proj.bias"])
    alpha, beta = torch.exp(p["ff.net.0.alpha"]), torch.exp(p["ff.net.0.beta"])
    u = u + (1.0 / (beta + 1e-9)) * torch.pow(torch.sin(u * alpha), 2)
    return F.linear(u, p["ff.net.2.weight"], p["ff.net.2.bias"]) + x


def estimator(sd, prefix, x, mu, t, heads):
    """Decoder.forward (decoder.py:413-487), one unmasked utterance; x, mu (T, n_feats); T even."""
    p = Sub(sd, prefix)
    temb = sinusoidal_pos_emb(t, x.shape[1] + mu.shape[1])
    temb = F.linear(temb, p["time_mlp.linear_1.weight"], p["time_mlp.linear_1.bias"])
    temb = F.linear(F.silu(temb), p["time_mlp.linear_2.weight"], p["time_mlp.linear_2.bias"])
    h = torch.cat([x, mu], dim=1)
    hiddens = []
    n_down = p.count("down_blocks")
    for i in range(n_down):
        b = p.sub(f"down_blocks.{i}.")
        h = resnet_block(h, temb, b.sub("0."))
        for j in range(b.count("1")):
            h = transformer_block(h, b.sub(f"1.{j}."), heads)
        hiddens.append(h)
        w = b["2.weight"] if b.has("2.weight") else b["2.conv.weight"]
        bias = b["2.bias"] if b.has("2.bias") else b["2.conv.bias"]
        h = conv_tc(h, w, bias, stride=1 if b.has("2.weight") else 2, padding=1)
    for i in range(p.count("mid_blocks")):
        b = p.sub(f"mid_blocks.{i}.")
        h = resnet_block(h, temb, b.sub("0."))
        for j in range(b.count("1")):
            h = transformer_block(h, b.sub(f"1.{j}."), heads)
    for i in range(p.count("up_blocks")):
        b = p.sub(f"up_blocks.{i}.")
        h = resnet_block(torch.cat([h, hiddens.pop()], dim=1), temb, b.sub("0."))
        for j in range(b.count("1")):
            h = transformer_block(h, b.sub(f"1.{j}."), heads)
        if b.has("2.weight"):
            h = conv_tc(h, b["2.weight"], b["2.bias"], padding=1)
        else:  # Upsample1D: ConvTranspose1d(C, C, 4, 2, 1)
            h = F.conv_transpose1d(h.t().unsqueeze(0), b["2.conv.weight"], b["2.conv.bias"], stride=2, padding=1)[0].t()
    h = block1d(h, p.sub("final_block."))
    return conv_tc(h, p["final_proj.weight"], p["final_proj.bias"])


def matcha_inference(sd, text, enc_heads, dec_heads, noise, n_timesteps=10, temperature=0.667, durations=None,
                     spembs=None, taps=None, hard_lr=False):
    """MatchaTTS_MAS.inference (matchatts_mas.py:552-642) for feats=None, with injected noise (T', odim).
    hard_lr=True: the tts1 `MatchaTTS` class instead (matchatts.py:423-427,482-558: LengthRegulator on the predicted durations)."""
    emb = sd["encoder.embed.0.weight"][text]
    hs = conformer_stack(emb, Sub(sd, "encoder."), enc_heads)
    if spembs is not None and "projection.weight" in sd:
        hs = hs + F.linear(F.normalize(spembs.unsqueeze(0))[0], sd["projection.weight"], sd["projection.bias"])
    logd = duration_predictor_log(hs, Sub(sd, "duration_predictor."))
    d_pred = duration_from_log(logd)
    d_used = d_pred if durations is None else durations
    up = length_regulate(hs, d_used)[0] if hard_lr else gaussian_upsample(hs, d_used)
    mu = F.linear(up, sd["encoder_proj.weight"], sd["encoder_proj.bias"])
    T = mu.shape[0] - mu.shape[0] % 2           # matchatts_mas.py:521-526: even length
    mu = mu[:T]
    if taps is not None:
        taps["mu"] = mu
    x = noise[:T] * temperature
    t_span = torch.linspace(0, 1, n_timesteps + 1)
    t, dt = t_span[0], t_span[1] - t_span[0]
    for step in range(1, len(t_span)):           # flow_matching.py:68-95
        x = x + dt * estimator(sd, "decoder.estimator.", x, mu, t, dec_heads)
        t = t + dt
        if step < len(t_span) - 1:
            dt = t_span[step + 1] - t
    return dict(feat_gen=x, duration=d_pred, log_duration=logd)
