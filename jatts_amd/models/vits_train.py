"""Differentiable train-mode forward of the mel-VITS (reference jatts/models/vits.py:413-579 with is_inference=False; trainer
jatts/trainers/vits.py:23-140) on the MI355X path -- SURVEY §8 f.4.  Same arithmetic as VITS.forward() (eval), rebuilt from the HIP
forward / backward pairs of jatts_amd.autograd: text encoder (conformer with the new relative positions) -> prior statistics;
posterior encoder (WaveNet: dilated conv + global conditioning + tanh/sigmoid gate, weight-normalised convolutions) -> z = m_q +
eps exp(logs_q); forward flow (mean-only residual coupling + channel flips); alignment module + monotonic alignment search;
duration predictor; masked Gaussian upsampling of the prior statistics; conformer decoder -> mel.  The WaveNet parts run on the
ragged batch of valid frames (the reference multiplies by the frame mask after every layer, which is that layout); the conformers
and the alignment module run on the padded batch like the reference.
"""
import math

import torch

from .. import autograd as A
from .. import hip
from .fastspeech2_train import _conformer, _Ctx, _predictor, spk_integrate


def _wn_weight(c, stem):
    """torch.nn.utils.weight_norm: w = g * v / ||v|| (norm over each output channel), or the plain weight."""
    if (stem + ".weight") in c.p:
        return c.p[stem + ".weight"]
    return A.WeightNorm.apply(c.p[stem + ".weight_g"], c.p[stem + ".weight_v"])


def _wavenet(c, prefix, h, g_spk, rb, rbs, n_layers, rate):
    """WaveNet.forward (wavenet/wavenet.py:115-153) + ResidualBlock.forward (residual_block.py:112-167) on the ragged valid frames:
    -> sum of the skip outputs * sqrt(1 / n_layers)."""
    Ad = h.shape[1]
    skips = None
    for layer in range(n_layers):
        q = f"{prefix}conv_layers.{layer}."
        w = _wn_weight(c, q + "conv")
        k = w.shape[-1]
        y = A.Conv1dFunction.apply(c.drop(h, rate), w, c.p[q + "conv.bias"], rb, 1, (k - 1) // 2)
        y = A.AddSeqVector.apply(y, A.Conv1dFunction.apply(g_spk, _wn_weight(c, q + "conv1x1_glo"), None, rbs, 1, 0), rb)
        o = A.Conv1dFunction.apply(A.Gate.apply(y, rb), _wn_weight(c, q + "conv1x1_out"), c.p[q + "conv1x1_out.bias"], rb, 1, 0)
        h, skips = A.SplitAdd.apply(o, h, skips)
    return skips * math.sqrt(1.0 / n_layers)


def train_forward(model, text, text_lengths, feats, feats_lengths, spembs, post_noise=None, seed=0):
    """-> the reference's return dict {outs, d_outs, ys, hs, olens_in, bin_loss, log_p_attn, ds, m_p, logs_p, z, y_mask, z_p, m_q,
    logs_q} (channel-first where the reference is), differentiable."""
    if spembs is None:
        raise ValueError("spembs is required (the reference crashes without it, vits.py:485)")
    dev = model.feat_out.weight.device
    if dev.type != "cuda":
        raise hip._abi.JattsHipError("jatts_amd.VITS trains on the GPU only (no CPU fallback); call .to('cuda')")
    hip._abi.load()
    c = _Ctx(model, seed)
    R = model.dropout_rates
    Ad, od = model.adim, model.odim
    ilens = [int(v) for v in text_lengths.tolist()]
    olens = [int(v) for v in feats_lengths.tolist()]
    B, Tm, To = len(ilens), max(ilens), max(olens)
    xs = text[:, :Tm].to(dev)
    ys = feats[:, :To].to(dev).float().contiguous()
    rbt, rbs, rbf = hip.RaggedBatch([Tm] * B, dev), hip.RaggedBatch([1] * B, dev), hip.RaggedBatch([To] * B, dev)
    # every length-derived device tensor is built here, before GPU work is queued (no host -> device copy in mid-forward)
    kv = hip.h2d(ilens, torch.int32, dev)
    kvo = hip.h2d(olens, torch.int32, dev)
    from ..alignments import frame_token_indices
    from .matchatts_train import beta_binomial_prior_dev
    tsel, fsel = frame_token_indices(ilens, olens, Tm, To, dev)
    prior = beta_binomial_prior_dev(ilens, olens, dev)                                         # ForwardSumLoss's static prior (cached on the device)
    # the posterior draw (vits.py:479 randn_like on the model's device): generated ON the device -- a host randn of B x To x A elements (9.4 M at
    # the recipe's batch) plus its upload cost ~50 ms of a 210 ms step
    nz_all = (torch.randn(B, To, Ad, device=dev) if post_noise is None else post_noise[:, :To].float().to(dev)).reshape(B * To, Ad)
    spk = spembs.to(dev).float().reshape(B, -1).contiguous()
    tmk = torch.arange(Tm, device=dev).unsqueeze(0) < kv.unsqueeze(1)
    fm = (torch.arange(To, device=dev).unsqueeze(0) < kvo.unsqueeze(1)).float()
    # ---- text encoder (text_encoder.py:104-140): emb * sqrt(A), conformer (x * sqrt(A) again inside its positional encoding)
    x = A.Embedding.apply(xs.reshape(-1).to(torch.int64).contiguous(), c.p["text_encoder.emb.weight"], float(Ad), -1)
    x = c.drop(x, R["te_pos"])
    hs = _conformer(c, "text_encoder.encoder.", x, rbt, kv, model.te_heads, dict(pos=R["te_pos"], layer=R["te"], ffn=R["te"], attn=R["te_attn"]),
                    rel_style="new")
    stats_p = A.MaskRows.apply(c.conv(hs, "text_encoder.proj", rbt), rbt, kv)                  # proj(x) * x_mask: m_p | logs_p
    hs = spk_integrate(c, model, hs, spk, rbt, rbs)
    # ---- posterior encoder + forward flow on the valid frames (posterior_encoder.py:96-130, residual_coupling.py:189-227)
    rbo = hip.RaggedBatch(olens, dev)
    yv = ys.reshape(B * To, od).index_select(0, fsel).contiguous()
    n_post = sum(1 for k in c.p if k.startswith("posterior_encoder.encoder.") and k.endswith("conv.bias"))
    h = A.Conv1dFunction.apply(yv, _wn_weight(c, "posterior_encoder.input_conv"), c.p["posterior_encoder.input_conv.bias"], rbo, 1, 0)
    sk = _wavenet(c, "posterior_encoder.encoder.", h, spk, rbo, rbs, n_post, R["post"])
    stats_q = A.Conv1dFunction.apply(sk, _wn_weight(c, "posterior_encoder.proj"), c.p["posterior_encoder.proj.bias"], rbo, 1, 0)
    m_q, logs_q = stats_q[:, :Ad], stats_q[:, Ad:]
    nz = nz_all.index_select(0, fsel)
    z = m_q + nz * torch.exp(logs_q)
    half = Ad // 2
    zp = z
    for i in range(model.flow_flows):
        q = f"flow.flows.{2 * i}."
        xa, xb = zp[:, :half], zp[:, half:]
        h = A.Conv1dFunction.apply(xa, _wn_weight(c, q + "input_conv"), c.p[q + "input_conv.bias"], rbo, 1, 0)
        sk = _wavenet(c, q + "encoder.", h, spk, rbo, rbs, model.flow_layers, R["flow"])
        m = A.Conv1dFunction.apply(sk, _wn_weight(c, q + "proj"), c.p[q + "proj.bias"], rbo, 1, 0)
        zp = torch.flip(torch.cat([xa, m + xb], dim=1), dims=[1])                              # coupling (mean only), then FlipFlow
    # ---- alignment module on the padded batch, monotonic alignment search, binarisation loss (as MatchaTTS_MAS)
    a = "alignment_module."
    tfe = c.conv(A.Act.apply(c.conv(hs, a + "t_conv1", rbt), "relu"), a + "t_conv2", rbt)
    ffe = A.Act.apply(c.conv(ys.reshape(B * To, od), a + "f_conv1", rbf), "relu")
    ffe = c.conv(A.Act.apply(c.conv(ffe, a + "f_conv2", rbf), "relu"), a + "f_conv3", rbf)
    log_p_attn = A.AlignLogProb.apply(ffe, tfe, B, ilens, tsel, tmk)
    from ..alignments import viterbi_path
    ds, path = viterbi_path(log_p_attn.detach(), ilens, olens, tsel, fsel)
    picked = torch.gather(log_p_attn, 2, path.unsqueeze(-1)).squeeze(-1).masked_fill(fm == 0, 0.0)
    bin_loss = -(picked.sum(1) / kvo.float()).mean()
    d_outs = A.MaskRows.apply(_predictor(c, "duration_predictor.", hs, rbt, R["dur"]), rbt, kv)
    # ---- prior statistics upsampled with the MAS durations (length_regulator.py:110-154; padded frames sit at t = 0)
    tpos = torch.arange(To, device=dev).float().unsqueeze(0) * fm
    cen = ds.cumsum(-1) - ds / 2
    p_up = torch.softmax((-0.1 * (tpos.unsqueeze(-1) - cen.unsqueeze(1)) ** 2).masked_fill(~tmk.unsqueeze(1), float("-inf")), dim=2)
    up = A.BMM.apply(p_up.unsqueeze(1), stats_p.view(B, 1, Tm, 2 * Ad), False).squeeze(1)   # (B, To, 2A): jatts_bgemm

    def pad_frames(v):     # ragged (valid frames, C) -> padded (B, To, C) with zeros (differentiable row scatter)
        out = torch.zeros(B * To, v.shape[1], dtype=v.dtype, device=dev)
        return out.index_copy(0, fsel, v).view(B, To, -1)
    # ---- decoder on the padded batch (z is zero at padded frames; key mask = olens), feat_out
    z_pad = pad_frames(z)
    zin = c.drop(z_pad.view(B * To, Ad) * math.sqrt(Ad), R["dec_pos"])
    zs = _conformer(c, "decoder.", zin, rbf, kvo, model.aheads, dict(pos=R["dec_pos"], layer=R["dec"], ffn=R["dec"], attn=R["dec_attn"]),
                    rel_style="new")
    outs = c.conv(zs, "feat_out", rbf).view(B, To, od)
    y_mask = fm.unsqueeze(1)
    tr = lambda v: v.transpose(1, 2)   # noqa: E731  the reference keeps these channel-first
    return {"m_q": tr(pad_frames(m_q)), "logs_q": tr(pad_frames(logs_q)), "outs": outs, "d_outs": d_outs.view(B, Tm), "ys": ys,
            "hs": tr(hs.view(B, Tm, Ad)), "olens_in": feats_lengths, "bin_loss": bin_loss, "log_p_attn": log_p_attn, "ds": ds,
            "m_p": tr(up[..., :Ad]), "logs_p": tr(up[..., Ad:]), "z": tr(z_pad), "y_mask": y_mask, "z_p": tr(pad_frames(zp)),
            "_prior": prior}


def criterion(ret, ilens, olens, duration_loss=True, forward_sum=False, bin_loss=False, lambda_align=2.0, lambda_mel=1.0):
    """The loss block of VITSTrainer._train_step (trainers/vits.py:47-110): lambda_mel x MelLoss (L1 on `outs`) + KLDivergenceLoss
    (losses/kldivergence_loss.py:17-49), the duration loss once `steps > dp_train_start_steps`, lambda_align x ForwardSumLoss while
    `steps < dp_train_start_steps`, lambda_align x the binarisation loss once `steps > bin_loss_start_steps`."""
    from .matchatts_train import beta_binomial_prior_dev
    outs = ret["outs"]
    dev = outs.device
    B, To, od = outs.shape
    Tm = ret["d_outs"].shape[1]
    rbf, rbt = hip.RaggedBatch([To] * B, dev), hip.RaggedBatch([Tm] * B, dev)
    # (cached pinned uploads: a .to(device) of a pageable CPU tensor here blocks the host until the whole queued forward has run)
    vo = olens.to(torch.int32) if olens.is_cuda else hip.h2d([int(v) for v in olens.tolist()], torch.int32, dev)
    vi = ilens.to(torch.int32) if ilens.is_cuda else hip.h2d([int(v) for v in ilens.tolist()], torch.int32, dev)
    n_o = float(int(olens.sum())) * od
    mel = A.MaskedLoss.apply(outs.reshape(B * To, od), ret["ys"].reshape(B * To, od).contiguous(), rbf, vo, 0, 1.0 / n_o, -1.0)
    zm = ret["y_mask"]
    kl = ret["logs_p"] - ret["logs_q"] - 0.5 + 0.5 * (ret["z_p"] - ret["m_p"]) ** 2 * torch.exp(-2.0 * ret["logs_p"])
    kl = A.SumAll.apply(kl * zm) / A.SumAll.apply(zm)                  # (own full reductions: capture-safe, see autograd.AddBias)
    out = dict(mel_loss=mel, kl_loss=kl)
    total = lambda_mel * mel + kl
    if duration_loss:
        tgt = ret["ds"][:, :Tm].to(dev).float().reshape(B * Tm, 1).contiguous()
        out["duration_loss"] = A.MaskedLoss.apply(ret["d_outs"].reshape(B * Tm, 1), tgt, rbt, vi, 1, 1.0 / float(int(ilens.sum())), 1.0)
        total = total + out["duration_loss"]
    if forward_sum:
        il, ol = [int(v) for v in ilens.tolist()], [int(v) for v in olens.tolist()]
        prior = ret["_prior"] if "_prior" in ret else beta_binomial_prior_dev(il, ol, dev)
        out["forward_sum_loss"] = A.ForwardSum.apply(ret["log_p_attn"] + prior, ilens, olens, -1.0)
        total = total + lambda_align * out["forward_sum_loss"]
    if bin_loss:
        out["bin_loss"] = ret["bin_loss"]
        total = total + lambda_align * ret["bin_loss"]
    out["loss"] = total
    return out
