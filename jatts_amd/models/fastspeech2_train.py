"""Differentiable train-mode forward of FastSpeech2 on the MI355X path: the same padded-batch arithmetic as
FastSpeech2.forward() (reference fastspeech2.py:473-653 with is_inference=False), built from the HIP forward / backward pairs
of jatts_amd.autograd so that ``loss.backward()`` runs HIP kernels (plus rocBLAS batched GEMMs for the [T x T] attention
products).  Train-mode behaviour follows the reference modules: batch-statistics BatchNorm in the conformer conv module and the
postnet (running stats updated), dropout at every site the reference has one (counter-based masks; rates from the ctor), the
`stop_gradient_from_*_predictor` detaches (fastspeech2.py:599-606).  Parameters are the module's own (reference state_dict
names), so torch optimisers / jatts_amd.training.Trainer / allreduce_gradients see them directly.
"""
import math

import torch

from .. import autograd as A
from .. import hip
from ._conformer import BN_EPS, LN_EPS, PE_TABLE_LEN, legacy_rel_pos_table, rel_pos_table_new

BN_MOMENTUM = 0.1


class _Ctx:
    """Parameter lookup + dropout bookkeeping of one forward pass."""

    def __init__(self, model, seed):
        self.p = dict(model.named_parameters())
        self.b = dict(model.named_buffers())
        self.train = model.training
        rank = 0
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            rank = torch.distributed.get_rank()          # data-parallel replicas draw different dropout masks
        self.seed = (int(seed) * 4099 + rank) * 1000003
        # graph mode (training.py): the per-step base seed lives in a device tensor the trainer rewrites before every replay; the call
        # sites then pass only their offset
        self.seed_dev = getattr(model, "_seed_dev", None)
        if self.seed_dev is not None:
            self.seed = 0
        self.n_drop = 0

    def drop(self, x, rate):
        if not self.train or rate <= 0.0:
            return x
        self.n_drop += 1
        return A.Dropout.apply(x, rate, self.seed + self.n_drop, self.seed_dev)

    def act_drop(self, x, mode, rate):
        """dropout(act(x)): one launch (ActDropout); the dropout site keeps its place in the seed sequence."""
        if not self.train or rate <= 0.0:
            return A.Act.apply(x, mode)
        self.n_drop += 1
        return A.ActDropout.apply(x, mode, rate, self.seed + self.n_drop, self.seed_dev)

    def resid_drop(self, x, h, rate, alpha=1.0):
        """x + alpha * dropout(h): one launch (ResidualDropAdd) instead of dropout, scale and add."""
        rate = rate if self.train else 0.0
        if rate > 0.0:
            self.n_drop += 1
        return A.ResidualDropAdd.apply(x, h, alpha, max(rate, 0.0), self.seed + self.n_drop, self.seed_dev)

    def conv(self, x, name, rb, dil=1, pad=None, bias=True):
        w = self.p[name + ".weight"]
        if w.dim() == 2:                     # nn.Linear == Conv1d with k = 1
            w = w.unsqueeze(-1)
        k = w.shape[-1]
        return A.Conv1dFunction.apply(x, w, self.p[name + ".bias"] if bias else None, rb, dil, (k - 1) // 2 * dil if pad is None else pad)

    def ln(self, x, name):
        return A.LayerNorm.apply(x, self.p[name + ".weight"], self.p[name + ".bias"], LN_EPS)

    def bn(self, x, name):
        if self.train:
            y = A.BatchNormTrain.apply(x, self.p[name + ".weight"], self.p[name + ".bias"], self.b[name + ".running_mean"],
                                       self.b[name + ".running_var"], BN_MOMENTUM, BN_EPS)
            if (name + ".num_batches_tracked") in self.b:
                self.b[name + ".num_batches_tracked"] += 1
            return y
        s = self.p[name + ".weight"] * torch.rsqrt(self.b[name + ".running_var"] + BN_EPS)     # eval: a per-channel affine (tiny [C] ops)
        return x * s + (self.p[name + ".bias"] - self.b[name + ".running_mean"] * s)


def spk_integrate(c, model, hs, spembs, rb, rbs):
    """_integrate_with_spk_embed with gradients (fastspeech2.py:737-761): "add" = hs + projection(normalize(s)); "concat" = the Linear over
    cat[hs, normalize(s)] split by columns, W_h hs + (W_s s + b) (jatts_amd.models._conformer.SpkProjection is the inference form)."""
    sp = hip.l2_normalize(spembs.to(hs.device).float().reshape(rb.n_seq, -1).contiguous(), hip.F32)
    if getattr(model, "spk_embed_integration_type", "add") == "add":
        return A.AddSeqVector.apply(hs, c.conv(sp, "projection", rbs), rb)
    w, Ad = c.p["projection.weight"], hs.shape[-1]
    vec = A.Conv1dFunction.apply(sp, w[:, Ad:].unsqueeze(-1).contiguous(), c.p["projection.bias"], rbs, 1, 0)
    return A.AddSeqVector.apply(A.Conv1dFunction.apply(hs, w[:, :Ad].unsqueeze(-1).contiguous(), None, rb, 1, 0), vec, rb)


_POS_TABLES = {}


def _pos_table(rel_style, T, Ad, device):
    """rel_style "new": RelPositionalEncoding's (2T-1, A) table (positional_encoding.py:265-309); "legacy": pe[:, :T] of the reversed table
    (positional_encoding.py:221-235).  Constant per (style, T, A): kept on the device (also what makes the step capturable)."""
    key = (rel_style, T, Ad, str(device))
    t = _POS_TABLES.get(key)
    if t is None:
        t = (rel_pos_table_new(T, Ad) if rel_style == "new" else legacy_rel_pos_table(T, Ad, max(PE_TABLE_LEN, T))).to(device)
        if len(_POS_TABLES) >= 64:
            _POS_TABLES.pop(next(iter(_POS_TABLES)))
        _POS_TABLES[key] = t
    return hip.keep(t)        # (a graph being captured pins it: the cache evicts)


def _conformer(c, prefix, x, rb, kv, H, rates, rel_style="legacy"):
    """conformer/encoder.py:233-289 after the input layer: x (rows, A) is already x * sqrt(A) (+ positional dropout).
    rel_style "legacy": LegacyRelPositionalEncoding / LegacyRelPositionMultiHeadedAttention (FastSpeech2, Matcha-TTS);
    "new": RelPositionalEncoding / RelPositionMultiHeadedAttention (VITS: 2T-1 relative positions, attention.py:209-305)."""
    B, T = rb.n_seq, rb.max_len
    Ad = x.shape[1]
    dk = Ad // H
    pos = _pos_table(rel_style, T, Ad, x.device)       # device copy cached per (style, T, A): rebuilt on the host + uploaded once, not per layer stack per step
    n_pos = pos.shape[0]
    pos = c.drop(pos, rates["pos"])
    rbp = hip.RaggedBatch([n_pos], x.device)
    i = 0
    while (prefix + f"encoders.{i}.norm_mha.weight") in c.p:
        q = prefix + f"encoders.{i}."
        macaron = (q + "norm_ff_macaron.weight") in c.p
        ff_scale = 0.5 if macaron else 1.0

        def ffn(x, nm, ln):
            h = c.ln(x, q + ln)
            h = c.act_drop(c.conv(h, q + nm + ".w_1", rb), "relu", rates["ffn"])
            h = c.conv(h, q + nm + ".w_2", rb)
            return c.resid_drop(x, h, rates["layer"], ff_scale)
        if macaron:
            x = ffn(x, "feed_forward_macaron", "norm_ff_macaron")
        # legacy relative-position self-attention (attention.py:164-206)
        h = c.ln(x, q + "norm_mha")
        a = q + "self_attn."
        wqkv = torch.cat([c.p[a + "linear_q.weight"], c.p[a + "linear_k.weight"], c.p[a + "linear_v.weight"]], 0).unsqueeze(-1)
        bqkv = torch.cat([c.p[a + "linear_q.bias"], c.p[a + "linear_k.bias"], c.p[a + "linear_v.bias"]], 0)
        qkv2 = A.Conv1dFunction.apply(h, wqkv, bqkv, rb, 1, 0)                               # (rows, 3 A)
        qu, qv, kh, vh = A.QKVSplit.apply(qkv2, c.p[a + "pos_bias_u"], c.p[a + "pos_bias_v"], B, T, H)      # each (B, H, T, dk), one launch
        ph = A.Conv1dFunction.apply(pos, c.p[a + "linear_pos.weight"].unsqueeze(-1), None, rbp, 1, 0).view(n_pos, H, dk).permute(1, 0, 2)
        ph = ph.contiguous()
        ac = A.BMM.apply(qu, kh, True)                                                      # q k^T, q p^T, P v: jatts_bgemm (exact-f32 MFMA), forward
        bd = A.BMM.apply(qv, ph, True)                                                      # and backward; p_h is shared over the batch
        p_attn = A.ShiftSoftmax.apply(ac, bd, kv, 1.0 / math.sqrt(dk), 2 if rel_style == "new" else 1)
        p_attn = c.drop(p_attn, rates["attn"])
        ctxv = A.BMM.apply(p_attn, vh, False).permute(0, 2, 1, 3).reshape(B * T, Ad)
        x = c.resid_drop(x, c.conv(ctxv, a + "linear_out", rb), rates["layer"])
        # convolution module (convolution.py:56-79)
        if (q + "conv_module.pointwise_conv1.weight") in c.p:
            m = q + "conv_module."
            h = c.ln(x, q + "norm_conv")
            h = A.GLU.apply(c.conv(h, m + "pointwise_conv1", rb))
            h = A.DepthwiseConv.apply(h, c.p[m + "depthwise_conv.weight"], c.p[m + "depthwise_conv.bias"], rb)
            h = A.Act.apply(c.bn(h, m + "norm"), "swish")
            x = c.resid_drop(x, c.conv(h, m + "pointwise_conv2", rb), rates["layer"])
        x = ffn(x, "feed_forward", "norm_ff")
        if (q + "norm_final.weight") in c.p:
            x = c.ln(x, q + "norm_final")
        i += 1
    if (prefix + "after_norm.weight") in c.p:
        x = c.ln(x, prefix + "after_norm")
    return x


def _predictor(c, prefix, x, rb, rate):
    """duration_predictor.py:60-97 / variance_predictor.py:47-85: [Conv1d -> ReLU -> LayerNorm -> Dropout] x n -> Linear(-> 1)."""
    i = 0
    while (prefix + f"conv.{i}.0.weight") in c.p:
        x = A.Act.apply(c.conv(x, prefix + f"conv.{i}.0", rb), "relu")
        x = c.drop(c.ln(x, prefix + f"conv.{i}.2"), rate)
        i += 1
    return A.RowDot.apply(x, c.p[prefix + "linear.weight"], c.p[prefix + "linear.bias"])


def _embed1(c, name, v, rb):
    """nn.Conv1d(1, adim, k, padding=(k-1)//2) on a per-token scalar (fastspeech2.py:366-393)."""
    w = c.p[name + ".weight"]
    if w.shape[-1] == 1:
        return A.OuterRows.apply(v, w, c.p[name + ".bias"])
    return c.conv(v.reshape(-1, 1), name, rb)


def train_forward(model, text, text_lengths, feats, feats_lengths, durations, durations_lengths, pitch, pitch_lengths, energy,
                  energy_lengths, spembs=None, sids=None, seed=0):
    """-> the reference's return dict {before_outs, after_outs, d_outs, p_outs, e_outs, ys, olens}, differentiable."""
    dev = model.feat_out.weight.device
    if dev.type != "cuda":
        raise hip._abi.JattsHipError("jatts_amd.FastSpeech2 trains on the GPU only (no CPU fallback); call .to('cuda')")
    hip._abi.load()
    c = _Ctx(model, seed)
    R = model.dropout_rates
    Ad = model.adim
    ilens = [int(v) for v in text_lengths.tolist()]
    olens = feats_lengths
    B, Tm = len(ilens), max(ilens)
    xs = text[:, :Tm].to(dev)
    ys = feats[:, : int(feats_lengths.max())]
    ds = durations[:, : int(durations_lengths.max())].to(dev)
    ps = pitch[:, : int(pitch_lengths.max())].to(dev).float()
    es = energy[:, : int(energy_lengths.max())].to(dev).float()
    if ds.shape[1] != Tm or ps.shape[1] != Tm or es.shape[1] != Tm:
        raise ValueError("durations / pitch / energy must be padded to the text length")
    rb = hip.RaggedBatch([Tm] * B, dev)
    # length-derived device tensors and the one host read-back (output lengths = duration sums) happen here, before GPU work is queued
    kv = hip.h2d(ilens, torch.int32, dev)
    kvo = hip.h2d([int(v) for v in olens.tolist()], torch.int32, dev) if not olens.is_cuda else olens.to(torch.int32).contiguous()
    ids = xs.reshape(-1).to(torch.int64).contiguous()
    x = A.Embedding.apply(ids, c.p["encoder.embed.0.weight"], math.sqrt(Ad), model.padding_idx)
    if getattr(model, "_static_olens", None) is not None:      # graph mode: the trainer checked once that sum(durations) == these lengths
        ol_h = list(model._static_olens)
    else:
        ol_h = [int(v) for v in ds.sum(1).tolist()]
        hip.check_bad_ids(dev)     # ids outside the table (zero rows, counted by the kernel) raise here like nn.Embedding: the sync above is needed anyway
    x = c.drop(x, R["enc_pos"])
    hs = _conformer(c, "encoder.", x, rb, kv, model.aheads, dict(pos=R["enc_pos"], layer=R["enc"], ffn=R["enc"], attn=R["enc_attn"]))
    if model.spks is not None:                                       # fastspeech2.py:589-592
        rbs = hip.RaggedBatch([1] * B, dev)
        sid = A.Embedding.apply(sids.to(dev).view(-1).to(torch.int64).contiguous(), c.p["sid_emb.weight"], 1.0, -1)
        hs = A.AddSeqVector.apply(hs, sid, rb)
    if model.spk_embed_dim is not None:                              # :594-597, _integrate_with_spk_embed "add" (:750-753)
        rbs = hip.RaggedBatch([1] * B, dev)
        hs = spk_integrate(c, model, hs, spembs, rb, rbs)
    p_outs = A.MaskRows.apply(_predictor(c, "pitch_predictor.", hs.detach() if model.stop_gradient_from_pitch_predictor else hs, rb,
                                         R["pitch"]), rb, kv)
    e_outs = A.MaskRows.apply(_predictor(c, "energy_predictor.", hs.detach() if model.stop_gradient_from_energy_predictor else hs, rb,
                                         R["energy"]), rb, kv)
    d_outs = A.MaskRows.apply(_predictor(c, "duration_predictor.", hs, rb, R["dur"]), rb, kv)
    p_emb = c.drop(_embed1(c, "pitch_embed.0", ps.reshape(-1).contiguous(), rb), R["pitch_embed"])
    e_emb = c.drop(_embed1(c, "energy_embed.0", es.reshape(-1).contiguous(), rb), R["energy_embed"])
    hs = hs + e_emb + p_emb
    d_used = ds.reshape(-1).to(torch.int64).contiguous()
    if sum(ol_h) == 0:     # (length_regulator.py:85-94: the whole-batch all-zero rule -> every duration becomes 1)
        _, cum, _, _ = hip.lr_durations(rb, d_used, 1.0, zero_rule=1)
        ol_h = [Tm] * B
    else:
        _, cum, _, _ = hip.lr_durations(rb, d_used, 1.0, zero_rule=0)
    To = max(ol_h)
    rbo = hip.RaggedBatch([To] * B, dev)
    yl = A.LengthRegulate.apply(hs, rb, cum, rbo) * math.sqrt(Ad)
    yl = c.drop(yl, R["dec_pos"])
    zs = _conformer(c, "decoder.", yl, rbo, kvo, model.aheads, dict(pos=R["dec_pos"], layer=R["dec"], ffn=R["dec"], attn=R["dec_attn"]))
    before = c.conv(zs, "feat_out", rbo)
    after = None
    i = 0
    if "postnet.postnet.0.0.weight" in c.p:
        n = 0
        while f"postnet.postnet.{n}.0.weight" in c.p:
            n += 1
        h = before
        for i in range(n):
            q = f"postnet.postnet.{i}."
            h = c.conv(h, q + "0", rbo, bias=False)
            if (q + "1.weight") in c.p:
                h = c.bn(h, q + "1")
            if i < n - 1:
                h = A.Act.apply(h, "tanh")
            h = c.drop(h, R["postnet"])
        after = before + h
    od = model.odim
    return {
        "before_outs": before.view(B, To, od),
        "after_outs": None if after is None else after.view(B, To, od),
        "d_outs": d_outs.view(B, Tm), "p_outs": p_outs.view(B, Tm, 1), "e_outs": e_outs.view(B, Tm, 1),
        "ys": ys, "olens": olens,
    }


def criterion(ret, durations, pitch, energy, ilens, use_masking=True):
    """Differentiable `_train_step` loss block (trainers/fastspeech2.py:62-84) -> dict of scalars incl. "loss"."""
    before, after, ys, olens = ret["before_outs"], ret["after_outs"], ret["ys"], ret["olens"]
    dev = before.device
    B, To, od = before.shape
    Tm = ret["d_outs"].shape[1]
    rbo, rbt = hip.RaggedBatch([To] * B, dev), hip.RaggedBatch([Tm] * B, dev)
    _len32 = lambda t: t.to(torch.int32) if t.is_cuda else hip.h2d([int(v) for v in t.tolist()], torch.int32, dev)  # noqa: E731  (cached: no copy under capture)
    vo = _len32(olens) if use_masking else None
    vi = _len32(ilens) if use_masking else None
    n_o = float(int(olens.sum()) if use_masking else B * To) * od
    n_i = float(int(ilens.sum()) if use_masking else B * Tm)
    ys2 = ys.to(dev).float().reshape(B * To, od).contiguous()
    mel = A.MaskedLoss.apply(before.reshape(B * To, od), ys2, rbo, vo, 0, 1.0 / n_o, -1.0)
    if after is not None:
        mel = mel + A.MaskedLoss.apply(after.reshape(B * To, od), ys2, rbo, vo, 0, 1.0 / n_o, -1.0)
    flat = lambda t: t.to(dev).float().reshape(B * Tm, 1).contiguous()  # noqa: E731
    dur = A.MaskedLoss.apply(ret["d_outs"].reshape(B * Tm, 1), flat(durations[:, :Tm]), rbt, vi, 1, 1.0 / n_i, 1.0)
    pit = A.MaskedLoss.apply(ret["p_outs"].reshape(B * Tm, 1), flat(pitch[:, :Tm]), rbt, vi, 1, 1.0 / n_i, -1.0)
    ene = A.MaskedLoss.apply(ret["e_outs"].reshape(B * Tm, 1), flat(energy[:, :Tm]), rbt, vi, 1, 1.0 / n_i, -1.0)
    return dict(mel_loss=mel, duration_loss=dur, pitch_loss=pit, energy_loss=ene, loss=mel + dur + pit + ene)
