"""Conformer encoder/decoder stack on the HIP path (host-side schedule only).

Mirrors jatts.modules.conformer.encoder.Encoder.forward (encoder.py:233-289) and
EncoderLayer.forward (encoder_layer.py:78-178) for eval mode, normalize_before=True,
legacy relative positional attention, on PACKED RAGGED batches: every utterance is
processed exactly as the reference's B=1 ``inference()`` does (no pad leakage; SURVEY
§8 note N1).  All arithmetic is in libjatts_hip.so; this file only packs weights once
and sequences kernel launches.
"""
import math
import os

import torch

from .. import hip
from ..hip import ACT_NONE, ACT_RELU

QKV_ONE_LAUNCH = os.environ.get("JATTS_QKV_ONE_LAUNCH", "1") != "0"     # A/B switch (tools/); the two forms are bit-identical (tests/test_kernels_gpu.py)
LN_EPS = 1e-12  # modules/transformer/layer_norm.py:23
BN_EPS = 1e-5   # torch.nn.BatchNorm1d default
PE_TABLE_LEN = 5000  # modules/positional_encoding.py:26


def legacy_rel_pos_table(n, d, table_len):
    """pe[p] = sin/cos((L-1-p) w_i), p < n  (positional_encoding.py:36-57, reverse=True).
    Built on the host in f32 exactly like the reference builds its buffer."""
    position = torch.arange(table_len - 1, -1, -1.0, dtype=torch.float32)[:n].unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))
    pe = torch.zeros(n, d, dtype=torch.float32)
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe


def rel_pos_table_new(cap, d):
    """RelPositionalEncoding (positional_encoding.py:265-309): row m encodes relative position
    cap-1-m, m in [0, 2cap-1); independent of the reference's table length."""
    pos = torch.arange(cap - 1, -cap, -1.0, dtype=torch.float32).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))
    pe = torch.zeros(2 * cap - 1, d, dtype=torch.float32)
    pe[:, 0::2] = torch.sin(pos * div_term)
    pe[:, 1::2] = torch.cos(pos * div_term)
    return pe


class PackedConv:
    """A Conv1d/Linear weight packed for jatts_conv1d."""

    def __init__(self, w, b, dtype, device, scale=None, shift=None, c_mult=64):
        # w: (n, c, k) or (n, c); optional per-output-channel affine folded in (BatchNorm eval)
        w = w.detach().float()
        if w.dim() == 2:
            w = w.unsqueeze(-1)
        b = None if b is None else b.detach().float()
        if scale is not None:
            w = w * scale.view(-1, 1, 1)
            b = (b * scale if b is not None else torch.zeros_like(scale)) + shift
        self.n_out, c, self.k = w.shape
        self.c_in = hip.round_up(c, c_mult)
        # set_precision("fp32_split"): f32 tensors, split f16 hi/lo MFMA operands (hip.split_weights is on while the model prepares)
        # set_precision("fp32_bf16x3"): f32 tensors, three exact bf16 terms per operand, seven (fp32_bf16x3_6p: six) MFMA products (hip.EmulWeight)
        self.w = hip.f32_operand(w.to(device), c_mult) if dtype == hip.F32 else hip.pack_conv_weight(w.to(device), dtype, c_mult)
        self.b = None if b is None else b.to(device).contiguous()


class SpkProjection:
    """_integrate_with_spk_embed (models/fastspeech2.py:737-761; the same method in matchatts.py:560-584, matchatts_mas.py:644-668,
    vits.py:681-705).  "add": hs += projection(normalize(spembs)).  "concat": projection(cat[hs, normalize(spembs) broadcast over time])
    -- the Linear over the concatenation is split by columns, W [hs; s] = W_h hs + (W_s s + b): one k = 1 conv over the frames plus a
    per-utterance vector, so the (B, T, adim + D) concatenation never exists."""

    def __init__(self, sd, adim, kind, dtype, device):
        if kind not in ("add", "concat"):
            raise NotImplementedError("support only add or concat.")       # the reference's own message (:759)
        w, b = sd["projection.weight"], sd["projection.bias"]
        self.kind, self.A, self.dtype = kind, adim, dtype
        if kind == "add":
            self.h, self.s = None, PackedConv(w, b, dtype, device)
        else:
            if w.shape[1] <= adim:
                raise ValueError("projection.weight: 'concat' expects (adim, adim + spk_embed_dim)")
            self.h, self.s = PackedConv(w[:, :adim], None, dtype, device), PackedConv(w[:, adim:], b, dtype, device)

    def __call__(self, rb, hs, spembs):
        """hs: f32 (rows, adim) of the ragged batch rb; spembs: (n_seq, D).  -> hs with the speaker integrated (in place for "add")."""
        dev, B = hs.device, rb.n_seq
        rbs = hip.RaggedBatch([1] * B, dev)
        sp = hip.l2_normalize(spembs.to(dev).float().reshape(B, -1).contiguous(), self.dtype, ldy=self.s.c_in)
        vec = hip.conv1d(rbs, sp, self.s.w, self.s.c_in, self.A, 1, dtype=self.dtype, bias=self.s.b, out_f32=True)
        if self.h is not None:
            hs = hip.conv1d(rb, hip.affine_cast(hs, self.dtype), self.h.w, self.h.c_in, self.A, 1, dtype=self.dtype, out_f32=True)
        hip.add_seq_vector(rb, hs, vec)
        return hs


class ConformerRunner:
    def __init__(self, sd, prefix, n_heads, dtype, device, rel_style="legacy"):
        """rel_style: "legacy" = LegacyRelPositionalEncoding + LegacyRelPositionMultiHeadedAttention
        (FastSpeech2 / Matcha, fastspeech2.py fallback), "new" = RelPositionalEncoding +
        RelPositionMultiHeadedAttention (VITS text encoder and decoder)."""
        self.dtype, self.device, self.H = dtype, device, n_heads
        self.split = dtype == hip.F32 and hip._SPLIT_WEIGHTS[0] == 1     # set_precision("fp32_split"): the attention takes the split arithmetic
        self.wmode = hip._SPLIT_WEIGHTS[0] if dtype == hip.F32 else 0    # the run-time packed position operands follow the weight mode
        self.rel_style = rel_style
        g = lambda k: sd[prefix + k]  # noqa: E731
        self.n_layers = 0
        while (prefix + f"encoders.{self.n_layers}.norm_mha.weight") in sd:
            self.n_layers += 1
        self.A = g("encoders.0.norm_mha.weight").shape[0]
        self.dk = self.A // n_heads
        if self.A % 64 or self.dk % 32:
            raise NotImplementedError("attention_dim must be a multiple of 64 and d_k a multiple of 32 (MFMA tiles)")
        f32 = lambda t: t.detach().float().to(device).contiguous()  # noqa: E731
        self.layers = []
        for i in range(self.n_layers):
            p = f"encoders.{i}."
            L = {}
            has = lambda k: (prefix + p + k) in sd  # noqa: E731
            for nm in ("norm_ff", "norm_mha", "norm_ff_macaron", "norm_conv", "norm_final"):
                if has(nm + ".weight"):
                    L[nm] = (f32(g(p + nm + ".weight")), f32(g(p + nm + ".bias")))
            for ff in ("feed_forward", "feed_forward_macaron"):
                if has(ff + ".w_1.weight"):
                    L[ff] = (PackedConv(g(p + ff + ".w_1.weight"), g(p + ff + ".w_1.bias"), dtype, device),
                             PackedConv(g(p + ff + ".w_2.weight"), g(p + ff + ".w_2.bias"), dtype, device))
            a = p + "self_attn."
            wq, wk = g(a + "linear_q.weight"), g(a + "linear_k.weight")
            L["qk"] = PackedConv(torch.cat([wq, wk], 0), torch.cat([g(a + "linear_q.bias"), g(a + "linear_k.bias")], 0),
                                 dtype, device)
            L["v"] = PackedConv(g(a + "linear_v.weight"), g(a + "linear_v.bias"), dtype, device)
            if (2 * self.A) % 256 == 0:     # Q | K | V as ONE launch (jatts_conv_desc.n_split): Q | K row-major, V transposed
                L["qkv"] = PackedConv(torch.cat([wq, wk, g(a + "linear_v.weight")], 0),
                                      torch.cat([g(a + "linear_q.bias"), g(a + "linear_k.bias"), g(a + "linear_v.bias")], 0), dtype, device)
            L["o"] = PackedConv(g(a + "linear_out.weight"), g(a + "linear_out.bias"), dtype, device)
            L["rel"] = has("self_attn.linear_pos.weight")
            if L["rel"]:
                L["pos"] = PackedConv(g(a + "linear_pos.weight"), None, dtype, device)
                L["u"] = f32(g(a + "pos_bias_u"))
                L["vb"] = f32(g(a + "pos_bias_v"))
            if has("conv_module.pointwise_conv1.weight"):
                c = p + "conv_module."
                L["pw1"] = PackedConv(g(c + "pointwise_conv1.weight"), g(c + "pointwise_conv1.bias"), dtype, device)
                L["pw2"] = PackedConv(g(c + "pointwise_conv2.weight"), g(c + "pointwise_conv2.bias"), dtype, device)
                wd = g(c + "depthwise_conv.weight").detach().float()
                L["dw_w"] = f32(wd.reshape(wd.shape[0], wd.shape[-1]))
                s = g(c + "norm.weight").detach().float() / torch.sqrt(g(c + "norm.running_var").detach().float() + BN_EPS)
                t = g(c + "norm.bias").detach().float() + (g(c + "depthwise_conv.bias").detach().float()
                                                            - g(c + "norm.running_mean").detach().float()) * s
                L["dw_s"], L["dw_t"], L["dw_k"] = f32(s), f32(t), wd.shape[-1]
            self.layers.append(L)
        self.after_norm = None
        if (prefix + "after_norm.weight") in sd:
            self.after_norm = (f32(g("after_norm.weight")), f32(g("after_norm.bias")))
        self.pe_len = PE_TABLE_LEN   # sticky, like the reference's regrown buffer (extend_pe)
        self._pos_cache = {}         # cap -> per-layer (per-head packed P, cv)

    # -- positional projections: P_l = linear_pos_l(pe[:cap]); batch independent, cached per cap
    def _pos(self, t_max):
        if t_max > self.pe_len:  # positional_encoding.py:36-43: table regrown, values change
            self.pe_len = t_max
            self._pos_cache.clear()
        cap = hip.round_up(t_max, 128)
        if cap in self._pos_cache:
            return cap, hip.keep(self._pos_cache[cap])      # (a capturing graph pins what it was handed: the cache evicts)
        if self.rel_style == "new":
            pe = rel_pos_table_new(cap, self.A)          # (2cap-1, A)
        else:
            n_valid = min(cap, self.pe_len)
            pe = torch.zeros(cap, self.A, dtype=torch.float32)
            pe[:n_valid] = legacy_rel_pos_table(n_valid, self.A, self.pe_len)
        n_pos = pe.shape[0]
        pe_t = hip.affine_cast(pe.to(self.device), self.dtype)
        rb = hip.RaggedBatch([n_pos], self.device)
        per_layer = []
        for L in self.layers:
            if not L["rel"]:
                per_layer.append(None)
                continue
            P = hip.conv1d(rb, pe_t, L["pos"].w, L["pos"].c_in, self.A, 1, dtype=self.dtype)  # (cap, A)
            cv = hip.rowdot(P, self.A, n_pos, self.H, self.dk, L["vb"]).t().contiguous()      # (H, n_pos)
            # per-head weight operand of the BD GEMM (n = position m, contraction d_k): pure re-layout
            pk = ((lambda w: hip.SplitWeight(w, 64)) if self.wmode == 1 else (lambda w: hip.EmulWeight(w, 64, hip.EMUL_CODE[self.wmode])) if self.wmode in hip.EMUL_CODE
                  else (lambda w: hip.pack_conv_weight(w, self.dtype)))
            heads = [pk(P[:, h * self.dk:(h + 1) * self.dk].float().unsqueeze(-1)) for h in range(self.H)]
            per_layer.append((heads, cv))
        if len(self._pos_cache) >= 8:
            self._pos_cache.pop(next(iter(self._pos_cache)))
        self._pos_cache[cap] = per_layer
        return cap, hip.keep(per_layer)

    def _ffn(self, rb, x, ln, ff, scale):
        xn = hip.layernorm(x, ln[0], ln[1], self.dtype, LN_EPS)
        w1, w2 = ff
        h = hip.conv1d(rb, xn, w1.w, w1.c_in, w1.n_out, w1.k, dtype=self.dtype, bias=w1.b, act=ACT_RELU)
        hip.conv1d(rb, h, w2.w, w2.c_in, w2.n_out, w2.k, dtype=self.dtype, bias=w2.b, act=ACT_NONE,
                   alpha=scale, resid=x, out=x, out_f32=True)

    def _mha(self, rb, x, L, pos, kv_len=None):
        A, H, dk = self.A, self.H, self.dk
        xn = hip.layernorm(x, L["norm_mha"][0], L["norm_mha"][1], self.dtype, LN_EPS)
        vcol, ldvt = rb.vt_layout()
        if "qkv" in L and QKV_ONE_LAUNCH:
            qk, vt = hip.conv1d(rb, xn, L["qkv"].w, A, 3 * A, 1, dtype=self.dtype, bias=L["qkv"].b, split=(2 * A, ldvt, vcol))   # (R, 2A), (A, ldvt)
        else:
            qk = hip.conv1d(rb, xn, L["qk"].w, A, 2 * A, 1, dtype=self.dtype, bias=L["qk"].b)       # (R, 2A)
            vt = hip.conv1d(rb, xn, L["v"].w, A, A, 1, dtype=self.dtype, bias=L["v"].b, transposed=True,
                            out_ld=ldvt, y_seq_col0=vcol)                                            # (A, ldvt)
        g = ku = None
        ldg = 0
        rel_mode, rel_center = 1, 0
        if L["rel"]:
            cap, (heads, cv) = pos
            n_pos = cv.shape[1]                       # cap (legacy) or 2cap-1 (new)
            ldg = hip.round_up(n_pos, 32)
            if self.rel_style == "new":
                rel_mode, rel_center = 2, cap - 1
            ku = hip.rowdot(qk, 2 * A, rb.total, H, dk, L["u"], col0=A)                          # u . k_j
            g = torch.empty(rb.total, H * ldg, dtype=hip.torch_dtype(self.dtype), device=x.device)
            for h in range(H):  # g[row][h][m] = q_row,h . p_h[m] + v_h . p_h[m]
                hip.conv1d(rb, qk, heads[h], hip.round_up(dk, 64), n_pos, 1, dtype=self.dtype, bias=cv[h], ldx=2 * A,
                           x_col0=h * dk, out=g, out_ld=H * ldg, out_col0=h * ldg)
        ctx = hip.relpos_attention(rb, qk, 2 * A, qk, 2 * A, vt, ldvt, g, ldg, ku, 1.0 / math.sqrt(dk),
                                   H, dk, hip.F32S if self.split else self.dtype, q_col0=0, k_col0=A, rel_mode=rel_mode, rel_center=rel_center, vt_col0=vcol,
                                   kv_len=kv_len)
        hip.conv1d(rb, ctx, L["o"].w, A, A, 1, dtype=self.dtype, bias=L["o"].b, resid=x, out=x, out_f32=True)

    def _convmod(self, rb, x, L):
        A = self.A
        xn = hip.layernorm(x, L["norm_conv"][0], L["norm_conv"][1], self.dtype, LN_EPS)
        pw = hip.conv1d(rb, xn, L["pw1"].w, A, 2 * A, 1, dtype=self.dtype, bias=L["pw1"].b)
        dw = hip.glu_dwconv_bn_swish(rb, pw, A, L["dw_k"], L["dw_w"], L["dw_s"], L["dw_t"], self.dtype)
        hip.conv1d(rb, dw, L["pw2"].w, A, A, 1, dtype=self.dtype, bias=L["pw2"].b, resid=x, out=x, out_f32=True)

    def run(self, rb, x, final_dtype=None, taps=None, kv_len=None):
        """x: f32 (rows, A) = input-layer output BEFORE the x*sqrt(A) scaling is applied by the caller.
        Returns after_norm output as f32 (or ``final_dtype``).  x is updated in place.
        kv_len (int32 device tensor, n_seq): key padding mask of a PADDED batch (the reference's batched forward():
        encoder.py:233-289 with masks) -- only the attention sees it, exactly as in the reference."""
        pos = self._pos(rb.max_len) if any(L["rel"] for L in self.layers) else None  # (cap, per_layer)
        for i, L in enumerate(self.layers):
            macaron = "feed_forward_macaron" in L
            ff_scale = 0.5 if macaron else 1.0
            if macaron:
                self._ffn(rb, x, L["norm_ff_macaron"], L["feed_forward_macaron"], ff_scale)
            self._mha(rb, x, L, (pos[0], pos[1][i]) if pos else None, kv_len)
            if "pw1" in L:
                self._convmod(rb, x, L)
            self._ffn(rb, x, L["norm_ff"], L["feed_forward"], ff_scale)
            if "norm_final" in L:
                hip.layernorm(x, L["norm_final"][0], L["norm_final"][1], hip.F32, LN_EPS, out=x)
            if taps is not None:
                taps[f"layer{i}"] = x.clone()
        if self.after_norm is None:
            return x
        od = hip.F32 if final_dtype is None else final_dtype
        return hip.layernorm(x, self.after_norm[0], self.after_norm[1], od, LN_EPS)
