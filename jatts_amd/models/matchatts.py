"""Matcha-TTS on the MI355X HIP path — drop-ins for ``jatts.models.MatchaTTS_MAS`` (tts2 recipes, Gaussian
upsampling, SURVEY §8 A14-A15 / BASELINE config 3) and ``jatts.models.MatchaTTS`` (tts1 variant: hard
length regulator, no aligner).

Same constructor kwargs (reference models/matchatts_mas.py:52-114), same state_dict schema (conformer
encoder + duration predictor + encoder_proj + [alignment module] + CFM U-Net ``decoder.estimator.*`` with
diffusers-style ``attn1.to_q/to_k/to_v/to_out.0`` keys), same ``inference(text, ..., n_timesteps,
temperature)`` contract (:552-642).  ``noise`` can be injected for parity (the reference draws
torch.randn_like inside CFM.inference, flow_matching.py:64).

Per Euler step the U-Net runs as: Conv1d k3 (MFMA) -> GroupNorm+Mish(+time vector) kernel -> ... ->
LayerNorm -> fused SDPA (the attention kernel without relative-position terms) -> SnakeBeta FF (MFMA +
elementwise) -> stride-2 conv as a 2-tap conv on the paired-row view -> ConvTranspose as polyphase conv;
the Euler update x += dt * v is the epilogue of the final 1x1 projection.
"""
import logging
import math
from typing import Optional, Sequence

import torch

from .. import hip
from ..hip import ACT_MISH, ACT_NONE, ACT_RELU, ACT_SWISH
from . import _schema as S
from ._conformer import QKV_ONE_LAUNCH, ConformerRunner, PackedConv, SpkProjection
from .fastspeech2 import _Predictor

GN_EPS = 1e-5  # torch.nn.GroupNorm / LayerNorm defaults (decoder.py:71, transformer.py:213)


def _split_in(w, sizes):
    """Split a conv weight (o, sum(sizes), k) along the input channels."""
    out, o = [], 0
    for n in sizes:
        out.append(w[:, o:o + n])
        o += n
    return out


class _Resnet:
    """ResnetBlock1D (decoder.py:80-97) with its input given as a list of channel groups (concat-free)."""

    def __init__(self, sd, p, in_sizes, dt, dev):
        f32 = lambda t: t.detach().float().to(dev).contiguous()  # noqa: E731
        w1 = sd[p + "block1.block.0.weight"].detach().float()
        wr = sd[p + "res_conv.weight"].detach().float()
        self.c_out = w1.shape[0]
        self.in_sizes = in_sizes
        self.conv1 = [PackedConv(w, sd[p + "block1.block.0.bias"] if i == 0 else None, dt, dev)
                      for i, w in enumerate(_split_in(w1, in_sizes))]
        self.res = [PackedConv(w, sd[p + "res_conv.bias"] if i == 0 else None, dt, dev)
                    for i, w in enumerate(_split_in(wr, in_sizes))]
        self.gn1 = (f32(sd[p + "block1.block.1.weight"]), f32(sd[p + "block1.block.1.bias"]))
        self.conv2 = PackedConv(sd[p + "block2.block.0.weight"], sd[p + "block2.block.0.bias"], dt, dev)
        self.gn2 = (f32(sd[p + "block2.block.1.weight"]), f32(sd[p + "block2.block.1.bias"]))
        self.mlp = PackedConv(sd[p + "mlp.1.weight"], sd[p + "mlp.1.bias"], dt, dev)

    def run(self, rb, rbs, xs, tmish, dt, tv_cache=None, valid=None):
        """xs: list of operand-dtype tensors (rows, ld_i) matching in_sizes; tmish: (n_seq, Ct) = mish(time emb).
        Returns f32 (rows, c_out).  valid (int32 (n_seq,), padded batches only): the `x * mask` / `output * mask` steps of
        Block1D / ResnetBlock1D (decoder.py:75-77,93-96) -- xs must already be masked by the caller."""
        C = self.c_out
        h = None
        for x, cv in zip(xs, self.conv1):
            h = hip.conv1d(rb, x, cv.w, cv.c_in, C, 3, dtype=dt, bias=cv.b, resid=h, out=h, out_f32=True,
                           ldx=x.shape[1])
        # the time vector depends only on (Euler step, n_timesteps, batch size): cached across batches by the caller
        tv = tv_cache.get(id(self)) if tv_cache is not None else None
        if tv is None:
            tv = hip.conv1d(rbs, tmish, self.mlp.w, self.mlp.c_in, C, 1, dtype=dt, bias=self.mlp.b, out_f32=True)
            if tv_cache is not None:
                tv_cache[id(self)] = tv
        h = hip.groupnorm_mish(rb, h, C, 8, self.gn1[0], self.gn1[1], dt, GN_EPS, addvec=tv)
        if valid is not None:
            hip.zero_pad_rows(rb, h, valid)          # block1(x) * mask, + time vector, then block2's own x * mask
        h = hip.conv1d(rb, h, self.conv2.w, self.conv2.c_in, C, 3, dtype=dt, bias=self.conv2.b)
        out = hip.groupnorm_mish(rb, h, C, 8, self.gn2[0], self.gn2[1], hip.F32, GN_EPS)
        if valid is not None:
            hip.zero_pad_rows(rb, out, valid)        # block2 output * mask; res_conv(x * mask) below adds its bias at padded rows
        for x, cv in zip(xs, self.res):
            hip.conv1d(rb, x, cv.w, cv.c_in, C, 1, dtype=dt, bias=cv.b, resid=out, out=out, out_f32=True, ldx=x.shape[1])
        return out


class _TBlock:
    """BasicTransformerBlock (transformer.py:276-364): LN -> self-attention -> +x ; LN -> SnakeBeta FF -> +x."""

    def __init__(self, sd, p, heads, dt, dev):
        f32 = lambda t: t.detach().float().to(dev).contiguous()  # noqa: E731
        self.heads = heads
        self.split = dt == hip.F32 and hip._SPLIT_WEIGHTS[0] == 1      # set_precision("fp32_split"): the attention takes the split arithmetic too
        self.n1 = (f32(sd[p + "norm1.weight"]), f32(sd[p + "norm1.bias"]))
        self.n3 = (f32(sd[p + "norm3.weight"]), f32(sd[p + "norm3.bias"]))
        wq, wk = sd[p + "attn1.to_q.weight"], sd[p + "attn1.to_k.weight"]
        self.inner = wq.shape[0]
        self.dh = self.inner // heads
        if self.dh % 32:
            raise NotImplementedError("attention head dim must be a multiple of 32")
        self.qk = PackedConv(torch.cat([wq, wk], 0), None, dt, dev)
        self.v = PackedConv(sd[p + "attn1.to_v.weight"], None, dt, dev)
        # Q | K | V as ONE launch (jatts_conv_desc.n_split; transformer.py:222-260 issues three Linear calls): Q | K row-major, V transposed
        self.qkv = PackedConv(torch.cat([wq, wk, sd[p + "attn1.to_v.weight"]], 0), None, dt, dev) if (2 * self.inner) % 256 == 0 else None
        self.o = PackedConv(sd[p + "attn1.to_out.0.weight"], sd[p + "attn1.to_out.0.bias"], dt, dev)
        self.ff1 = PackedConv(sd[p + "ff.net.0.proj.weight"], sd[p + "ff.net.0.proj.bias"], dt, dev)
        self.ff2 = PackedConv(sd[p + "ff.net.2.weight"], sd[p + "ff.net.2.bias"], dt, dev)
        self.alpha = f32(torch.exp(sd[p + "ff.net.0.alpha"].detach().float()))
        self.inv_beta = f32(1.0 / (torch.exp(sd[p + "ff.net.0.beta"].detach().float()) + 1e-9))

    def run(self, rb, x, dt, key_bias=None):
        """x: f32 (rows, C), updated in place.  key_bias f32 (rows, heads) or None: added to every score of that key AFTER the
        1/sqrt(d) scaling -- diffusers adds `attention_mask` to the scores [recalled]; Matcha passes its 1/0 frame mask."""
        C, I = x.shape[1], self.inner
        n = hip.layernorm(x, self.n1[0], self.n1[1], dt, GN_EPS)
        vcol, ldvt = rb.vt_layout()
        if self.qkv is not None and QKV_ONE_LAUNCH:
            qk, vt = hip.conv1d(rb, n, self.qkv.w, self.qkv.c_in, 3 * I, 1, dtype=dt, split=(2 * I, ldvt, vcol))
        else:
            qk = hip.conv1d(rb, n, self.qk.w, self.qk.c_in, 2 * I, 1, dtype=dt)
            vt = hip.conv1d(rb, n, self.v.w, self.v.c_in, I, 1, dtype=dt, transposed=True, out_ld=ldvt, y_seq_col0=vcol)
        a = hip.relpos_attention(rb, qk, 2 * I, qk, 2 * I, vt, ldvt, None, 0, key_bias, self.dh ** -0.5, self.heads,
                                 self.dh, hip.F32S if self.split else dt, q_col0=0, k_col0=I, rel_mode=0, vt_col0=vcol)
        hip.conv1d(rb, a, self.o.w, self.o.c_in, C, 1, dtype=dt, bias=self.o.b, resid=x, out=x, out_f32=True)
        n = hip.layernorm(x, self.n3[0], self.n3[1], dt, GN_EPS)
        # SnakeBeta rides in the epilogue of the conv that feeds it (one launch and one round trip of the widest tensor less)
        u = hip.conv1d(rb, n, self.ff1.w, self.ff1.c_in, self.ff1.n_out, 1, dtype=dt, bias=self.ff1.b, snake=(self.alpha, self.inv_beta))
        hip.conv1d(rb, u, self.ff2.w, self.ff2.c_in, C, 1, dtype=dt, bias=self.ff2.b, resid=x, out=x, out_f32=True)


class _MatchaBase(torch.nn.Module):
    _MAS = True

    def __init__(
        self, idim: int, odim: int, adim: int = 384, aheads: int = 4, elayers: int = 6, eunits: int = 1536,
        positionwise_layer_type: str = "conv1d", positionwise_conv_kernel_size: int = 1, use_scaled_pos_enc: bool = True,
        use_batch_norm: bool = True, encoder_normalize_before: bool = True, encoder_concat_after: bool = False,
        reduction_factor: int = 1, encoder_type: str = "transformer", transformer_enc_dropout_rate: float = 0.1,
        transformer_enc_positional_dropout_rate: float = 0.1, transformer_enc_attn_dropout_rate: float = 0.1,
        conformer_rel_pos_type: str = "legacy", conformer_pos_enc_layer_type: str = "rel_pos",
        conformer_self_attn_layer_type: str = "rel_selfattn", conformer_activation_type: str = "swish",
        use_macaron_style_in_conformer: bool = True, use_cnn_in_conformer: bool = True, zero_triu: bool = False,
        conformer_enc_kernel_size: int = 7, conformer_dec_kernel_size: int = 31, decoder_channels=(256, 256),
        decoder_dropout: float = 0.05, decoder_attention_head_dim: int = 64, decoder_n_blocks: int = 1,
        decoder_num_mid_blocks: int = 2, decoder_num_heads: int = 2, decoder_act_fn: str = "snakebeta",
        duration_predictor_type: str = "deterministic", duration_predictor_layers: int = 2,
        duration_predictor_chans: int = 384, duration_predictor_kernel_size: int = 3,
        duration_predictor_dropout_rate: float = 0.1, spks: Optional[int] = None, spk_embed_dim: Optional[int] = None,
        spk_embed_integration_type: str = "add", use_gst: bool = False, gst_tokens: int = 10, gst_heads: int = 4,
        gst_conv_layers: int = 6, gst_conv_chans_list: Sequence[int] = (32, 32, 64, 64, 128, 128),
        gst_conv_kernel_size: int = 3, gst_conv_stride: int = 2, gst_gru_layers: int = 1, gst_gru_units: int = 128,
        init_type: str = "xavier_uniform", init_enc_alpha: float = 1.0, use_masking: bool = False,
        use_weighted_masking: bool = False, **unused,
    ):
        super().__init__()
        if encoder_type != "conformer":
            raise ValueError(f"{encoder_type} is not supported (only 'conformer'; the transformer branch is dead code)")
        if conformer_rel_pos_type != "legacy" or zero_triu or encoder_concat_after or not encoder_normalize_before:
            raise NotImplementedError("only the reference defaults: legacy rel-pos, normalize_before, no concat_after")
        if duration_predictor_type != "deterministic" or use_gst or reduction_factor != 1:
            raise NotImplementedError("stochastic duration predictor / GST / reduction_factor > 1 are not supported")
        if decoder_act_fn != "snakebeta":
            raise NotImplementedError("only decoder_act_fn='snakebeta' (the recipes' setting)")
        if len(decoder_channels) != 2:
            raise NotImplementedError("decoder_channels must have two levels (one down/up-sampling), as in the recipes")
        self.idim, self.odim, self.adim, self.aheads = idim, odim, adim, aheads
        self.dec_heads, self.dec_channels = decoder_num_heads, tuple(decoder_channels)
        self.n_blocks, self.n_mid = decoder_n_blocks, decoder_num_mid_blocks
        self.sigma_min = 1e-4   # CFM default (flow_matching.py:31); the reference never overrides it
        self.spk_embed_dim = spk_embed_dim if (spk_embed_dim is not None and spk_embed_dim > 0) else None
        self.spk_embed_integration_type = spk_embed_integration_type
        if self.spk_embed_dim is not None and spk_embed_integration_type not in ("add", "concat"):
            raise NotImplementedError("support only add or concat.")
        spec = S.new_spec()
        spec["encoder.embed.0.weight"] = ((idim, adim), "param")
        S.conformer_spec(spec, "encoder.", adim, aheads, eunits, elayers, positionwise_layer_type,
                         positionwise_conv_kernel_size, use_macaron_style_in_conformer, use_cnn_in_conformer,
                         conformer_enc_kernel_size)
        if spks is not None and spks > 1:
            spec["sid_emb.weight"] = ((spks, adim), "param")
        self.spks = spks if (spks is not None and spks > 1) else None
        if self.spk_embed_dim is not None:
            S._lin(spec, "projection", adim, self.spk_embed_dim + (adim if spk_embed_integration_type == "concat" else 0))
        S._lin(spec, "encoder_proj", odim, adim)
        S.predictor_spec(spec, "duration_predictor.", adim, duration_predictor_layers, duration_predictor_chans,
                         duration_predictor_kernel_size)
        if self._MAS:
            for nm, o, i, k in (("t_conv1", adim, adim, 3), ("t_conv2", adim, adim, 1), ("f_conv1", adim, odim, 3),
                                ("f_conv2", adim, adim, 3), ("f_conv3", adim, adim, 1)):
                S._conv(spec, "alignment_module." + nm, o, i, k)
        # CFM estimator (decoder.py:243-411)
        e = "decoder.estimator."
        in_ch = 2 * odim
        ch = self.dec_channels
        ted = ch[0] * 4
        inner = decoder_num_heads * decoder_attention_head_dim
        S._lin(spec, e + "time_mlp.linear_1", ted, in_ch)
        S._lin(spec, e + "time_mlp.linear_2", ted, ted)

        def resnet(p, ci, co):
            S._lin(spec, p + "mlp.1", co, ted)
            S._conv(spec, p + "block1.block.0", co, ci, 3)
            S._norm(spec, p + "block1.block.1", co)
            S._conv(spec, p + "block2.block.0", co, co, 3)
            S._norm(spec, p + "block2.block.1", co)
            S._conv(spec, p + "res_conv", co, ci, 1)

        def tblock(p, dim):
            S._norm(spec, p + "norm1", dim)
            for nm in ("to_q", "to_k", "to_v"):
                S._lin(spec, p + "attn1." + nm, inner, dim, bias=False)
            S._lin(spec, p + "attn1.to_out.0", dim, inner)
            S._norm(spec, p + "norm3", dim)
            spec[p + "ff.net.0.alpha"] = ((4 * dim,), "param")
            spec[p + "ff.net.0.beta"] = ((4 * dim,), "param")
            S._lin(spec, p + "ff.net.0.proj", 4 * dim, dim)
            S._lin(spec, p + "ff.net.2", dim, 4 * dim)

        oc = in_ch
        for i in range(len(ch)):
            ic, oc = oc, ch[i]
            resnet(e + f"down_blocks.{i}.0.", ic, oc)
            for j in range(decoder_n_blocks):
                tblock(e + f"down_blocks.{i}.1.{j}.", oc)
            S._conv(spec, e + f"down_blocks.{i}.2" + (".conv" if i < len(ch) - 1 else ""), oc, oc, 3)
        for i in range(decoder_num_mid_blocks):
            resnet(e + f"mid_blocks.{i}.0.", ch[-1], ch[-1])
            for j in range(decoder_n_blocks):
                tblock(e + f"mid_blocks.{i}.1.{j}.", ch[-1])
        up = ch[::-1] + (ch[0],)
        for i in range(len(up) - 1):
            resnet(e + f"up_blocks.{i}.0.", 2 * up[i], up[i + 1])
            for j in range(decoder_n_blocks):
                tblock(e + f"up_blocks.{i}.1.{j}.", up[i + 1])
            if i < len(up) - 2:
                spec[e + f"up_blocks.{i}.2.conv.weight"] = ((up[i + 1], up[i + 1], 4), "param")  # ConvTranspose1d
                spec[e + f"up_blocks.{i}.2.conv.bias"] = ((up[i + 1],), "param")
            else:
                S._conv(spec, e + f"up_blocks.{i}.2", up[i + 1], up[i + 1], 3)
        S._conv(spec, e + "final_block.block.0", up[-1], up[-1], 3)
        S._norm(spec, e + "final_block.block.1", up[-1])
        S._conv(spec, e + "final_proj", odim, up[-1], 1)
        S.build_from_spec(self, spec)
        # train-mode behaviour (models/matchatts_train.py): the reference's dropout sites
        self.dropout_rates = dict(enc=transformer_enc_dropout_rate, enc_pos=transformer_enc_positional_dropout_rate,
                                  enc_attn=transformer_enc_attn_dropout_rate, dur=duration_predictor_dropout_rate, decoder=decoder_dropout)
        self._train_calls = 0
        self.precision = "fp32"   # the reference's arithmetic; set_precision("fp16") selects the fast mode
        self._prep = None
        self.eval()

    def train(self, mode: bool = True):
        """train(True) also turns the parameters' requires_grad on (they are created frozen for the inference path)."""
        super().train(mode)
        if mode:
            self.requires_grad_(True)
        return self

    def set_precision(self, precision):
        # fp32_split: f32 tensors, every Conv1d / Linear but the duration predictor's on split f16 hi/lo MFMA operands (hip.SplitWeight;
        # csrc/conv1d_split.h); fp32_bf16x3 (fp32_bf16x3_6p): the same convs on three exact bf16 terms per operand, seven (six) partial products
        # per product (hip.EmulWeight; csrc/conv1d_emul.h)
        if precision not in hip.PRECISIONS:
            raise ValueError(precision)
        if precision != self.precision:
            self.precision, self._prep = precision, None
        return self

    def load_state_dict(self, *a, **k):
        self._prep = None
        return super().load_state_dict(*a, **k)

    def _apply(self, fn, *a, **k):
        self._prep = None
        return super()._apply(fn, *a, **k)

    def _prepare(self):
        dev = self.encoder_proj.weight.device
        if dev.type != "cuda":
            raise hip._abi.JattsHipError("jatts_amd Matcha-TTS runs on the GPU only (no CPU fallback); call .to('cuda')")
        key = (self.precision, str(dev))
        if self._prep is not None and self._prep["key"] == key:
            return self._prep
        hip._abi.load()
        dt = hip.F16 if self.precision == "fp16" else hip.F32
        with hip.split_weights(self.precision):
            return self._prepare_packed(dev, key, dt)

    def _prepare_packed(self, dev, key, dt):
        sd = self.state_dict()
        f32 = lambda t: t.detach().float().to(dev).contiguous()  # noqa: E731
        P = {"key": key, "dtype": dt, "dev": dev}
        P["emb"] = f32(sd["encoder.embed.0.weight"])
        P["enc"] = ConformerRunner(sd, "encoder.", self.aheads, dt, dev)  # legacy rel-pos (matchatts_mas.py:196-218)
        with hip.split_weights(False):     # exact f32 always: durations are integers
            P["dur"] = _Predictor(sd, "duration_predictor.", hip.F32, dev)   # always f32 (integer durations)
        P["eproj"] = PackedConv(sd["encoder_proj.weight"], sd["encoder_proj.bias"], dt, dev)
        if self.spk_embed_dim is not None:
            P["proj"] = SpkProjection(sd, self.adim, self.spk_embed_integration_type, dt, dev)
        if self.spks is not None:
            P["sid_emb"] = f32(sd["sid_emb.weight"])
        e = "decoder.estimator."
        od, ch = self.odim, self.dec_channels
        P["t1"] = PackedConv(sd[e + "time_mlp.linear_1.weight"], sd[e + "time_mlp.linear_1.bias"], dt, dev)
        P["t2"] = PackedConv(sd[e + "time_mlp.linear_2.weight"], sd[e + "time_mlp.linear_2.bias"], dt, dev)
        tb = lambda p: [_TBlock(sd, p + f"1.{j}.", self.dec_heads, dt, dev) for j in range(self.n_blocks)]  # noqa: E731
        P["d0"] = (_Resnet(sd, e + "down_blocks.0.0.", [od, od], dt, dev), tb(e + "down_blocks.0."))
        wd = sd[e + "down_blocks.0.2.conv.weight"].detach().float()        # Conv1d(C, C, 3, stride 2, pad 1)
        C0 = wd.shape[0]
        w2 = torch.zeros(C0, 2 * C0, 2)
        w2[:, C0:, 0] = wd[:, :, 0]   # tap 0 reads paired row j-1: its second half is x[2j-1]
        w2[:, :C0, 1] = wd[:, :, 1]   # tap 1 reads paired row j: x[2j] | x[2j+1]
        w2[:, C0:, 1] = wd[:, :, 2]
        P["down"] = PackedConv(w2, sd[e + "down_blocks.0.2.conv.bias"], dt, dev)
        P["d1"] = (_Resnet(sd, e + "down_blocks.1.0.", [ch[0]], dt, dev), tb(e + "down_blocks.1."))
        P["d1c"] = PackedConv(sd[e + "down_blocks.1.2.weight"], sd[e + "down_blocks.1.2.bias"], dt, dev)
        P["mid"] = [(_Resnet(sd, e + f"mid_blocks.{i}.0.", [ch[1]], dt, dev), tb(e + f"mid_blocks.{i}."))
                    for i in range(self.n_mid)]
        P["u0"] = (_Resnet(sd, e + "up_blocks.0.0.", [ch[1], ch[1]], dt, dev), tb(e + "up_blocks.0."))
        wu, pad = hip.convtranspose_as_conv(sd[e + "up_blocks.0.2.conv.weight"].detach().float(), 2, 1)
        P["up"] = (PackedConv(wu, sd[e + "up_blocks.0.2.conv.bias"].detach().float().repeat(2), dt, dev), pad)
        P["u1"] = (_Resnet(sd, e + "up_blocks.1.0.", [ch[0], ch[0]], dt, dev), tb(e + "up_blocks.1."))
        P["u1c"] = PackedConv(sd[e + "up_blocks.1.2.weight"], sd[e + "up_blocks.1.2.bias"], dt, dev)
        P["fb"] = (PackedConv(sd[e + "final_block.block.0.weight"], sd[e + "final_block.block.0.bias"], dt, dev),
                   f32(sd[e + "final_block.block.1.weight"]), f32(sd[e + "final_block.block.1.bias"]))
        P["fp"] = PackedConv(sd[e + "final_proj.weight"], sd[e + "final_proj.bias"], dt, dev)
        self._prep = P
        return P

    # one U-Net evaluation; x_t / mu_t: operand dtype (R, ld) with odim valid columns; x32: f32 state (R, odim)
    def _estimator_step(self, P, rb, rb2, rbs, x32, mu_t, temb_rows, dt_step, tkey=None, valid=None):
        """One U-Net evaluation.  Inference (valid None): ragged batch, x32 += dt_step * v(x32) in place.  Training forward
        (valid = (int32 lengths at full rate, at half rate, key-bias rows at full rate, at half rate)): PADDED batch with the
        reference's mask multiplications (decoder.py:413-487); returns v(x) * mask as f32 and leaves x32 alone."""
        dt = P["dtype"]
        od = self.odim
        ld_in = P["d0"][0].conv1[0].c_in
        x_t = hip.affine_cast(x32, dt, ldy=ld_in)
        v1 = v2 = kb1 = kb2 = None
        if valid is not None:
            v1, v2, kb1, kb2 = valid
            hip.zero_pad_rows(rb, x_t, v1)            # pack([x, mu]) * mask
        # time MLP + the per-ResNet time projections: functions of the Euler step only (t is a scalar shared by the
        # batch, flow_matching.py:77-93) -> computed once per (n_timesteps, step, batch size) and reused
        tcache = P.setdefault("tcache", {}).setdefault(tkey, {}) if tkey is not None else None
        tm = tcache.get("tm") if tcache is not None else None
        if tm is None:
            t1 = hip.conv1d(rbs, temb_rows, P["t1"].w, P["t1"].c_in, P["t1"].n_out, 1, dtype=dt, bias=P["t1"].b, act=ACT_SWISH)
            tm = hip.conv1d(rbs, t1, P["t2"].w, P["t2"].c_in, P["t2"].n_out, 1, dtype=dt, bias=P["t2"].b, act=ACT_MISH)
            if tcache is not None:
                tcache["tm"] = tm

        def stage(blk, rbx, xs, vl, kb):
            res, tbs = blk
            h = res.run(rbx, rbs, xs, tm, dt, tv_cache=tcache, valid=vl)
            for t in tbs:
                t.run(rbx, h, dt, key_bias=kb)
            return h

        def masked(rbx, t, vl, len_mul=1):     # `x * mask` in front of a conv / a skip connection (no-op on ragged batches)
            return t if vl is None else hip.zero_pad_rows(rbx, t, vl, len_mul)

        tdt = hip.torch_dtype(dt)

        def operand(t):     # conv operand dtype.  f32 / split modes on a ragged batch: the f32 stream itself (no copy); the padded training-
            return t if (valid is None and t.dtype == tdt) else hip.affine_cast(t, dt)   # time forward masks its operands in place -> own buffer

        h0 = stage(P["d0"], rb, [x_t, mu_t], v1, kb1)            # (R, C0) f32; skip connection 0
        h0_t = masked(rb, operand(h0), v1)
        C0 = h0.shape[1]
        dn = P["down"]
        h = hip.conv1d(rb2, h0_t.view(-1, 2 * C0), dn.w, dn.c_in, C0, 2, dtype=dt, bias=dn.b, pad=1)   # (R/2, C0)
        h1 = stage(P["d1"], rb2, [masked(rb2, h, v2)], v2, kb2)  # skip connection 1
        h1_t = masked(rb2, operand(h1), v2)
        c = P["d1c"]
        h = hip.conv1d(rb2, h1_t, c.w, c.c_in, c.n_out, 3, dtype=dt, bias=c.b)
        for blk in P["mid"]:
            h = operand(stage(blk, rb2, [masked(rb2, h, v2)], v2, kb2))
        h = stage(P["u0"], rb2, [masked(rb2, h, v2), h1_t], v2, kb2)
        up, pad = P["up"]
        C1 = h.shape[1]
        h = hip.conv1d(rb2, masked(rb2, operand(h), v2), up.w, up.c_in, 2 * C1, up.k, dtype=dt, bias=up.b, pad=pad)
        h = h.view(-1, C1)                                        # (R, C1): polyphase rows are already interleaved
        h = stage(P["u1"], rb, [masked(rb, h, v1), h0_t], v1, kb1)
        c = P["u1c"]
        h = hip.conv1d(rb, masked(rb, operand(h), v1), c.w, c.c_in, c.n_out, 3, dtype=dt, bias=c.b)
        fbc, g, b = P["fb"]
        h = hip.conv1d(rb, masked(rb, h, v1), fbc.w, fbc.c_in, fbc.n_out, 3, dtype=dt, bias=fbc.b)
        h = masked(rb, hip.groupnorm_mish(rb, h, fbc.n_out, 8, g, b, dt, GN_EPS), v1)
        fp = P["fp"]
        if valid is not None:   # the vector field itself, masked (decoder.py:485-487)
            return masked(rb, hip.conv1d(rb, h, fp.w, fp.c_in, od, 1, dtype=dt, bias=fp.b, out_f32=True), v1)
        # Euler update fused into the projection epilogue: x += dt * (W h + b)   (flow_matching.py:86-88)
        hip.conv1d(rb, h, fp.w, fp.c_in, od, 1, dtype=dt, bias=fp.b, alpha=dt_step, resid=x32, out=x32, out_f32=True)

    @torch.no_grad()
    def inference_batch(self, texts, n_timesteps: int = 10, temperature: float = 0.667, spembs=None, sids=None,
                        noise=None, durations=None, taps=None):
        P = self._prepare()
        dt, dev, A, od = P["dtype"], P["dev"], self.adim, self.odim
        B = len(texts)
        lens = [int(t.numel()) for t in texts]
        rb = hip.RaggedBatch(lens, dev)
        ids = torch.cat([t.reshape(-1) for t in texts]).to(device=dev, dtype=torch.int64)
        n_bad = torch.zeros(1, dtype=torch.int64, device=dev)   # out-of-range ids: counted by the kernel, raised at the LR host sync
        hs = P["enc"].run(rb, hip.embed_scale(ids, P["emb"], math.sqrt(A), n_bad))
        rbs = hip.RaggedBatch([1] * B, dev)
        if self.spks is not None:
            hip.add_seq_vector(rb, hs, P["sid_emb"][sids.to(dev).view(-1).long()].contiguous())
        if self.spk_embed_dim is not None:
            hs = P["proj"](rb, hs, spembs)
        if taps is not None:
            taps["hs"] = hs.clone()                                 # text encoding (+ speaker): what the alignment module scores
        logd, d_pred = hip.predictor_head(P["dur"].trunk(rb, hs), P["dur"].w, P["dur"].b,
                                          want_duration=True)
        d_used = d_pred
        if durations is not None:
            d_used = torch.cat([d.reshape(-1) for d in durations]).to(device=dev, dtype=torch.int64).contiguous()
        d_eff, cum, olens = hip.lr_sizes(rb, d_used, check=n_bad)                # host sync: output sizes (all-zero utterances -> all ones)
        olens = [n - n % 2 for n in olens]                         # matchatts_mas.py:521-526: even lengths
        if min(olens) <= 0:
            raise RuntimeError("an utterance has fewer than 2 output frames")
        rbo = hip.RaggedBatch(olens, dev)
        if self._MAS:   # GaussianUpsampling (length_regulator.py:111-154); frame f only depends on f and d
            up = hip.gaussian_upsample(rb, d_eff, rbo, hs)
        else:           # MatchaTTS (tts1): hard LengthRegulator (matchatts.py:427)
            up = hip.lr_gather(rb, cum, rbo, hs)
        ep = P["eproj"]
        mu = hip.conv1d(rbo, hip.affine_cast(up, dt), ep.w, ep.c_in, od, 1, dtype=dt, bias=ep.b, out_f32=True)
        if taps is not None:
            taps["mu"] = mu.clone()
        if noise is None:
            nz = torch.randn(rbo.total, od, device=dev)
        else:
            nz = torch.cat([n[:m].reshape(-1, od) for n, m in zip(noise, olens)]).to(dev).float().contiguous()
        x = hip.affine_cast(nz, hip.F32, scale=torch.full((od,), float(temperature), device=dev))
        mu_t = hip.affine_cast(mu, dt, ldy=P["d0"][0].conv1[1].c_in)
        rb2 = hip.RaggedBatch([n // 2 for n in olens], dev)
        # Euler schedule exactly as flow_matching.py:68-93 (float32 accumulation of t)
        t_span = torch.linspace(0, 1, n_timesteps + 1)
        t, dt_s = t_span[0], t_span[1] - t_span[0]
        half = od  # SinusoidalPosEmb(in_channels = 2*odim): half_dim = odim
        freq = torch.exp(torch.arange(half).float() * -(math.log(10000) / (half - 1)))
        c_t = P["t1"].c_in
        # the whole schedule's time embeddings in ONE asynchronous upload before the loop (a per-step .to(dev) from pageable memory
        # drains the launch queue ten times per batch)
        sched, tembs = [], torch.zeros(n_timesteps, B, c_t)
        for step in range(1, n_timesteps + 1):
            emb = 1000.0 * t * freq
            tembs[step - 1, :, :2 * half] = torch.cat((emb.sin(), emb.cos()))
            sched.append(float(dt_s))
            t = t + dt_s
            if step < n_timesteps:
                dt_s = t_span[step + 1] - t
        tembs = tembs.pin_memory().to(dev, non_blocking=True).to(hip.torch_dtype(dt))
        for step in range(1, n_timesteps + 1):
            self._estimator_step(P, rbo, rb2, rbs, x, mu_t, tembs[step - 1], sched[step - 1], tkey=(int(n_timesteps), step, B))
        return dict(feat_gen=x, olens=olens, feats_rb=rbo, text_rb=rb, duration=d_pred, log_duration=logd)

    def inference(self, text, feats=None, durations=None, spembs=None, sids=None, lids=None, n_timesteps: int = 10,
                  temperature: float = 0.667, use_teacher_forcing: bool = False, noise=None):
        """Same contract as jatts.models.MatchaTTS_MAS.inference (matchatts_mas.py:552-642) for feats=None."""
        if use_teacher_forcing:
            raise NotImplementedError("use_teacher_forcing is broken in the reference itself (matchatts_mas.py:597-612 reads undefined p, e)")
        taps = {} if feats is not None else None
        r = self.inference_batch([text], n_timesteps=n_timesteps, temperature=temperature,
                                 spembs=None if spembs is None else spembs.unsqueeze(0), sids=sids,
                                 noise=None if noise is None else [noise], taps=taps)
        out = dict(feat_gen=r["feat_gen"], duration=r["duration"])
        if self._MAS:
            out.update(log_p_attn=None, ds=None)
            if feats is not None:   # alignment of the given features against the text encoding (matchatts_mas.py:449-455); B = 1: no padding
                from ..alignments import pack_alignment_convs, padded_alignment
                P = self._prepare()
                if "align" not in P:
                    P["align"] = pack_alignment_convs(self.state_dict(), P["dev"])
                lp, ds, _ = padded_alignment(P["align"], taps["hs"], feats.to(P["dev"]).float().unsqueeze(0).contiguous(),
                                             [int(text.numel())], [int(feats.shape[0])], self.adim)
                out.update(log_p_attn=lp[0], ds=ds[0])
        return out

    def forward(self, text, text_lengths, feats, feats_lengths, durations=None, durations_lengths=None, spembs=None, sids=None,
                lids=None, joint_training=False, cfm_t=None, cfm_noise=None):
        """The reference's training-time call, forward only (matchatts_mas.py:337-550 with is_inference=False): padded batch ->
        alignment module + monotonic alignment search -> duration predictor -> masked Gaussian upsampling -> encoder_proj ->
        conditional-flow-matching loss.  Same arguments and return dict {d_outs, ys, hs, olens_in, bin_loss, log_p_attn, ds,
        cfm_loss}.  ``cfm_t`` (B,) / ``cfm_noise`` (B, T, odim): the two random draws of CFM.compute_loss
        (flow_matching.py:115-117), injectable for parity; drawn with torch.rand / randn when omitted.
        Padded-batch arithmetic as in the reference: key masks in the encoder attention, mask multiplications inside the
        U-Net (GroupNorm statistics run over the padded length), the attention mask of the U-Net's transformer blocks added to
        the scores.  In train() mode with gradients enabled the tts1 MatchaTTS returns the differentiable HIP forward of
        models/matchatts_train.py instead (criterion / backward: jatts_amd.training.MatchaTTSTrainer)."""
        if self.training and torch.is_grad_enabled():
            from .matchatts_train import train_forward
            self._train_calls += 1
            self._prep = None
            return train_forward(self, text, text_lengths, feats, feats_lengths, durations, durations_lengths, spembs=spembs, sids=sids,
                                 cfm_t=cfm_t, cfm_noise=cfm_noise, seed=self._train_calls)
        with torch.no_grad():
            return self._forward_eval(text, text_lengths, feats, feats_lengths, durations, durations_lengths, spembs, sids, lids,
                                      joint_training, cfm_t, cfm_noise)

    def _forward_eval(self, text, text_lengths, feats, feats_lengths, durations=None, durations_lengths=None, spembs=None, sids=None,
                      lids=None, joint_training=False, cfm_t=None, cfm_noise=None):
        P = self._prepare()
        dt, dev, A, od = P["dtype"], P["dev"], self.adim, self.odim
        ilens = [int(v) for v in text_lengths.tolist()]
        olens = [int(v) for v in feats_lengths.tolist()]
        B, Tm, To = len(ilens), max(ilens), max(olens)
        xs = text[:, :Tm].to(dev)
        ys = feats[:, :To].to(dev).float().contiguous()
        rbt = hip.RaggedBatch([Tm] * B, dev)                       # padded text geometry
        kv = hip.h2d(ilens, torch.int32, dev)
        rbs = hip.RaggedBatch([1] * B, dev)
        hs = P["enc"].run(rbt, hip.embed_scale(xs.reshape(-1).to(torch.int64).contiguous(), P["emb"], math.sqrt(A)), kv_len=kv)
        if self.spks is not None:
            hip.add_seq_vector(rbt, hs, P["sid_emb"][sids.to(dev).view(-1).long()].contiguous())
        if self.spk_embed_dim is not None:
            hs = P["proj"](rbt, hs, spembs)
        olens_in = [n - n % 2 for n in olens]
        Te = max(olens_in)
        if self._MAS:
            # ---- alignment module on the PADDED batch + batched monotonic alignment search (jatts_amd.alignments.padded_alignment)
            from ..alignments import pack_alignment_convs, padded_alignment
            if "align" not in P:
                P["align"] = pack_alignment_convs(self.state_dict(), dev)
            log_p_attn, ds, bin_loss = padded_alignment(P["align"], hs, ys, ilens, olens, A)
            rbv = hip.RaggedBatch(ilens, dev)                          # valid tokens, packed (row selection: plumbing)
            sel = torch.cat([torch.arange(b * Tm, b * Tm + ilens[b], device=dev) for b in range(B)])
            # ---- duration predictor (log domain, masked) and masked Gaussian upsampling
            d_outs = hip.zero_pad_rows(rbt, hip.predictor_head(P["dur"].trunk(rbt, hs), P["dur"].w, P["dur"].b), kv).view(B, Tm)
            d_int = torch.cat([ds[b, : ilens[b]] for b in range(B)]).to(torch.int64).contiguous()
            rbo = hip.RaggedBatch(olens, dev)
            up = hip.gaussian_upsample(rbv, d_int, rbo, hs.index_select(0, sel).contiguous())    # valid frames of every utterance
            # h_masks zero the frame index of padded frames (length_regulator.py:139-141): they all equal frame 0 of their utterance
            cu = rbo.cu_host
            rows = hip.h2d([cu[b] + (t if t < olens[b] else 0) for b in range(B) for t in range(Te)], torch.int64, dev)
            up_pad = up.index_select(0, rows).contiguous()
        else:   # tts1: ground-truth durations, hard length regulator, zero padding (length_regulator.py:70-97)
            if durations is None:
                raise ValueError("MatchaTTS.forward needs durations")
            d_outs = hip.zero_pad_rows(rbt, hip.predictor_head(P["dur"].trunk(rbt, hs), P["dur"].w, P["dur"].b), kv).view(B, Tm)
            d_flat = durations[:, : int(durations_lengths.max())].to(dev).reshape(-1).to(torch.int64).contiguous()
            if d_flat.numel() != B * Tm:
                raise ValueError("durations must be padded to the text length")
            _, cum, _, _ = hip.lr_durations(rbt, d_flat, 1.0, zero_rule=0)
            up_pad = hip.lr_gather(rbt, cum, hip.RaggedBatch([Te] * B, dev), hs)
            ds = log_p_attn = bin_loss = None
        ep = P["eproj"]
        rbe = hip.RaggedBatch([Te] * B, dev)
        mu = hip.conv1d(rbe, hip.affine_cast(up_pad, dt), ep.w, ep.c_in, od, 1, dtype=dt, bias=ep.b,
                        out_f32=True)                              # hs of the return dict: (B, Te, odim)
        ys_e = ys[:, :Te].contiguous()
        # ---- CFM loss (flow_matching.py:99-127)
        t = (torch.rand(B) if cfm_t is None else cfm_t.reshape(B).float().cpu())
        z = (torch.randn(B, Te, od) if cfm_noise is None else cfm_noise[:, :Te].float()).to(dev).reshape(B * Te, od).contiguous()
        y, u = hip.cfm_mix(rbe, ys_e.view(B * Te, od), z, t.to(dev).contiguous(), self.sigma_min)
        v1 = hip.h2d(olens_in, torch.int32, dev)
        v2 = hip.h2d([n // 2 for n in olens_in], torch.int32, dev)
        heads, scale = self.dec_heads, P["d0"][1][0].dh ** -0.5 if P["d0"][1] else 1.0

        def key_bias(T, vl):   # the (B, T) 1/0 frame mask as an additive score bias (after scaling): ku = mask / scale
            m = (torch.arange(T, device=dev).unsqueeze(0) < vl.unsqueeze(1)).float().reshape(-1, 1) / scale
            return m.expand(-1, heads).contiguous()
        half = od
        freq = torch.exp(torch.arange(half).float() * -(math.log(10000) / (half - 1)))
        emb = 1000.0 * t.unsqueeze(1) * freq.unsqueeze(0)
        temb = torch.zeros(B, P["t1"].c_in)
        temb[:, :2 * half] = torch.cat((emb.sin(), emb.cos()), dim=-1)
        mu_t = hip.zero_pad_rows(rbe, hip.affine_cast(mu, dt, ldy=P["d0"][0].conv1[1].c_in), v1)
        rb2 = hip.RaggedBatch([Te // 2] * B, dev)
        pred = self._estimator_step(P, rbe, rb2, rbs, y, mu_t, temb.to(dev).to(hip.torch_dtype(dt)), 0.0,
                                    valid=(v1, v2, key_bias(Te, v1), key_bias(Te // 2, v2)))
        # F.mse_loss(pred, u, reduction="sum") / (sum(mask) * n_feats): the sum runs over the padded frames too (pred is 0 there)
        cfm_loss = hip.sq_err_sum(pred, u, 1.0 / (float(sum(olens_in)) * od))
        ret = {"d_outs": d_outs, "ys": ys_e, "hs": mu.view(B, Te, od), "olens_in": torch.tensor(olens_in), "cfm_loss": cfm_loss}
        if self._MAS:
            ret.update(bin_loss=bin_loss, log_p_attn=log_p_attn, ds=ds)
        return ret


class MatchaTTS_MAS(_MatchaBase):
    _MAS = True


class MatchaTTS(_MatchaBase):
    """tts1 variant (reference models/matchatts.py): external-duration training, hard LengthRegulator."""
    _MAS = False
