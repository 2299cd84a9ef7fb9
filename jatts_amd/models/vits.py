"""mel-VITS on the MI355X HIP path — drop-in for ``jatts.models.VITS`` at stage 4.

Same constructor kwargs (reference models/vits.py:47-127), same state_dict schema (incl. the
weight-norm ``weight_g/weight_v`` pairs of the flow / posterior encoder and the training-only
posterior encoder + alignment module, so checkpoints load strictly), same ``inference()`` contract
(:581-679).  The reference's single-speaker stage 4 is broken (`spembs=None` crashes at vits.py:485,
SURVEY §3.3): speaker embeddings are required here too.

``inference_batch`` processes a packed ragged batch; ``noise`` (list of (T_feats, adim) standard
normal tensors) can be injected for parity — the reference draws torch.randn_like internally (:479).
"""
import logging
import math
from typing import Optional, Sequence

import torch

from .. import hip
from . import _schema as S
from ._conformer import LN_EPS, ConformerRunner, PackedConv, SpkProjection
from .fastspeech2 import _Predictor


def _fold_wn(sd, stem):
    if stem + ".weight" in sd:
        return sd[stem + ".weight"].detach().float()
    g, v = sd[stem + ".weight_g"].detach().float(), sd[stem + ".weight_v"].detach().float()
    norm = v.reshape(v.shape[0], -1).norm(dim=1).reshape(-1, *([1] * (v.dim() - 1)))
    return g * v / norm


def _wavenet_spec(spec, prefix, layers, ch, k, glob, wn=True):
    """wavenet.py:58-96 + residual_block.py:64-110 (gate = 2*ch, skip = ch, no aux)."""
    def conv(name, o, i, kk, bias):
        if bias:
            spec[name + ".bias"] = ((o,), "param")
        if wn:
            spec[name + ".weight_g"] = ((o, 1, 1), "param")
            spec[name + ".weight_v"] = ((o, i, kk), "param")
        else:
            spec[name + ".weight"] = ((o, i, kk), "param")
    for layer in range(layers):
        q = f"{prefix}conv_layers.{layer}."
        conv(q + "conv", 2 * ch, ch, k, True)
        if glob:
            conv(q + "conv1x1_glo", 2 * ch, glob, 1, False)
        conv(q + "conv1x1_out", 2 * ch, ch, 1, True)


class VITS(torch.nn.Module):
    def __init__(
        self, idim: int, odim: int, adim: int = 384, aheads: int = 4, reduction_factor: int = 1,
        text_encoder_attention_heads: int = 2, text_encoder_ffn_expand: int = 4, text_encoder_blocks: int = 6,
        text_encoder_positionwise_layer_type: str = "conv1d", text_encoder_positionwise_conv_kernel_size: int = 1,
        text_encoder_positional_encoding_layer_type: str = "rel_pos",
        text_encoder_self_attention_layer_type: str = "rel_selfattn", text_encoder_activation_type: str = "swish",
        text_encoder_normalize_before: bool = True, text_encoder_dropout_rate: float = 0.1,
        text_encoder_positional_dropout_rate: float = 0.0, text_encoder_attention_dropout_rate: float = 0.0,
        text_encoder_conformer_kernel_size: int = 7, use_macaron_style_in_text_encoder: bool = True,
        use_conformer_conv_in_text_encoder: bool = True, dlayers: int = 6, dunits: int = 1536,
        decoder_positionwise_layer_type: str = "conv1d", decoder_positionwise_conv_kernel_size: int = 1,
        decoder_normalize_before: bool = True, decoder_concat_after: bool = False,
        transformer_dec_dropout_rate: float = 0.1, transformer_dec_positional_dropout_rate: float = 0.1,
        transformer_dec_attn_dropout_rate: float = 0.1, conformer_rel_pos_type: str = "legacy",
        conformer_pos_enc_layer_type: str = "rel_pos", conformer_self_attn_layer_type: str = "rel_selfattn",
        conformer_activation_type: str = "swish", use_macaron_style_in_conformer: bool = True,
        use_cnn_in_conformer: bool = True, conformer_dec_kernel_size: int = 31,
        duration_predictor_type: str = "deterministic", duration_predictor_layers: int = 2,
        duration_predictor_chans: int = 384, duration_predictor_kernel_size: int = 3,
        duration_predictor_dropout_rate: float = 0.1, posterior_encoder_kernel_size: int = 5,
        posterior_encoder_layers: int = 16, posterior_encoder_stacks: int = 1,
        posterior_encoder_base_dilation: int = 1, posterior_encoder_dropout_rate: float = 0.0,
        use_weight_norm_in_posterior_encoder: bool = True, flow_flows: int = 4, flow_kernel_size: int = 5,
        flow_base_dilation: int = 1, flow_layers: int = 4, flow_dropout_rate: float = 0.0,
        use_weight_norm_in_flow: bool = True, use_only_mean_in_flow: bool = True, spks: Optional[int] = None,
        spk_embed_dim: Optional[int] = None, spk_embed_integration_type: str = "add", use_gst: bool = False,
        gst_tokens: int = 10, gst_heads: int = 4, gst_conv_layers: int = 6,
        gst_conv_chans_list: Sequence[int] = (32, 32, 64, 64, 128, 128), gst_conv_kernel_size: int = 3,
        gst_conv_stride: int = 2, gst_gru_layers: int = 1, gst_gru_units: int = 128,
        init_type: str = "xavier_uniform", init_enc_alpha: float = 1.0, use_masking: bool = False,
        use_weighted_masking: bool = False,
    ):
        super().__init__()
        self.idim, self.odim, self.adim = idim, odim, adim
        self.aheads, self.te_heads = aheads, text_encoder_attention_heads
        self.eos = idim - 1
        if duration_predictor_type != "deterministic":
            raise NotImplementedError("stochastic duration predictor is dead code in the reference (vits.py:290)")
        if use_gst or reduction_factor != 1:
            raise NotImplementedError("GST / reduction_factor > 1 are outside the stage-4 hot path")
        if not use_only_mean_in_flow or flow_base_dilation != 1:
            raise NotImplementedError("only use_only_mean_in_flow=True, flow_base_dilation=1 (reference defaults)")
        if not spk_embed_dim or spk_embed_dim <= 0:
            raise NotImplementedError("the reference's VITS stage 4 needs speaker embeddings (vits.py:485)")
        if spk_embed_integration_type not in ("add", "concat"):
            raise NotImplementedError("support only add or concat.")
        self.spk_embed_integration_type = spk_embed_integration_type
        if text_encoder_positional_encoding_layer_type != "rel_pos" or conformer_pos_enc_layer_type != "rel_pos":
            raise NotImplementedError("only rel_pos / rel_selfattn conformers")
        self.spk_embed_dim = spk_embed_dim
        self.flow_flows, self.flow_layers, self.flow_kernel_size = flow_flows, flow_layers, flow_kernel_size
        spec = S.new_spec()
        spec["text_encoder.emb.weight"] = ((idim, adim), "param")
        S.conformer_spec(spec, "text_encoder.encoder.", adim, text_encoder_attention_heads,
                         adim * text_encoder_ffn_expand, text_encoder_blocks, text_encoder_positionwise_layer_type,
                         text_encoder_positionwise_conv_kernel_size, use_macaron_style_in_text_encoder,
                         use_conformer_conv_in_text_encoder, text_encoder_conformer_kernel_size,
                         attn_type="rel_selfattn", normalize_before=text_encoder_normalize_before)
        S._conv(spec, "text_encoder.proj", 2 * adim, adim, 1)
        if spks is not None and spks > 1:
            raise NotImplementedError("sid embeddings are not used by the reference VITS forward")
        S._lin(spec, "projection", adim, spk_embed_dim + (adim if spk_embed_integration_type == "concat" else 0))
        # posterior encoder (training / reconstruction only; kept for strict checkpoint loading)
        S._conv(spec, "posterior_encoder.input_conv", adim, odim, 1)
        _wavenet_spec(spec, "posterior_encoder.encoder.", posterior_encoder_layers, adim, posterior_encoder_kernel_size,
                      spk_embed_dim, use_weight_norm_in_posterior_encoder)
        S._conv(spec, "posterior_encoder.proj", 2 * adim, adim, 1)
        for i in range(flow_flows):
            q = f"flow.flows.{2 * i}."
            S._conv(spec, q + "input_conv", adim, adim // 2, 1)
            _wavenet_spec(spec, q + "encoder.", flow_layers, adim, flow_kernel_size, spk_embed_dim, use_weight_norm_in_flow)
            S._conv(spec, q + "proj", adim // 2, adim, 1)
        S.predictor_spec(spec, "duration_predictor.", adim, duration_predictor_layers, duration_predictor_chans,
                         duration_predictor_kernel_size)
        for nm, o, i, k in (("t_conv1", adim, adim, 3), ("t_conv2", adim, adim, 1), ("f_conv1", adim, odim, 3),
                            ("f_conv2", adim, adim, 3), ("f_conv3", adim, adim, 1)):
            S._conv(spec, "alignment_module." + nm, o, i, k)  # MAS aligner: training only
        S.conformer_spec(spec, "decoder.", adim, aheads, dunits, dlayers, decoder_positionwise_layer_type,
                         decoder_positionwise_conv_kernel_size, use_macaron_style_in_conformer, use_cnn_in_conformer,
                         conformer_dec_kernel_size, attn_type="rel_selfattn", normalize_before=decoder_normalize_before)
        S._lin(spec, "feat_out", odim, adim)
        S.build_from_spec(self, spec)
        # train-mode behaviour (models/vits_train.py): the reference's dropout sites
        self.dropout_rates = dict(te=text_encoder_dropout_rate, te_pos=text_encoder_positional_dropout_rate,
                                  te_attn=text_encoder_attention_dropout_rate, dec=transformer_dec_dropout_rate,
                                  dec_pos=transformer_dec_positional_dropout_rate, dec_attn=transformer_dec_attn_dropout_rate,
                                  dur=duration_predictor_dropout_rate, post=posterior_encoder_dropout_rate, flow=flow_dropout_rate)
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
        dev = self.feat_out.weight.device
        if dev.type != "cuda":
            raise hip._abi.JattsHipError("jatts_amd.VITS runs on the GPU only (no CPU fallback); call .to('cuda')")
        key = (self.precision, str(dev))
        if self._prep is not None and self._prep["key"] == key:
            return self._prep
        hip._abi.load()
        dt = hip.F16 if self.precision == "fp16" else hip.F32
        with hip.split_weights(self.precision):
            return self._prepare_packed(dev, key, dt)

    def _prepare_packed(self, dev, key, dt):
        sd = self.state_dict()
        A = self.adim
        f32 = lambda t: t.detach().float().to(dev).contiguous()  # noqa: E731
        P = {"key": key, "dtype": dt, "dev": dev}
        P["emb"] = f32(sd["text_encoder.emb.weight"])
        # vits.py:203-221,311-331: both conformers get "rel_pos"/"rel_selfattn" WITHOUT FastSpeech2's legacy fallback
        P["tenc"] = ConformerRunner(sd, "text_encoder.encoder.", self.te_heads, dt, dev, rel_style="new")
        P["dec"] = ConformerRunner(sd, "decoder.", self.aheads, dt, dev, rel_style="new")
        P["te_proj"] = PackedConv(sd["text_encoder.proj.weight"], sd["text_encoder.proj.bias"], dt, dev)
        P["proj"] = SpkProjection(sd, self.adim, self.spk_embed_integration_type, dt, dev)
        with hip.split_weights(False):     # exact f32 always: durations are integers
            P["dur"] = _Predictor(sd, "duration_predictor.", hip.F32, dev)   # always f32 (integer durations)
        flows = []
        for i in range(self.flow_flows):
            q = f"flow.flows.{2 * i}."
            layers = self._wn_layers(sd, q + "encoder.", self.flow_layers, dt, dev)
            flows.append(dict(inp=PackedConv(sd[q + "input_conv.weight"], sd[q + "input_conv.bias"], dt, dev),
                              proj=PackedConv(sd[q + "proj.weight"], sd[q + "proj.bias"], dt, dev), layers=layers))
        P["flows"] = flows
        P["feat_out"] = PackedConv(sd["feat_out.weight"], sd["feat_out.bias"], dt, dev)
        P["sqrtA"] = torch.full((A,), math.sqrt(A), dtype=torch.float32, device=dev)
        self._prep = P
        return P

    def _wavenet(self, P, rbo, rbs, h, layers, sp_t):
        """WaveNet residual stack (wavenet.py:115-153, residual_block.py:112-167) on a ragged batch: h f32 (rows, A) is the
        residual stream (updated in place), returns the skip sum f32.  The reference's `* x_mask` after every conv1x1_out is
        the ragged layout itself (rows past an utterance's end do not exist)."""
        dt, A = P["dtype"], self.adim
        skips = torch.zeros_like(h)
        for ly in layers:
            g = hip.conv1d(rbs, sp_t, ly["glo"].w, ly["glo"].c_in, 2 * A, 1, dtype=dt, out_f32=True)  # (B, 2A)
            cv = ly["conv"]
            y = hip.conv1d(rbo, hip.affine_cast(h, dt), cv.w, cv.c_in, 2 * A, cv.k, dtype=dt, bias=cv.b)
            gate = hip.gated_tanh_sigmoid(rbo, y, g, A, dt)
            hip.conv1d(rbo, gate, ly["skip"].w, ly["skip"].c_in, A, 1, dtype=dt, bias=ly["skip"].b, resid=skips,
                       out=skips, out_f32=True)
            hip.conv1d(rbo, gate, ly["res"].w, ly["res"].c_in, A, 1, dtype=dt, bias=ly["res"].b, resid=h, out=h,
                       out_f32=True)
        return skips

    def _post(self, P):
        """Posterior encoder + alignment convolutions: prepared on first use (training-time forward(), inference(feats=...))."""
        post = P.get("post")
        if post is None:
            dt, dev, sd = P["dtype"], P["dev"], self.state_dict()
            from ..alignments import pack_alignment_convs
            n_layers = sum(1 for k in sd if k.startswith("posterior_encoder.encoder.") and k.endswith("conv.bias"))
            post = P["post"] = dict(
                inp=PackedConv(sd["posterior_encoder.input_conv.weight"], sd["posterior_encoder.input_conv.bias"], dt, dev),
                layers=self._wn_layers(sd, "posterior_encoder.encoder.", n_layers, dt, dev),
                proj=PackedConv(sd["posterior_encoder.proj.weight"], sd["posterior_encoder.proj.bias"], dt, dev))
            P["align"] = pack_alignment_convs(sd, dev)
        return post

    def _posterior(self, P, rbo, rbs, yv, sp_t, noise_rows):
        """PosteriorEncoder.forward on ragged valid frames (posterior_encoder.py:96-130): -> (z, stats_q = m_q | logs_q)."""
        dt, A, post = P["dtype"], self.adim, self._post(P)
        q = post["inp"]
        h = hip.conv1d(rbo, hip.affine_cast(yv, dt, ldy=q.c_in), q.w, q.c_in, A, 1, dtype=dt, bias=q.b, out_f32=True)
        sk = self._wavenet(P, rbo, rbs, h, post["layers"], sp_t)
        scale_q = torch.full((A,), math.sqrt(1.0 / len(post["layers"])), device=P["dev"])
        pp = post["proj"]
        stats_q = hip.conv1d(rbo, hip.affine_cast(sk, dt, scale=scale_q), pp.w, pp.c_in, 2 * A, 1, dtype=dt, bias=pp.b, out_f32=True)
        return hip.gaussian_sample(stats_q, noise_rows, 1.0), stats_q                    # z = m_q + eps * exp(logs_q)

    def _wn_layers(self, sd, prefix, n_layers, dt, dev):
        A, layers = self.adim, []
        for l in range(n_layers):
            c = prefix + f"conv_layers.{l}."
            wo = _fold_wn(sd, c + "conv1x1_out")           # (2A, A, 1): [residual | skip] rows
            bo = sd[c + "conv1x1_out.bias"].detach().float()
            layers.append(dict(
                conv=PackedConv(_fold_wn(sd, c + "conv"), sd[c + "conv.bias"], dt, dev),
                glo=PackedConv(_fold_wn(sd, c + "conv1x1_glo"), None, dt, dev),
                res=PackedConv(wo[:A], bo[:A], dt, dev), skip=PackedConv(wo[A:], bo[A:], dt, dev)))
        return layers

    @torch.no_grad()
    def inference_batch(self, texts, spembs, noise=None, noise_scale: float = 0.667, durations=None, taps=None):
        """texts: list of LongTensor; spembs: (B, spk_embed_dim); noise: optional list of (T_feats_b, adim)."""
        P = self._prepare()
        dt, dev, A = P["dtype"], P["dev"], self.adim
        B = len(texts)
        lens = [int(t.numel()) for t in texts]
        rb = hip.RaggedBatch(lens, dev)
        ids = torch.cat([t.reshape(-1) for t in texts]).to(device=dev, dtype=torch.int64)
        n_bad = torch.zeros(1, dtype=torch.int64, device=dev)   # out-of-range ids: counted by the kernel, raised at the LR host sync
        # TextEncoder: emb * sqrt(A) (text_encoder.py:123), then RelPositionalEncoding scales by sqrt(A) again
        x = hip.embed_scale(ids, P["emb"], float(A), n_bad)
        hs = P["tenc"].run(rb, x)                                          # f32 (R, A)
        if taps is not None:
            taps["text_encoder_out"] = hs.clone()
        hs_t = hip.affine_cast(hs, dt)
        tp = P["te_proj"]
        stats = hip.conv1d(rb, hs_t, tp.w, tp.c_in, 2 * A, 1, dtype=dt, bias=tp.b, out_f32=True)   # (R, 2A): m_p | logs_p
        # speaker embedding: hs += projection(normalize(spembs))  (vits.py:441-443, :706-712)
        spembs = spembs.to(dev).float().reshape(B, -1).contiguous()
        rbs = hip.RaggedBatch([1] * B, dev)
        hs = P["proj"](rb, hs, spembs)
        if taps is not None:
            taps["hs"] = hs.clone()                                        # text encoding + speaker: what the alignment module scores
        logd, d_pred = hip.predictor_head(P["dur"].trunk(rb, hs), P["dur"].w, P["dur"].b,
                                          want_duration=True)
        d_used = d_pred
        if durations is not None:
            d_used = torch.cat([d.reshape(-1) for d in durations]).to(device=dev, dtype=torch.int64).contiguous()
        d_used, _, olens = hip.lr_sizes(rb, d_used, check=n_bad)                        # per-utterance frame counts (host sync)
        if min(olens) <= 0:
            raise RuntimeError("an utterance has zero output frames (all durations 0)")
        rbo = hip.RaggedBatch(olens, dev)
        # Gaussian upsampling of m_p and logs_p together (length_regulator.py:111-154)
        up = hip.gaussian_upsample(rb, d_used, rbo, stats)                 # (Rf, 2A)
        if noise is None:
            nz = torch.randn(rbo.total, A, device=dev)
        else:
            nz = torch.cat([n.reshape(-1, A) for n in noise]).to(dev).float().contiguous()
        z = hip.gaussian_sample(up, nz, noise_scale)   # z_p = m_p + eps * exp(logs_p) * noise_scale (vits.py:478-480)
        if taps is not None:
            taps["z_p"] = z.clone()
        # global conditioning vectors g = conv1x1_glo(spembs) per flow layer (residual_block.py:150-154)
        sp_t = hip.affine_cast(spembs, dt, ldy=P["flows"][0]["layers"][0]["glo"].c_in)
        half = A // 2
        for fl in reversed(P["flows"]):
            z = hip.flip_channels(z)                                      # FlipFlow
            xa_t = hip.affine_cast(z, dt)                                  # coupling reads the first half only
            fi = fl["inp"]
            h = hip.conv1d(rbo, xa_t, fi.w, fi.c_in, A, 1, dtype=dt, bias=fi.b, out_f32=True, ldx=A)  # (Rf, A) f32
            skips = self._wavenet(P, rbo, rbs, h, fl["layers"], sp_t)
            # m = proj(skips * sqrt(1/layers));  xb <- xb - m   (use_only_mean: logs = 0)
            sk_t = hip.affine_cast(skips, dt, scale=P.setdefault(
                "skip_scale", torch.full((A,), math.sqrt(1.0 / len(fl["layers"])), device=dev)))
            pr = fl["proj"]
            hip.conv1d(rbo, sk_t, pr.w, pr.c_in, half, 1, dtype=dt, bias=pr.b, alpha=-1.0, resid=z, resid_col0=half,
                       out=z, out_ld=A, out_col0=half, out_f32=True)
        if taps is not None:
            taps["z"] = z.clone()
        zs_t = P["dec"].run(rbo, hip.affine_cast(z, hip.F32, scale=P["sqrtA"]), final_dtype=dt)
        fo = P["feat_out"]
        out = hip.conv1d(rbo, zs_t, fo.w, fo.c_in, fo.n_out, 1, dtype=dt, bias=fo.b, out_f32=True)
        return dict(feat_gen=out, olens=olens, feats_rb=rbo, text_rb=rb, duration=d_pred, log_duration=logd)

    def inference(self, text, feats=None, durations=None, spembs=None, sids=None, lids=None, n_timesteps=None,
                  temperature=None, noise_scale: float = 0.667, use_teacher_forcing: bool = False, noise=None, post_noise=None):
        """Same contract as jatts.models.VITS.inference (vits.py:581-679) for feats=None."""
        if use_teacher_forcing:
            raise NotImplementedError("use_teacher_forcing is broken in the reference itself (vits.py:620-636 reads undefined p, e)")
        if spembs is None:
            raise ValueError("spembs is required (the reference crashes without it, vits.py:485)")
        taps = {} if feats is not None else None
        r = self.inference_batch([text], spembs.unsqueeze(0), noise=None if noise is None else [noise],
                                 noise_scale=noise_scale, taps=taps)
        out = dict(feat_gen=r["feat_gen"], duration=r["duration"], log_p_attn=None, ds=None)
        if feats is not None:
            # the given features aligned against the text encoding (vits.py:449-455) and reconstructed through posterior encoder +
            # decoder (`outs_bar`, :546-556); B = 1: no padding.  post_noise: the posterior encoder's randn_like draw.
            from ..alignments import padded_alignment
            P = self._prepare()
            dt, dev, A = P["dtype"], P["dev"], self.adim
            self._post(P)
            ys = feats.to(dev).float().unsqueeze(0).contiguous()
            Tf = int(ys.shape[1])
            lp, ds, _ = padded_alignment(P["align"], taps["hs"], ys, [int(text.numel())], [Tf], A)
            rbo, rbs = hip.RaggedBatch([Tf], dev), hip.RaggedBatch([1], dev)
            sp = spembs.to(dev).float().reshape(1, -1).contiguous()
            sp_t = hip.affine_cast(sp, dt, ldy=P["flows"][0]["layers"][0]["glo"].c_in)
            nz = (torch.randn(Tf, A) if post_noise is None else post_noise.float()).to(dev).contiguous()
            z_bar, _ = self._posterior(P, rbo, rbs, ys[0], sp_t, nz)
            zs = P["dec"].run(rbo, hip.affine_cast(z_bar, hip.F32, scale=P["sqrtA"]), final_dtype=dt)
            fo = P["feat_out"]
            out.update(log_p_attn=lp[0], ds=ds[0],
                       outs_bar=hip.conv1d(rbo, zs, fo.w, fo.c_in, fo.n_out, 1, dtype=dt, bias=fo.b, out_f32=True))
        return out

    def forward(self, text, text_lengths, feats, feats_lengths, durations=None, durations_lengths=None, spembs=None, sids=None,
                lids=None, joint_training=False, post_noise=None):
        """The reference's training-time call, forward only (vits.py:342-579 with is_inference=False): padded batch -> text encoder,
        posterior encoder (WaveNet) + sampling, forward flow, alignment module + monotonic alignment search, duration predictor,
        masked Gaussian upsampling of the prior statistics, conformer decoder.  Same arguments and return dict {outs, d_outs, ys,
        hs, olens_in, bin_loss, log_p_attn, ds, m_p, logs_p, z, y_mask, z_p, m_q, logs_q}, tensors in the reference's layouts.
        ``post_noise`` (B, T_feats, adim): the posterior encoder's randn_like draw (posterior_encoder.py:128), injectable.
        The conformers run on the PADDED batch (key masks only, convolutions read the padding, as in the reference); the
        WaveNet parts run ragged -- there the reference masks after every layer, which is the ragged layout."""
        if self.training and torch.is_grad_enabled():   # differentiable HIP forward (models/vits_train.py; jatts_amd.training.VITSTrainer)
            from .vits_train import train_forward
            self._train_calls += 1
            self._prep = None
            return train_forward(self, text, text_lengths, feats, feats_lengths, spembs, post_noise=post_noise, seed=self._train_calls)
        with torch.no_grad():
            return self._forward_eval(text, text_lengths, feats, feats_lengths, spembs, post_noise)

    def _forward_eval(self, text, text_lengths, feats, feats_lengths, spembs, post_noise=None):
        if spembs is None:
            raise ValueError("spembs is required (the reference crashes without it, vits.py:485)")
        P = self._prepare()
        dt, dev, A, od = P["dtype"], P["dev"], self.adim, self.odim
        ilens = [int(v) for v in text_lengths.tolist()]
        olens = [int(v) for v in feats_lengths.tolist()]
        B, Tm, To = len(ilens), max(ilens), max(olens)
        xs = text[:, :Tm].to(dev)
        ys = feats[:, :To].to(dev).float().contiguous()
        rbt, rbs = hip.RaggedBatch([Tm] * B, dev), hip.RaggedBatch([1] * B, dev)
        kv = hip.h2d(ilens, torch.int32, dev)
        # ---- text encoder on the padded batch (text_encoder.py:104-140)
        hs = P["tenc"].run(rbt, hip.embed_scale(xs.reshape(-1).to(torch.int64).contiguous(), P["emb"], float(A)), kv_len=kv)
        tp = P["te_proj"]
        stats_p = hip.zero_pad_rows(rbt, hip.conv1d(rbt, hip.affine_cast(hs, dt), tp.w, tp.c_in, 2 * A, 1, dtype=dt, bias=tp.b,
                                                    out_f32=True), kv)                 # proj(x) * x_mask: m_p | logs_p
        spembs = spembs.to(dev).float().reshape(B, -1).contiguous()
        hs = P["proj"](rbt, hs, spembs)
        # ---- posterior encoder + forward flow on the valid frames (ragged)
        self._post(P)
        rbo = hip.RaggedBatch(olens, dev)
        fsel = torch.cat([torch.arange(b * To, b * To + olens[b], device=dev) for b in range(B)])     # valid frame rows (plumbing)
        yv = ys.reshape(B * To, od).index_select(0, fsel).contiguous()
        sp_t = hip.affine_cast(spembs, dt, ldy=P["flows"][0]["layers"][0]["glo"].c_in)               # g = spembs, unnormalised
        nz = (torch.randn(B, To, A) if post_noise is None else post_noise[:, :To].float()).to(dev).reshape(B * To, A)
        z, stats_q = self._posterior(P, rbo, rbs, yv, sp_t, nz.index_select(0, fsel).contiguous())
        half = A // 2
        zp = z.clone()
        skip_scale = P.setdefault("skip_scale", torch.full((A,), math.sqrt(1.0 / self.flow_layers), device=dev))
        for fl in P["flows"]:                                              # coupling (xb <- m + xb), then FlipFlow
            fi = fl["inp"]
            h = hip.conv1d(rbo, hip.affine_cast(zp, dt), fi.w, fi.c_in, A, 1, dtype=dt, bias=fi.b, out_f32=True, ldx=A)
            sk = self._wavenet(P, rbo, rbs, h, fl["layers"], sp_t)
            pr = fl["proj"]
            hip.conv1d(rbo, hip.affine_cast(sk, dt, scale=skip_scale), pr.w, pr.c_in, half, 1, dtype=dt, bias=pr.b, alpha=1.0,
                       resid=zp, resid_col0=half, out=zp, out_ld=A, out_col0=half, out_f32=True)
            zp = hip.flip_channels(zp)
        # ---- alignment module on the padded batch + monotonic alignment search (jatts_amd.alignments.padded_alignment)
        from ..alignments import padded_alignment
        log_p_attn, ds, bin_loss = padded_alignment(P["align"], hs, ys, ilens, olens, A)
        rbf, rbv = hip.RaggedBatch([To] * B, dev), hip.RaggedBatch(ilens, dev)
        tsel = torch.cat([torch.arange(b * Tm, b * Tm + ilens[b], device=dev) for b in range(B)])
        d_outs = hip.zero_pad_rows(rbt, hip.predictor_head(P["dur"].trunk(rbt, hs), P["dur"].w, P["dur"].b), kv).view(B, Tm)
        # ---- prior statistics upsampled with the MAS durations (padded frames take the value of frame 0, length_regulator.py:139-141)
        d_int = torch.cat([ds[b, : ilens[b]] for b in range(B)]).to(torch.int64).contiguous()
        up = hip.gaussian_upsample(rbv, d_int, rbo, stats_p.index_select(0, tsel).contiguous())      # (valid frames, 2A)
        cu = rbo.cu_host
        rows = hip.h2d([cu[b] + (t if t < olens[b] else 0) for b in range(B) for t in range(To)], torch.int64, dev)
        up_pad = up.index_select(0, rows).view(B, To, 2 * A)
        # ---- decoder on the padded batch (z is zero at padded frames; key mask = olens)
        def pad_frames(v):     # ragged (valid frames, C) -> padded (B, To, C) with zeros
            out = torch.zeros(B * To, v.shape[1], dtype=v.dtype, device=dev)
            out[fsel] = v
            return out.view(B, To, -1)
        z_pad = pad_frames(z)
        kvo = hip.h2d(olens, torch.int32, dev)
        zs_t = P["dec"].run(rbf, hip.affine_cast(z_pad.view(B * To, A), hip.F32, scale=P["sqrtA"]), final_dtype=dt, kv_len=kvo)
        fo = P["feat_out"]
        outs = hip.conv1d(rbf, zs_t, fo.w, fo.c_in, fo.n_out, 1, dtype=dt, bias=fo.b, out_f32=True).view(B, To, od)
        sq = pad_frames(stats_q)
        y_mask = (torch.arange(To, device=dev).unsqueeze(0) < kvo.unsqueeze(1)).float().unsqueeze(1)
        tr = lambda v: v.transpose(1, 2)   # noqa: E731  the reference keeps these channel-first
        return {"m_q": tr(sq[..., :A]), "logs_q": tr(sq[..., A:]), "outs": outs, "d_outs": d_outs, "ys": ys,
                "hs": tr(hs.view(B, Tm, A)), "olens_in": feats_lengths, "bin_loss": bin_loss, "log_p_attn": log_p_attn, "ds": ds,
                "m_p": tr(up_pad[..., :A]), "logs_p": tr(up_pad[..., A:]), "z": tr(z_pad), "y_mask": y_mask, "z_p": tr(pad_frames(zp))}
