"""FastSpeech2 on the MI355X HIP path — drop-in for ``jatts.models.FastSpeech2``.

Same constructor kwargs (reference models/fastspeech2.py:46-128), same state_dict key
schema (so ``load_state_dict(torch.load(ckpt)["model"])`` works), same
``inference()`` signature and return dict (:655-735).  ``inference_batch()`` is the
data-parallel entry point: a packed ragged batch in which every utterance is computed
exactly as the reference's B=1 ``inference()`` would (SURVEY §8 note N1).

The class owns parameters only.  All arithmetic runs in libjatts_hip.so; there is no
PyTorch/CPU fallback — calling it without the extension or on CPU tensors raises.
"""
import logging
import math
from typing import Dict, Optional, Sequence

import torch

from .. import hip
from ..graphs import GraphCache
from ..hip import ACT_NONE, ACT_RELU, ACT_TANH
from . import _schema as S
from ._conformer import BN_EPS, LN_EPS, ConformerRunner, PackedConv, SpkProjection


class _Predictor:
    """Conv1d -> ReLU -> LayerNorm(channels) stack + Linear(->1)
    (duration_predictor.py:60-97, variance_predictor.py:47-85)."""

    def __init__(self, sd, prefix, dtype, device):
        self.dtype = dtype
        self.convs = []
        i = 0
        while (prefix + f"conv.{i}.0.weight") in sd:
            self.convs.append((
                PackedConv(sd[prefix + f"conv.{i}.0.weight"], sd.get(prefix + f"conv.{i}.0.bias"), dtype, device),
                sd[prefix + f"conv.{i}.2.weight"].detach().float().to(device).contiguous(),
                sd[prefix + f"conv.{i}.2.bias"].detach().float().to(device).contiguous(),
            ))
            i += 1
        self.w = sd[prefix + "linear.weight"].detach().float().reshape(-1).to(device).contiguous()
        self.b = float(sd[prefix + "linear.bias"].detach().float().reshape(-1)[0])

    def trunk(self, rb, x_t):
        h = x_t
        for conv, g, b in self.convs:
            h = hip.conv1d(rb, h, conv.w, conv.c_in, conv.n_out, conv.k, dtype=self.dtype, bias=conv.b, act=ACT_RELU)
            h = hip.layernorm(h, g, b, self.dtype, LN_EPS)
        return h


class FastSpeech2(torch.nn.Module):
    def __init__(
        self,
        idim: int, odim: int, adim: int = 384, aheads: int = 4, elayers: int = 6, eunits: int = 1536,
        dlayers: int = 6, dunits: int = 1536, postnet_layers: int = 5, postnet_chans: int = 512,
        postnet_filts: int = 5, postnet_dropout_rate: float = 0.5, positionwise_layer_type: str = "conv1d",
        positionwise_conv_kernel_size: int = 1, use_scaled_pos_enc: bool = True, use_batch_norm: bool = True,
        encoder_normalize_before: bool = True, decoder_normalize_before: bool = True,
        encoder_concat_after: bool = False, decoder_concat_after: bool = False, reduction_factor: int = 1,
        encoder_type: str = "transformer", decoder_type: str = "transformer",
        transformer_enc_dropout_rate: float = 0.1, transformer_enc_positional_dropout_rate: float = 0.1,
        transformer_enc_attn_dropout_rate: float = 0.1, transformer_dec_dropout_rate: float = 0.1,
        transformer_dec_positional_dropout_rate: float = 0.1, transformer_dec_attn_dropout_rate: float = 0.1,
        conformer_rel_pos_type: str = "legacy", conformer_pos_enc_layer_type: str = "rel_pos",
        conformer_self_attn_layer_type: str = "rel_selfattn", conformer_activation_type: str = "swish",
        use_macaron_style_in_conformer: bool = True, use_cnn_in_conformer: bool = True, zero_triu: bool = False,
        conformer_enc_kernel_size: int = 7, conformer_dec_kernel_size: int = 31,
        duration_predictor_layers: int = 2, duration_predictor_chans: int = 384,
        duration_predictor_kernel_size: int = 3, duration_predictor_dropout_rate: float = 0.1,
        energy_predictor_layers: int = 2, energy_predictor_chans: int = 384, energy_predictor_kernel_size: int = 3,
        energy_predictor_dropout: float = 0.5, energy_embed_kernel_size: int = 9, energy_embed_dropout: float = 0.5,
        stop_gradient_from_energy_predictor: bool = False,
        pitch_predictor_layers: int = 2, pitch_predictor_chans: int = 384, pitch_predictor_kernel_size: int = 3,
        pitch_predictor_dropout: float = 0.5, pitch_embed_kernel_size: int = 9, pitch_embed_dropout: float = 0.5,
        stop_gradient_from_pitch_predictor: bool = False,
        spks: Optional[int] = None, spk_embed_dim: Optional[int] = None, spk_embed_integration_type: str = "add",
        use_gst: bool = False, gst_tokens: int = 10, gst_heads: int = 4, gst_conv_layers: int = 6,
        gst_conv_chans_list: Sequence[int] = (32, 32, 64, 64, 128, 128), gst_conv_kernel_size: int = 3,
        gst_conv_stride: int = 2, gst_gru_layers: int = 1, gst_gru_units: int = 128,
        init_type: str = "xavier_uniform", init_enc_alpha: float = 1.0, init_dec_alpha: float = 1.0,
        use_masking: bool = False, use_weighted_masking: bool = False,
    ):
        super().__init__()
        self.idim, self.odim, self.adim, self.aheads = idim, odim, adim, aheads
        self.eos = idim - 1
        self.reduction_factor = reduction_factor
        self.padding_idx = 0
        if encoder_type != "conformer" or decoder_type != "conformer":
            # The reference's `transformer` branch raises NameError (TransformerEncoder is never
            # imported, fastspeech2.py:274,402): only conformer is live.
            raise ValueError(f"{encoder_type}/{decoder_type} is not supported (only 'conformer').")
        if conformer_rel_pos_type != "legacy":
            raise NotImplementedError("only conformer_rel_pos_type='legacy' (the reference default) is supported")
        if conformer_pos_enc_layer_type not in ("rel_pos", "legacy_rel_pos") or \
                conformer_self_attn_layer_type not in ("rel_selfattn", "legacy_rel_selfattn"):
            raise NotImplementedError("only (legacy) relative positional attention is supported")
        if use_gst:
            raise NotImplementedError("GST style encoder is outside the stage-4 hot path")
        if reduction_factor != 1:
            raise NotImplementedError("reduction_factor > 1 is not supported")
        if zero_triu or encoder_concat_after or decoder_concat_after or not (
                encoder_normalize_before and decoder_normalize_before):
            raise NotImplementedError("only normalize_before=True, concat_after=False, zero_triu=False")
        self.spks = spks if (spks is not None and spks > 1) else None
        self.spk_embed_dim = spk_embed_dim if (spk_embed_dim is not None and spk_embed_dim > 0) else None
        self.spk_embed_integration_type = spk_embed_integration_type
        if self.spk_embed_dim is not None and spk_embed_integration_type not in ("add", "concat"):
            raise NotImplementedError("support only add or concat.")

        spec = S.new_spec()
        spec["encoder.embed.0.weight"] = ((idim, adim), "param")
        S.conformer_spec(spec, "encoder.", adim, aheads, eunits, elayers, positionwise_layer_type,
                         positionwise_conv_kernel_size, use_macaron_style_in_conformer, use_cnn_in_conformer,
                         conformer_enc_kernel_size)
        if self.spks is not None:
            spec["sid_emb.weight"] = ((spks, adim), "param")
        if self.spk_embed_dim is not None:
            S._lin(spec, "projection", adim, self.spk_embed_dim + (adim if spk_embed_integration_type == "concat" else 0))
        S.predictor_spec(spec, "duration_predictor.", adim, duration_predictor_layers, duration_predictor_chans,
                         duration_predictor_kernel_size)
        S.predictor_spec(spec, "pitch_predictor.", adim, pitch_predictor_layers, pitch_predictor_chans,
                         pitch_predictor_kernel_size)
        S._conv(spec, "pitch_embed.0", adim, 1, pitch_embed_kernel_size)
        S.predictor_spec(spec, "energy_predictor.", adim, energy_predictor_layers, energy_predictor_chans,
                         energy_predictor_kernel_size)
        S._conv(spec, "energy_embed.0", adim, 1, energy_embed_kernel_size)
        S.conformer_spec(spec, "decoder.", adim, aheads, dunits, dlayers, positionwise_layer_type,
                         positionwise_conv_kernel_size, use_macaron_style_in_conformer, use_cnn_in_conformer,
                         conformer_dec_kernel_size)
        S._lin(spec, "feat_out", odim * reduction_factor, adim)
        if postnet_layers > 0:
            S.postnet_spec(spec, "postnet.", odim, postnet_layers, postnet_chans, postnet_filts, use_batch_norm)
        S.build_from_spec(self, spec)
        # train-mode behaviour (models/fastspeech2_train.py): the dropout sites of the reference and the predictor detaches
        self.dropout_rates = dict(
            enc=transformer_enc_dropout_rate, enc_pos=transformer_enc_positional_dropout_rate, enc_attn=transformer_enc_attn_dropout_rate,
            dec=transformer_dec_dropout_rate, dec_pos=transformer_dec_positional_dropout_rate, dec_attn=transformer_dec_attn_dropout_rate,
            dur=duration_predictor_dropout_rate, pitch=pitch_predictor_dropout, energy=energy_predictor_dropout,
            pitch_embed=pitch_embed_dropout, energy_embed=energy_embed_dropout, postnet=postnet_dropout_rate)
        self.stop_gradient_from_pitch_predictor = stop_gradient_from_pitch_predictor
        self.stop_gradient_from_energy_predictor = stop_gradient_from_energy_predictor
        self._train_calls = 0
        self.precision = "fp32"   # the reference's arithmetic; set_precision("fp16") selects the fast mode
        self._prep = None
        self.eval()

    # ------------------------------------------------------------------ weight preparation
    def set_precision(self, precision: str):
        """'fp16' (f16 MFMA operands, f32 accumulate — fast mode) or 'fp32' (exact-f32 MFMA, parity mode)."""
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

    def train(self, mode: bool = True):
        """train(True) also turns the parameters' requires_grad on (they are created frozen for the inference path)."""
        super().train(mode)
        if mode:
            self.requires_grad_(True)
        return self

    def _apply(self, fn, *a, **k):
        self._prep = None
        return super()._apply(fn, *a, **k)

    def _prepare(self):
        dev = self.feat_out.weight.device
        if dev.type != "cuda":
            raise hip._abi.JattsHipError("jatts_amd.FastSpeech2 runs on the GPU only (no CPU fallback); call .to('cuda')")
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
        P["enc"] = ConformerRunner(sd, "encoder.", self.aheads, dt, dev)
        P["dec"] = ConformerRunner(sd, "decoder.", self.aheads, dt, dev)
        with hip.split_weights(False):     # exact f32 always: durations are integers
            P["dur"] = _Predictor(sd, "duration_predictor.", hip.F32, dev)   # always f32: durations are integers, a
        # rounding-boundary flip under f16 would change the utterance length (the trunk is 0.3 % of the FLOPs)
        P["pitch"] = _Predictor(sd, "pitch_predictor.", dt, dev)
        P["energy"] = _Predictor(sd, "energy_predictor.", dt, dev)
        for nm in ("pitch_embed", "energy_embed"):
            w = sd[nm + ".0.weight"]
            P[nm] = (f32(w.reshape(w.shape[0], w.shape[-1])), f32(sd[nm + ".0.bias"]))
        P["feat_out"] = PackedConv(sd["feat_out.weight"], sd["feat_out.bias"], dt, dev)
        post = []
        i = 0
        while f"postnet.postnet.{i}.0.weight" in sd:
            q = f"postnet.postnet.{i}."
            if (q + "1.running_mean") in sd:  # BatchNorm(eval) folded into the conv
                s = sd[q + "1.weight"].float() / torch.sqrt(sd[q + "1.running_var"].float() + BN_EPS)
                t = sd[q + "1.bias"].float() - sd[q + "1.running_mean"].float() * s
                post.append(PackedConv(sd[q + "0.weight"], None, dt, dev, scale=s, shift=t))
            else:
                post.append(PackedConv(sd[q + "0.weight"], None, dt, dev))
            i += 1
        P["postnet"] = post
        if self.spks is not None:
            P["sid_emb"] = f32(sd["sid_emb.weight"])
        if self.spk_embed_dim is not None:
            P["proj"] = SpkProjection(sd, self.adim, self.spk_embed_integration_type, dt, dev)
        self._prep = P
        return P

    # ------------------------------------------------------------------------ hot path
    @torch.no_grad()
    def inference_batch(self, texts, spembs=None, sids=None, alpha: float = 1.0, durations=None,
                        taps=None):
        """Batched stage-4 text2mel.

        texts: list of LongTensor (T_b,) on the GPU.  spembs: (B, spk_embed_dim) or None.
        sids: (B,) or None.  durations: optional list of LongTensor overriding predicted durations.
        Returns dict with packed tensors: feat_gen (sum T_feats, odim) f32, before, olens (list),
        duration/pitch/energy/log_duration packed over tokens, and ``feats_rb``/``text_rb`` geometry.
        """
        P = self._prepare()
        dev = P["dev"]
        lens = [int(t.numel()) for t in texts]
        if min(lens) <= 0:
            raise ValueError("empty text")
        rb = hip.RaggedBatch(lens, dev)
        ids = torch.cat([t.reshape(-1) for t in texts]).to(device=dev, dtype=torch.int64).contiguous()
        d_over = None
        if durations is not None:
            d_over = torch.cat([d.reshape(-1) for d in durations]).to(device=dev, dtype=torch.int64).contiguous()
            if d_over.numel() != rb.total:
                raise ValueError("durations do not match texts")
        if self.spks is not None and sids is None:
            raise ValueError("sids required (spks is set)")
        if self.spk_embed_dim is not None and spembs is None:
            raise ValueError("spembs required (spk_embed_dim is set)")
        sp = None if self.spk_embed_dim is None else spembs.to(dev).float().reshape(rb.n_seq, -1).contiguous()
        sd = None if self.spks is None else sids.to(dev).view(-1).long().contiguous()
        # B = 1 (the reference's own call shape, tts_decode.py:230): both halves replay as hipGraphs keyed by (T_text) / (T_text, T_feats)
        # (jatts_amd/graphs.py: first sight eager, second captures; bit-identical to the eager launches)
        gc = None
        if rb.n_seq == 1 and taps is None:
            gc = P.get("graphs")
            if gc is None:
                gc = P["graphs"] = GraphCache()
        extra = tuple(t for t in (sp, sd, d_over) if t is not None)

        def front(ids_, *more):
            it = iter(more)
            sp_ = next(it) if sp is not None else None
            sd_ = next(it) if sd is not None else None
            do_ = next(it) if d_over is not None else None
            return self._front(P, rb, ids_, sp_, sd_, do_, alpha, taps)

        if gc is not None:
            hs, p_outs, e_outs, logd, d_pred, d_eff, cum, sizes = gc.run(("front", lens[0], float(alpha), sp is not None, sd is not None, d_over is not None),
                                                                         front, (ids,) + extra)
        else:
            hs, p_outs, e_outs, logd, d_pred, d_eff, cum, sizes = front(ids, *extra)
        # length regulator (length_regulator.py:70-97): the one host sync of the path — output sizes
        olens_h = hip.lr_sizes_host(rb, sizes, True)    # an all-zero utterance gets all ones, as the reference's B=1 call
        rbo = hip.RaggedBatch(olens_h, dev)
        if gc is not None:
            after, before = gc.run(("back", lens[0], olens_h[0]), lambda hs_, cum_: self._back(P, rb, rbo, hs_, cum_, None), (hs, cum))
        else:
            after, before = self._back(P, rb, rbo, hs, cum, taps)
        return dict(feat_gen=after, before=before, olens=olens_h, feats_rb=rbo, text_rb=rb, duration=d_pred,
                    duration_used=d_eff, pitch=p_outs, energy=e_outs, log_duration=logd)

    def _front(self, P, rb, ids, spembs, sids, d_over, alpha, taps):
        """Token ids -> encoder -> variance adaptor -> length-regulator sizes, all on the device (capturable).
        -> (hs, pitch, energy, log duration, predicted duration, durations used, their running sums, sizes for hip.lr_sizes_host)."""
        dt = P["dtype"]
        A = self.adim
        # encoder.embed: Embedding -> LegacyRelPositionalEncoding (x * sqrt(adim))  encoder.py:133-137.  Out-of-range ids are
        # counted by the kernel and raised (IndexError, as torch.nn.Embedding) at the length-regulator host sync
        n_bad = torch.zeros(1, dtype=torch.int64, device=ids.device)
        x = hip.embed_scale(ids, P["emb"], math.sqrt(A), n_bad)
        hs = P["enc"].run(rb, x, taps=taps)                                  # f32 (R, A)
        if taps is not None:
            taps["encoder_out"] = hs.clone()
        if self.spks is not None:
            vec = P["sid_emb"][sids].contiguous()   # row select (plumbing)
            hip.add_seq_vector(rb, hs, vec)
        if self.spk_embed_dim is not None:
            hs = P["proj"](rb, hs, spembs)
        hs_t = hip.affine_cast(hs, dt)
        p_outs = hip.predictor_head(P["pitch"].trunk(rb, hs_t), P["pitch"].w, P["pitch"].b)
        e_outs = hip.predictor_head(P["energy"].trunk(rb, hs_t), P["energy"].w, P["energy"].b)
        logd, d_pred = hip.predictor_head(P["dur"].trunk(rb, hs), P["dur"].w, P["dur"].b,
                                          want_duration=True)
        hip.variance_embed_add(rb, hs, p_outs, P["pitch_embed"][0], P["pitch_embed"][1],
                               e_outs, P["energy_embed"][0], P["energy_embed"][1])
        if taps is not None:
            taps["variance_out"] = hs.clone()
        d_eff, cum, sizes = hip.lr_sizes_dev(rb, d_pred if d_over is None else d_over, alpha, check=n_bad)
        return hs, p_outs, e_outs, logd, d_pred, d_eff, cum, sizes

    def _back(self, P, rb, rbo, hs, cum, taps):
        """Length-regulator gather -> decoder -> feat_out -> postnet (capturable).  -> (after, before)."""
        dt, dev = P["dtype"], P["dev"]
        A = self.adim
        if taps is not None:
            ys, fidx = hip.lr_gather(rb, cum, rbo, hs, want_index=True)
            taps["lr_out"], taps["frame_index"] = ys.clone(), fidx
        else:
            ys = hip.lr_gather(rb, cum, rbo, hs)
        # decoder: input_layer=None -> pos-enc only: x * sqrt(adim)   encoder.py:138-141
        if "sqrtA" not in P:
            P["sqrtA"] = torch.full((A,), math.sqrt(A), dtype=torch.float32, device=dev)
        ys = hip.affine_cast(ys, hip.F32, scale=P["sqrtA"])
        zs_t = P["dec"].run(rbo, ys, final_dtype=dt)                          # T (Rf, A)
        fo = P["feat_out"]
        before = hip.conv1d(rbo, zs_t, fo.w, fo.c_in, fo.n_out, 1, dtype=dt, bias=fo.b, out_f32=True)  # (Rf, odim)
        after = before
        if P["postnet"]:
            h = hip.affine_cast(before, dt, ldy=P["postnet"][0].c_in)
            n = len(P["postnet"])
            for i, pc in enumerate(P["postnet"]):
                last = i == n - 1
                if last:
                    after = hip.conv1d(rbo, h, pc.w, pc.c_in, pc.n_out, pc.k, dtype=dt, bias=pc.b, act=ACT_NONE,
                                       resid=before, out_f32=True)
                else:
                    h = hip.conv1d(rbo, h, pc.w, pc.c_in, pc.n_out, pc.k, dtype=dt, bias=pc.b, act=ACT_TANH)
        if taps is not None:
            taps["decoder_out"] = zs_t.float()
        return after, before

    def inference(
        self, text: torch.Tensor, feats: Optional[torch.Tensor] = None, durations: Optional[torch.Tensor] = None,
        spembs: torch.Tensor = None, sids: Optional[torch.Tensor] = None, lids: Optional[torch.Tensor] = None,
        pitch: Optional[torch.Tensor] = None, energy: Optional[torch.Tensor] = None, alpha: float = 1.0,
        use_teacher_forcing: bool = False,
    ) -> Dict[str, torch.Tensor]:
        """Same contract as jatts.models.FastSpeech2.inference (fastspeech2.py:655-735)."""
        if use_teacher_forcing:
            # ground-truth duration / pitch / energy (fastspeech2.py:704-717): _forward(is_inference=False) on a batch of one
            if durations is None or pitch is None or energy is None:
                raise ValueError("use_teacher_forcing needs durations, pitch and energy")
            T = int(text.numel())
            n = torch.tensor([T])
            To = int(durations.sum())
            ys = torch.zeros(1, To, self.odim) if feats is None else feats.unsqueeze(0)
            r = self.forward(text.view(1, T), n, ys, torch.tensor([To]), durations.view(1, T), n, pitch.view(1, T, 1), n,
                             energy.view(1, T, 1), n, spembs=None if spembs is None else spembs.unsqueeze(0), sids=sids)
            outs = r["after_outs"] if r["after_outs"] is not None else r["before_outs"]
            return dict(feat_gen=outs[0], duration=r["d_outs"][0], pitch=r["p_outs"][0], energy=r["e_outs"][0])
        r = self.inference_batch([text], spembs=None if spembs is None else spembs.unsqueeze(0),
                                 sids=sids, alpha=alpha)
        return dict(feat_gen=r["feat_gen"], duration=r["duration"], pitch=r["pitch"].unsqueeze(-1),
                    energy=r["energy"].unsqueeze(-1))

    def forward(
        self, text: torch.Tensor, text_lengths: torch.Tensor, feats: torch.Tensor, feats_lengths: torch.Tensor,
        durations: torch.Tensor, durations_lengths: torch.Tensor, pitch: torch.Tensor, pitch_lengths: torch.Tensor,
        energy: torch.Tensor, energy_lengths: torch.Tensor, spembs: Optional[torch.Tensor] = None,
        sids: Optional[torch.Tensor] = None, lids: Optional[torch.Tensor] = None, joint_training: bool = False,
    ) -> Dict[str, torch.Tensor]:
        """The reference's training-time call (fastspeech2.py:473-564).  In train() mode with gradients enabled this is the
        differentiable HIP forward of models/fastspeech2_train.py (f32; batch-statistics BatchNorm, dropout); otherwise the
        no-grad eval-mode pass below."""
        if self.training and torch.is_grad_enabled():
            from .fastspeech2_train import train_forward
            self._train_calls += 1
            self._prep = None   # the parameters are about to change under the packed inference weights
            return train_forward(self, text, text_lengths, feats, feats_lengths, durations, durations_lengths, pitch, pitch_lengths,
                                 energy, energy_lengths, spembs=spembs, sids=sids, seed=self._train_calls)
        with torch.no_grad():
            return self._forward_eval(text, text_lengths, feats, feats_lengths, durations, durations_lengths, pitch, pitch_lengths,
                                      energy, energy_lengths, spembs, sids, lids, joint_training)

    def _forward_eval(
        self, text: torch.Tensor, text_lengths: torch.Tensor, feats: torch.Tensor, feats_lengths: torch.Tensor,
        durations: torch.Tensor, durations_lengths: torch.Tensor, pitch: torch.Tensor, pitch_lengths: torch.Tensor,
        energy: torch.Tensor, energy_lengths: torch.Tensor, spembs: Optional[torch.Tensor] = None,
        sids: Optional[torch.Tensor] = None, lids: Optional[torch.Tensor] = None, joint_training: bool = False,
    ) -> Dict[str, torch.Tensor]:
        """The reference's training-time call (fastspeech2.py:473-564 -> _forward(is_inference=False) :566-653), forward
        only: teacher-forced durations / pitch / energy on a PADDED batch.  Same arguments, same return dict
        {before_outs, after_outs, d_outs, p_outs, e_outs, ys, olens}.

        Unlike inference_batch (ragged, every utterance as the reference's B=1 call), this reproduces the reference's
        batched arithmetic: every sequence is computed at the padded length, only the attention sees the key mask
        (encoder.py:233-289), so padding rows flow through the convolutions exactly as they do there; the predictors'
        outputs are multiplied by the non-pad mask (variance_predictor.py:81-83), the length regulator zero-pads to the
        longest output (length_regulator.py:96-97), and outputs at padded frames are returned as computed."""
        P = self._prepare()
        dt, dev = P["dtype"], P["dev"]
        A = self.adim
        ilens = [int(v) for v in text_lengths.tolist()]
        olens = feats_lengths
        B, Tm = len(ilens), max(ilens)
        xs = text[:, :Tm].to(dev)                                    # "for data-parallel" truncations (:520-524)
        ys = feats[:, : int(feats_lengths.max())]
        ds = durations[:, : int(durations_lengths.max())].to(dev)
        ps = pitch[:, : int(pitch_lengths.max())].to(dev).float()
        es = energy[:, : int(energy_lengths.max())].to(dev).float()
        if ds.shape[1] != Tm or ps.shape[1] != Tm or es.shape[1] != Tm:
            raise ValueError("durations / pitch / energy must be padded to the text length")
        rb = hip.RaggedBatch([Tm] * B, dev)                          # padded geometry: every sequence Tm rows
        kv = hip.h2d(ilens, torch.int32, dev)
        ids = xs.reshape(-1).to(torch.int64).contiguous()
        x = hip.embed_scale(ids, P["emb"], math.sqrt(A))
        hs = P["enc"].run(rb, x, kv_len=kv)                          # f32 (B*Tm, A)
        if self.spks is not None:
            hip.add_seq_vector(rb, hs, P["sid_emb"][sids.to(dev).view(-1).long()].contiguous())
        if self.spk_embed_dim is not None:
            hs = P["proj"](rb, hs, spembs)
        hs_t = hip.affine_cast(hs, dt)
        p_outs = hip.zero_pad_rows(rb, hip.predictor_head(P["pitch"].trunk(rb, hs_t), P["pitch"].w, P["pitch"].b), kv)
        e_outs = hip.zero_pad_rows(rb, hip.predictor_head(P["energy"].trunk(rb, hs_t), P["energy"].w, P["energy"].b), kv)
        d_outs = hip.zero_pad_rows(rb, hip.predictor_head(P["dur"].trunk(rb, hs), P["dur"].w, P["dur"].b), kv)   # log domain
        # ground-truth pitch / energy embeddings (:618-622), then the length regulator on the ground-truth durations
        hip.variance_embed_add(rb, hs, ps.reshape(-1).contiguous(), P["pitch_embed"][0], P["pitch_embed"][1],
                               es.reshape(-1).contiguous(), P["energy_embed"][0], P["energy_embed"][1])
        d_used = ds.reshape(-1).to(torch.int64).contiguous()
        d_eff, cum, ol, _ = hip.lr_durations(rb, d_used, 1.0, zero_rule=0)
        ol_h = ol.tolist()
        hip.check_bad_ids(dev)   # out-of-range token ids: IndexError as nn.Embedding, read at the sync above
        if sum(ol_h) == 0:   # the batched call applies the all-zero rule only when the whole batch sums to 0 (:85-94)
            logging.warning("predicted durations includes all 0 sequences. fill the first element with 1.")
            d_eff, cum, ol, _ = hip.lr_durations(rb, d_used, 1.0, zero_rule=1)
            ol_h = ol.tolist()
        To = max(ol_h)
        rbo = hip.RaggedBatch([To] * B, dev)
        yl = hip.lr_gather(rb, cum, rbo, hs)                         # zero rows past each utterance's frames (pad_list)
        if "sqrtA" not in P:
            P["sqrtA"] = torch.full((A,), math.sqrt(A), dtype=torch.float32, device=dev)
        yl = hip.affine_cast(yl, hip.F32, scale=P["sqrtA"])
        kvo = olens.to(device=dev, dtype=torch.int32).contiguous()  # h_masks = _source_mask(olens) (:636-637)
        zs_t = P["dec"].run(rbo, yl, final_dtype=dt, kv_len=kvo)
        fo = P["feat_out"]
        before = hip.conv1d(rbo, zs_t, fo.w, fo.c_in, fo.n_out, 1, dtype=dt, bias=fo.b, out_f32=True)
        after = None
        if P["postnet"]:
            h = hip.affine_cast(before, dt, ldy=P["postnet"][0].c_in)
            n = len(P["postnet"])
            for i, pc in enumerate(P["postnet"]):
                if i == n - 1:
                    after = hip.conv1d(rbo, h, pc.w, pc.c_in, pc.n_out, pc.k, dtype=dt, bias=pc.b, act=ACT_NONE,
                                       resid=before, out_f32=True)
                else:
                    h = hip.conv1d(rbo, h, pc.w, pc.c_in, pc.n_out, pc.k, dtype=dt, bias=pc.b, act=ACT_TANH)
        od = self.odim
        return {
            "before_outs": before.view(B, To, od),
            "after_outs": None if after is None else after.view(B, To, od),
            "d_outs": d_outs.view(B, Tm), "p_outs": p_outs.view(B, Tm, 1), "e_outs": e_outs.view(B, Tm, 1),
            "ys": ys, "olens": olens,
        }
