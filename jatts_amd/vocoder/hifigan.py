"""HiFi-GAN v1 generator on the MI355X HIP path.

Drop-in for ``parallel_wavegan.models.HiFiGANGenerator`` as the reference uses it
(/root/reference/jatts/vocoder/vocoder.py:41-44,64): same constructor kwargs and
state_dict keys [recalled from the public package, which is not vendored in the
reference — parity unpinned, see oracle/hifigan_oracle.py], ``remove_weight_norm()``,
``inference(c, normalize_before=False)``.  ``inference_batch`` is the data-parallel
entry: one packed ragged batch of mels -> packed waveforms.

Kernel schedule per batch: affine/cast -> input conv (MFMA implicit GEMM) -> per
upsample stage [polyphase ConvTranspose as MFMA conv (LeakyReLU + MRF mean fused on the
input side) -> 3 ResBlocks x 3 fused dilation units] -> output conv + tanh.
"""
import contextlib
import os

import torch

from .. import hip
from ..graphs import GraphCache
from ..models._conformer import PackedConv
from ..models import _schema as S


class PackedSplitConv:
    """A ResBlock conv packed for the JATTS_F32S unit: hi/lo f16 halves of w[n] * 2^s[n] + the inverse scales (hip.pack_conv_weight_split)."""

    def __init__(self, w, b):
        self.n_out, self.c_in, self.k = w.shape
        self.w, self.inv = hip.pack_conv_weight_split(w, 32)
        self.b = b.detach().float().contiguous()


# fragment order of the emulated units' weights: "16" = the v_mfma_f32_16x16x32_bf16 kernels (round 6: the power-limited matrix pipe sustains 14 % more of that
# form, csrc/resunit_emul16_impl.h), "32" = the round-5 v_mfma_f32_32x32x16_bf16 kernels (A/B runs; the fused ResBlock launches always take them)
EMUL_UNIT_FORM = os.environ.get("JATTS_RESUNIT_EMUL_FORM", "16")


class PackedEmulConv:
    """A ResBlock conv packed for the JATTS_F32E unit: the three exact bf16 terms of every weight, no scales; ``k32``: in the fragment order of the
    16 x 16 x 32 kernels (hip.pack_unit_weight_bf16x3_k32, w_layout 1) instead of the 32 x 32 x 16 ones' (hip.pack_conv_weight_bf16x3, w_layout 0)."""

    def __init__(self, w, b, k32=False, also32=False):
        self.n_out, self.c_in, self.k = w.shape
        self.layout = 1 if k32 else 0
        self.w, self.inv = (hip.pack_unit_weight_bf16x3_k32(w) if k32 else hip.pack_conv_weight_bf16x3(w, 32)), None
        self.w32 = hip.pack_conv_weight_bf16x3(w, 32) if (k32 and also32) else self.w      # what a fused ResBlock launch takes (always the 32 x 32 x 16 order)
        self.b = b.detach().float().contiguous()


class HiFiGANGenerator(torch.nn.Module):
    def __init__(self, in_channels=80, out_channels=1, channels=512, kernel_size=7,
                 upsample_scales=(8, 8, 2, 2), upsample_kernel_sizes=(16, 16, 4, 4),
                 resblock_kernel_sizes=(3, 7, 11), resblock_dilations=((1, 3, 5), (1, 3, 5), (1, 3, 5)),
                 use_additional_convs=True, bias=True, nonlinear_activation="LeakyReLU",
                 nonlinear_activation_params=None, use_weight_norm=True, **unused):
        super().__init__()
        if nonlinear_activation != "LeakyReLU":
            raise NotImplementedError("only LeakyReLU generators are supported")
        if out_channels != 1:
            raise NotImplementedError("out_channels must be 1")
        if not use_additional_convs or not bias:
            raise NotImplementedError("use_additional_convs=True and bias=True required (HiFi-GAN v1)")
        if len(upsample_scales) != len(upsample_kernel_sizes):
            raise ValueError("upsample_scales / upsample_kernel_sizes mismatch")
        self.slope = float((nonlinear_activation_params or {"negative_slope": 0.1}).get("negative_slope", 0.1))
        self.in_channels, self.channels, self.kernel_size = in_channels, channels, kernel_size
        self.upsample_scales = tuple(int(s) for s in upsample_scales)
        self.upsample_kernel_sizes = tuple(int(k) for k in upsample_kernel_sizes)
        self.resblock_kernel_sizes = tuple(int(k) for k in resblock_kernel_sizes)
        self.resblock_dilations = tuple(tuple(int(d) for d in ds) for ds in resblock_dilations)
        if channels % (1 << len(self.upsample_scales)) or channels > 512:
            raise NotImplementedError("channels must be divisible by 2**n_upsamples and <= 512")
        for s_, uk in zip(self.upsample_scales, self.upsample_kernel_sizes):
            if uk != 2 * s_:   # the polyphase upsampling emits exactly L*s samples: ConvTranspose1d(padding=s//2+s%2, output_padding=s%2) only for k = 2s
                raise NotImplementedError(f"upsample_kernel_size {uk} != 2 * scale {s_}: only the k = 2s geometry of the HiFi-GAN recipes is supported")
        if not 1 <= len(self.resblock_kernel_sizes) <= 3:
            raise NotImplementedError("1..3 ResBlocks per stage (the MRF mix kernels take up to 3 inputs)")
        self.hop = 1
        for s in self.upsample_scales:
            self.hop *= s
        spec = S.new_spec()
        S._conv(spec, "input_conv", channels, in_channels, kernel_size)
        nb = len(self.resblock_kernel_sizes)
        c = channels
        for i, uk in enumerate(self.upsample_kernel_sizes):
            spec[f"upsamples.{i}.1.weight"] = ((c, c // 2, uk), "param")
            spec[f"upsamples.{i}.1.bias"] = ((c // 2,), "param")
            c //= 2
            for j, rk in enumerate(self.resblock_kernel_sizes):
                for d in range(len(self.resblock_dilations[j])):
                    S._conv(spec, f"blocks.{i * nb + j}.convs1.{d}.1", c, c, rk)
                    S._conv(spec, f"blocks.{i * nb + j}.convs2.{d}.1", c, c, rk)
        S._conv(spec, "output_conv.1", out_channels, c, kernel_size)
        S.build_from_spec(self, spec)
        self.precision = "fp32"   # the reference's arithmetic; set_precision("fp16") selects the fast mode
        self._prep = None
        self.eval()

    # checkpoints are saved with weight norm (weight_g / weight_v); fold on load
    def load_state_dict(self, state_dict, strict=True, **kw):
        sd = {}
        for k, v in state_dict.items():
            if k.endswith("weight_g"):
                stem = k[:-len("weight_g")]
                wv = state_dict[stem + "weight_v"]
                norm = wv.reshape(wv.shape[0], -1).norm(dim=1).reshape(-1, *([1] * (wv.dim() - 1)))
                sd[stem + "weight"] = v * wv / norm
            elif k.endswith("weight_v") or k in ("mean", "scale"):
                continue
            else:
                sd[k] = v
        self._prep = None
        return super().load_state_dict(sd, strict=strict, **kw)

    def remove_weight_norm(self):
        """Weight norm is folded at load time; kept for API compatibility (vocoder.py:43)."""
        return None

    def set_precision(self, precision):
        """"fp32": exact-f32 MFMA (the reference's arithmetic, the default); "fp16": f16 operands and activations (fast mode);
        "fp32_split" (round 4): f32 activations everywhere, the ResBlock dilation units (97 % of the generator's FLOPs) and the input /
        upsampling convs on error-corrected split-precision MFMA operands (JATTS_F32S: hi/lo f16 halves, f32 accumulate, power-of-two
        scales) -- measured at or below the exact-f32 path's error against fp64 (tests/test_kernels_gpu.py::test_hifigan_resunit_split,
        test_conv1d_split; tests/test_benchsize_gpu.py::test_hifigan_split_mode_at_bench_size);
        "fp32_bf16x3" (round 5): f32 activations everywhere, the same kernels' operands carried EXACTLY as three bf16 terms with seven MFMA
        products per product (JATTS_F32E: no scales, a one-term contraction within 2^-23 = 2 x an f32 FMA's bound; csrc/resunit_emul_impl.h,
        conv1d_emul.h); "fp32_bf16x3_6p": six products (JATTS_F32E6: dropped terms <= 2^-23 per product, 1/7 fewer pipe cycles)."""
        if precision not in hip.PRECISIONS:
            raise ValueError(precision)
        if precision != self.precision:
            self.precision, self._prep = precision, None
        return self

    def _apply(self, fn, *a, **k):
        self._prep = None
        return super()._apply(fn, *a, **k)

    def _prepare(self):
        dev = self.input_conv.weight.device
        if dev.type != "cuda":
            raise hip._abi.JattsHipError("jatts_amd HiFiGANGenerator runs on the GPU only (no CPU fallback)")
        key = (self.precision, str(dev))
        if self._prep is not None and self._prep["key"] == key:
            return self._prep
        hip._abi.load()
        dt = hip.F16 if self.precision == "fp16" else hip.F32
        wmode = hip.WEIGHT_MODE[self.precision]
        split, emul = wmode == 1, wmode in hip.EMUL_CODE
        sd = self.state_dict()
        f32 = lambda t: t.detach().float().to(dev).contiguous()  # noqa: E731
        P = {"key": key, "dtype": dt, "dev": dev, "unit_dtype": hip.F32S if split else hip.EMUL_CODE[wmode] if emul else dt}
        nb = len(self.resblock_kernel_sizes)
        P["ups"], P["blocks"] = [], []
        supported = {hip.F16: (32, 64, 128, 256, 512), hip.F32: (32, 64, 128, 256)}[dt]
        P["unit_channels"] = supported

        # Stage widths the fused unit has no tile for (HiFi-GAN V2/V3: 16, 8 ... channels) are zero-padded to the next
        # supported width at load time: padded channels carry weight 0 and bias 0 everywhere (conv -> 0, LeakyReLU(0) = 0,
        # residual + 0, zero input weights downstream), so the result is exact.
        def cp(c):
            for v in supported:
                if c <= v:
                    return v
            raise NotImplementedError(f"{c} channels exceed the widest fused-unit tile for this precision")

        def padw(w, n, c):   # (n0, c0, k) -> (n, c, k)
            w = w.detach().float()
            o = torch.zeros(n, c, w.shape[2], dtype=torch.float32)
            o[: w.shape[0], : w.shape[1]] = w
            return o

        def padb(b, n):
            o = torch.zeros(n, dtype=torch.float32)
            o[: b.numel()] = b.detach().float()
            return o

        c_prev = hip.round_up(self.channels, 64)   # input conv feeds the generic conv: 64-channel chunks
        with hip.split_weights(wmode):      # fp32_split / fp32_bf16x3: the input / upsampling convs take the split / emulated conv kernel too
            P["in"] = PackedConv(padw(sd["input_conv.weight"], c_prev, sd["input_conv.weight"].shape[1]),
                                 padb(sd["input_conv.bias"], c_prev), dt, dev)
        for i, (s, uk) in enumerate(zip(self.upsample_scales, self.upsample_kernel_sizes)):
            w = sd[f"upsamples.{i}.1.weight"].detach().float()            # ConvTranspose1d: (c_in, c_out, k)
            c_out = cp(w.shape[1])
            if i + 1 < len(self.upsample_scales):
                c_out = max(c_out, 64)   # this stage feeds the next polyphase conv (jatts_conv1d: 64-channel chunks)
            wp = torch.zeros(c_prev, c_out, w.shape[2], dtype=torch.float32)
            wp[: w.shape[0], : w.shape[1]] = w
            wc, pad = hip.convtranspose_as_conv(wp, s, s // 2 + s % 2)
            with hip.split_weights(wmode):
                pc = PackedConv(wc, padb(sd[f"upsamples.{i}.1.bias"], c_out).repeat(s), dt, dev)
            P["ups"].append((pc, pad, s, c_out))
            stage = []
            for j, rk in enumerate(self.resblock_kernel_sizes):
                units = []
                for di, d in enumerate(self.resblock_dilations[j]):
                    q = f"blocks.{i * nb + j}."
                    k32 = EMUL_UNIT_FORM == "16" and c_out % 32 == 0
                    mk = (PackedSplitConv if split else (lambda w, b: PackedEmulConv(w, b, k32, (c_out, rk) in self.fused_blocks_emul)) if emul
                          else (lambda w, b: PackedConv(w, b, dt, dev, c_mult=32)))   # fused unit takes c_in == channels
                    c1 = mk(padw(sd[q + f"convs1.{di}.1.weight"], c_out, c_out).to(dev), padb(sd[q + f"convs1.{di}.1.bias"], c_out).to(dev))
                    c2 = mk(padw(sd[q + f"convs2.{di}.1.weight"], c_out, c_out).to(dev), padb(sd[q + f"convs2.{di}.1.bias"], c_out).to(dev))
                    units.append((c1, c2, rk, d))
                stage.append(units)
            P["blocks"].append(stage)
            c_prev = c_out
        wo = padw(sd["output_conv.1.weight"], 1, c_prev)                  # (1, C, k) -> [k][C]
        P["out_w"] = f32(wo[0].t())
        P["out_b"] = float(sd["output_conv.1.bias"].detach().float()[0])
        self._prep = P
        return P

    # (channels, kernel size) of the ResBlocks issued as ONE fused launch (jatts_hifigan_resblock): the shapes where it
    # measured faster than three unit launches (profiles/r02_notes.md); JATTS_HIFIGAN_FUSE=0 switches it off
    fused_blocks = frozenset() if os.environ.get("JATTS_HIFIGAN_FUSE", "1") == "0" else frozenset({(32, 3), (32, 7), (64, 3)})
    # f32 (round 3): only the k = 3 block of the 32-channel stage, where a conv is 6 K-steps and the per-unit launches spend as long in their
    # staging / store phases as in their MFMAs: 4.11 vs 4.42 ms (0.72 vs 0.66 of the f32 MFMA peak).  C = 64 k = 3 measured 4 % SLOWER fused
    # (7.78 vs 7.48 ms: the 24-row chain halo of a 256-column window) and stays on the per-unit path (tools/bench_unit.py --resblock
    # --dtype f32); JATTS_HIFIGAN_FUSE_F32=0 switches the fused launch off (A/B runs)
    fused_blocks_f32 = frozenset() if os.environ.get("JATTS_HIFIGAN_FUSE_F32", "1") == "0" else frozenset({(32, 3)})
    # fp32_split (round 4): the HBM-bound blocks -- x in + y out once per ResBlock (csrc/resblock_split_impl.h); JATTS_HIFIGAN_FUSE_SPLIT=0: per-unit launches
    fused_blocks_split = (frozenset() if os.environ.get("JATTS_HIFIGAN_FUSE_SPLIT", "1") == "0"
                          else frozenset(tuple(int(v) for v in t.split("x")) for t in os.environ.get("JATTS_HIFIGAN_FUSE_SPLIT_SET", "32x3,32x7,64x3").split(",")))

    # fp32_bf16x3 / fp32_bf16x3_6p (round 5, csrc/resblock_emul_impl.h): only C = 32, k = 3 measured faster fused (3.60 vs 3.93 ms; C = 32 k = 7 0.82x, C = 64 k = 3 0.93-0.96x:
    # the 6-byte tile leaves one workgroup per CU or a 72 %-useful window; profiles/r05_notes.md); JATTS_HIFIGAN_FUSE_EMUL=0: per-unit launches
    fused_blocks_emul = (frozenset() if os.environ.get("JATTS_HIFIGAN_FUSE_EMUL", "1") == "0"
                         else frozenset(tuple(int(v) for v in t.split("x")) for t in os.environ.get("JATTS_HIFIGAN_FUSE_EMUL_SET", "32x3").split(",")))

    # tuning knob (profiles/r01_notes.md): run the independent ResBlock chains of a stage on separate HIP streams
    concurrent = os.environ.get("JATTS_HIFIGAN_STREAMS", "0") == "1"

    def _side_streams(self, n):
        pool = self.__dict__.setdefault("_streams", [])
        while len(pool) < n:
            pool.append(torch.cuda.Stream())
        return pool[:n]

    @torch.no_grad()
    def inference_batch(self, rb, mel, scale=None, shift=None, taps=None):
        """rb: RaggedBatch over mel frames; mel: f32 (rows, in_channels) packed.
        scale/shift: optional per-channel affine applied first (Vocoder.decode normalisation).
        Returns f32 (rows * hop,) packed waveforms (utterance b owns samples cu[b]*hop ...)."""
        P = self._prepare()
        if rb.n_seq == 1 and taps is None and not self.concurrent:
            # B = 1 (Vocoder.decode, the reference's call shape vocoder.py:56-67): the ~150 launches of one utterance replay as ONE hipGraph per frame
            # count (jatts_amd/graphs.py: first sight eager, second captures, bit-identical)
            gc = P.get("graphs")
            if gc is None:
                gc = P["graphs"] = GraphCache()
            affine = tuple(t for t in (scale, shift) if t is not None)

            def body(mel_, *aff):
                it = iter(aff)
                return self._generate(P, rb, mel_, next(it) if scale is not None else None, next(it) if shift is not None else None, None)
            return gc.run(("generate", rb.total, scale is not None, shift is not None), body, (mel,) + affine)
        return self._generate(P, rb, mel, scale, shift, taps)

    def _generate(self, P, rb, mel, scale, shift, taps):
        dt, udt = P["dtype"], P["unit_dtype"]
        pin = P["in"]
        x = hip.affine_cast(mel, dt, scale=scale, shift=shift, ldy=pin.c_in)
        x = hip.conv1d(rb, x, pin.w, pin.c_in, pin.n_out, pin.k, dtype=dt, bias=pin.b)
        if taps is not None:
            taps["input_conv"] = x.float()
        xs, in_scale, rate = [x], 1.0, 1
        supported = P["unit_channels"]
        for i, (pc, pad, s, c_out) in enumerate(P["ups"]):
            rows = rb.total * rate
            up = hip.conv1d(rb, xs, pc.w, pc.c_in, pc.n_out, pc.k, dtype=dt, bias=pc.b, pad=pad,
                            pre_lrelu=self.slope, in_scale=in_scale, len_mul=rate)       # (rows, s*c_out)
            rate *= s
            up = up.view(rows * s, c_out)
            if taps is not None:
                taps[f"up{i}"] = up.float()
            outs = []
            blocks = P["blocks"][i]
            fuse_mean = c_out in supported and len(blocks) in (2, 3)
            # The ResBlocks of a stage are independent chains: with JATTS_HIFIGAN_STREAMS=1 all but the last run on
            # side streams so that workgroups of memory-heavy (k=3) and MFMA-heavy (k=11) units share the CUs.
            # Every buffer of the stage is allocated up front on the launch stream and held until the join.
            side = self._side_streams(len(blocks) - 1) if (self.concurrent and c_out in supported) else []
            main = torch.cuda.current_stream() if side else None
            bufs = [[torch.empty_like(up), torch.empty_like(up)] for _ in blocks]
            if side:
                fork = torch.cuda.Event()
                fork.record(main)
            done = []
            for j, units in enumerate(blocks):
                cur = up
                st = side[j] if j < len(side) else None
                # HBM-bound shapes: the whole ResBlock in one launch (x read once, y written once; residual in registers)
                fset = (self.fused_blocks_split if udt == hip.F32S else self.fused_blocks_emul if udt in hip.EMUL
                        else (self.fused_blocks if dt == hip.F16 else self.fused_blocks_f32))
                if (c_out, units[0][2]) in fset and len(units) <= 3 and st is None \
                        and sum((units[0][2] - 1) // 2 * (u[3] + 1) for u in units) <= (64 if (dt == hip.F16 or udt == hip.F32S or udt in hip.EMUL) else 16):
                    lastb = fuse_mean and j == len(blocks) - 1
                    hip.hifigan_resblock(rb, rate, cur, bufs[j][0], [(getattr(c1, "w32", c1.w), c1.b, getattr(c2, "w32", c2.w), c2.b, d) for c1, c2, _, d in units],
                                         c_out, units[0][2], self.slope, udt, add=outs if lastb else None,
                                         out_scale=1.0 / len(blocks) if lastb else 1.0,
                                         ws=[(c1.inv, c2.inv) for c1, c2, _, _ in units] if udt == hip.F32S else None)
                    outs.append(bufs[j][0])
                    continue
                if st is not None:
                    st.wait_event(fork)
                with torch.cuda.stream(st) if st is not None else contextlib.nullcontext():
                    for di, (c1, c2, rk, d) in enumerate(units):
                        nxt = bufs[j][di & 1]
                        # (c_out is always a fused-unit width: _prepare zero-pads narrower stages and refuses wider ones)
                        last = fuse_mean and j == len(blocks) - 1 and di == len(units) - 1
                        if last:
                            for ev in done:
                                torch.cuda.current_stream().wait_event(ev)
                        # the last unit of the last ResBlock writes the MRF mean (cs / num_blocks) directly
                        hip.hifigan_resunit(rb, rate, cur, nxt, c1.w, c1.b, c2.w, c2.b, c_out, rk, d, self.slope, udt,
                                            add=outs if last else None, out_scale=1.0 / len(blocks) if last else 1.0,
                                            ws=(c1.inv, c2.inv) if udt == hip.F32S else None, w_layout=getattr(c1, "layout", 0))
                        cur = nxt
                    if st is not None:
                        ev = torch.cuda.Event()
                        ev.record(st)
                        done.append(ev)
                outs.append(cur)
            if side and not fuse_mean:
                for ev in done:
                    main.wait_event(ev)
            if fuse_mean:
                xs, in_scale = [outs[-1]], 1.0
            else:
                xs, in_scale = outs, 1.0 / len(outs)
            if taps is not None:
                taps[f"mrf{i}"] = sum(o.float() for o in xs) * in_scale
        return hip.hifigan_output(rb, rate, xs, in_scale, 0.01, xs[0].shape[1], self.kernel_size,
                                  P["out_w"], P["out_b"], dt)

    @torch.no_grad()
    def inference(self, c, normalize_before=False):
        """c: (T, in_channels) -> (T * hop, 1), as parallel_wavegan's HiFiGANGenerator.inference."""
        if normalize_before:
            raise NotImplementedError("normalize_before=True needs registered stats; jatts passes False (vocoder.py:64)")
        if not isinstance(c, torch.Tensor):
            c = torch.tensor(c, dtype=torch.float)
        dev = self.input_conv.weight.device
        c = c.to(dev).float().contiguous()
        rb = hip.RaggedBatch([c.shape[0]], dev)
        return self.inference_batch(rb, c).view(-1, 1)
