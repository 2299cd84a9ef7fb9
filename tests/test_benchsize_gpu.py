"""Oracle / reference parity AT THE SHAPES bench.py TIMES: the largest comparisons used to be ~50 frames, so the
128 x 128 f32 conv tile (12.9 % of the profiled step) and the 768-frame vocoder were reached by self-comparisons only.

  * FastSpeech2: utterances 0 / 37 of the bench batch through the REAL reference's B=1 inference() (tests/golden/fs2_bench768.npz, made by
    tests/golden/make_golden_r3.py) vs f32 inference_batch alone and inside the 64-utterance bench batch;
  * HiFi-GAN v1 22 kHz full width on that 768-frame mel vs oracle.hifigan_generate (unpinned oracle), alone and inside
    the batch of 64;
  * jatts_conv1d f32 vs fp64 F.conv1d at launches of > 600 workgroups with every f32 kernel variant forced (LDS-staged 128 x 64 and
    128 x 128 tiles, the register-streamed kernel with both ring depths) and under the product heuristic (variant 0 = the register-streamed kernel
    wherever it applies).
"""
import math

import pytest
import torch
import torch.nn.functional as F

from helpers import golden_state, load_golden, maxdiff, relerr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bench_stack(cuda, lib):
    from jatts_amd.models import FastSpeech2
    from jatts_amd.synthetic import FS2_JSUT, HIFIGAN_V1_22K, pin_duration_head, synth_hifigan_state, synth_texts
    from jatts_amd.vocoder import Vocoder
    z, keys = load_golden("fs2_bench768.npz")
    m = FastSpeech2(idim=45, **FS2_JSUT)
    m.load_state_dict(pin_duration_head(golden_state(keys, 0), 6))
    m = m.to(cuda).set_precision("fp32")
    ones, zeros = [1.0] * 80, [0.0] * 80
    vsd = synth_hifigan_state(HIFIGAN_V1_22K, 0)
    voc = Vocoder(vsd, {"sampling_rate": 22050, "generator_type": "HiFiGANGenerator", "generator_params": HIFIGAN_V1_22K},
                  {"mean": zeros, "scale": ones}, cuda, trg_stats={"mean": zeros, "scale": ones})
    voc.set_precision("fp32")
    texts = [t.to(cuda) for t in synth_texts(64, 128, 45, seed=1)]      # bench.py's batch (rank 0)
    return z, m, voc, vsd, texts


VOC_PARAMS = {"22k": ("HIFIGAN_V1_22K", 22050, 256), "24k": ("HIFIGAN_V1_24K", 24000, 300)}
_VOCS = {}


def _vocoder(bench_stack, sr):
    """(Vocoder, its state dict, generator params, hop) of BASELINE's 22.05 kHz / hop-256 generator (the fixture's) or of the 24 kHz / hop-300 one the
    JSUT / JVS recipes load (scales 5,5,4,3, kernels 10,10,8,6: conf/fastspeech2.v1.yaml:4-6,96-99), full width, synthetic weights."""
    from jatts_amd import synthetic
    from jatts_amd.vocoder import Vocoder
    z, m, voc, vsd, texts = bench_stack
    name, rate, hop = VOC_PARAMS[sr]
    params = getattr(synthetic, name)
    if sr == "22k":
        return voc, vsd, params, hop
    if sr not in _VOCS:
        ones, zeros = [1.0] * 80, [0.0] * 80
        sd = synthetic.synth_hifigan_state(params, 0)
        v = Vocoder(sd, {"sampling_rate": rate, "generator_type": "HiFiGANGenerator", "generator_params": params},
                    {"mean": zeros, "scale": ones}, texts[0].device, trg_stats={"mean": zeros, "scale": ones})
        _VOCS[sr] = (v.set_precision("fp32"), sd)
    assert _VOCS[sr][0].model.hop == hop
    return _VOCS[sr][0], _VOCS[sr][1], params, hop


def test_fs2_bench_utterances_match_the_reference(bench_stack):
    """f32 mel of a 128-phoneme / 768-frame bench utterance vs the real reference (abs 2e-3 on values up to 4.8), computed alone and
    as part of the 64-utterance batch bench.py times; integer outputs exact."""
    z, m, voc, vsd, texts = bench_stack
    utts = [int(u) for u in z["utts"]]
    rb = m.inference_batch(texts)
    assert rb["olens"] == [768] * 64
    for j, u in enumerate(utts):
        assert torch.equal(texts[u].cpu(), torch.tensor(z[f"u{j}_text"]))
        ref = torch.tensor(z[f"u{j}_feat_gen"])
        r1 = m.inference_batch([texts[u]])
        assert torch.equal(r1["duration"].cpu(), torch.tensor(z[f"u{j}_duration"]))
        e1 = maxdiff(r1["feat_gen"], ref)
        eb = maxdiff(rb["feat_gen"][768 * u:768 * (u + 1)], ref)
        assert e1 <= 2e-3 and eb <= 2e-3, f"bench utterance {u}: alone {e1:.3e}, in the batch {eb:.3e}"
        assert maxdiff(r1["pitch"].reshape(-1), z[f"u{j}_pitch"].reshape(-1)) <= 2e-3
        assert maxdiff(r1["energy"].reshape(-1), z[f"u{j}_energy"].reshape(-1)) <= 2e-3
        assert torch.equal(rb["duration"][128 * u:128 * (u + 1)].cpu(), torch.tensor(z[f"u{j}_duration"]))


@pytest.mark.parametrize("mode", ["fp32_split", "fp32_bf16x3"])
def test_fs2_split_mode_bench_utterances_match_the_reference(bench_stack, mode):
    """set_precision("fp32_split") on the acoustic model (round 4: every conv but the duration predictor's on split f16 hi/lo MFMA operands;
    round 5, "fp32_bf16x3": the same convs on three exact bf16 terms per operand, seven MFMA products)
    against the SAME real-reference golden and tolerance as the exact-f32 path (abs 2e-3, durations exact), alone and inside the batch of 64
    (bit-identical to each other), with its error next to the exact-f32 path's: at most twice as far from the reference."""
    import json
    import os
    z, m, voc, vsd, texts = bench_stack
    utts = [int(u) for u in z["utts"]]
    rec = {}
    try:
        m.set_precision(mode)
        rb = m.inference_batch(texts)
        assert rb["olens"] == [768] * 64
        for j, u in enumerate(utts):
            ref = torch.tensor(z[f"u{j}_feat_gen"])
            m.set_precision(mode)
            r1 = m.inference_batch([texts[u]])
            m.set_precision("fp32")
            rf = m.inference_batch([texts[u]])
            assert torch.equal(r1["duration"].cpu(), torch.tensor(z[f"u{j}_duration"]))
            assert torch.equal(r1["feat_gen"], rb["feat_gen"][768 * u:768 * (u + 1)]), "split mode: utterance alone != inside the batch"
            es, ef = maxdiff(r1["feat_gen"], ref), maxdiff(rf["feat_gen"], ref)
            rec[f"utt{u}"] = {"split_vs_reference_max": es, "exact_f32_vs_reference_max": ef, "split_vs_exact_f32_max": maxdiff(r1["feat_gen"], rf["feat_gen"]),
                              "mel_abs_max": float(ref.abs().max())}
            assert es <= 2e-3, f"bench utterance {u}: split {es:.3e}"
            assert es <= 2.0 * ef + 1e-6, f"bench utterance {u}: split {es:.3e} vs exact f32 {ef:.3e} against the reference"
            assert maxdiff(r1["pitch"].reshape(-1), z[f"u{j}_pitch"].reshape(-1)) <= 2e-3
            assert maxdiff(r1["energy"].reshape(-1), z[f"u{j}_energy"].reshape(-1)) <= 2e-3
    finally:
        m.set_precision("fp32")
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, f"r06_{mode}_errors_fs2.json"), "w") as f:
        json.dump(rec, f, indent=1)


@pytest.mark.parametrize("sr", ["22k", "24k"])
def test_hifigan_bench_size_matches_the_oracle(bench_stack, sr):
    """HiFi-GAN v1 (22.05 kHz, hop 256 / 24 kHz, hop 300; 512 channels) on a 768-frame mel: 196 608 / 230 400 samples vs oracle.hifigan_generate
    (f32, abs 2e-4 on a signal in [-1, 1]), alone and as utterances 0 / 37 inside the batch of 64 the bench times."""
    from jatts_amd import hip
    from oracle.hifigan_oracle import hifigan_generate
    z, m, _, _, texts = bench_stack
    voc, vsd, HIFIGAN_V1_22K, hop = _vocoder(bench_stack, sr)      # (the generator under test; the name is kept for the lines below)
    utts = [int(u) for u in z["utts"]]
    dev = texts[0].device
    rbatch = m.inference_batch(texts)
    mel_b = rbatch["feat_gen"].clone()
    refs = {}
    for j, u in enumerate(utts):                              # the golden (reference) mel takes the two utterances' places
        mel_b[768 * u:768 * (u + 1)] = torch.tensor(z[f"u{j}_feat_gen"]).to(dev)
        torch.set_num_threads(min(32, torch.get_num_threads()))
        with torch.no_grad():
            refs[u] = hifigan_generate(vsd, torch.tensor(z[f"u{j}_feat_gen"]), HIFIGAN_V1_22K["upsample_scales"],
                                       HIFIGAN_V1_22K["resblock_dilations"])
        assert refs[u].numel() == 768 * hop and float(refs[u].abs().max()) > 1e-3
    yb = voc.decode_batch(rbatch["feats_rb"], mel_b)
    assert yb.numel() == 64 * 768 * hop
    for j, u in enumerate(utts):
        y1 = voc.decode_batch(hip.RaggedBatch([768], dev), torch.tensor(z[f"u{j}_feat_gen"]).to(dev))
        e1 = maxdiff(y1.reshape(-1), refs[u].reshape(-1))
        eb = maxdiff(yb[768 * hop * u:768 * hop * (u + 1)].reshape(-1), refs[u].reshape(-1))
        assert e1 <= 2e-4 and eb <= 2e-4, f"{sr} utterance {u}: alone {e1:.3e}, in the batch {eb:.3e}"


@pytest.mark.parametrize("sr", ["22k", "24k"])
@pytest.mark.parametrize("mode", ["fp32_split", "fp32_bf16x3"])
def test_hifigan_split_mode_at_bench_size(bench_stack, mode, sr):
    """The fp32_split vocoder (round 4: ResBlock units on split f16 hi/lo MFMA operands; round 5, "fp32_bf16x3": on three exact bf16 terms per
    operand, seven MFMA products) on the same 768-frame mels: the SAME tolerance as
    the exact-f32 path against the f32 oracle (abs 2e-4), and against the oracle run in FP64 its maximum error must not exceed twice the
    exact-f32 path's -- the condition under which it is not a narrower arithmetic than the reference's.  Alone and inside the batch of 64,
    bit-identical to each other.  The measured errors go to gpurun_out/r04_split_errors.json (profiles/r04_notes.md quotes them)."""
    import json
    import os
    from jatts_amd import hip
    from oracle.hifigan_oracle import hifigan_generate
    z, m, _, _, texts = bench_stack
    voc, vsd, HIFIGAN_V1_22K, hop = _vocoder(bench_stack, sr)      # (the generator under test; the name is kept for the lines below)
    utts = [int(u) for u in z["utts"]]
    dev = texts[0].device
    rbatch = m.inference_batch(texts)
    mel_b = rbatch["feat_gen"].clone()
    for j, u in enumerate(utts):
        mel_b[768 * u:768 * (u + 1)] = torch.tensor(z[f"u{j}_feat_gen"]).to(dev)
    vsd64 = {k: v.double() for k, v in vsd.items()}
    rec = {}
    try:
        voc.set_precision(mode)
        yb = voc.decode_batch(rbatch["feats_rb"], mel_b)
        for j, u in enumerate(utts):
            mel = torch.tensor(z[f"u{j}_feat_gen"])
            torch.set_num_threads(min(32, torch.get_num_threads()))
            with torch.no_grad():
                ref32 = hifigan_generate(vsd, mel, HIFIGAN_V1_22K["upsample_scales"], HIFIGAN_V1_22K["resblock_dilations"]).reshape(-1)
                ref64 = hifigan_generate(vsd64, mel.double(), HIFIGAN_V1_22K["upsample_scales"], HIFIGAN_V1_22K["resblock_dilations"]).reshape(-1)
            voc.set_precision(mode)
            ys = voc.decode_batch(hip.RaggedBatch([768], dev), mel.to(dev)).reshape(-1)
            voc.set_precision("fp32")
            yf = voc.decode_batch(hip.RaggedBatch([768], dev), mel.to(dev)).reshape(-1)
            assert torch.equal(ys, yb[768 * hop * u:768 * hop * (u + 1)].reshape(-1)), "split mode: utterance alone != inside the batch"
            e32 = maxdiff(ys, ref32)
            es64, ef64 = float((ys.double().cpu() - ref64).abs().max()), float((yf.double().cpu() - ref64).abs().max())
            rs64 = float((ys.double().cpu() - ref64).pow(2).mean().sqrt())
            rf64 = float((yf.double().cpu() - ref64).pow(2).mean().sqrt())
            rec[f"utt{u}"] = {"split_vs_f32_oracle_max": e32, "split_vs_fp64_max": es64, "exact_f32_vs_fp64_max": ef64,
                              "split_vs_fp64_rms": rs64, "exact_f32_vs_fp64_rms": rf64, "f32_oracle_vs_fp64_max": float((ref32.double() - ref64).abs().max()),
                              "wave_abs_max": float(ref64.abs().max())}
            assert e32 <= 2e-4, f"utterance {u}: split vs the f32 oracle {e32:.3e}"
            assert es64 <= 2.0 * ef64 + 1e-7, f"utterance {u}: split {es64:.3e} vs exact f32 {ef64:.3e} against fp64"
    finally:
        voc.set_precision("fp32")
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, f"r06_{mode}_errors{'' if sr == '22k' else '_24k'}.json"), "w") as f:
        json.dump(rec, f, indent=1)


# ------------------------------------------------------------------------------------------------------------ f32 conv, every variant
def _ref_conv64(x, w, b, lens, dil, pad, k):
    outs, o = [], 0
    for L in lens:
        xs = F.pad(x[o:o + L].t().unsqueeze(0).double(), (pad, (k - 1) * dil - pad))
        outs.append(F.conv1d(xs, w.double(), b.double(), dilation=dil)[0].t())
        o += L
    return torch.cat(outs)


BIG = [  # c_in, n_out, k, dil, lens, act, resid     (128 x 128 tiles: ceil(L / 128) per sequence x ceil(n_out / 128) > 600 workgroups)
    (384, 1536, 3, 1, [1024] * 8, "relu", False),            # FFN w_1: 8 x 8 x 12 = 768
    (1536, 384, 3, 1, [1024] * 26, None, True),              # FFN w_2 with the residual stream: 26 x 8 x 3 = 624
    (384, 384, 1, 1, [768] * 34 + [700], None, True),        # attention / conv-module projections, one ragged sequence: 35 x 6 x 3 = 630
    (512, 2048, 1, 1, [768] * 7, None, False),               # Matcha U-Net FFN: 7 x 6 x 16 = 672
    (192, 768, 1, 1, [640, 513, 768, 31] * 5, None, False),  # ragged lengths, c_in = 3 chunks
    (256, 256, 5, 1, [1500, 77, 2048], "tanh", False),       # postnet-like, small launch (the 128 x 64 tile under the heuristic)
    (128, 136, 3, 2, [300, 1000], None, False),              # n_out not a multiple of 32, dilation 2
]
VARIANTS = [0, 1, 2, 3, 4, 5]


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("case", BIG, ids=[f"{c[0]}to{c[1]}k{c[2]}" for c in BIG])
def test_conv1d_f32_every_variant_at_bench_size(cuda, lib, case, variant):
    from jatts_amd import hip
    c_in, n_out, k, dil, lens, act, resid = case
    g = torch.Generator().manual_seed(c_in * 7 + n_out)
    R = sum(lens)
    x = torch.randn(R, c_in, generator=g)
    w = torch.randn(n_out, c_in, k, generator=g) / math.sqrt(c_in * k)
    b = torch.randn(n_out, generator=g)
    res = torch.randn(R, n_out, generator=g) if resid else None
    pad = (k - 1) // 2 * dil
    ref = _ref_conv64(x, w, b, lens, dil, pad, k)
    ref = {"relu": torch.relu, "tanh": torch.tanh, None: lambda t: t}[act](ref)
    alpha = 0.5 if resid else 1.0
    ref = ref * alpha + (res.double() if resid else 0)
    rb = hip.RaggedBatch(lens, cuda)
    wp = hip.pack_conv_weight(w.to(cuda), hip.F32)
    y = hip.conv1d(rb, x.to(cuda), wp, c_in, n_out, k, dtype=hip.F32, dil=dil, bias=b.to(cuda),
                   act={"relu": hip.ACT_RELU, "tanh": hip.ACT_TANH, None: hip.ACT_NONE}[act], alpha=alpha,
                   resid=None if res is None else res.to(cuda), out_f32=True, variant=variant)
    e = relerr(y, ref)
    assert e <= 3e-6, f"conv1d f32 {case[:4]} variant {variant}: rel err {e:.3e}"
    assert maxdiff(y, ref) <= 5e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("variant", [0, 3, 4, 5])
def test_conv1d_f32_reflect_padding_and_transposed_output(cuda, lib, variant):
    """Reflect padding (ECAPA-TDNN convs: the register-streamed kernels hand it to the LDS-staged one) and the transposed V^T output
    (served by the register-streamed kernels through the generic fragment-order epilogue)."""
    from jatts_amd import hip
    g = torch.Generator().manual_seed(variant)
    lens, c_in, n_out, k, dil = [97, 300, 41], 128, 192, 5, 2
    R = sum(lens)
    x = torch.randn(R, c_in, generator=g)
    w = torch.randn(n_out, c_in, k, generator=g) / math.sqrt(c_in * k)
    b = torch.randn(n_out, generator=g)
    pad = (k - 1) // 2 * dil
    outs, o = [], 0
    for L in lens:
        xs = F.pad(x[o:o + L].t().unsqueeze(0).double(), (pad, pad), mode="reflect")
        outs.append(F.conv1d(xs, w.double(), b.double(), dilation=dil)[0].t())
        o += L
    ref = torch.cat(outs)
    rb = hip.RaggedBatch(lens, cuda)
    wp = hip.pack_conv_weight(w.to(cuda), hip.F32)
    y = hip.conv1d(rb, x.to(cuda), wp, c_in, n_out, k, dtype=hip.F32, dil=dil, bias=b.to(cuda), reflect=True, variant=variant)
    assert relerr(y, ref) <= 3e-6
    ref0 = _ref_conv64(x, w, b, lens, dil, pad, k)
    yt = hip.conv1d(rb, x.to(cuda), wp, c_in, n_out, k, dtype=hip.F32, dil=dil, bias=b.to(cuda), transposed=True, variant=variant)
    assert relerr(yt.t(), ref0) <= 3e-6


@pytest.mark.parametrize("variant", [0, 2, 3])
@pytest.mark.parametrize("shape", [(256, 1024, 3, [2048] * 6 + [1000]), (128, 128, 3, [4096, 777]), (512, 2048, 1, [768] * 7)])
def test_conv1d_f32_lrelu_prologue_at_bench_size(cuda, lib, shape, variant):
    """The HiFi-GAN upsampling convs' form (LeakyReLU(0.1) on the input, polyphase ConvTranspose as a stride-1 conv): register-streamed kernel with
    the prologue on the B fragments (variant 3 / the product heuristic at these sizes) and the LDS-staged kernel (variant 2) vs fp64."""
    from jatts_amd import hip
    c_in, n_out, k, lens = shape
    g = torch.Generator().manual_seed(c_in + n_out)
    R = sum(lens)
    x = torch.randn(R, c_in, generator=g)
    w = torch.randn(n_out, c_in, k, generator=g) / math.sqrt(c_in * k)
    b = torch.randn(n_out, generator=g)
    ref = _ref_conv64(F.leaky_relu(x.double(), 0.1), w, b, lens, 1, (k - 1) // 2, k)
    rb = hip.RaggedBatch(lens, cuda)
    y = hip.conv1d(rb, x.to(cuda), hip.pack_conv_weight(w.to(cuda), hip.F32), c_in, n_out, k, dtype=hip.F32, bias=b.to(cuda), pre_lrelu=0.1,
                   out_f32=True, variant=variant)
    assert relerr(y, ref) <= 3e-6, f"{shape[:3]} variant {variant}: {relerr(y, ref):.3e}"


EDGE = [  # c_in (window of a wider row), n_out, k, dil, lens, x_col0, ldx, act, resid_ld
    (64, 72, 3, 1, [1, 2, 5, 129, 64], 0, 64, "relu", None),          # one-row sequences, n_out = 72 (multiple of 8, not of 32)
    (128, 384, 1, 1, [127, 1, 130], 64, 256, None, 512),               # column window of a wider matrix, residual with its own row stride
    (192, 200, 5, 3, [33, 400, 7], 0, 192, "tanh", None),              # dilation 3 with sequences shorter than the receptive field
    (64, 1000, 1, 1, [300], 0, 64, "swish", None),                     # wide output, one sequence
    (256, 136, 7, 1, [50, 3, 200], 128, 384, None, None),              # n_out % 8 == 0 but % 32 != 0, k = 7, shifted window
]


@pytest.mark.parametrize("variant", [0, 3, 5])
@pytest.mark.parametrize("case", EDGE, ids=[f"{c[0]}to{c[1]}k{c[2]}d{c[3]}" for c in EDGE])
def test_conv1d_direct_edge_geometry(cuda, lib, case, variant):
    """The register-streamed kernel's addressing corners: zero padding by descriptor range check at both sequence ends, sequences
    shorter than a tile / than the receptive field, column windows of wider rows (x_col0, ldx > c_in), residual rows with their own
    stride, partial last n-fragments, every fused activation -- against fp64."""
    from jatts_amd import hip
    c_in, n_out, k, dil, lens, x_col0, ldx, act, rld = case
    g = torch.Generator().manual_seed(c_in + n_out + k)
    R = sum(lens)
    xw = torch.randn(R, ldx, generator=g)
    x = xw[:, x_col0:x_col0 + c_in]
    w = torch.randn(n_out, c_in, k, generator=g) / math.sqrt(c_in * k)
    b = torch.randn(n_out, generator=g)
    pad = (k - 1) // 2 * dil
    ref = _ref_conv64(x, w, b, lens, dil, pad, k)
    ref = {"relu": torch.relu, "tanh": torch.tanh, "swish": lambda t: t * torch.sigmoid(t), None: lambda t: t}[act](ref)
    resw = torch.randn(R, rld, generator=g) if rld else None
    alpha = 0.5 if rld else 1.0
    if rld:
        ref = ref * alpha + resw[:, 8:8 + n_out].double()
    rb = hip.RaggedBatch(lens, cuda)
    wp = hip.pack_conv_weight(w.to(cuda), hip.F32)
    resd = resw.to(cuda) if rld else None
    y = hip.conv1d(rb, xw.to(cuda), wp, c_in, n_out, k, dtype=hip.F32, dil=dil, bias=b.to(cuda), ldx=ldx, x_col0=x_col0,
                   act={"relu": hip.ACT_RELU, "tanh": hip.ACT_TANH, "swish": hip.ACT_SWISH, None: hip.ACT_NONE}[act], alpha=alpha,
                   resid=resd, resid_col0=8 if rld else 0, out_f32=True, variant=variant)
    assert torch.isfinite(y).all()
    assert relerr(y, ref) <= 5e-6, f"{case} variant {variant}: rel err {relerr(y, ref):.3e}"


# ------------------------------------------------------------------------------------------------ configs 3 and 5 at the bench's length
def _record_error(name, prec, e_alone, e_batch):
    """max |mel - reference| per arithmetic -> gpurun_out/r06_model_errors.json (profiles/r06_notes.md quotes the table)."""
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "r06_model_errors.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    rec = json.load(open(path)) if os.path.exists(path) else {}
    rec.setdefault(name, {})[prec] = {"alone": e_alone, "in_batch_of_8": e_batch}
    with open(path, "w") as f:
        json.dump(rec, f, indent=1)


def _seeded_noise(z):
    shape = [int(v) for v in z["noise_shape"]]
    # the reference drew randn_like(x) with x of shape (1, C, T); the fixture records (T, C) = noise[0].t()
    return torch.randn(1, shape[1], shape[0], generator=torch.Generator().manual_seed(int(z["noise_seed"])))[0].t().contiguous()


@pytest.mark.parametrize("prec,atol", [("fp32", 5e-3), ("fp32_split", 5e-3), ("fp32_bf16x3", 5e-3), ("fp16", 0.15)])     # split / emulated modes: the exact-f32 tolerance
def test_matcha_bench_utterance_matches_the_reference(cuda, lib, prec, atol):
    """BASELINE configs[2] at the bench's utterance length: the config-3 model (U-Net 512/512, head dim 256, 10 Euler steps) on a
    128-phoneme bench utterance -> 768 frames, against the REAL reference (matcha_bench128.npz; diffusers attention = SDPA stand-in),
    alone and as one of 8 utterances of a batch."""
    from jatts_amd.models import MatchaTTS_MAS
    from jatts_amd.synthetic import MATCHA_MAS_JSUT, matcha_golden_tweaks, pin_duration_head, synth_texts
    z, keys = load_golden("matcha_bench128.npz")
    m = MatchaTTS_MAS(idim=45, **MATCHA_MAS_JSUT)
    m.load_state_dict(pin_duration_head(matcha_golden_tweaks(golden_state(keys, 0)), 6))
    m = m.to(cuda).set_precision(prec)
    text = torch.tensor(z["u0_text"]).to(cuda)
    assert torch.equal(text.cpu(), synth_texts(64, 128, 45, seed=1)[5])
    noise = _seeded_noise(z)
    ref = z["u0_feat_gen"]
    r = m.inference_batch([text], n_timesteps=10, temperature=0.667, noise=[noise])
    assert torch.equal(r["duration"].cpu(), torch.tensor(z["u0_duration"])) and r["feat_gen"].shape == ref.shape == (768, 80)
    e1 = maxdiff(r["feat_gen"], ref)
    others = [t.to(cuda) for t in synth_texts(64, 128, 45, seed=1)[8:15]]
    g = torch.Generator().manual_seed(77)
    rb = m.inference_batch(others[:3] + [text] + others[3:], n_timesteps=10, temperature=0.667,
                           noise=[torch.randn(768, 80, generator=g) for _ in range(3)] + [noise] + [torch.randn(768, 80, generator=g) for _ in range(4)])
    eb = maxdiff(rb["feat_gen"][3 * 768:4 * 768], ref)
    _record_error("matcha_bench128", prec, e1, eb)
    assert e1 <= atol and eb <= atol, f"{prec}: alone {e1:.3e}, in a batch {eb:.3e}"


@pytest.mark.parametrize("prec,atol", [("fp32", 3e-3), ("fp32_split", 3e-3), ("fp32_bf16x3", 3e-3), ("fp16", 8e-2)])     # split / emulated modes: the exact-f32 tolerance
def test_vits_bench_utterance_matches_the_reference(cuda, lib, prec, atol):
    """BASELINE configs[4] at the bench's utterance length: mel-VITS with a 192-d speaker embedding on a 128-phoneme bench utterance ->
    768 frames, against the REAL reference (vits_bench128.npz), alone and inside a batch of 8."""
    from jatts_amd.models import VITS
    from jatts_amd.synthetic import VITS_JSUT, pin_duration_head, synth_texts
    z, keys = load_golden("vits_bench128.npz")
    m = VITS(idim=45, spk_embed_dim=192, **VITS_JSUT)
    m.load_state_dict(pin_duration_head(golden_state(keys, 0), 6))
    m = m.to(cuda).set_precision(prec)
    text = torch.tensor(z["u0_text"]).to(cuda)
    spk = torch.tensor(z["u0_spemb"])
    noise = _seeded_noise(z)
    ref = z["u0_feat_gen"]
    r = m.inference_batch([text], spk.unsqueeze(0), noise=[noise])
    assert torch.equal(r["duration"].cpu(), torch.tensor(z["u0_duration"])) and r["feat_gen"].shape == ref.shape == (768, 80)
    e1 = maxdiff(r["feat_gen"], ref)
    others = [t.to(cuda) for t in synth_texts(64, 128, 45, seed=1)[8:15]]
    g = torch.Generator().manual_seed(78)
    spks = torch.cat([torch.randn(3, 192, generator=g), spk.unsqueeze(0), torch.randn(4, 192, generator=g)])
    rb = m.inference_batch(others[:3] + [text] + others[3:], spks,
                           noise=[torch.randn(768, 384, generator=g) for _ in range(3)] + [noise] + [torch.randn(768, 384, generator=g) for _ in range(4)])
    eb = maxdiff(rb["feat_gen"][3 * 768:4 * 768], ref)
    _record_error("vits_bench128", prec, e1, eb)
    assert e1 <= atol and eb <= atol, f"{prec}: alone {e1:.3e}, in a batch {eb:.3e}"


@pytest.mark.parametrize("kind", ["fs2", "matcha", "matcha_mas", "vits"])
def test_graph_mode_matches_eager_at_the_recipe_batch(cuda, lib, kind):
    """The captured training step against the eager one AT THE BENCH'S BATCH (32 x 128 phonemes x 6 frames, the recipes' models): the
    small-config graph tests cannot see reductions that only go multi-block at this size -- torch's strided / full sums keep block
    semaphores zeroed with a memset that did not replay inside a captured step (pos_bias_u / v, the alignment module's text side and the
    KL term came back as 1e38; jatts_amd/autograd.py AddBias / SumAll / AlignLogProb are the own-kernel replacements).  Same draws,
    dropout on: the clipped gradient norm and every loss agree step by step, and no gradient entry is out of range."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    g = torch.Generator().manual_seed(9)
    trainers = []
    for graph in (False, True):
        m, b, cls, extra, _, _ = bench.train_setup(cuda, kind)
        trainers.append(cls(m, lr=1e-4, grad_norm=1.0, warmup_steps=0, capture_graph=graph, **extra))
    B, To = b["ys"].shape[0], b["ys"].shape[1]
    if kind in ("matcha", "matcha_mas"):
        b["cfm_t"], b["cfm_noise"] = torch.rand(B, generator=g), torch.randn(B, To, 80, generator=g)
    if kind == "vits":
        b["post_noise"] = torch.randn(B, To, trainers[0].model.adim, generator=g)
    a, c = trainers
    for step in range(5):
        la, lc = a.train_step(b), c.train_step(b)
        assert set(la) == set(lc)
        assert float(c.flat_g.abs().max()) < 1e4, (kind, step, float(c.flat_g.abs().max()))
        for k in la:
            tol = 2e-4 * (1 + 2 * step)
            assert abs(float(la[k]) - float(lc[k])) <= tol * max(1.0, abs(float(la[k]))), (kind, step, k, float(la[k]), float(lc[k]))
        # gradients and parameters too (round 4: every reduction of the step is fixed-order, so eager and replay run the same arithmetic;
        # what is left is the torch / rocBLAS side of the captured step): relative L2, growing with Adam's amplification
        rg = float((a.flat_g - c.flat_g).norm() / a.flat_g.norm().clamp_min(1e-30))
        rp = float((a.flat_p - c.flat_p).norm() / a.flat_p.norm().clamp_min(1e-30))
        assert rg <= 1e-5 * (1 + 4 * step) and rp <= 1e-5 * (1 + 4 * step), (kind, step, rg, rp)
    assert any(st["graph"] is not None for st in c._graphs.values())


@pytest.mark.parametrize("kind", ["fs2", "matcha", "matcha_mas", "vits", "fs2:fp32_split", "vits:fp32_split"])
def test_two_eager_trainers_are_bit_identical_at_the_recipe_batch(cuda, lib, kind):
    """Run-to-run reproducibility (the reference is bit-identical run to run): two trainers built from the same seed and fed the
    same batch and the same draws produce bit-identical losses, gradients and parameters step by step AT THE RECIPE BATCH, where every
    parameter-gradient reduction spans many workgroups.  Round 3 added those partial sums with f32 atomics (arrival order); round 4 adds
    them in a fixed order (csrc/det_reduce.h)."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    g = torch.Generator().manual_seed(9)
    kind, _, precision = kind.partition(":")
    trainers = []
    for _ in range(2):
        m, b, cls, extra, _, _ = bench.train_setup(cuda, kind)
        trainers.append(cls(m, lr=1e-4, grad_norm=1.0, warmup_steps=0, capture_graph=False, precision=precision or "fp32", **extra))
    B, To = b["ys"].shape[0], b["ys"].shape[1]
    if kind in ("matcha", "matcha_mas"):
        b["cfm_t"], b["cfm_noise"] = torch.rand(B, generator=g), torch.randn(B, To, 80, generator=g)
    if kind == "vits":
        b["post_noise"] = torch.randn(B, To, trainers[0].model.adim, generator=g)
    a, c = trainers
    for step in range(4):
        la, lc = a.train_step(b), c.train_step(b)
        for k in la:
            assert float(la[k]) == float(lc[k]), (kind, step, k, float(la[k]), float(lc[k]))
        assert torch.equal(a.flat_g, c.flat_g), (kind, step, float((a.flat_g - c.flat_g).abs().max()))
        assert torch.equal(a.flat_p, c.flat_p), (kind, step)
