"""Full-size (BASELINE.json configs[1]: 64 utterances x 128 phonemes x 6 frames, FastSpeech2-JSUT + HiFi-GAN v1) checks
through size-independent properties -- the oracle cannot run this size in seconds:
  * utterance independence (SURVEY 8a note N1: the parity target is the B=1 path): every utterance of the packed batch is
    BIT-identical to the same utterance synthesised alone, in any batch order, ragged or not;
  * determinism: two runs of the same batch are bit-identical;
  * geometry: samples = sum(durations) * hop per utterance, |y| <= 1, finite;
  * the integer part (durations, length regulator) is exact: pinned duration head -> 6 frames per phoneme.
"""
import pytest
import torch

from helpers import maxdiff

pytestmark = pytest.mark.gpu


def _make_stack(cuda, prec):
    from jatts_amd.models import FastSpeech2
    from jatts_amd.synthetic import FS2_JSUT, HIFIGAN_V1_22K, pin_duration_head, synth_hifigan_state, synth_state_dict
    from jatts_amd.vocoder import Vocoder
    m = FastSpeech2(idim=45, **FS2_JSUT)
    m.load_state_dict(pin_duration_head(synth_state_dict(m.state_dict(), 0), 6))
    m = m.to(cuda).set_precision(prec)
    ones, zeros = [1.0] * 80, [0.0] * 80
    voc = Vocoder(synth_hifigan_state(HIFIGAN_V1_22K, 0),
                  {"sampling_rate": 22050, "generator_type": "HiFiGANGenerator", "generator_params": HIFIGAN_V1_22K},
                  {"mean": zeros, "scale": ones}, cuda, trg_stats={"mean": zeros, "scale": ones})
    voc.set_precision(prec)
    return m, voc


_VOC24 = {}


def _voc24(cuda, prec):
    """The 24 kHz / hop-300 generator the JSUT / JVS recipes load (scales 5,5,4,3; conf/fastspeech2.v1.yaml:96-99), one per arithmetic."""
    from jatts_amd.synthetic import HIFIGAN_V1_24K, synth_hifigan_state
    from jatts_amd.vocoder import Vocoder
    if prec not in _VOC24:
        ones, zeros = [1.0] * 80, [0.0] * 80
        _VOC24[prec] = Vocoder(synth_hifigan_state(HIFIGAN_V1_24K, 0), {"sampling_rate": 24000, "generator_type": "HiFiGANGenerator", "generator_params": HIFIGAN_V1_24K},
                               {"mean": zeros, "scale": ones}, cuda, trg_stats={"mean": zeros, "scale": ones}).set_precision(prec)
    return _VOC24[prec]


@pytest.fixture(scope="module")
def stack(cuda, lib):
    return _make_stack(cuda, "fp16")


@pytest.fixture(scope="module")
def stack32(cuda, lib):
    """The f32 (headline) arithmetic: its own model / vocoder pair, so the fp16 tests of this module never see a precision switch."""
    return _make_stack(cuda, "fp32")


def _synth(stack, texts):
    m, voc = stack
    r = m.inference_batch(texts)
    y = voc.decode_batch(r["feats_rb"], r["feat_gen"])
    hop = voc.model.hop
    lens = [n * hop for n in r["olens"]]
    outs, o = [], 0
    for n in lens:
        outs.append(y[o:o + n])
        o += n
    assert o == y.numel()
    return r, outs


@pytest.fixture(scope="module")
def stack_split(cuda, lib):
    """fp32_split (round 4): f32 tensors, split f16 hi/lo MFMA operands in every conv and fused unit."""
    return _make_stack(cuda, "fp32_split")


@pytest.fixture(scope="module")
def stack_emul(cuda, lib):
    """fp32_bf16x3 (round 5): f32 tensors, three exact bf16 terms per operand and seven MFMA products in every conv and fused unit."""
    return _make_stack(cuda, "fp32_bf16x3")


@pytest.fixture(scope="module")
def stack_emul6(cuda, lib):
    """fp32_bf16x3_6p: six MFMA products, one accumulator; its k = 1 convs pick their tile by launch size (csrc/conv1d_emul.hip), which must not change a bit."""
    return _make_stack(cuda, "fp32_bf16x3_6p")


@pytest.mark.parametrize("prec", ["fp16", "fp32", "fp32_split", "fp32_bf16x3", "fp32_bf16x3_6p"])
@pytest.mark.parametrize("ragged", [False, True], ids=["64x128", "64xU(64..128)"])
@pytest.mark.parametrize("sr", ["22k", "24k"])
def test_full_batch_properties(cuda, stack, stack32, stack_split, stack_emul, stack_emul6, sr, ragged, prec):
    """fp32_bf16x3 = the arithmetic bench.py's headline measures, fp32 = the reference's own (register-streamed f32 convs, f32 fused units):
    determinism, utterance independence and permutation equivariance hold bit for bit in every arithmetic, through the 22.05 kHz / hop-256 generator
    of BASELINE's metric and the 24 kHz / hop-300 one of the recipes (odd strides 5 and 3 in the polyphase upsampling convs)."""
    from jatts_amd.synthetic import synth_texts
    stack = {"fp32": stack32, "fp32_split": stack_split, "fp32_bf16x3": stack_emul, "fp32_bf16x3_6p": stack_emul6, "fp16": stack}[prec]
    if sr == "24k":
        stack = (stack[0], _voc24(cuda, prec))
    hop = stack[1].model.hop
    assert hop == (256 if sr == "22k" else 300)
    texts = [t.to(cuda) for t in synth_texts(64, 128, 45, seed=1)]
    if ragged:
        g = torch.Generator().manual_seed(5)
        texts = [t[: int(torch.randint(64, 129, (1,), generator=g))] for t in texts]
    r, outs = _synth(stack, texts)
    # integer path: exact
    assert torch.equal(r["duration"].cpu(), torch.full((sum(len(t) for t in texts),), 6, dtype=torch.int64))
    assert r["olens"] == [6 * len(t) for t in texts]
    for t, y in zip(texts, outs):
        assert y.numel() == 6 * len(t) * hop
    ycat = torch.cat(outs)
    assert torch.isfinite(ycat).all() and float(ycat.abs().max()) <= 1.0 and float(ycat.abs().max()) > 0.0
    # determinism
    _, outs2 = _synth(stack, texts)
    assert all(torch.equal(a, b) for a, b in zip(outs, outs2))
    # utterance independence: alone == inside the batch, bit for bit
    for i in (0, 17, 63):
        _, single = _synth(stack, [texts[i]])
        assert torch.equal(single[0], outs[i]), f"utterance {i} depends on its batch neighbours"
    # permutation equivariance
    perm = torch.randperm(64, generator=torch.Generator().manual_seed(3)).tolist()
    _, outs_p = _synth(stack, [texts[i] for i in perm])
    assert all(torch.equal(outs_p[j], outs[i]) for j, i in enumerate(perm))


def test_zero_length_sequences_are_skipped(cuda, lib):
    """Ragged batches may contain empty sequences (an utterance whose durations round to zero frames)."""
    from jatts_amd import hip
    g = torch.Generator().manual_seed(0)
    lens, C = [0, 37, 0, 130, 0], 128
    R = sum(lens)
    rb = hip.RaggedBatch(lens, cuda)
    x = (torch.randn(R, C, generator=g) * 0.3).to(cuda).half()
    w = [hip.pack_conv_weight((torch.randn(C, C, 3, generator=g) / (3 * C) ** 0.5).to(cuda), hip.F16, 32) for _ in range(2)]
    b = torch.zeros(C, device=cuda)
    y = torch.full_like(x, float("nan"))
    hip.hifigan_resunit(rb, 1, x, y, w[0], b, w[1], b, C, 3, 1, 0.1, hip.F16)
    assert torch.isfinite(y).all()
    rb2 = hip.RaggedBatch([37, 130], cuda)
    y2 = torch.empty_like(x)
    hip.hifigan_resunit(rb2, 1, x, y2, w[0], b, w[1], b, C, 3, 1, 0.1, hip.F16)
    assert torch.equal(y, y2)
    wc = hip.pack_conv_weight((torch.randn(64, C, 3, generator=g) / (3 * C) ** 0.5).to(cuda), hip.F16)
    o1 = hip.conv1d(rb, x, wc, C, 64, 3, dtype=hip.F16)
    o2 = hip.conv1d(rb2, x, wc, C, 64, 3, dtype=hip.F16)
    assert torch.equal(o1, o2)
    H, dk = 2, 64
    vt = x.t().contiguous()   # unaligned packed V^T (vt_col0 = NULL): scalar staging path
    a1 = hip.relpos_attention(rb, x, C, x, C, vt, R, None, 0, None, 0.125, H, dk, hip.F16, rel_mode=0)
    a2 = hip.relpos_attention(rb2, x, C, x, C, vt, R, None, 0, None, 0.125, H, dk, hip.F16, rel_mode=0)
    assert torch.equal(a1, a2) and torch.isfinite(a1).all()


@pytest.mark.parametrize("mode", ["F32", "F32E", "F32E6"])
def test_zero_and_one_row_sequences_f32_and_emulated(cuda, lib, mode):
    """Empty sequences and one-row sequences in the exact-f32 and emulated unit / conv kernels (1-D ragged grids: a zero-length sequence owns no
    tile; a one-row sequence is all halo): finite, identical to the batch without the empty ones."""
    from jatts_amd import hip
    dt = getattr(hip, mode)
    pack = (lambda w, cm: hip.pack_conv_weight_bf16x3(w, cm)) if dt in hip.EMUL else (lambda w, cm: hip.pack_conv_weight(w, hip.F32, cm))
    g = torch.Generator().manual_seed(1)
    lens, C = [0, 1, 37, 0, 1, 130, 0], 128
    R = sum(lens)
    rb, rb2 = hip.RaggedBatch(lens, cuda), hip.RaggedBatch([1, 37, 1, 130], cuda)
    x = (torch.randn(R, C, generator=g) * 0.3).to(cuda)
    w = [pack((torch.randn(C, C, 7, generator=g) / (7 * C) ** 0.5).to(cuda), 32) for _ in range(2)]
    b = torch.randn(C, generator=g).to(cuda) * 0.1
    y, y2 = torch.full_like(x, float("nan")), torch.full_like(x, float("nan"))
    hip.hifigan_resunit(rb, 1, x, y, w[0], b, w[1], b, C, 7, 3, 0.1, dt)
    hip.hifigan_resunit(rb2, 1, x, y2, w[0], b, w[1], b, C, 7, 3, 0.1, dt)
    assert torch.isfinite(y).all() and torch.equal(y, y2)
    wc = pack((torch.randn(192, C, 3, generator=g) / (3 * C) ** 0.5).to(cuda), 64)
    o1 = hip.conv1d(rb, x, wc, C, 192, 3, dtype=dt, bias=torch.zeros(192, device=cuda))
    o2 = hip.conv1d(rb2, x, wc, C, 192, 3, dtype=dt, bias=torch.zeros(192, device=cuda))
    assert torch.isfinite(o1).all() and torch.equal(o1, o2)


def test_two_stream_pipeline_is_bit_identical(cuda, stack):
    """jatts_amd.pipeline.Stage4Pipeline (text2mel of batch k+1 overlapping the vocoder of batch k) == sequential loop."""
    from jatts_amd.pipeline import Stage4Pipeline
    from jatts_amd.synthetic import synth_texts
    m, voc = stack
    batches = [[t.to(cuda) for t in synth_texts(16, 64 + 16 * i, 45, seed=10 + i)] for i in range(4)]
    want = []
    for b in batches:
        r = m.inference_batch(b)
        want.append(voc.decode_batch(r["feats_rb"], r["feat_gen"]).clone())
    torch.cuda.synchronize()
    got = [y.clone() for _, y in Stage4Pipeline(m, voc).run(batches)]
    torch.cuda.synchronize()
    assert len(got) == len(want) and all(torch.equal(a, b) for a, b in zip(got, want))


# ------------------------------------------------------------------ f16 fast mode against the f32 path at BASELINE sizes
def _split(y, lens):
    out, o = [], 0
    for n in lens:
        out.append(y[o:o + n])
        o += n
    return out


def test_config2_f16_against_f32_full_size(cuda, stack):
    """configs[1] at its own size (64 x 768 frames -> 12.6 M samples): the f16 fast mode against the f32 (reference
    arithmetic) path on the same batch.  Tolerances: mel max abs 3e-2 (values up to ~6), waveform max abs 5e-3 and
    rms 1e-3 on [-1, 1] (measured: 5e-3 / 1e-3 / 1.5e-4)."""
    from jatts_amd.synthetic import synth_texts
    m, voc = stack
    texts = [t.to(cuda) for t in synth_texts(64, 128, 45, seed=1)]
    res = {}
    try:
        for p in ("fp32", "fp16"):
            m.set_precision(p)
            voc.set_precision(p)
            r = m.inference_batch(texts)
            res[p] = (r["feat_gen"].float().clone(), voc.decode_batch(r["feats_rb"], r["feat_gen"]).float().clone(), r["olens"])
    finally:
        m.set_precision("fp16")
        voc.set_precision("fp16")
    assert res["fp32"][2] == res["fp16"][2] == [768] * 64
    dm = (res["fp16"][0] - res["fp32"][0]).abs()
    dw = res["fp16"][1] - res["fp32"][1]
    assert float(dm.max()) <= 3e-2, float(dm.max())
    assert float(dw.abs().max()) <= 5e-3 and float(dw.pow(2).mean().sqrt()) <= 1e-3, (float(dw.abs().max()), float(dw.pow(2).mean().sqrt()))


def _vocoder(cuda, prec):
    from jatts_amd.synthetic import HIFIGAN_V1_22K, synth_hifigan_state
    from jatts_amd.vocoder import Vocoder
    ones, zeros = [1.0] * 80, [0.0] * 80
    voc = Vocoder(synth_hifigan_state(HIFIGAN_V1_22K, 0),
                  {"sampling_rate": 22050, "generator_type": "HiFiGANGenerator", "generator_params": HIFIGAN_V1_22K},
                  {"mean": zeros, "scale": ones}, cuda, trg_stats={"mean": zeros, "scale": ones})
    return voc.set_precision(prec)


def test_config3_matcha_full_size(cuda, lib):
    """BASELINE configs[2]: MatchaTTS_MAS (U-Net 512/512, head dim 256), 64 utterances x 128 phonemes, 10 Euler steps.
    Properties in f32 (determinism, every utterance equal to the same utterance synthesised alone) and the f16 fast mode
    against f32 on the same batch (mel max abs <= 0.1 after 10 U-Net evaluations; measured 4.4e-3)."""
    from jatts_amd.models import MatchaTTS_MAS
    from jatts_amd.synthetic import MATCHA_MAS_JSUT, synth_state_dict, synth_texts
    m = MatchaTTS_MAS(idim=45, **MATCHA_MAS_JSUT)
    m.load_state_dict(synth_state_dict(m.state_dict(), 0))
    m = m.to(cuda).set_precision("fp32")
    texts = [t.to(cuda) for t in synth_texts(64, 128, 45, seed=1)]
    g = torch.Generator().manual_seed(5)
    texts = [t[: int(torch.randint(64, 129, (1,), generator=g))] for t in texts]          # ragged
    dur = [torch.full((len(t),), 6, dtype=torch.int64, device=cuda) for t in texts]
    noise = [torch.randn(6 * len(t), 80, generator=g).to(cuda) for t in texts]
    run = lambda idx: m.inference_batch([texts[i] for i in idx], n_timesteps=10, temperature=0.667,  # noqa: E731
                                        durations=[dur[i] for i in idx], noise=[noise[i] for i in idx])
    r = run(range(64))
    assert r["olens"] == [6 * len(t) for t in texts]
    mel = r["feat_gen"].clone()
    assert torch.isfinite(mel).all()
    assert torch.equal(run(range(64))["feat_gen"], mel), "not deterministic"
    parts = _split(mel, r["olens"])
    for i in (0, 29, 63):
        alone = run([i])["feat_gen"]
        assert maxdiff(alone, parts[i]) <= 1e-4, f"utterance {i} depends on its batch neighbours: {maxdiff(alone, parts[i]):.3e}"
    voc = _vocoder(cuda, "fp32")
    y32 = voc.decode_batch(r["feats_rb"], mel).clone()
    assert y32.numel() == sum(r["olens"]) * 256 and float(y32.abs().max()) <= 1.0
    m.set_precision("fp16")
    r16 = run(range(64))
    assert r16["olens"] == r["olens"]
    e = float((r16["feat_gen"].float() - mel).abs().max())
    assert e <= 0.1, e
    y16 = _vocoder(cuda, "fp16").decode_batch(r16["feats_rb"], r16["feat_gen"])
    assert float((y16 - y32).abs().max()) <= 2e-2


def test_config5_vits_full_size(cuda, lib):
    """BASELINE configs[4] per-GPU share: mel-VITS with 192-d speaker embeddings, 32 utterances x 128 phonemes."""
    from jatts_amd.models import VITS
    from jatts_amd.synthetic import VITS_JSUT, synth_state_dict, synth_texts
    m = VITS(idim=45, spk_embed_dim=192, **VITS_JSUT)
    m.load_state_dict(synth_state_dict(m.state_dict(), 0))
    m = m.to(cuda).set_precision("fp32")
    g = torch.Generator().manual_seed(7)
    texts = [t.to(cuda) for t in synth_texts(32, 128, 45, seed=3)]
    texts = [t[: int(torch.randint(64, 129, (1,), generator=g))] for t in texts]
    spk = torch.randn(32, 192, generator=g).to(cuda)
    dur = [torch.full((len(t),), 6, dtype=torch.int64, device=cuda) for t in texts]
    noise = [torch.randn(6 * len(t), 384, generator=g).to(cuda) for t in texts]
    run = lambda idx: m.inference_batch([texts[i] for i in idx], spk[list(idx)], noise_scale=0.667,  # noqa: E731
                                        durations=[dur[i] for i in idx], noise=[noise[i] for i in idx])
    r = run(range(32))
    assert r["olens"] == [6 * len(t) for t in texts]
    mel = r["feat_gen"].clone()
    assert torch.isfinite(mel).all()
    assert torch.equal(run(range(32))["feat_gen"], mel), "not deterministic"
    parts = _split(mel, r["olens"])
    for i in (0, 13, 31):
        alone = run([i])["feat_gen"]
        assert maxdiff(alone, parts[i]) <= 1e-4, f"utterance {i} depends on its batch neighbours: {maxdiff(alone, parts[i]):.3e}"
    y32 = _vocoder(cuda, "fp32").decode_batch(r["feats_rb"], mel).clone()
    m.set_precision("fp16")
    r16 = run(range(32))
    e = float((r16["feat_gen"].float() - mel).abs().max())
    assert e <= 5e-2, e
    y16 = _vocoder(cuda, "fp16").decode_batch(r16["feats_rb"], r16["feat_gen"])
    assert float((y16 - y32).abs().max()) <= 1e-2
