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

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def stack(cuda, lib):
    from jatts_amd.models import FastSpeech2
    from jatts_amd.synthetic import FS2_JSUT, HIFIGAN_V1_22K, pin_duration_head, synth_hifigan_state, synth_state_dict
    from jatts_amd.vocoder import Vocoder
    m = FastSpeech2(idim=45, **FS2_JSUT)
    m.load_state_dict(pin_duration_head(synth_state_dict(m.state_dict(), 0), 6))
    m = m.to(cuda).set_precision("fp16")
    ones, zeros = [1.0] * 80, [0.0] * 80
    voc = Vocoder(synth_hifigan_state(HIFIGAN_V1_22K, 0),
                  {"sampling_rate": 22050, "generator_type": "HiFiGANGenerator", "generator_params": HIFIGAN_V1_22K},
                  {"mean": zeros, "scale": ones}, cuda, trg_stats={"mean": zeros, "scale": ones})
    voc.set_precision("fp16")
    return m, voc


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


@pytest.mark.parametrize("ragged", [False, True], ids=["64x128", "64xU(64..128)"])
def test_full_batch_properties(cuda, stack, ragged):
    from jatts_amd.synthetic import synth_texts
    texts = [t.to(cuda) for t in synth_texts(64, 128, 45, seed=1)]
    if ragged:
        g = torch.Generator().manual_seed(5)
        texts = [t[: int(torch.randint(64, 129, (1,), generator=g))] for t in texts]
    r, outs = _synth(stack, texts)
    # integer path: exact
    assert torch.equal(r["duration"].cpu(), torch.full((sum(len(t) for t in texts),), 6, dtype=torch.int64))
    assert r["olens"] == [6 * len(t) for t in texts]
    for t, y in zip(texts, outs):
        assert y.numel() == 6 * len(t) * 256
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
