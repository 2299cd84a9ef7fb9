"""The B = 1 drop-in path by hipGraph replay (jatts_amd/graphs.py; reference call sites jatts/bin/tts_decode.py:230,249): `model.inference(x)` and
`vocoder.decode(mel)` answered from a captured graph are BIT-IDENTICAL to the eager launches -- same kernels, same order, same arithmetic -- for
every arithmetic, for new inputs of a captured signature, and the tensors handed out are not overwritten by the next replay."""
import pytest
import torch

from helpers import golden_state, load_golden
from jatts_amd.synthetic import FS2_SMALL, HIFIGAN_V1_22K, HIFIGAN_V1_24K, synth_hifigan_state

pytestmark = pytest.mark.gpu


def _eager(fn):
    from jatts_amd import graphs
    on, graphs.ENABLED = graphs.ENABLED, False
    try:
        return fn()
    finally:
        graphs.ENABLED = on


def _fs2(cuda, prec, **kw):
    from jatts_amd.models import FastSpeech2
    _, keys = load_golden("fs2_small.npz")
    m = FastSpeech2(idim=20, **FS2_SMALL, **kw)
    if kw:
        from jatts_amd.synthetic import synth_state_dict
        m.load_state_dict(synth_state_dict(m.state_dict(), 0))
    else:
        m.load_state_dict(golden_state(keys, 0))
    return m.to(cuda).set_precision(prec)


@pytest.mark.parametrize("prec", ["fp32", "fp32_bf16x3", "fp32_bf16x3_6p", "fp32_split", "fp16"])
def test_fs2_inference_graph_is_bit_identical(cuda, lib, prec):
    m = _fs2(cuda, prec)
    g = torch.Generator().manual_seed(3)
    texts = [torch.randint(1, 20, (17,), generator=g).to(cuda) for _ in range(5)] + [torch.randint(1, 20, (9,), generator=g).to(cuda) for _ in range(3)]
    ref = _eager(lambda: [m.inference(t) for t in texts])
    got = [m.inference(t) for t in texts]          # 17 phonemes: eager, capture, replay x3; 9 phonemes: eager, capture, replay
    gc = m._prep["graphs"]
    assert gc.stats["captured"] >= 2 and gc.stats["replayed"] >= 3 and gc.stats["failed"] == 0, gc.stats
    for r, o in zip(ref, got):
        assert set(o) == {"feat_gen", "duration", "pitch", "energy"}
        for k in o:
            assert torch.equal(r[k], o[k]), f"{prec} {k}: replay differs from the eager launches"
    # the tensors of an earlier call survive later replays of the same graph
    keep = m.inference(texts[0])
    snap = {k: v.clone() for k, v in keep.items()}
    m.inference(texts[1])
    m.inference(texts[2])
    for k in keep:
        assert torch.equal(keep[k], snap[k])
    assert torch.equal(keep["feat_gen"], ref[0]["feat_gen"])


def test_fs2_graph_with_speaker_embedding_alpha_and_durations(cuda, lib):
    m = _fs2(cuda, "fp32", spk_embed_dim=16, spk_embed_integration_type="concat")
    g = torch.Generator().manual_seed(5)
    text = [torch.randint(1, 20, (13,), generator=g).to(cuda) for _ in range(4)]
    spk = [torch.randn(16, generator=g).to(cuda) for _ in range(4)]
    for alpha in (1.0, 1.3):
        ref = _eager(lambda: [m.inference(t, spembs=s, alpha=alpha) for t, s in zip(text, spk)])
        got = [m.inference(t, spembs=s, alpha=alpha) for t, s in zip(text, spk)]
        for r, o in zip(ref, got):
            assert torch.equal(r["feat_gen"], o["feat_gen"]) and torch.equal(r["duration"], o["duration"])
    # teacher-forced durations through inference_batch (B = 1): its own signature
    d = torch.full((13,), 3, dtype=torch.int64)
    ref = _eager(lambda: [m.inference_batch([t], spembs=s.unsqueeze(0), durations=[d])["feat_gen"] for t, s in zip(text, spk)])
    got = [m.inference_batch([t], spembs=s.unsqueeze(0), durations=[d])["feat_gen"] for t, s in zip(text, spk)]
    for r, o in zip(ref, got):
        assert torch.equal(r, o)
    assert m._prep["graphs"].stats["failed"] == 0


def test_bad_token_id_still_raises_under_replay(cuda, lib):
    m = _fs2(cuda, "fp32")
    t = torch.randint(1, 20, (11,), generator=torch.Generator().manual_seed(1)).to(cuda)
    for _ in range(3):
        m.inference(t)
    bad = t.clone()
    bad[4] = 20
    with pytest.raises(IndexError):
        m.inference(bad)
    m.inference(t)        # and the counter was reset: the next good utterance passes


@pytest.mark.parametrize("prec", ["fp32", "fp32_bf16x3", "fp16"])
@pytest.mark.parametrize("params", [HIFIGAN_V1_22K, HIFIGAN_V1_24K], ids=["22k", "24k"])
def test_vocoder_decode_graph_is_bit_identical(cuda, lib, prec, params):
    from jatts_amd.vocoder import Vocoder
    p = dict(params, channels=128)
    sd = synth_hifigan_state(p, seed=0)
    g = torch.Generator().manual_seed(2)
    mean, scale = torch.randn(80, generator=g).tolist(), (torch.rand(80, generator=g) + 0.5).tolist()
    v = Vocoder(sd, {"sampling_rate": 24000, "generator_type": "HiFiGANGenerator", "generator_params": p}, {"mean": mean, "scale": scale}, cuda,
                trg_stats={"mean": [0.1] * 80, "scale": [1.1] * 80}).set_precision(prec)
    mels = [torch.randn(23, 80, generator=g).to(cuda) for _ in range(4)] + [torch.randn(40, 80, generator=g).to(cuda) for _ in range(3)]
    ref = _eager(lambda: [v.decode(c)[0] for c in mels])
    got = [v.decode(c)[0] for c in mels]
    gc = v.model._prep["graphs"]
    assert gc.stats["captured"] == 2 and gc.stats["replayed"] == 3 and gc.stats["failed"] == 0, gc.stats
    for c, r, o in zip(mels, ref, got):
        assert o.shape == r.shape and o.numel() == c.shape[0] * v.model.hop
        assert torch.equal(r, o), f"{prec}: replayed waveform differs from the eager launches"
    keep = v.decode(mels[0])[0]
    snap = keep.clone()
    v.decode(mels[1])
    assert torch.equal(keep, snap) and torch.equal(keep, ref[0])


def test_graph_cache_evicts_least_recently_used(cuda, lib):
    from jatts_amd.graphs import GraphCache
    gc = GraphCache(max_graphs=2)
    x = torch.arange(8, dtype=torch.float32, device=cuda)
    for key in ("a", "b", "c", "a"):
        for _ in range(3):
            y = gc.run(key, lambda t: t * 2 + 1, (x,))
            assert torch.equal(y, x * 2 + 1)
    assert len(gc) == 2 and gc.stats["captured"] == 4      # "a" was evicted by "c" and captured again
