"""the speaker-embedding front end (log-mel filterbank + ECAPA-TDNN) on the HIP path against the CPU oracle
(oracle/ecapa_oracle.py -- restated from the public SpeechBrain recipe; speechbrain is not under /root/reference: parity
unpinned).  f32 throughout (exact-f32 MFMA); tolerances: features 2e-3 dB abs (the DFT runs as a 400-term f32 contraction
where torch uses an FFT), embedding 2e-3 relative."""
import wave

import numpy as np
import pytest
import torch

from helpers import maxdiff, relerr
from jatts_amd.synthetic import synth_state_dict

pytestmark = pytest.mark.gpu

SMALL = dict(channels=(512, 512, 512, 512, 1536), attention_channels=64, se_channels=64, lin_neurons=192)


def _waves():
    g = torch.Generator().manual_seed(0)
    out = []
    for n in (19200, 11000, 25601):
        t = torch.arange(n) / 16000.0
        out.append(0.3 * torch.sin(2 * np.pi * (120 + 40 * len(out)) * t) * (1 + 0.5 * torch.sin(2 * np.pi * 3 * t)) + 0.05 * torch.randn(n, generator=g))
    return out


def test_fbank_features_match_oracle(cuda, lib):
    from jatts_amd.spkemb import FbankFrontEnd
    from oracle.ecapa_oracle import fbank_features
    waves = _waves()
    rb, feats = FbankFrontEnd(cuda)(waves)
    assert rb.lens == [1 + w.numel() // 160 for w in waves]
    o = 0
    for w, n in zip(waves, rb.lens):
        ref = fbank_features(w)
        assert ref.shape == (n, 80)
        assert maxdiff(feats[o:o + n, :80], ref) <= 2e-3
        assert not feats[o:o + n, 80:].any()
        o += n


@pytest.mark.parametrize("cfg", [SMALL, {}], ids=["small", "spkrec-ecapa-voxceleb"])
def test_ecapa_embedding_matches_oracle(cuda, lib, cfg):
    from jatts_amd.spkemb import SpkEmbExtractor
    from oracle.ecapa_oracle import encode
    ex = SpkEmbExtractor(cuda, **cfg)
    sd = synth_state_dict(ex.model.state_dict(), 5)
    ex.model.load_state_dict(sd)
    waves = _waves()
    emb = ex.encode_batch(waves)
    assert emb.shape == (3, 192)
    for b, w in enumerate(waves):
        ref = encode(sd, w)
        assert relerr(emb[b], ref) <= 2e-3, (b, relerr(emb[b], ref))
    one = ex.encode_batch([waves[1]])            # an utterance does not depend on its batch neighbours
    assert maxdiff(one[0], emb[1]) <= 1e-5


def test_extractor_contract_and_cache(cuda, lib, tmp_path):
    """forward(wav_path) -> np.ndarray (192,), as SpeechBrainSpkEmbExtractor.forward (spkemb_speechbrain.py:20-28); cached per file."""
    from jatts_amd.spkemb import SpkEmbExtractor
    ex = SpkEmbExtractor(cuda, **SMALL)
    ex.model.load_state_dict(synth_state_dict(ex.model.state_dict(), 5))
    w = _waves()[0]
    p = str(tmp_path / "ref.wav")
    with wave.open(p, "wb") as f:
        f.setnchannels(1); f.setsampwidth(2); f.setframerate(16000)
        f.writeframes((w.clamp(-1, 1) * 32767).round().to(torch.int16).numpy().tobytes())
    e = ex.forward(p)
    assert isinstance(e, np.ndarray) and e.shape == (192,) and e.dtype == np.float32 and np.isfinite(e).all()
    assert ex.forward(p) is not None and list(ex._cache) == [p]
    both = ex.forward_many([p, p])
    assert np.array_equal(both[0], e) and np.array_equal(both[1], e)
