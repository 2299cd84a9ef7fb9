"""GPU parity of `spk_embed_integration_type="concat"` (models/fastspeech2.py:754-758 and the same branch of matchatts_mas.py / vits.py;
VERDICT r4 missing #5) against the REAL reference's outputs (tests/golden/spk_concat_small.npz, make_golden_r5.py): the projection over
cat[hs, normalize(spembs)] runs as one k = 1 conv over the frames plus a per-utterance vector (jatts_amd.models._conformer.SpkProjection).
Tolerances: the add-mode goldens' (FastSpeech2 mel 2e-3 abs, VITS 3e-3, Matcha 5e-3; durations exact)."""
import json

import pytest
import torch

from helpers import golden_state, load_golden, maxdiff
from jatts_amd.synthetic import FS2_SMALL, matcha_golden_tweaks

pytestmark = pytest.mark.gpu
SPK = 16


def _keys(z, name):
    return json.loads(str(z[name]))


def _noise(z, p):
    shape = [int(v) for v in z[p + "_noise_shape"]]            # the reference drew randn_like(x), x (1, C, T); the models take (T, C)
    return torch.randn(shape, generator=torch.Generator().manual_seed(int(z[p + "_noise_seed"])))[0].t().contiguous()


@pytest.mark.parametrize("prec", ["fp32", "fp32_bf16x3", "fp32_split"])
def test_fs2_concat_matches_the_reference(cuda, lib, prec):
    from jatts_amd.models import FastSpeech2
    z, _ = load_golden("spk_concat_small.npz")
    m = FastSpeech2(idim=20, **FS2_SMALL, spk_embed_dim=SPK, spk_embed_integration_type="concat")
    assert tuple(m.state_dict()["projection.weight"].shape) == (FS2_SMALL["adim"], FS2_SMALL["adim"] + SPK)
    keys = _keys(z, "fs2_keys")
    assert [k for k, _ in keys] == list(m.state_dict().keys())
    m.load_state_dict(golden_state(keys, 11))
    m = m.to(cuda).set_precision(prec)
    texts = [torch.tensor(z[f"u{u}_text"]).to(cuda) for u in range(2)]
    spks = [torch.tensor(z[f"u{u}_spemb"]) for u in range(2)]
    for u in range(2):
        r = m.inference(texts[u], spembs=spks[u].to(cuda))
        assert torch.equal(r["duration"].cpu(), torch.tensor(z[f"fs2_u{u}_duration"]))
        assert maxdiff(r["feat_gen"], z[f"fs2_u{u}_feat_gen"]) <= 2e-3
        assert maxdiff(r["pitch"].reshape(-1), z[f"fs2_u{u}_pitch"].reshape(-1)) <= 2e-3
        assert maxdiff(r["energy"].reshape(-1), z[f"fs2_u{u}_energy"].reshape(-1)) <= 2e-3
    rb = m.inference_batch(texts, spembs=torch.stack(spks))      # ragged batch == per utterance
    o = 0
    for u in range(2):
        n = rb["olens"][u]
        assert maxdiff(rb["feat_gen"][o:o + n], z[f"fs2_u{u}_feat_gen"]) <= 2e-3
        o += n
    if prec != "fp32":
        return
    # the train-time call: batched, padded, teacher-forced forward() (fastspeech2.py:473-564)
    t = lambda k: torch.tensor(z[k])  # noqa: E731
    il, ol = t("fwd_ilens"), t("fwd_olens")
    r = m(t("fwd_xs"), il, t("fwd_ys"), ol, t("fwd_ds"), il, t("fwd_ps"), il, t("fwd_es"), il, spembs=torch.stack(spks))
    for k in ("d_outs", "p_outs", "e_outs"):
        assert maxdiff(r[k], z["fwd_" + k]) <= 2e-3, k
    for k in ("before_outs", "after_outs"):
        for b, n in enumerate(ol.tolist()):
            assert maxdiff(r[k][b, :n], z["fwd_" + k][b, :n]) <= 2e-3, (k, b)


@pytest.mark.parametrize("prec", ["fp32", "fp32_bf16x3"])
def test_vits_concat_matches_the_reference(cuda, lib, prec):
    from jatts_amd.models import VITS
    z, _ = load_golden("spk_concat_small.npz")
    m = VITS(idim=20, **json.loads(str(z["vits_config"])))
    m.load_state_dict(golden_state(_keys(z, "vits_keys"), 12))
    m = m.to(cuda).set_precision(prec)
    for u in range(2):
        r = m.inference_batch([torch.tensor(z[f"u{u}_text"]).to(cuda)], torch.tensor(z[f"u{u}_spemb"]).unsqueeze(0), noise=[_noise(z, f"vits_u{u}")])
        assert torch.equal(r["duration"].cpu(), torch.tensor(z[f"vits_u{u}_duration"]))
        assert maxdiff(r["feat_gen"], z[f"vits_u{u}_feat_gen"]) <= 3e-3


@pytest.mark.parametrize("prec", ["fp32", "fp32_bf16x3"])
def test_matcha_concat_matches_the_reference(cuda, lib, prec):
    from jatts_amd.models import MatchaTTS_MAS
    z, _ = load_golden("spk_concat_small.npz")
    m = MatchaTTS_MAS(idim=20, **json.loads(str(z["matcha_config"])))
    m.load_state_dict(matcha_golden_tweaks(golden_state(_keys(z, "matcha_keys"), 13)))
    m = m.to(cuda).set_precision(prec)
    nt, temp = int(z["matcha_n_timesteps"]), float(z["matcha_temperature"])
    for u in range(2):
        r = m.inference_batch([torch.tensor(z[f"u{u}_text"]).to(cuda)], spembs=torch.tensor(z[f"u{u}_spemb"]).unsqueeze(0), n_timesteps=nt,
                              temperature=temp, noise=[_noise(z, f"matcha_u{u}")])
        assert torch.equal(r["duration"].cpu(), torch.tensor(z[f"matcha_u{u}_duration"]))
        assert maxdiff(r["feat_gen"], z[f"matcha_u{u}_feat_gen"]) <= 5e-3


def test_unknown_integration_type_is_refused(lib):
    from jatts_amd.models import FastSpeech2
    with pytest.raises(NotImplementedError):
        FastSpeech2(idim=20, **FS2_SMALL, spk_embed_dim=SPK, spk_embed_integration_type="mul")
