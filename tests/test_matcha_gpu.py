"""GPU parity: jatts_amd.models.MatchaTTS_MAS (BASELINE config 3) against a golden
captured from the reference code with injected noise (tests/golden/matcha_small.npz; the diffusers
attention inside the U-Net transformer blocks is a standard-SDPA stand-in there: parity unpinned for that
piece, see oracle/matcha_oracle.py).  4 Euler steps.
Tolerances: fp32 mode max|mel - ref| <= 5e-3; fp16 mode <= 0.15 abs and 3e-2 relative L2 (the ODE
integrates 4 U-Net evaluations of ~60 f16 layers each)."""
import json

import numpy as np
import pytest
import torch

from helpers import golden_state, load_golden, maxdiff, relerr
from jatts_amd.synthetic import matcha_golden_tweaks

pytestmark = pytest.mark.gpu


def _model(cuda, prec):
    from jatts_amd.models import MatchaTTS_MAS
    z, keys = load_golden("matcha_small.npz")
    m = MatchaTTS_MAS(idim=20, **json.loads(str(z["config"])))
    m.load_state_dict(matcha_golden_tweaks(golden_state(keys, 3)))
    return z, m.to(cuda).set_precision(prec)


@pytest.mark.parametrize("prec,atol,rtol", [("fp32", 5e-3, 1e-3), ("fp16", 0.15, 3e-2)])
def test_matcha_matches_reference_golden(cuda, lib, prec, atol, rtol):
    z, m = _model(cuda, prec)
    nt, temp = int(z["n_timesteps"]), float(z["temperature"])
    texts = [torch.tensor(z[f"u{u}_text"]).to(cuda) for u in range(2)]
    noises = [torch.tensor(z[f"u{u}_noise"]) for u in range(2)]
    durs = [torch.tensor(z[f"u{u}_duration"]) for u in range(2)]
    for u in range(2):
        r = m.inference_batch([texts[u]], n_timesteps=nt, temperature=temp, noise=[noises[u]], durations=[durs[u]])
        if prec == "fp32":
            assert torch.equal(r["duration"].cpu(), durs[u])
        ref = z[f"u{u}_feat_gen"]
        assert r["feat_gen"].shape == ref.shape
        e = maxdiff(r["feat_gen"], ref)
        assert e <= atol, f"u{u} {prec}: max|d| = {e:.3e}"
        assert relerr(r["feat_gen"], ref) <= rtol
    rb = m.inference_batch(texts, n_timesteps=nt, temperature=temp, noise=noises, durations=durs)
    o = 0
    for u in range(2):
        n = rb["olens"][u]
        assert n % 2 == 0 and maxdiff(rb["feat_gen"][o:o + n], z[f"u{u}_feat_gen"]) <= atol
        o += n
    out = m.inference(texts[0], n_timesteps=nt, temperature=temp, noise=noises[0])
    assert set(out) == {"feat_gen", "duration", "log_p_attn", "ds"}


def test_matcha_oracle_agrees_at_10_steps(cuda, lib):
    """Config-3 settings (ODE steps 10, temperature 0.667) against the CPU oracle."""
    from oracle.matcha_oracle import matcha_inference
    z, m = _model(cuda, "fp32")
    keys = json.loads(str(z["keys"]))
    sd = matcha_golden_tweaks(golden_state(keys, 3))
    text = torch.tensor(z["u0_text"])
    noise = torch.randn(200, 80, generator=torch.Generator().manual_seed(9))
    ref = matcha_inference(sd, text, 2, 2, noise, n_timesteps=10, temperature=0.667)
    r = m.inference_batch([text.to(cuda)], n_timesteps=10, temperature=0.667, noise=[noise])
    assert torch.equal(r["duration"].cpu(), ref["duration"])
    assert maxdiff(r["feat_gen"], ref["feat_gen"]) <= 1e-2


@pytest.mark.parametrize("prec,atol,rtol", [("fp32", 5e-3, 1e-3), ("fp16", 0.15, 3e-2)])
def test_matcha_full_width_matches_reference_golden(cuda, lib, prec, atol, rtol):
    """BASELINE config 3's own model (MATCHA_MAS_JSUT: U-Net channels 512/512, attention head dim 256 -- the
    relattn_kernel<d_k=256> path), 10 Euler steps, temperature 0.667, against the reference run (matcha_jsut.npz)."""
    from jatts_amd.models import MatchaTTS_MAS
    from jatts_amd.synthetic import MATCHA_MAS_JSUT
    z, keys = load_golden("matcha_jsut.npz")
    m = MatchaTTS_MAS(idim=45, **MATCHA_MAS_JSUT)
    m.load_state_dict(matcha_golden_tweaks(golden_state(keys, 0)))
    m = m.to(cuda).set_precision(prec)
    text, noise, dur = torch.tensor(z["u0_text"]).to(cuda), torch.tensor(z["u0_noise"]), torch.tensor(z["u0_duration"])
    r = m.inference_batch([text], n_timesteps=int(z["n_timesteps"]), temperature=float(z["temperature"]), noise=[noise])
    assert torch.equal(r["duration"].cpu(), dur), "predicted durations differ from the reference (both precisions: f32 trunk)"
    ref = z["u0_feat_gen"]
    assert r["feat_gen"].shape == ref.shape
    e = maxdiff(r["feat_gen"], ref)
    assert e <= atol, f"{prec}: max|d| = {e:.3e}"
    assert relerr(r["feat_gen"], ref) <= rtol


@pytest.mark.parametrize("prec,atol,rtol", [("fp32", 5e-3, 1e-3), ("fp16", 0.15, 3e-2)])
def test_matcha_tts1_matches_reference_golden(cuda, lib, prec, atol, rtol):
    """The tts1 `MatchaTTS` class (reference models/matchatts.py: hard LengthRegulator instead of Gaussian upsampling)."""
    from jatts_amd.models import MatchaTTS
    z, keys = load_golden("matcha_tts1_small.npz")
    m = MatchaTTS(idim=20, **json.loads(str(z["config"])))
    m.load_state_dict(matcha_golden_tweaks(golden_state(keys, 4)))
    m = m.to(cuda).set_precision(prec)
    nt, temp = int(z["n_timesteps"]), float(z["temperature"])
    texts = [torch.tensor(z[f"u{u}_text"]).to(cuda) for u in range(2)]
    noises = [torch.tensor(z[f"u{u}_noise"]) for u in range(2)]
    for u in range(2):
        out = m.inference(texts[u], n_timesteps=nt, temperature=temp, noise=noises[u])
        assert set(out) == {"feat_gen", "duration"}
        assert torch.equal(out["duration"].cpu(), torch.tensor(z[f"u{u}_duration"]))
        ref = z[f"u{u}_feat_gen"]
        assert out["feat_gen"].shape == ref.shape
        e = maxdiff(out["feat_gen"], ref)
        assert e <= atol, f"u{u} {prec}: max|d| = {e:.3e}"
        assert relerr(out["feat_gen"], ref) <= rtol
    rb = m.inference_batch(texts, n_timesteps=nt, temperature=temp, noise=noises)
    o = 0
    for u in range(2):
        n = rb["olens"][u]
        assert maxdiff(rb["feat_gen"][o:o + n], z[f"u{u}_feat_gen"]) <= atol
        o += n


@pytest.mark.parametrize("prec", ["fp32", "fp16"])
def test_matcha_mas_forward_matches_reference_golden(cuda, lib, prec):
    """MatchaTTS_MAS.forward() -- the reference's training-time pass on a padded ragged batch (alignment module, monotonic
    alignment search, masked Gaussian upsampling, CFM loss with the U-Net's mask multiplications), captured from the reference
    with the two random draws of CFM.compute_loss injected (matcha_forward_small.npz).  Integer parts (MAS durations) exact."""
    from jatts_amd.models import MatchaTTS_MAS
    z, keys = load_golden("matcha_forward_small.npz")
    m = MatchaTTS_MAS(idim=20, **json.loads(str(z["config"])))
    m.load_state_dict(matcha_golden_tweaks(golden_state(keys, 3)))
    m = m.to(cuda).set_precision(prec)
    t = lambda k: torch.tensor(z[k])  # noqa: E731
    il, ol = t("text_lengths"), t("feats_lengths")
    r = m(t("text"), il, t("feats"), ol, cfm_t=t("t"), cfm_noise=t("z"))
    assert set(r) == {"d_outs", "ys", "hs", "olens_in", "bin_loss", "log_p_attn", "ds", "cfm_loss"}
    assert torch.equal(r["olens_in"], t("ref_olens_in")) and torch.equal(r["ys"].cpu(), t("ref_ys"))
    assert torch.equal(r["ds"].cpu(), t("ref_ds")), "monotonic alignment search durations differ"
    lp, ref = r["log_p_attn"].cpu(), t("ref_log_p_attn")
    assert torch.equal(torch.isinf(lp), torch.isinf(ref))
    fin = ~torch.isinf(ref)
    tol = {"fp32": 2e-3, "fp16": 2e-3}[prec]     # the alignment module always runs in f32
    assert float((lp[fin] - ref[fin]).abs().max()) <= tol
    assert abs(float(r["bin_loss"]) - float(z["ref_bin_loss"])) <= 1e-3
    assert maxdiff(r["d_outs"], z["ref_d_outs"]) <= (2e-3 if prec == "fp32" else 3e-2)
    assert maxdiff(r["hs"], z["ref_hs"]) <= (3e-3 if prec == "fp32" else 5e-2)
    rel = abs(float(r["cfm_loss"]) - float(z["ref_cfm_loss"])) / float(z["ref_cfm_loss"])
    assert rel <= (1e-3 if prec == "fp32" else 2e-2), (float(r["cfm_loss"]), float(z["ref_cfm_loss"]))


@pytest.mark.parametrize("prec", ["fp32", "fp16"])
def test_matcha_tts1_forward_matches_reference_golden(cuda, lib, prec):
    """tts1 MatchaTTS.forward(): ground-truth durations, hard LengthRegulator with zero padding, CFM loss (matchatts.py:317-480)."""
    from jatts_amd.models import MatchaTTS
    z, keys = load_golden("matcha_tts1_forward_small.npz")
    m = MatchaTTS(idim=20, **json.loads(str(z["config"])))
    m.load_state_dict(matcha_golden_tweaks(golden_state(keys, 4)))
    m = m.to(cuda).set_precision(prec)
    t = lambda k: torch.tensor(z[k])  # noqa: E731
    il, ol = t("text_lengths"), t("feats_lengths")
    r = m(t("text"), il, t("feats"), ol, t("durations"), il, cfm_t=t("t"), cfm_noise=t("z"))
    assert set(r) == {"d_outs", "ys", "hs", "olens_in", "cfm_loss"}
    assert torch.equal(r["olens_in"], t("ref_olens_in")) and torch.equal(r["ys"].cpu(), t("ref_ys"))
    assert maxdiff(r["d_outs"], z["ref_d_outs"]) <= (2e-3 if prec == "fp32" else 3e-2)
    assert maxdiff(r["hs"], z["ref_hs"]) <= (3e-3 if prec == "fp32" else 5e-2)
    rel = abs(float(r["cfm_loss"]) - float(z["ref_cfm_loss"])) / float(z["ref_cfm_loss"])
    assert rel <= (1e-3 if prec == "fp32" else 2e-2), (float(r["cfm_loss"]), float(z["ref_cfm_loss"]))


def test_inference_with_feats_alignment_branches(cuda, lib):
    """inference(text, feats=...): MatchaTTS_MAS returns the alignment of the given features (log_p_attn, ds), VITS additionally the
    posterior reconstruction outs_bar -- against the reference (infer_feats_small.npz; noise draws injected)."""
    from jatts_amd.models import VITS, MatchaTTS_MAS
    z, _ = load_golden("infer_feats_small.npz")
    t = lambda k: torch.tensor(z[k])  # noqa: E731
    m = MatchaTTS_MAS(idim=20, **json.loads(str(z["matcha_config"])))
    m.load_state_dict(matcha_golden_tweaks(golden_state(json.loads(str(z["matcha_keys"])), 3)))
    m = m.to(cuda)
    r = m.inference(t("m_text").to(cuda), feats=t("m_feats"), n_timesteps=4, temperature=0.667, noise=t("m_noise"))
    assert torch.equal(r["ds"].cpu(), t("m_ds")) and torch.equal(r["duration"].cpu(), t("m_duration"))
    assert maxdiff(r["log_p_attn"], z["m_log_p_attn"]) <= 2e-3 and maxdiff(r["feat_gen"], z["m_feat_gen"]) <= 5e-3
    v = VITS(idim=20, **json.loads(str(z["vits_config"])))
    v.load_state_dict(golden_state(json.loads(str(z["vits_keys"])), 2))
    v = v.to(cuda)
    r = v.inference(t("v_text").to(cuda), feats=t("v_feats"), spembs=t("v_spemb").to(cuda), noise=t("v_noise"), post_noise=t("v_post_noise"))
    assert set(r) == {"feat_gen", "duration", "log_p_attn", "ds", "outs_bar"}
    assert torch.equal(r["ds"].cpu(), t("v_ds")) and torch.equal(r["duration"].cpu(), t("v_duration"))
    assert maxdiff(r["log_p_attn"], z["v_log_p_attn"]) <= 2e-3
    assert maxdiff(r["feat_gen"], z["v_feat_gen"]) <= 3e-3 and maxdiff(r["outs_bar"], z["v_outs_bar"]) <= 3e-3
