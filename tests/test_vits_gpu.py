"""GPU parity: jatts_amd.models.VITS (mel-VITS) against golden vectors captured from
the real reference with the sampling noise injected (tests/golden/vits_small.npz).
Tolerances: fp32 mode max|mel - ref| <= 3e-3; fp16 mode <= 8e-2 abs and 2e-2 relative L2."""
import json

import numpy as np
import pytest
import torch

from helpers import golden_state, load_golden, maxdiff, relerr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("prec,atol,rtol", [("fp32", 3e-3, 5e-4), ("fp16", 8e-2, 2e-2)])
def test_vits_matches_reference_golden(cuda, lib, prec, atol, rtol):
    from jatts_amd.models import VITS
    z, keys = load_golden("vits_small.npz")
    cfg = json.loads(str(z["config"]))
    m = VITS(idim=20, **cfg)
    m.load_state_dict(golden_state(keys, 2))
    m = m.to(cuda).set_precision(prec)
    texts, sps, noises, durs = [], [], [], []
    for u in range(2):
        texts.append(torch.tensor(z[f"u{u}_text"]).to(cuda))
        sps.append(torch.tensor(z[f"u{u}_spemb"]))
        noises.append(torch.tensor(z[f"u{u}_noise"]))
        durs.append(torch.tensor(z[f"u{u}_duration"]))
    # per-utterance API (reference contract) with injected noise; durations teacher-forced in fp16 (H3)
    for u in range(2):
        r = m.inference_batch([texts[u]], sps[u].unsqueeze(0), noise=[noises[u]], durations=[durs[u]])
        if prec == "fp32":
            assert torch.equal(r["duration"].cpu(), durs[u]), "predicted durations differ from the reference"
        ref = z[f"u{u}_feat_gen"]
        assert r["feat_gen"].shape == ref.shape
        assert maxdiff(r["feat_gen"], ref) <= atol, f"u{u} {prec}: max|d| = {maxdiff(r['feat_gen'], ref):.3e}"
        assert relerr(r["feat_gen"], ref) <= rtol
    # ragged batch == per-utterance
    rb = m.inference_batch(texts, torch.stack(sps), noise=noises, durations=durs)
    o = 0
    for u in range(2):
        n = rb["olens"][u]
        assert maxdiff(rb["feat_gen"][o:o + n], z[f"u{u}_feat_gen"]) <= atol
        o += n
    out = m.inference(texts[0], spembs=sps[0].to(cuda), noise=noises[0])
    assert set(out) == {"feat_gen", "duration", "log_p_attn", "ds"} and out["log_p_attn"] is None


def test_vits_state_dict_schema(golden_dir):
    from jatts_amd.models import VITS
    z = np.load(golden_dir + "/vits_small.npz")
    keys = json.loads(str(z["keys"]))
    sd = VITS(idim=20, **json.loads(str(z["config"]))).state_dict()
    assert [k for k, _ in keys] == list(sd.keys())
    assert all(tuple(s) == tuple(sd[k].shape) for k, s in keys)


@pytest.mark.parametrize("prec,atol,rtol", [("fp32", 3e-3, 5e-4), ("fp16", 8e-2, 2e-2)])
def test_vits_full_width_192d_matches_reference_golden(cuda, lib, prec, atol, rtol):
    """BASELINE config 5's own model (VITS_JSUT + 192-d speaker embedding) against the reference run (vits_jsut.npz)."""
    from jatts_amd.models import VITS
    from jatts_amd.synthetic import VITS_JSUT
    z, keys = load_golden("vits_jsut.npz")
    m = VITS(idim=45, spk_embed_dim=192, **VITS_JSUT)
    m.load_state_dict(golden_state(keys, 0))
    m = m.to(cuda).set_precision(prec)
    for u in range(2):
        r = m.inference_batch([torch.tensor(z[f"u{u}_text"]).to(cuda)], torch.tensor(z[f"u{u}_spemb"]).unsqueeze(0),
                              noise=[torch.tensor(z[f"u{u}_noise"])])
        assert torch.equal(r["duration"].cpu(), torch.tensor(z[f"u{u}_duration"])), "predicted durations differ (f32 trunk in both modes)"
        ref = z[f"u{u}_feat_gen"]
        assert r["feat_gen"].shape == ref.shape
        assert maxdiff(r["feat_gen"], ref) <= atol, f"u{u} {prec}: max|d| = {maxdiff(r['feat_gen'], ref):.3e}"
        assert relerr(r["feat_gen"], ref) <= rtol


@pytest.mark.parametrize("prec", ["fp32", "fp16"])
def test_vits_forward_matches_reference_golden(cuda, lib, prec):
    """VITS.forward() -- the reference's training-time pass on a padded ragged batch (posterior encoder, forward flow, alignment
    module + MAS, masked Gaussian upsampling, decoder), captured from the reference with the posterior noise injected."""
    from jatts_amd.models import VITS
    z, keys = load_golden("vits_forward_small.npz")
    m = VITS(idim=20, **json.loads(str(z["config"])))
    m.load_state_dict(golden_state(keys, 2))
    m = m.to(cuda).set_precision(prec)
    t = lambda k: torch.tensor(z[k])  # noqa: E731
    il, ol = t("text_lengths"), t("feats_lengths")
    r = m(t("text"), il, t("feats"), ol, spembs=t("spembs"), post_noise=t("noise"))
    assert set(r) == {"outs", "d_outs", "ys", "hs", "olens_in", "bin_loss", "log_p_attn", "ds", "m_p", "logs_p", "z", "y_mask", "z_p",
                      "m_q", "logs_q"}
    assert torch.equal(r["ds"].cpu(), t("ref_ds")), "monotonic alignment search durations differ"
    assert torch.equal(r["olens_in"], t("ref_olens_in")) and torch.equal(r["y_mask"].cpu(), t("ref_y_mask"))
    lp, ref = r["log_p_attn"].cpu(), t("ref_log_p_attn")
    assert torch.equal(torch.isinf(lp), torch.isinf(ref))
    fin = ~torch.isinf(ref)
    assert float((lp[fin] - ref[fin]).abs().max()) <= 2e-3
    assert abs(float(r["bin_loss"]) - float(z["ref_bin_loss"])) <= 1e-3
    tol = {"fp32": 3e-3, "fp16": 8e-2}[prec]
    for k in ("hs", "m_p", "logs_p", "m_q", "logs_q", "z", "z_p", "d_outs"):
        assert r[k].shape == z["ref_" + k].shape, k
        assert maxdiff(r[k], z["ref_" + k]) <= tol, (k, maxdiff(r[k], z["ref_" + k]))
    for b, n in enumerate(ol.tolist()):          # valid frames tight; padded frames carry the same padding leakage
        assert maxdiff(r["outs"][b, :n], z["ref_outs"][b, :n]) <= tol, (b, maxdiff(r["outs"][b, :n], z["ref_outs"][b, :n]))
    assert maxdiff(r["outs"], z["ref_outs"]) <= 5 * tol
