#!/usr/bin/env python3
"""Round-3 golden vectors from the REAL reference (/root/reference): the shapes bench.py times.

    python tests/golden/make_golden_r3.py          (build container only; the reference never travels)

vits_bench128.npz / matcha_bench128.npz: BASELINE configs[4] / configs[2] models on a 128-phoneme bench utterance (noise is regenerated
from its seed by the tests: torch.randn(shape, generator=manual_seed(seed)) -- not stored).
fs2_bench768.npz: conf/fastspeech2.v1.yaml-width FastSpeech2 (jatts_amd.synthetic.FS2_JSUT), synthetic weights (seed 0)
with the duration head pinned to 6 frames per phoneme, utterances 0 and 37 of bench.py's batch
(synth_texts(64, 128, 45, seed=1)) through the reference's own B=1 `inference()` -> (768, 80) mel each.  Weights are
rebuilt from (name, shape, seed) by the tests, never stored.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import ROOT, import_reference, np_  # noqa: E402

sys.path.insert(0, ROOT)
from jatts_amd.synthetic import FS2_JSUT, pin_duration_head, synth_state_dict, synth_texts  # noqa: E402

UTTS = (0, 37)


def matcha_vits_bench_width():
    """configs[2] / configs[4] at the bench's utterance length: utterance 5 of the bench batch (128 phonemes) through the REAL reference's
    MatchaTTS_MAS (config-3 model, 10 Euler steps, temperature 0.667; diffusers attention = the SDPA stand-in of make_golden.py) and
    mel-VITS (config-5 model, 192-d speaker embedding), sampling noise injected."""
    import make_golden as G
    import make_golden_r2 as R2
    from jatts_amd.synthetic import MATCHA_MAS_JSUT, VITS_JSUT, matcha_golden_tweaks
    text = synth_texts(64, 128, 45, seed=1)[5]
    VITS = G.import_reference_vits()
    model = VITS(idim=45, spk_embed_dim=192, **VITS_JSUT).eval()
    ref_sd = model.state_dict()
    model.load_state_dict(pin_duration_head(synth_state_dict(ref_sd, 0), 6))     # every phoneme -> the same frame count, as in bench.py
    spemb = torch.randn(192, generator=torch.Generator().manual_seed(3))
    r, noise = R2.with_noise(710, lambda: model.inference(text, spembs=spemb))
    out = {"keys": json.dumps([[k, list(v.shape)] for k, v in ref_sd.items()]), "u0_text": np_(text), "u0_spemb": np_(spemb),
           "noise_seed": np.int64(710), "noise_shape": np.array(list(noise[0].t().shape), dtype=np.int64),
           "u0_feat_gen": np_(r["feat_gen"]), "u0_duration": np_(r["duration"])}
    assert torch.equal(noise[0].t(), torch.randn(noise.shape, generator=torch.Generator().manual_seed(710))[0].t())
    print("vits_bench128: frames", r["feat_gen"].shape[0], "absmax", float(r["feat_gen"].abs().max()))
    from oracle.vits_oracle import vits_inference
    o = vits_inference(model.state_dict(), text, 2, 2, spemb, noise[0].t())
    print("  oracle-vs-ref mel max|d| =", float((o["feat_gen"] - r["feat_gen"]).abs().max()))
    np.savez_compressed(os.path.join(HERE, "vits_bench128.npz"), **out)
    del model
    Matcha = G.import_reference_matcha()
    model = Matcha(idim=45, **MATCHA_MAS_JSUT).eval()
    ref_sd = model.state_dict()
    model.load_state_dict(pin_duration_head(matcha_golden_tweaks(synth_state_dict(ref_sd, 0)), 6))
    r, noise = R2.with_noise(510, lambda: model.inference(text, n_timesteps=10, temperature=0.667))
    out = {"keys": json.dumps([[k, list(v.shape)] for k, v in ref_sd.items()]), "n_timesteps": np.int64(10), "temperature": np.float32(0.667),
           "u0_text": np_(text), "noise_seed": np.int64(510), "noise_shape": np.array(list(noise[0].t().shape), dtype=np.int64),
           "u0_feat_gen": np_(r["feat_gen"]), "u0_duration": np_(r["duration"])}
    print("matcha_bench128: frames", r["feat_gen"].shape[0], "absmax", float(r["feat_gen"].abs().max()))
    from oracle.matcha_oracle import matcha_inference
    o = matcha_inference(model.state_dict(), text, 2, 2, noise[0].t(), n_timesteps=10, temperature=0.667)
    print("  oracle-vs-ref mel max|d| =", float((o["feat_gen"] - r["feat_gen"]).abs().max()))
    np.savez_compressed(os.path.join(HERE, "matcha_bench128.npz"), **out)


def main():
    torch.set_num_threads(8)
    FastSpeech2 = import_reference()
    model = FastSpeech2(idim=45, **FS2_JSUT).eval()
    ref_sd = model.state_dict()
    sd = pin_duration_head(synth_state_dict(ref_sd, 0), 6)
    model.load_state_dict(sd)
    texts = synth_texts(64, 128, 45, seed=1)
    out = {"keys": json.dumps([[k, list(v.shape)] for k, v in ref_sd.items()]), "utts": np.array(UTTS, dtype=np.int64)}
    from oracle.fs2_oracle import fs2_inference
    for j, u in enumerate(UTTS):
        with torch.no_grad():
            r = model.inference(texts[u])
        assert r["feat_gen"].shape == (768, 80) and bool((r["duration"] == 6).all())
        out[f"u{j}_text"] = np_(texts[u])
        out[f"u{j}_feat_gen"] = np_(r["feat_gen"])
        out[f"u{j}_duration"] = np_(r["duration"])
        out[f"u{j}_pitch"] = np_(r["pitch"])
        out[f"u{j}_energy"] = np_(r["energy"])
        o = fs2_inference(sd, texts[u], 2)
        print(f"bench utt {u}: oracle-vs-ref mel max|d| =", float((o["feat_gen"] - r["feat_gen"]).abs().max()),
              "mel absmax", float(r["feat_gen"].abs().max()))
    path = os.path.join(HERE, "fs2_bench768.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path))
    matcha_vits_bench_width()
    for f in ("vits_bench128.npz", "matcha_bench128.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
