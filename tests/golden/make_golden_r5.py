#!/usr/bin/env python3
"""Round-5 golden vectors from the REAL reference (/root/reference): spk_embed_integration_type="concat".

    python tests/golden/make_golden_r5.py          (build container only; the reference never travels)

spk_concat_small.npz: `_integrate_with_spk_embed`'s "concat" branch (models/fastspeech2.py:754-758, the same lines in matchatts_mas.py
and vits.py) through the reference's own B=1 `inference()` of FastSpeech2 / MatchaTTS_MAS / mel-VITS at the small golden widths, 16-d
speaker embeddings, two utterances each; FastSpeech2 also through the batched teacher-forced `forward()` (the train-time call).  Weights
are rebuilt from (name, shape, seed) by the tests, never stored; Matcha / VITS sampling noise is regenerated from its seed.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402
import make_golden_r2 as R2  # noqa: E402
from make_golden import ROOT, np_  # noqa: E402

sys.path.insert(0, ROOT)
from jatts_amd.synthetic import FS2_SMALL, matcha_golden_tweaks, synth_state_dict  # noqa: E402

SPK = 16


def keys_of(sd):
    return json.dumps([[k, list(v.shape)] for k, v in sd.items()])


def main():
    torch.set_num_threads(8)
    out = {}
    g = torch.Generator().manual_seed(55)
    texts = [torch.randint(1, 20, (n,), generator=g) for n in (21, 12)]
    spks = [torch.randn(SPK, generator=torch.Generator().manual_seed(900 + u)) for u in range(2)]
    for u in range(2):
        out[f"u{u}_text"], out[f"u{u}_spemb"] = np_(texts[u]), np_(spks[u])

    # ---- FastSpeech2
    FastSpeech2 = G.import_reference()
    model = FastSpeech2(idim=20, **dict(FS2_SMALL, spk_embed_dim=SPK, spk_embed_integration_type="concat")).eval()
    ref_sd = model.state_dict()
    assert tuple(ref_sd["projection.weight"].shape) == (FS2_SMALL["adim"], FS2_SMALL["adim"] + SPK)
    model.load_state_dict(synth_state_dict(ref_sd, 11))
    out["fs2_keys"] = keys_of(ref_sd)
    for u in range(2):
        with torch.no_grad():
            r = model.inference(texts[u], spembs=spks[u])
        out[f"fs2_u{u}_feat_gen"], out[f"fs2_u{u}_duration"] = np_(r["feat_gen"]), np_(r["duration"])
        out[f"fs2_u{u}_pitch"], out[f"fs2_u{u}_energy"] = np_(r["pitch"]), np_(r["energy"])
        print(f"fs2 concat u{u}: frames {r['feat_gen'].shape[0]} absmax {float(r['feat_gen'].abs().max()):.3f}")
    # batched teacher-forced forward (eval mode): durations from the inference runs, random pitch / energy / feats
    B, Tm = 2, max(int(t.numel()) for t in texts)
    ilens = torch.tensor([int(t.numel()) for t in texts])
    xs = torch.zeros(B, Tm, dtype=torch.long)
    ds = torch.zeros(B, Tm, dtype=torch.long)
    for u in range(2):
        xs[u, :ilens[u]] = texts[u]
        ds[u, :ilens[u]] = torch.tensor(out[f"fs2_u{u}_duration"]).clamp(min=1)
    olens = ds.sum(1)
    gg = torch.Generator().manual_seed(56)
    ys = torch.randn(B, int(olens.max()), 80, generator=gg)
    ps, es = torch.randn(B, Tm, 1, generator=gg), torch.randn(B, Tm, 1, generator=gg)
    with torch.no_grad():
        ret = model(xs, ilens, ys, olens, ds, ilens, ps, ilens, es, ilens, spembs=torch.stack(spks))
    out.update(fwd_xs=np_(xs), fwd_ilens=np_(ilens), fwd_ds=np_(ds), fwd_olens=np_(olens), fwd_ys=np_(ys), fwd_ps=np_(ps), fwd_es=np_(es),
               fwd_after_outs=np_(ret["after_outs"]), fwd_before_outs=np_(ret["before_outs"]), fwd_d_outs=np_(ret["d_outs"]),
               fwd_p_outs=np_(ret["p_outs"]), fwd_e_outs=np_(ret["e_outs"]))
    del model

    # ---- mel-VITS
    VITS = G.import_reference_vits()
    cfg = dict(G.VITS_SMALL, spk_embed_dim=SPK, spk_embed_integration_type="concat")
    model = VITS(idim=20, **cfg).eval()
    ref_sd = model.state_dict()
    model.load_state_dict(synth_state_dict(ref_sd, 12))
    out["vits_keys"], out["vits_config"] = keys_of(ref_sd), json.dumps(cfg)
    for u in range(2):
        r, noise = R2.with_noise(920 + u, lambda: model.inference(texts[u], spembs=spks[u]))
        out[f"vits_u{u}_feat_gen"], out[f"vits_u{u}_duration"] = np_(r["feat_gen"]), np_(r["duration"])
        out[f"vits_u{u}_noise_seed"], out[f"vits_u{u}_noise_shape"] = np.int64(920 + u), np.array(list(noise.shape), dtype=np.int64)
        print(f"vits concat u{u}: frames {r['feat_gen'].shape[0]} absmax {float(r['feat_gen'].abs().max()):.3f}")
    del model

    # ---- MatchaTTS_MAS
    Matcha = G.import_reference_matcha()
    cfg = dict(G.MATCHA_SMALL, spk_embed_dim=SPK, spk_embed_integration_type="concat")
    model = Matcha(idim=20, **cfg).eval()
    ref_sd = model.state_dict()
    model.load_state_dict(matcha_golden_tweaks(synth_state_dict(ref_sd, 13)))
    out["matcha_keys"], out["matcha_config"] = keys_of(ref_sd), json.dumps(cfg)
    out["matcha_n_timesteps"], out["matcha_temperature"] = np.int64(4), np.float32(0.667)
    for u in range(2):
        r, noise = R2.with_noise(940 + u, lambda: model.inference(texts[u], spembs=spks[u], n_timesteps=4, temperature=0.667))
        out[f"matcha_u{u}_feat_gen"], out[f"matcha_u{u}_duration"] = np_(r["feat_gen"]), np_(r["duration"])
        out[f"matcha_u{u}_noise_seed"], out[f"matcha_u{u}_noise_shape"] = np.int64(940 + u), np.array(list(noise.shape), dtype=np.int64)
        print(f"matcha concat u{u}: frames {r['feat_gen'].shape[0]} absmax {float(r['feat_gen'].abs().max()):.3f}")
    np.savez_compressed(os.path.join(HERE, "spk_concat_small.npz"), **out)
    print("spk_concat_small.npz", os.path.getsize(os.path.join(HERE, "spk_concat_small.npz")))


if __name__ == "__main__":
    main()
