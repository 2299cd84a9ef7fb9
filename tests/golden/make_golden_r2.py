#!/usr/bin/env python3
"""Round-2 golden vectors, again captured by importing the REAL reference from /root/reference (build container
only; the fixtures are data: ids, durations, injected noise, outputs -- weights are rebuilt from seeds):

  matcha_jsut.npz        full-width BASELINE config 3 model (MATCHA_MAS_JSUT: U-Net 512/512, attention head dim 256),
                         10 Euler steps, temperature 0.667, one short utterance          [diffusers attention = SDPA stand-in]
  vits_jsut.npz          full-width BASELINE config 5 model (VITS_JSUT + 192-d speaker embedding), two short utterances
  matcha_tts1_small.npz  the tts1 `MatchaTTS` class (hard LengthRegulator), small config, 4 Euler steps
  fs2_forward_small.npz  FastSpeech2.forward() -- the training-time, teacher-forced, PADDED batched pass
                         (fastspeech2.py:473-564): before_outs / after_outs / d_outs / p_outs / e_outs for a ragged batch
  matcha_forward_small.npz, vits_forward_small.npz  MatchaTTS_MAS.forward() / VITS.forward() on padded ragged batches, random
                         draws injected (CFM t and noise; posterior sampling noise)

    python tests/golden/make_golden_r2.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402

from jatts_amd.synthetic import (FS2_SMALL, MATCHA_MAS_JSUT, VITS_JSUT, matcha_golden_tweaks,  # noqa: E402
                                 synth_state_dict)

np_ = G.np_


def with_noise(seed, fn):
    """Run fn() with torch.randn_like replaced by a seeded generator; returns (result, the noise drawn)."""
    holder = {}
    real = torch.randn_like

    def fake(t, *a, **k):
        holder["noise"] = torch.randn(t.shape, generator=torch.Generator().manual_seed(seed))
        return holder["noise"]

    torch.randn_like = fake
    try:
        with torch.no_grad():
            r = fn()
    finally:
        torch.randn_like = real
    return r, holder["noise"]


def matcha_full(Matcha):
    model = Matcha(idim=45, **MATCHA_MAS_JSUT).eval()
    ref_sd = model.state_dict()
    sd = matcha_golden_tweaks(synth_state_dict(ref_sd, 0))
    model.load_state_dict(sd)
    text = torch.randint(1, 45, (18,), generator=torch.Generator().manual_seed(21))
    r, noise = with_noise(500, lambda: model.inference(text, n_timesteps=10, temperature=0.667))
    out = {"keys": json.dumps([[k, list(v.shape)] for k, v in ref_sd.items()]), "n_timesteps": np.int64(10),
           "temperature": np.float32(0.667), "u0_text": np_(text), "u0_noise": np_(noise[0].t()),
           "u0_feat_gen": np_(r["feat_gen"]), "u0_duration": np_(r["duration"])}
    print("matcha_jsut: frames", r["feat_gen"].shape[0], "absmax", float(r["feat_gen"].abs().max()))
    return out, model


def vits_full(VITS):
    model = VITS(idim=45, spk_embed_dim=192, **VITS_JSUT).eval()
    ref_sd = model.state_dict()
    model.load_state_dict(synth_state_dict(ref_sd, 0))
    out = {"keys": json.dumps([[k, list(v.shape)] for k, v in ref_sd.items()])}
    g = torch.Generator().manual_seed(22)
    for u, n in enumerate((12, 31)):
        text = torch.randint(1, 45, (n,), generator=g)
        spemb = torch.randn(192, generator=torch.Generator().manual_seed(600 + u))
        r, noise = with_noise(700 + u, lambda: model.inference(text, spembs=spemb))
        out.update({f"u{u}_text": np_(text), f"u{u}_spemb": np_(spemb), f"u{u}_noise": np_(noise[0].t()),
                    f"u{u}_feat_gen": np_(r["feat_gen"]), f"u{u}_duration": np_(r["duration"])})
        print(f"vits_jsut u{u}: frames", r["feat_gen"].shape[0], "absmax", float(r["feat_gen"].abs().max()))
    return out, model


MATCHA_TTS1_SMALL = dict(G.MATCHA_SMALL)


def matcha_tts1():
    for m in [k for k in sys.modules if k == "jatts.models.matchatts"]:
        del sys.modules[m]
    from jatts.models.matchatts import MatchaTTS

    model = MatchaTTS(idim=20, **MATCHA_TTS1_SMALL).eval()
    ref_sd = model.state_dict()
    sd = matcha_golden_tweaks(synth_state_dict(ref_sd, 4))
    model.load_state_dict(sd)
    out = {"keys": json.dumps([[k, list(v.shape)] for k, v in ref_sd.items()]), "config": json.dumps(MATCHA_TTS1_SMALL),
           "n_timesteps": np.int64(4), "temperature": np.float32(0.667)}
    g = torch.Generator().manual_seed(23)
    for u, n in enumerate((13, 22)):
        text = torch.randint(1, 20, (n,), generator=g)
        r, noise = with_noise(800 + u, lambda: model.inference(text, n_timesteps=4, temperature=0.667))
        out.update({f"u{u}_text": np_(text), f"u{u}_noise": np_(noise[0].t()), f"u{u}_feat_gen": np_(r["feat_gen"]),
                    f"u{u}_duration": np_(r["duration"])})
        print(f"matcha_tts1 u{u}: frames", r["feat_gen"].shape[0], "sum(d)", int(r["duration"].sum()),
              "absmax", float(r["feat_gen"].abs().max()))
    return out


def fs2_forward(FastSpeech2):
    """forward(): padded ragged batch, ground-truth durations / pitch / energy (fastspeech2.py:473-564)."""
    model = FastSpeech2(idim=20, **FS2_SMALL).eval()
    ref_sd = model.state_dict()
    model.load_state_dict(synth_state_dict(ref_sd, 0))
    g = torch.Generator().manual_seed(31)
    ilens = torch.tensor([17, 9, 23])
    B, Tm = len(ilens), int(ilens.max())
    text = torch.zeros(B, Tm, dtype=torch.long)
    ds = torch.zeros(B, Tm, dtype=torch.long)
    ps, es = torch.zeros(B, Tm, 1), torch.zeros(B, Tm, 1)
    for b, n in enumerate(ilens.tolist()):
        text[b, :n] = torch.randint(1, 20, (n,), generator=g)
        ds[b, :n] = torch.randint(0, 5, (n,), generator=g)
        ps[b, :n] = torch.randn(n, 1, generator=g)
        es[b, :n] = torch.randn(n, 1, generator=g)
    olens = ds.sum(1)
    feats = torch.randn(B, int(olens.max()), 80, generator=g)
    with torch.no_grad():
        r = model(text, ilens, feats, olens, ds, ilens, ps, ilens, es, ilens)
    out = {"keys": json.dumps([[k, list(v.shape)] for k, v in ref_sd.items()]), "text": np_(text), "text_lengths": np_(ilens),
           "feats": np_(feats), "feats_lengths": np_(olens), "durations": np_(ds), "pitch": np_(ps), "energy": np_(es)}
    for k in ("before_outs", "after_outs", "d_outs", "p_outs", "e_outs", "ys", "olens"):
        out["ref_" + k] = np_(r[k])
    print("fs2_forward: olens", olens.tolist(), {k: tuple(r[k].shape) for k in ("before_outs", "d_outs", "p_outs")})
    return out


def install_additive_mask_attention():
    import jatts.modules.matchatts.transformer as T

    def attn_forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
        B, L, _ = hidden_states.shape
        sp = lambda t: t.view(B, L, self.heads, -1).transpose(1, 2)  # noqa: E731
        q, k, v = sp(self.to_q(hidden_states)), sp(self.to_k(hidden_states)), sp(self.to_v(hidden_states))
        sc = q @ k.transpose(-1, -2) * self.scale
        if attention_mask is not None:
            sc = sc + attention_mask.to(sc.dtype)[:, None, None, :]
        return self.to_out[0]((torch.softmax(sc, dim=-1) @ v).transpose(1, 2).reshape(B, L, -1))

    T.Attention.forward = attn_forward


def matcha_forward(Matcha):
    """MatchaTTS_MAS.forward() (matchatts_mas.py:337-550, is_inference=False): alignment module + MAS + masked Gaussian
    upsampling + CFM loss on a padded ragged batch, with the two random draws of CFM.compute_loss (t ~ U, z ~ N) injected.
    The diffusers attention stand-in adds the (B, T) float mask to the scores, as diffusers does with `attention_mask`
    [recalled: Attention.prepare_attention_mask -> additive bias] -- unpinned, like the rest of that class."""
    install_additive_mask_attention()
    model = Matcha(idim=20, **G.MATCHA_SMALL).eval()
    ref_sd = model.state_dict()
    sd = matcha_golden_tweaks(synth_state_dict(ref_sd, 3))
    model.load_state_dict(sd)
    g = torch.Generator().manual_seed(41)
    ilens, olens = torch.tensor([14, 9, 17]), torch.tensor([53, 40, 62])
    B, Tm, To = 3, int(ilens.max()), int(olens.max())
    text = torch.zeros(B, Tm, dtype=torch.long)
    feats = torch.zeros(B, To, 80)
    for b in range(B):
        text[b, : ilens[b]] = torch.randint(1, 20, (int(ilens[b]),), generator=g)
        feats[b, : olens[b]] = torch.randn(int(olens[b]), 80, generator=g)
    t_fix = torch.rand(B, 1, 1, generator=g)
    real_rand = torch.rand
    torch.rand = lambda *a, **k: t_fix.clone()
    try:
        r, z = with_noise(900, lambda: model(text, ilens, feats, olens))
    finally:
        torch.rand = real_rand
    out = {"keys": json.dumps([[k, list(v.shape)] for k, v in ref_sd.items()]), "config": json.dumps(G.MATCHA_SMALL),
           "text": np_(text), "text_lengths": np_(ilens), "feats": np_(feats), "feats_lengths": np_(olens),
           "t": np_(t_fix.reshape(-1)), "z": np_(z.permute(0, 2, 1))}                 # z as (B, T, odim)
    for k in ("d_outs", "ys", "hs", "olens_in", "log_p_attn", "ds", "cfm_loss", "bin_loss"):
        out["ref_" + k] = np_(torch.as_tensor(r[k]))
    print("matcha_forward: cfm_loss", float(r["cfm_loss"]), "bin_loss", float(r["bin_loss"]), "olens_in", r["olens_in"].tolist(),
          "ds sums", r["ds"].sum(1).tolist())
    return out


def matcha_tts1_forward():
    """tts1 MatchaTTS.forward() (matchatts.py:317-480): ground-truth durations + hard LengthRegulator + CFM loss."""
    from jatts.models.matchatts import MatchaTTS

    install_additive_mask_attention()
    model = MatchaTTS(idim=20, **MATCHA_TTS1_SMALL).eval()
    ref_sd = model.state_dict()
    model.load_state_dict(matcha_golden_tweaks(synth_state_dict(ref_sd, 4)))
    g = torch.Generator().manual_seed(61)
    ilens = torch.tensor([11, 15, 8])
    B, Tm = 3, int(ilens.max())
    text = torch.zeros(B, Tm, dtype=torch.long)
    ds = torch.zeros(B, Tm, dtype=torch.long)
    for b in range(B):
        text[b, : ilens[b]] = torch.randint(1, 20, (int(ilens[b]),), generator=g)
        ds[b, : ilens[b]] = torch.randint(1, 6, (int(ilens[b]),), generator=g)
    olens = ds.sum(1)
    feats = torch.zeros(B, int(olens.max()), 80)
    for b in range(B):
        feats[b, : olens[b]] = torch.randn(int(olens[b]), 80, generator=g)
    t_fix = torch.rand(B, 1, 1, generator=g)
    real_rand = torch.rand
    torch.rand = lambda *a, **k: t_fix.clone()
    try:
        r, z = with_noise(960, lambda: model(text, ilens, feats, olens, ds, ilens))
    finally:
        torch.rand = real_rand
    out = {"keys": json.dumps([[k, list(v.shape)] for k, v in ref_sd.items()]), "config": json.dumps(MATCHA_TTS1_SMALL),
           "text": np_(text), "text_lengths": np_(ilens), "feats": np_(feats), "feats_lengths": np_(olens), "durations": np_(ds),
           "t": np_(t_fix.reshape(-1)), "z": np_(z.permute(0, 2, 1))}
    for k in ("d_outs", "ys", "hs", "olens_in", "cfm_loss"):
        out["ref_" + k] = np_(torch.as_tensor(r[k]))
    print("matcha_tts1_forward: cfm_loss", float(r["cfm_loss"]), "olens", olens.tolist(), "olens_in", r["olens_in"].tolist())
    return out


def vits_forward(VITS):
    """VITS.forward() (vits.py:342-579, is_inference=False): posterior encoder + forward flow + alignment module / MAS +
    masked Gaussian upsampling + decoder on a padded ragged batch; the posterior sampling noise is injected."""
    model = VITS(idim=20, **G.VITS_SMALL).eval()
    ref_sd = model.state_dict()
    model.load_state_dict(synth_state_dict(ref_sd, 2))
    g = torch.Generator().manual_seed(51)
    ilens, olens = torch.tensor([12, 16, 7]), torch.tensor([45, 58, 31])
    B, Tm, To = 3, int(ilens.max()), int(olens.max())
    text = torch.zeros(B, Tm, dtype=torch.long)
    feats = torch.zeros(B, To, 80)
    for b in range(B):
        text[b, : ilens[b]] = torch.randint(1, 20, (int(ilens[b]),), generator=g)
        feats[b, : olens[b]] = torch.randn(int(olens[b]), 80, generator=g)
    spembs = torch.randn(B, 16, generator=g)
    r, noise = with_noise(950, lambda: model(text, ilens, feats, olens, spembs=spembs))
    out = {"keys": json.dumps([[k, list(v.shape)] for k, v in ref_sd.items()]), "config": json.dumps(G.VITS_SMALL),
           "text": np_(text), "text_lengths": np_(ilens), "feats": np_(feats), "feats_lengths": np_(olens), "spembs": np_(spembs),
           "noise": np_(noise.permute(0, 2, 1))}                                        # (B, T_feats, adim)
    for k in ("outs", "d_outs", "ys", "hs", "olens_in", "bin_loss", "log_p_attn", "ds", "m_p", "logs_p", "z", "y_mask", "z_p",
              "m_q", "logs_q"):
        out["ref_" + k] = np_(torch.as_tensor(r[k]))
    print("vits_forward:", {k: tuple(out["ref_" + k].shape) for k in ("outs", "hs", "m_p", "z", "z_p", "m_q", "y_mask", "d_outs")},
          "bin_loss", float(r["bin_loss"]))
    return out



def with_noises(seed, fn):
    """Like with_noise, for several randn_like draws: draw i is seeded seed + i; returns (result, [draws])."""
    draws = []
    real = torch.randn_like

    def fake(t, *a, **k):
        draws.append(torch.randn(t.shape, generator=torch.Generator().manual_seed(seed + len(draws))))
        return draws[-1]

    torch.randn_like = fake
    try:
        with torch.no_grad():
            r = fn()
    finally:
        torch.randn_like = real
    return r, draws


def inference_with_feats(Matcha, VITS):
    """inference(text, feats=...): the alignment branch of MatchaTTS_MAS (log_p_attn, ds) and VITS (+ outs_bar, the posterior
    reconstruction) -- matchatts_mas.py:449-455, vits.py:449-455,546-556 -> infer_feats_small.npz."""
    g = torch.Generator().manual_seed(81)
    out = {}
    install_additive_mask_attention()
    m = Matcha(idim=20, **G.MATCHA_SMALL).eval()
    sd = m.state_dict()
    out["matcha_keys"] = json.dumps([[k, list(v.shape)] for k, v in sd.items()])
    out["matcha_config"] = json.dumps(G.MATCHA_SMALL)
    m.load_state_dict(matcha_golden_tweaks(synth_state_dict(sd, 3)))
    text = torch.randint(1, 20, (13,), generator=g)
    feats = torch.randn(47, 80, generator=g)
    r, z = with_noise(970, lambda: m.inference(text, feats=feats, n_timesteps=4, temperature=0.667))
    out.update(m_text=np_(text), m_feats=np_(feats), m_noise=np_(z[0].t()), m_feat_gen=np_(r["feat_gen"]), m_duration=np_(r["duration"]),
               m_log_p_attn=np_(r["log_p_attn"]), m_ds=np_(r["ds"]))
    v = VITS(idim=20, **G.VITS_SMALL).eval()
    sd = v.state_dict()
    out["vits_keys"] = json.dumps([[k, list(x.shape)] for k, x in sd.items()])
    out["vits_config"] = json.dumps(G.VITS_SMALL)
    v.load_state_dict(synth_state_dict(sd, 2))
    text = torch.randint(1, 20, (11,), generator=g)
    feats = torch.randn(39, 80, generator=g)
    spemb = torch.randn(16, generator=g)
    r, draws = with_noises(980, lambda: v.inference(text, feats=feats, spembs=spemb))
    assert len(draws) == 2                                    # prior sampling, then the posterior encoder of the reconstruction
    out.update(v_text=np_(text), v_feats=np_(feats), v_spemb=np_(spemb), v_noise=np_(draws[0][0].t()), v_post_noise=np_(draws[1][0].t()),
               v_feat_gen=np_(r["feat_gen"]), v_duration=np_(r["duration"]), v_log_p_attn=np_(r["log_p_attn"]), v_ds=np_(r["ds"]),
               v_outs_bar=np_(r["outs_bar"]))
    print("infer_feats:", {k: tuple(out[k].shape) for k in ("m_log_p_attn", "m_ds", "v_log_p_attn", "v_outs_bar", "v_feat_gen")})
    return out


def fs2_teacher_forcing(FastSpeech2):
    """FastSpeech2.inference(use_teacher_forcing=True, durations, pitch, energy) (fastspeech2.py:704-717)."""
    model = FastSpeech2(idim=20, **FS2_SMALL).eval()
    ref_sd = model.state_dict()
    model.load_state_dict(synth_state_dict(ref_sd, 0))
    g = torch.Generator().manual_seed(71)
    T = 19
    text = torch.randint(1, 20, (T,), generator=g)
    d = torch.randint(0, 5, (T,), generator=g)
    p, e = torch.randn(T, 1, generator=g), torch.randn(T, 1, generator=g)
    with torch.no_grad():
        r = model.inference(text, durations=d, pitch=p, energy=e, use_teacher_forcing=True)
    np.savez_compressed(os.path.join(HERE, "fs2_teacher_forcing_small.npz"), keys=json.dumps([[k, list(v.shape)] for k, v in ref_sd.items()]),
                        text=np_(text), durations=np_(d), pitch=np_(p), energy=np_(e), **{"ref_" + k: np_(v) for k, v in r.items()})


def fs2_losses():
    """The reference's FastSpeech2 criterion (jatts/losses: MelLoss/L1Loss, DurationPredictorLoss, PitchLoss, EnergyLoss, as
    trainers/fastspeech2.py:62-84 calls them) on the forward() golden -> tests/golden/fs2_losses_small.npz."""
    import importlib
    l1 = importlib.import_module("jatts.losses.l1l2_loss")
    dl = importlib.import_module("jatts.losses.duration_predictor_loss")
    vl = importlib.import_module("jatts.losses.variance_predictor_loss")
    z = np.load(os.path.join(HERE, "fs2_forward_small.npz"))
    t = lambda k: torch.tensor(z[k])  # noqa: E731
    il, ol = t("text_lengths"), t("feats_lengths")
    np.savez(os.path.join(HERE, "fs2_losses_small.npz"),
             mel_loss=np.float32(l1.MelLoss()(t("ref_after_outs"), t("ref_before_outs"), t("ref_ys"), ol)),
             duration_loss=np.float32(dl.DurationPredictorLoss()(t("ref_d_outs"), t("durations"), il)),
             pitch_loss=np.float32(vl.PitchLoss()(t("ref_p_outs"), t("pitch"), il)),
             energy_loss=np.float32(vl.EnergyLoss()(t("ref_e_outs"), t("energy"), il)))


def main():
    torch.set_num_threads(8)
    FastSpeech2 = G.import_reference()
    np.savez_compressed(os.path.join(HERE, "fs2_forward_small.npz"), **fs2_forward(FastSpeech2))
    fs2_losses()
    fs2_teacher_forcing(FastSpeech2)
    VITS = G.import_reference_vits()
    from oracle.vits_oracle import vits_inference
    vz, vmodel = vits_full(VITS)
    np.savez_compressed(os.path.join(HERE, "vits_jsut.npz"), **vz)
    vsd = vmodel.state_dict()
    for u in range(2):
        o = vits_inference(vsd, torch.tensor(vz[f"u{u}_text"]), 2, 2, torch.tensor(vz[f"u{u}_spemb"]), torch.tensor(vz[f"u{u}_noise"]))
        print(f"vits_jsut u{u}: oracle-vs-ref mel max|d| =", float((o["feat_gen"] - torch.tensor(vz[f"u{u}_feat_gen"])).abs().max()),
              "dur equal:", bool((o["duration"].numpy() == vz[f"u{u}_duration"]).all()))
    del vmodel, vsd
    Matcha = G.import_reference_matcha()
    from oracle.matcha_oracle import matcha_inference
    mz, mmodel = matcha_full(Matcha)
    np.savez_compressed(os.path.join(HERE, "matcha_jsut.npz"), **mz)
    o = matcha_inference(mmodel.state_dict(), torch.tensor(mz["u0_text"]), 2, 2, torch.tensor(mz["u0_noise"]), n_timesteps=10,
                         temperature=0.667)
    print("matcha_jsut: oracle-vs-ref mel max|d| =", float((o["feat_gen"] - torch.tensor(mz["u0_feat_gen"])).abs().max()),
          "dur equal:", bool((o["duration"].numpy() == mz["u0_duration"]).all()))
    del mmodel
    np.savez_compressed(os.path.join(HERE, "matcha_tts1_small.npz"), **matcha_tts1())
    np.savez_compressed(os.path.join(HERE, "matcha_forward_small.npz"), **matcha_forward(Matcha))
    np.savez_compressed(os.path.join(HERE, "vits_forward_small.npz"), **vits_forward(VITS))
    np.savez_compressed(os.path.join(HERE, "matcha_tts1_forward_small.npz"), **matcha_tts1_forward())
    np.savez_compressed(os.path.join(HERE, "infer_feats_small.npz"), **inference_with_feats(Matcha, VITS))
    for f in ("matcha_forward_small.npz", "vits_forward_small.npz", "matcha_tts1_forward_small.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)))
    for f in ("fs2_forward_small.npz", "vits_jsut.npz", "matcha_jsut.npz", "matcha_tts1_small.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
