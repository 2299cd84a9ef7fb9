#!/usr/bin/env python3
"""Generate golden vectors by importing the REAL reference from /root/reference.

Run in the build container only (the reference never travels to the GPU box):
    python tests/golden/make_golden.py
Outputs small .npz fixtures next to this file.  Inputs are data (token ids, seeds,
durations); weights are NOT stored — they are rebuilt from
jatts_amd.synthetic.synth_tensor(name, shape, seed).

Shims: `typeguard` is not installed (decorators become identity); the package
`jatts.models.__init__` pulls numba/diffusers/x_transformers, so an empty package
is pre-registered and only `jatts.models.fastspeech2` is imported; for
Vocoder.decode the absent `parallel_wavegan`/`h5py` are replaced by a recorder so
that ONLY the in-repo normalisation + call contract (vocoder.py:17-67) is captured.
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from jatts_amd.synthetic import FS2_JSUT, FS2_SMALL, matcha_golden_tweaks, synth_state_dict  # noqa: E402


def import_reference():
    sys.path.insert(0, REF)
    tg = types.ModuleType("typeguard")
    tg.typechecked = lambda f=None, **k: f if f is not None else (lambda g: g)
    sys.modules["typeguard"] = tg
    import jatts  # noqa: F401

    pkg = types.ModuleType("jatts.models")
    pkg.__path__ = [os.path.join(REF, "jatts", "models")]
    sys.modules["jatts.models"] = pkg
    from jatts.models.fastspeech2 import FastSpeech2

    return FastSpeech2


def import_reference_vits():
    """VITS needs no-op stubs for numba / conformer / diffusers (none executes at inference)."""
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _T:
        def __getitem__(self, k):
            return self

        def __call__(self, *a, **k):
            return self

    class _Any:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return self

        def __getattr__(self, k):
            return _Any()

    t = _T()
    stub("numba", jit=lambda *a, **k: (lambda f: f), float64=t, float32=t, int8=t, int32=t, int64=t, boolean=t)
    stub("conformer", ConformerBlock=object)
    for n in ("diffusers", "diffusers.models", "diffusers.models.activations", "diffusers.models.attention",
              "diffusers.models.attention_processor", "diffusers.models.lora", "diffusers.utils",
              "diffusers.utils.torch_utils", "diffusers.models.embeddings", "diffusers.models.normalization"):
        stub(n).__getattr__ = lambda k: _Any
    from jatts.models.vits import VITS

    return VITS


def import_reference_matcha():
    """MatchaTTS_MAS with diffusers' Attention / LoRACompatibleLinear replaced by a standard SDPA module
    (diffusers is absent; see oracle/matcha_oracle.py: that piece is parity-unpinned)."""
    import torch.nn as nn

    class Attention(nn.Module):
        def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, dropout=0.0, bias=False,
                     upcast_attention=False):
            super().__init__()
            inner = heads * dim_head
            self.heads, self.scale = heads, dim_head ** -0.5
            self.to_q = nn.Linear(query_dim, inner, bias=bias)
            self.to_k = nn.Linear(query_dim, inner, bias=bias)
            self.to_v = nn.Linear(query_dim, inner, bias=bias)
            self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(dropout)])

        def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
            B, T, _ = hidden_states.shape
            sp = lambda t: t.view(B, T, self.heads, -1).transpose(1, 2)  # noqa: E731
            q, k, v = sp(self.to_q(hidden_states)), sp(self.to_k(hidden_states)), sp(self.to_v(hidden_states))
            a = torch.softmax(q @ k.transpose(-1, -2) * self.scale, dim=-1)
            return self.to_out[0]((a @ v).transpose(1, 2).reshape(B, T, -1))

    ident = lambda c: c  # noqa: E731
    sys.modules["diffusers.models.attention_processor"].Attention = Attention
    sys.modules["diffusers.models.lora"].LoRACompatibleLinear = nn.Linear
    sys.modules["diffusers.utils.torch_utils"].maybe_allow_in_graph = ident
    for n in ("GEGLU", "GELU", "AdaLayerNorm", "AdaLayerNormZero", "ApproximateGELU"):
        setattr(sys.modules["diffusers.models.attention"], n, type(n, (nn.Module,), {}))
    for m in [k for k in sys.modules if k.startswith("jatts.modules.matchatts") or k == "jatts.models.matchatts_mas"]:
        del sys.modules[m]
    from jatts.models.matchatts_mas import MatchaTTS_MAS

    return MatchaTTS_MAS


MATCHA_SMALL = dict(odim=80, adim=64, aheads=2, elayers=2, eunits=128, positionwise_layer_type="conv1d",
                    positionwise_conv_kernel_size=3, encoder_type="conformer", duration_predictor_layers=2,
                    duration_predictor_chans=64, duration_predictor_kernel_size=3, conformer_enc_kernel_size=7,
                    decoder_channels=[128, 128], decoder_attention_head_dim=64, decoder_n_blocks=1,
                    decoder_num_mid_blocks=2, decoder_num_heads=2, decoder_act_fn="snakebeta")


def run_matcha(Matcha, texts, n_timesteps=4, temperature=0.667):
    model = Matcha(idim=20, **MATCHA_SMALL).eval()
    ref_sd = model.state_dict()
    sd = matcha_golden_tweaks(synth_state_dict(ref_sd, 3))
    model.load_state_dict(sd)
    out = {"keys": json.dumps([[k, list(v.shape)] for k, v in ref_sd.items()]), "config": json.dumps(MATCHA_SMALL),
           "n_timesteps": np.int64(n_timesteps), "temperature": np.float32(temperature)}
    real = torch.randn_like
    for u, text in enumerate(texts):
        holder = {}

        def fake(t, *a, **k):
            holder["noise"] = torch.randn(t.shape, generator=torch.Generator().manual_seed(400 + u))
            return holder["noise"]

        torch.randn_like = fake
        try:
            with torch.no_grad():
                r = model.inference(text, n_timesteps=n_timesteps, temperature=temperature)
        finally:
            torch.randn_like = real
        out[f"u{u}_text"] = np_(text)
        out[f"u{u}_noise"] = np_(holder["noise"][0].t())   # (T', odim)
        out[f"u{u}_feat_gen"] = np_(r["feat_gen"])
        out[f"u{u}_duration"] = np_(r["duration"])
    return out, model, sd


VITS_SMALL = dict(odim=80, adim=64, aheads=2, text_encoder_blocks=2, text_encoder_attention_heads=2, dlayers=2,
                  dunits=128, flow_flows=2, flow_layers=2, posterior_encoder_layers=2, duration_predictor_chans=64,
                  spk_embed_dim=16)


def run_vits(VITS, texts):
    model = VITS(idim=20, **VITS_SMALL).eval()
    ref_sd = model.state_dict()
    sd = synth_state_dict(ref_sd, 2)
    # the reference zero-initialises the coupling projections (residual_coupling.py:173-174): give the
    # golden model non-trivial ones so the flow is actually exercised
    model.load_state_dict(sd)
    out = {"keys": json.dumps([[k, list(v.shape)] for k, v in ref_sd.items()]),
           "config": json.dumps(VITS_SMALL)}
    real_randn_like = torch.randn_like
    for u, text in enumerate(texts):
        spemb = torch.randn(16, generator=torch.Generator().manual_seed(200 + u))
        holder = {}

        def fake_randn_like(t, *a, **k):
            holder["noise"] = torch.randn(t.shape, generator=torch.Generator().manual_seed(300 + u))
            return holder["noise"]

        torch.randn_like = fake_randn_like
        try:
            with torch.no_grad():
                r = model.inference(text, spembs=spemb)
        finally:
            torch.randn_like = real_randn_like
        out[f"u{u}_text"] = np_(text)
        out[f"u{u}_spemb"] = np_(spemb)
        out[f"u{u}_noise"] = np_(holder["noise"][0].t())  # (T_feats, A)
        out[f"u{u}_feat_gen"] = np_(r["feat_gen"])
        out[f"u{u}_duration"] = np_(r["duration"])
    return out, model


def np_(t):
    return t.detach().cpu().numpy()


def run_fs2(FastSpeech2, cfg, idim, seed, texts, extra_kwargs=None, with_taps=False, spk_dim=None):
    kw = dict(cfg)
    if spk_dim:
        kw.update(spk_embed_dim=spk_dim, spk_embed_integration_type="add")
    model = FastSpeech2(idim=idim, **kw).eval()
    ref_sd = model.state_dict()
    sd = synth_state_dict(ref_sd, seed)
    model.load_state_dict(sd)
    out = {"keys": json.dumps([[k, list(v.shape)] for k, v in ref_sd.items()])}
    taps = {}
    if with_taps:
        def hook(name):
            def fn(_m, _inp, o):
                o = o[0] if isinstance(o, tuple) else o
                o = o[0] if isinstance(o, tuple) else o
                taps.setdefault(name, []).append(np_(o)[0] if o.dim() == 3 else np_(o))
            return fn
        model.encoder.register_forward_hook(hook("encoder_out"))
        model.decoder.register_forward_hook(hook("decoder_out"))
        model.encoder.encoders[0].register_forward_hook(hook("enc_layer0"))
        model.encoder.encoders[0].self_attn.register_forward_hook(hook("enc_layer0_attn"))
        model.encoder.encoders[0].conv_module.register_forward_hook(hook("enc_layer0_convmod"))
        model.encoder.encoders[0].feed_forward_macaron.register_forward_hook(hook("enc_layer0_ffm"))
        model.duration_predictor.linear.register_forward_hook(hook("log_duration"))
        model.length_regulator.register_forward_hook(hook("lr_out"))
        model.feat_out.register_forward_hook(hook("before"))
    for u, text in enumerate(texts):
        kwargs = dict(extra_kwargs[u]) if extra_kwargs else {}
        spemb = None
        if spk_dim:
            spemb = torch.randn(spk_dim, generator=torch.Generator().manual_seed(100 + u))
            out[f"u{u}_spemb"] = np_(spemb)
        with torch.no_grad():
            r = model.inference(text, spembs=spemb, **kwargs)
        out[f"u{u}_text"] = np_(text)
        out[f"u{u}_feat_gen"] = np_(r["feat_gen"])
        out[f"u{u}_duration"] = np_(r["duration"])
        out[f"u{u}_pitch"] = np_(r["pitch"])
        out[f"u{u}_energy"] = np_(r["energy"])
        if "alpha" in kwargs:
            out[f"u{u}_alpha"] = np.float32(kwargs["alpha"])
        for k, v in taps.items():
            out[f"u{u}_{k}"] = v[-1]
        taps.clear()
    return out, model


def lr_cases():
    from jatts.modules.length_regulator import LengthRegulator

    lr = LengthRegulator()
    g = torch.Generator().manual_seed(7)
    out = {}
    cases = [
        (torch.tensor([[2, 0, 3, 1, 0]]), 1.0),
        (torch.tensor([[2, 0, 3, 1, 0]]), 1.5),
        (torch.zeros(1, 5, dtype=torch.long), 1.0),
        (torch.randint(0, 9, (1, 37), generator=g), 1.0),
        (torch.randint(0, 9, (1, 37), generator=g), 0.7),
        (torch.randint(0, 9, (1, 64), generator=g), 2.5),  # x.5 products -> half-to-even
        (torch.randint(0, 4, (3, 11), generator=g), 1.0),  # batched, ragged output lengths
        (torch.tensor([[0, 0, 0], [1, 2, 0]]), 1.0),       # one all-zero row, batch sum != 0
        (torch.zeros(2, 4, dtype=torch.long), 1.0),        # whole batch zero
    ]
    for n, (ds, alpha) in enumerate(cases):
        B, T = ds.shape
        xs = torch.arange(B * T * 3, dtype=torch.float32).reshape(B, T, 3) + 1.0
        y = lr(xs, ds.clone(), alpha)
        out[f"c{n}_ds"] = np_(ds)
        out[f"c{n}_alpha"] = np.float32(alpha)
        out[f"c{n}_xs"] = np_(xs)
        out[f"c{n}_out"] = np_(y)
    out["n_cases"] = np.int64(len(cases))
    return out


def mask_cases():
    from jatts.modules.utils import make_non_pad_mask, make_pad_mask

    out = {}
    for n, lens in enumerate([[5, 3, 2], [1], [4, 4], [7, 1, 3, 7]]):
        out[f"m{n}_lens"] = np.array(lens, dtype=np.int64)
        out[f"m{n}_pad"] = np_(make_pad_mask(lens))
        out[f"m{n}_nonpad"] = np_(make_non_pad_mask(lens))
    out["n_cases"] = np.int64(4)
    return out


def vocoder_decode_case():
    """Capture Vocoder.decode's normalisation and call contract with a recording generator."""
    captured = {}

    class FakeGen:
        def remove_weight_norm(self):
            captured["remove_weight_norm"] = True

        def eval(self):
            return self

        def to(self, _d):
            return self

        def inference(self, c, normalize_before=False):
            captured["c"] = c.clone()
            captured["normalize_before"] = normalize_before
            return torch.zeros(c.shape[0] * 4, 1)

    pwg = types.ModuleType("parallel_wavegan")
    pwg_u = types.ModuleType("parallel_wavegan.utils")
    pwg_u.load_model = lambda ckpt, cfg: FakeGen()
    sys.modules["parallel_wavegan"] = pwg
    sys.modules["parallel_wavegan.utils"] = pwg_u
    g = torch.Generator().manual_seed(11)
    voc_mean = torch.randn(80, generator=g).numpy()
    voc_scale = (0.5 + torch.rand(80, generator=g)).numpy()
    ju = types.ModuleType("jatts.utils")
    ju.read_hdf5 = lambda path, key: {"mean": voc_mean, "scale": voc_scale}[key]
    sys.modules["jatts.utils"] = ju
    import importlib

    vmod = importlib.import_module("jatts.vocoder.vocoder")
    import tempfile, yaml

    with tempfile.NamedTemporaryFile("w", suffix=".yml", delete=False) as f:
        yaml.safe_dump({"sampling_rate": 24000}, f)
        cfg_path = f.name
    trg = {"mean": torch.randn(80, generator=g).numpy(), "scale": (0.5 + torch.rand(80, generator=g)).numpy()}
    voc = vmod.Vocoder("ckpt", cfg_path, "stats.h5", torch.device("cpu"), trg_stats=trg)
    c = torch.randn(13, 80, generator=g)
    y, sr = voc.decode(c)
    os.unlink(cfg_path)
    assert captured["normalize_before"] is False and captured["remove_weight_norm"]
    return {
        "c": np_(c), "c_norm": np_(captured["c"]), "voc_mean": voc_mean, "voc_scale": voc_scale,
        "trg_mean": trg["mean"], "trg_scale": trg["scale"], "sr": np.int64(sr), "y_len": np.int64(y.numel()),
    }


def main():
    torch.set_num_threads(8)
    FastSpeech2 = import_reference()
    from oracle.fs2_oracle import fs2_inference

    g = torch.Generator().manual_seed(3)
    # --- small config, with taps, 3 utterances incl. alpha != 1
    texts = [torch.randint(1, 20, (n,), generator=g) for n in (24, 9, 33)]
    small, model = run_fs2(FastSpeech2, FS2_SMALL, 20, 0, texts,
                           extra_kwargs=[{}, {"alpha": 1.3}, {}], with_taps=True)
    np.savez_compressed(os.path.join(HERE, "fs2_small.npz"), **small)
    sd = model.state_dict()
    for u, t in enumerate(texts):
        o = fs2_inference(sd, t, 2, alpha=float(small.get(f"u{u}_alpha", 1.0)))
        print(f"small u{u}: oracle-vs-ref mel max|d| =",
              float((o["feat_gen"] - torch.tensor(small[f"u{u}_feat_gen"])).abs().max()),
              "dur equal:", bool((o["duration"].numpy() == small[f"u{u}_duration"]).all()))
    # --- small config, multi-speaker ("add" integration)
    spk, model = run_fs2(FastSpeech2, FS2_SMALL, 20, 1, texts[:2], spk_dim=16)
    np.savez_compressed(os.path.join(HERE, "fs2_small_spk.npz"), **spk)
    # --- full jsut config (weights rebuilt from seed), 2 short utterances
    g = torch.Generator().manual_seed(4)
    texts = [torch.randint(1, 45, (n,), generator=g) for n in (16, 40)]
    full, model = run_fs2(FastSpeech2, FS2_JSUT, 45, 0, texts, with_taps=False)
    np.savez_compressed(os.path.join(HERE, "fs2_jsut.npz"), **full)
    sd = model.state_dict()
    for u, t in enumerate(texts):
        o = fs2_inference(sd, t, 2)
        print(f"jsut u{u}: oracle-vs-ref mel max|d| =",
              float((o["feat_gen"] - torch.tensor(full[f"u{u}_feat_gen"])).abs().max()),
              "frames", full[f"u{u}_feat_gen"].shape[0])
    # --- mel-VITS (small config, speaker embedding, injected noise)
    VITS = import_reference_vits()
    from oracle.vits_oracle import vits_inference
    g = torch.Generator().manual_seed(5)
    texts = [torch.randint(1, 20, (n,), generator=g) for n in (14, 27)]
    vz, vmodel = run_vits(VITS, texts)
    np.savez_compressed(os.path.join(HERE, "vits_small.npz"), **vz)
    vsd = vmodel.state_dict()
    for u, t in enumerate(texts):
        o = vits_inference(vsd, t, 2, 2, torch.tensor(vz[f"u{u}_spemb"]), torch.tensor(vz[f"u{u}_noise"]))
        print(f"vits u{u}: oracle-vs-ref mel max|d| =",
              float((o["feat_gen"] - torch.tensor(vz[f"u{u}_feat_gen"])).abs().max()),
              "dur equal:", bool((o["duration"].numpy() == vz[f"u{u}_duration"]).all()),
              "frames", vz[f"u{u}_feat_gen"].shape[0])
    # --- Matcha-TTS (MAS variant), small config, injected noise
    Matcha = import_reference_matcha()
    from oracle.matcha_oracle import matcha_inference
    g = torch.Generator().manual_seed(6)
    texts = [torch.randint(1, 20, (n,), generator=g) for n in (11, 19)]
    mz, mmodel, msd = run_matcha(Matcha, texts)
    np.savez_compressed(os.path.join(HERE, "matcha_small.npz"), **mz)
    for u, t in enumerate(texts):
        o = matcha_inference(mmodel.state_dict(), t, 2, 2, torch.tensor(mz[f"u{u}_noise"]), n_timesteps=4)
        print(f"matcha u{u}: oracle-vs-ref mel max|d| =",
              float((o["feat_gen"] - torch.tensor(mz[f"u{u}_feat_gen"])).abs().max()),
              "dur equal:", bool((o["duration"].numpy() == mz[f"u{u}_duration"]).all()),
              "frames", mz[f"u{u}_feat_gen"].shape[0], "absmax", float(np.abs(mz[f"u{u}_feat_gen"]).max()))
    np.savez_compressed(os.path.join(HERE, "lr_kat.npz"), **lr_cases())
    np.savez_compressed(os.path.join(HERE, "mask_kat.npz"), **mask_cases())
    np.savez_compressed(os.path.join(HERE, "vocoder_decode.npz"), **vocoder_decode_case())
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
