"""CPU: the oracle against the golden vectors captured from the REAL reference
(tests/golden/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

from helpers import golden_state, load_golden, maxdiff
from oracle import fs2_oracle as O
from oracle import hifigan_oracle as HO
from oracle import lr_oracle as LR


@pytest.mark.parametrize("name,seed,heads", [("fs2_small.npz", 0, 2), ("fs2_jsut.npz", 0, 2)])
def test_fs2_oracle_matches_reference(name, seed, heads):
    z, keys = load_golden(name)
    sd = golden_state(keys, seed)
    u = 0
    while f"u{u}_text" in z.files:
        text = torch.tensor(z[f"u{u}_text"])
        alpha = float(z[f"u{u}_alpha"]) if f"u{u}_alpha" in z.files else 1.0
        taps = {}
        o = O.fs2_inference(sd, text, heads, alpha=alpha, taps=taps)
        assert np.array_equal(o["duration"].numpy(), z[f"u{u}_duration"])
        # same torch build + same thread-independent kernels: bit-exact here; 1e-5 leaves room for BLAS threading
        assert maxdiff(o["feat_gen"], z[f"u{u}_feat_gen"]) <= 1e-5
        assert maxdiff(o["pitch"], z[f"u{u}_pitch"]) <= 1e-5
        assert maxdiff(o["energy"], z[f"u{u}_energy"]) <= 1e-5
        if f"u{u}_encoder_out" in z.files:
            assert maxdiff(taps["encoder_out"], z[f"u{u}_encoder_out"]) <= 1e-5
            assert maxdiff(taps["layer0"], z[f"u{u}_enc_layer0"]) <= 1e-5
            assert maxdiff(taps["decoder_out"], z[f"u{u}_decoder_out"]) <= 1e-5
            assert maxdiff(o["before"], z[f"u{u}_before"]) <= 1e-5
            assert maxdiff(o["log_duration"], z[f"u{u}_log_duration"].reshape(-1)) <= 1e-5
        u += 1
    assert u >= 2


def test_fs2_oracle_speaker_embedding():
    z, keys = load_golden("fs2_small_spk.npz")
    sd = golden_state(keys, 1)
    for u in range(2):
        o = O.fs2_inference(sd, torch.tensor(z[f"u{u}_text"]), 2, spembs=torch.tensor(z[f"u{u}_spemb"]))
        assert np.array_equal(o["duration"].numpy(), z[f"u{u}_duration"])
        assert maxdiff(o["feat_gen"], z[f"u{u}_feat_gen"]) <= 1e-5


def test_vits_oracle_matches_reference():
    """mel-VITS (A16): oracle vs the real reference with injected noise (tests/golden/vits_small.npz)."""
    import json
    from oracle.vits_oracle import rel_shift_new, vits_inference
    z, keys = load_golden("vits_small.npz")
    sd = golden_state(keys, 2)
    assert json.loads(str(z["config"]))["spk_embed_dim"] == 16
    for u in range(2):
        o = vits_inference(sd, torch.tensor(z[f"u{u}_text"]), 2, 2, torch.tensor(z[f"u{u}_spemb"]),
                           torch.tensor(z[f"u{u}_noise"]))
        assert np.array_equal(o["duration"].numpy(), z[f"u{u}_duration"])
        assert maxdiff(o["feat_gen"], z[f"u{u}_feat_gen"]) <= 1e-5
    # new rel_shift == the diagonal index map the HIP kernel uses
    bd = torch.randn(2, 5, 9)
    want = torch.stack([torch.stack([bd[:, i, 4 - i + j] for j in range(5)], -1) for i in range(5)], 1)
    assert torch.equal(rel_shift_new(bd), want)


def test_matcha_oracle_matches_reference():
    """Matcha-TTS MAS (A14-A15): oracle vs the reference run with injected noise, 4 Euler steps."""
    import json
    from jatts_amd.synthetic import matcha_golden_tweaks
    from oracle.matcha_oracle import matcha_inference
    z, keys = load_golden("matcha_small.npz")
    sd = matcha_golden_tweaks(golden_state(keys, 3))
    for u in range(2):
        o = matcha_inference(sd, torch.tensor(z[f"u{u}_text"]), 2, 2, torch.tensor(z[f"u{u}_noise"]),
                             n_timesteps=int(z["n_timesteps"]), temperature=float(z["temperature"]))
        assert np.array_equal(o["duration"].numpy(), z[f"u{u}_duration"])
        assert maxdiff(o["feat_gen"], z[f"u{u}_feat_gen"]) <= 1e-5


def test_oracles_match_reference_at_full_width():
    """Round-2 fixtures (make_golden_r2.py): the BASELINE config-3 / config-5 models at their own width (U-Net 512/512 with
    attention head dim 256; VITS with a 192-d speaker embedding) and the tts1 MatchaTTS class, oracle vs the reference run."""
    from jatts_amd.synthetic import matcha_golden_tweaks
    from oracle.matcha_oracle import matcha_inference
    from oracle.vits_oracle import vits_inference
    z, keys = load_golden("matcha_jsut.npz")
    o = matcha_inference(matcha_golden_tweaks(golden_state(keys, 0)), torch.tensor(z["u0_text"]), 2, 2, torch.tensor(z["u0_noise"]),
                         n_timesteps=int(z["n_timesteps"]), temperature=float(z["temperature"]))
    assert np.array_equal(o["duration"].numpy(), z["u0_duration"]) and maxdiff(o["feat_gen"], z["u0_feat_gen"]) <= 1e-4
    z, keys = load_golden("vits_jsut.npz")
    sd = golden_state(keys, 0)
    for u in range(2):
        o = vits_inference(sd, torch.tensor(z[f"u{u}_text"]), 2, 2, torch.tensor(z[f"u{u}_spemb"]), torch.tensor(z[f"u{u}_noise"]))
        assert np.array_equal(o["duration"].numpy(), z[f"u{u}_duration"]) and maxdiff(o["feat_gen"], z[f"u{u}_feat_gen"]) <= 1e-4
    z, keys = load_golden("matcha_tts1_small.npz")
    sd = matcha_golden_tweaks(golden_state(keys, 4))
    for u in range(2):
        o = matcha_inference(sd, torch.tensor(z[f"u{u}_text"]), 2, 2, torch.tensor(z[f"u{u}_noise"]), n_timesteps=4, hard_lr=True)
        assert np.array_equal(o["duration"].numpy(), z[f"u{u}_duration"]) and maxdiff(o["feat_gen"], z[f"u{u}_feat_gen"]) <= 1e-4


def test_rel_shift_closed_form_equals_view_trick():
    g = torch.Generator().manual_seed(0)
    for T in (1, 2, 3, 7, 16):
        bd = torch.randn(2, T, T, generator=g)
        assert torch.equal(O.rel_shift_legacy(bd), O.rel_shift_closed_form(bd))


def test_length_regulator_kats():
    """C oracle vs the reference LengthRegulator outputs (incl. SURVEY §8 A9 KATs)."""
    z, _ = load_golden("lr_kat.npz")
    for n in range(int(z["n_cases"])):
        ds, alpha, xs, ref = z[f"c{n}_ds"], float(z[f"c{n}_alpha"]), z[f"c{n}_xs"], z[f"c{n}_out"]
        B, T = ds.shape
        out, olens = LR.gather(xs, ds, [T] * B, alpha)
        assert out.shape == ref.shape, (n, out.shape, ref.shape)
        assert np.array_equal(out, ref), n
    assert list(LR.frame_index([2, 0, 3, 1, 0])) == [0, 0, 2, 2, 2, 3]
    d, _ = LR.effective_durations(np.array([[2, 0, 3, 1, 0]]), [5], 1.5)
    assert list(LR.frame_index(d[0])) == [0, 0, 0, 2, 2, 2, 2, 3, 3]
    d, _ = LR.effective_durations(np.zeros((1, 5), dtype=np.int64), [5])
    assert list(LR.frame_index(d[0])) == [0, 1, 2, 3, 4]


def test_length_regulate_torch_matches_c_oracle():
    g = torch.Generator().manual_seed(5)
    for alpha in (1.0, 0.5, 1.5, 2.5):
        d = torch.randint(0, 7, (50,), generator=g)
        x = torch.randn(50, 4, generator=g)
        y, d_eff = O.length_regulate(x, d, alpha)
        out, olens = LR.gather(x.numpy()[None], d.numpy()[None], [50], alpha)
        assert np.array_equal(out[0], y.numpy())


def test_vocoder_decode_normalisation_golden():
    z, _ = load_golden("vocoder_decode.npz")
    c = HO.vocoder_normalize(torch.tensor(z["c"]), torch.tensor(z["trg_mean"]), torch.tensor(z["trg_scale"]),
                             torch.tensor(z["voc_mean"]), torch.tensor(z["voc_scale"]))
    assert maxdiff(c, z["c_norm"]) <= 1e-6


def test_hifigan_oracle_structure():
    """Structural invariants of the (unpinned) HiFi-GAN restatement: length, range, weight-norm folding."""
    sd = HO.random_hifigan_state(channels=64, upsample_scales=(4, 2), upsample_kernel_sizes=(8, 4), std=0.1)
    c = torch.randn(9, 80)
    y = HO.hifigan_generate(sd, c, (4, 2))
    assert y.shape == (9 * 8,) and float(y.abs().max()) <= 1.0
    wn = {}
    for k, v in sd.items():
        if k.endswith(".weight"):
            norm = v.reshape(v.shape[0], -1).norm(dim=1).reshape(-1, *([1] * (v.dim() - 1)))
            wn[k[:-6] + "weight_g"], wn[k[:-6] + "weight_v"] = norm, v * 3.0
        else:
            wn[k] = v
    y2 = HO.hifigan_generate(wn, c, (4, 2))
    assert maxdiff(y, y2) <= 1e-5


@pytest.mark.parametrize("case", ["v1", "w128", "two_blocks", "wn"])
def test_hifigan_oracle_matches_an_independent_implementation(case):
    """The generator restatement against waveforms produced by Hugging Face transformers' FastSpeech2ConformerHifiGan -- an independent
    public implementation of the same published network (tests/golden/make_golden_hifigan_xcheck.py explains why the two agree for the
    22.05 kHz V1 strides, and that parallel_wavegan itself is absent).  fp64 oracle == fp64 fixture to rounding; the f32 oracle within the
    f32 run's own distance; activations after conv_pre / the first transposed conv as well; `wn` goes through fold_weight_norm."""
    from helpers import hifigan_xcheck_case
    params, sd, mel, z = hifigan_xcheck_case(case)
    taps = {}
    y64 = HO.hifigan_generate({k: v.double() for k, v in sd.items()}, mel.double(), params["upsample_scales"], params["resblock_dilations"], taps=taps)
    ref = z[f"{case}_wave_f64"]
    assert y64.shape == ref.shape and maxdiff(y64, ref) <= 1e-12
    assert maxdiff(taps["input_conv"], z[f"{case}_input_conv"]) <= 1e-5 and maxdiff(taps["up0"], z[f"{case}_up0"]) <= 1e-5
    y32 = HO.hifigan_generate(sd, mel, params["upsample_scales"], params["resblock_dilations"])
    assert maxdiff(y32, ref) <= 2e-6 and maxdiff(z[f"{case}_wave_f32"], ref) <= 2e-6


def test_hifigan_oracle_matches_transformers_live():
    """The same cross-check run live where `transformers` is importable (this image): another seed / length than the fixture's."""
    pytest.importorskip("transformers")
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from make_golden_hifigan_xcheck import hf_generator, run
    from jatts_amd.synthetic import HIFIGAN_V1_22K, synth_hifigan_state
    params = dict(HIFIGAN_V1_22K, channels=256)
    sd = synth_hifigan_state(params, seed=21)
    mel = torch.randn(11, 80, generator=torch.Generator().manual_seed(22))
    y, _ = run(hf_generator(params, sd), mel, torch.float64)
    yo = HO.hifigan_generate({k: v.double() for k, v in sd.items()}, mel.double(), params["upsample_scales"], params["resblock_dilations"])
    assert maxdiff(yo, y) <= 1e-12


def test_masks_golden():
    from jatts_amd.hostlogic import make_non_pad_mask, make_pad_mask

    z, _ = load_golden("mask_kat.npz")
    for n in range(int(z["n_cases"])):
        lens = z[f"m{n}_lens"].tolist()
        assert np.array_equal(make_pad_mask(lens).numpy(), z[f"m{n}_pad"])
        assert np.array_equal(make_non_pad_mask(lens).numpy(), z[f"m{n}_nonpad"])


def test_mas_oracle_matches_reference(golden_dir):
    """the numpy restatement of AlignmentModule / _monotonic_alignment_search / viterbi_decode against the
    reference's own outputs (tests/golden/mas_kat.npz, make_golden_mas.py).  Bit-exact paths and durations."""
    import json

    from jatts_amd.synthetic import synth_state_dict
    from oracle.mas_oracle import alignment_log_p, monotonic_alignment_search, viterbi_decode
    z = np.load(golden_dir + "/mas_kat.npz")
    for n in range(int(z["n_mas"])):
        lp, ref = z[f"mas{n}_logp"], z[f"mas{n}_path"]
        assert np.array_equal(monotonic_alignment_search(lp, literal=True), ref), n
        assert np.array_equal(monotonic_alignment_search(lp), ref), n   # float64 running sum of row 0: same paths
        assert (np.diff(ref) >= 0).all() and (np.diff(ref) <= 1).all() and ref[-1] == lp.shape[1] - 1
    sd = synth_state_dict({k: tuple(s) for k, s in json.loads(str(z["keys"]))}, 7)
    tl, fl = z["text_lengths"], z["feats_lengths"]
    ref = torch.tensor(z["log_p_attn"])
    for b in range(len(tl)):   # per utterance, unpadded: the parity target (make_golden_mas.py)
        lp = alignment_log_p(sd, torch.tensor(z["text"][b:b + 1, : tl[b]]), torch.tensor(z["feats"][b:b + 1, : fl[b]]))[0]
        assert float((lp - ref[b, : fl[b], : tl[b]]).abs().max()) <= 1e-6
    ds, bin_loss, _ = viterbi_decode(ref, tl, fl, literal=True)
    assert np.array_equal(ds.numpy(), z["ds"])
    assert abs(float(bin_loss) - float(z["bin_loss"])) <= 1e-6


def test_training_lr_schedules_match_torch_and_the_reference_formula():
    """jatts_amd.training.scheduled_lr (host logic of the trainers): "steplr" against torch.optim.lr_scheduler.StepLR stepped once per
    optimiser step (the Matcha recipes: step_size 10000, gamma 0.5), "warmuplr" against jatts/schedulers/warmup_lr.py:55-62."""
    import torch
    from jatts_amd.training import scheduled_lr
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=1e-4)
    sch = torch.optim.lr_scheduler.StepLR(opt, step_size=7, gamma=0.5)
    for step in range(1, 40):
        assert abs(scheduled_lr("steplr", 1e-4, step, step_size=7, gamma=0.5) - opt.param_groups[0]["lr"]) <= 1e-18
        opt.step()
        sch.step()
    for step in (1, 2, 3999, 4000, 4001, 100000):
        want = 0.0008 * 4000 ** 0.5 * min(step ** -0.5, step * 4000 ** -1.5)
        assert abs(scheduled_lr("warmuplr", 0.0008, step, warmup_steps=4000) - want) <= 1e-18
    assert scheduled_lr(None, 3e-4, 17) == 3e-4


def test_oracles_match_reference_at_the_bench_length():
    """Round-3 fixtures (tests/golden/make_golden_r3.py): the REAL reference on 128-phoneme bench utterances -> 768 frames, for the three
    acoustic models of BASELINE configs[1] / [2] / [4].  The CPU oracles reproduce them (FastSpeech2 and Matcha bit-exactly)."""
    from jatts_amd.synthetic import matcha_golden_tweaks, pin_duration_head
    from oracle.matcha_oracle import matcha_inference
    from oracle.vits_oracle import vits_inference
    torch.set_num_threads(8)

    def noise_of(z):
        shape = [int(v) for v in z["noise_shape"]]
        return torch.randn(1, shape[1], shape[0], generator=torch.Generator().manual_seed(int(z["noise_seed"])))[0].t().contiguous()
    z, keys = load_golden("fs2_bench768.npz")
    sd = pin_duration_head(golden_state(keys, 0), 6)
    o = O.fs2_inference(sd, torch.tensor(z["u1_text"]), 2)
    assert o["feat_gen"].shape == (768, 80) and maxdiff(o["feat_gen"], z["u1_feat_gen"]) == 0.0
    assert np.array_equal(o["duration"].numpy(), z["u1_duration"])
    z, keys = load_golden("vits_bench128.npz")
    o = vits_inference(pin_duration_head(golden_state(keys, 0), 6), torch.tensor(z["u0_text"]), 2, 2, torch.tensor(z["u0_spemb"]), noise_of(z))
    assert o["feat_gen"].shape == (768, 80) and maxdiff(o["feat_gen"], z["u0_feat_gen"]) <= 1e-5
    assert np.array_equal(o["duration"].numpy(), z["u0_duration"])
    z, keys = load_golden("matcha_bench128.npz")
    o = matcha_inference(pin_duration_head(matcha_golden_tweaks(golden_state(keys, 0)), 6), torch.tensor(z["u0_text"]), 2, 2, noise_of(z),
                         n_timesteps=10, temperature=0.667)
    assert o["feat_gen"].shape == (768, 80) and maxdiff(o["feat_gen"], z["u0_feat_gen"]) <= 1e-5
    assert np.array_equal(o["duration"].numpy(), z["u0_duration"])
