"""Stage-4 CLI (jatts_amd.bin.tts_decode): host-side pieces on CPU, the whole recipe step on the GPU."""
import csv
import os
import shutil
import wave

import numpy as np
import pytest
import torch
import yaml


def test_cli_flags_match_the_reference():
    from jatts_amd.bin.tts_decode import get_parser
    flags = {a.option_strings[0] for a in get_parser()._actions if a.option_strings}
    # reference jatts/bin/tts_decode.py:37-88
    assert {"--csv", "--stats", "--token-list", "--token-column", "--outdir", "--checkpoint", "--config", "--verbose"} <= flags


def test_token_converter_and_wav_writer(tmp_path):
    from jatts_amd.bin.tts_decode import TokenIDConverter, write_wav_pcm16
    tl = tmp_path / "tokens.txt"
    tl.write_text("<blank>\n<unk>\na\nk\no\n<sos/eos>\n")
    c = TokenIDConverter(str(tl))
    assert c.tokens2ids(["k", "o", "zz", "a"]) == [3, 4, 1, 2]
    y = np.array([0.0, 1.0, -1.0, 0.5, 2.0])
    p = str(tmp_path / "x.wav")
    write_wav_pcm16(p, y, 24000)
    with wave.open(p) as w:
        assert (w.getframerate(), w.getsampwidth(), w.getnchannels(), w.getnframes()) == (24000, 2, 1, 5)
        pcm = np.frombuffer(w.readframes(5), dtype="<i2")
    assert pcm.tolist() == [0, 32767, -32767, 16384, 32767]   # lrint(x * 32767) like libsndfile PCM_16, clipped


@pytest.mark.gpu
def test_pcm16_kernel_equals_host_conversion(cuda, lib):
    """jatts_pcm16 is bit-identical to the host twin of libsndfile's float -> PCM_16 conversion."""
    from jatts_amd import hip
    from jatts_amd.bin.tts_decode import to_pcm16
    g = torch.Generator().manual_seed(0)
    y = torch.cat([torch.tanh(torch.randn(100003, generator=g) * 2), torch.tensor([0.0, 1.0, -1.0, 0.5, 1.5, -7.0, 0.5 / 32767, 1.5 / 32767])])
    for off in (0, 1, 3):   # unaligned starts take the scalar tail path
        got = hip.pcm16(y[off:].contiguous().to(cuda)).cpu().numpy()
        assert np.array_equal(got, to_pcm16(y[off:].numpy()))


def _make_expdir(d, csv_name="dev.csv"):
    """A synthetic exp/<expname> directory in the reference's layout (config.yml, tokens.txt, stats, checkpoint, vocoder)."""
    from jatts_amd.models import FastSpeech2
    from jatts_amd.synthetic import FS2_SMALL, HIFIGAN_V1_24K, synth_hifigan_state, synth_state_dict
    tokens = ["<blank>", "<unk>"] + [f"p{i}" for i in range(17)] + ["<sos/eos>"]
    (d / "tokens.txt").write_text("\n".join(tokens) + "\n")
    g = torch.Generator().manual_seed(0)
    with open(d / csv_name, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=["sample_id", "phonemes"])
        w.writeheader()
        for i, n in enumerate((7, 15, 11)):
            w.writerow({"sample_id": f"utt{i}", "phonemes": " ".join(tokens[int(j)] for j in torch.randint(2, 19, (n,), generator=g))})
    m = FastSpeech2(idim=20, **FS2_SMALL)
    torch.save({"model": synth_state_dict(m.state_dict(), 0)}, d / "checkpoint-1steps.pkl")
    vparams = dict(HIFIGAN_V1_24K, channels=512)
    v = d / "hfg"     # the recipes keep the vocoder under downloads/hfg, not next to the TTS checkpoints
    os.makedirs(v, exist_ok=True)
    torch.save({"model": {"generator": synth_hifigan_state(vparams, 0)}}, v / "voc.pkl")
    with open(v / "voc.yml", "w") as f:
        yaml.safe_dump({"sampling_rate": 24000, "generator_type": "HiFiGANGenerator",
                        "generator_params": {k: (list(v) if isinstance(v, tuple) else v) for k, v in vparams.items()}}, f)
    np.savez(d / "stats.npz", mel_mean=np.zeros(80, np.float32), mel_scale=np.ones(80, np.float32))
    np.savez(v / "vstats.npz", mean=np.zeros(80, np.float32), scale=np.ones(80, np.float32))
    with open(d / "config.yml", "w") as f:
        yaml.safe_dump({"model_type": "FastSpeech2", "model_params": dict(FS2_SMALL, idim=20), "out_feat_type": "mel",
                        "feat_list": ["mel"], "vocoder": {"checkpoint": str(v / "voc.pkl"), "config": str(v / "voc.yml"),
                                                          "stats": str(v / "vstats.npz")}}, f)


@pytest.mark.gpu
def test_stage4_cli_end_to_end(cuda, lib, tmp_path):
    from jatts_amd.bin import tts_decode
    d = tmp_path
    _make_expdir(d)
    tts_decode.main(["--csv", str(d / "dev.csv"), "--stats", str(d / "stats.npz"), "--token-list", str(d / "tokens.txt"),
                     "--token-column", "phonemes", "--checkpoint", str(d / "checkpoint-1steps.pkl"),
                     "--outdir", str(d / "out"), "--verbose", "0", "--batch-size", "2"])
    for i in range(3):
        with wave.open(str(d / "out" / "wav" / f"utt{i}.wav")) as w:
            assert w.getframerate() == 24000 and w.getnframes() % 300 == 0 and w.getnframes() > 0


@pytest.mark.gpu
def test_stage4_cli_split_precision_matches_fp32(cuda, lib, tmp_path):
    """`--precision fp32_split` (round 4: f32 tensors, split-precision MFMA operands) through the whole CLI: the same utterance lengths and wav
    files as the exact-f32 run, PCM samples at most one 16-bit step apart (the two arithmetics differ by ~1e-6 on a [-1, 1] signal)."""
    from jatts_amd.bin import tts_decode
    d = tmp_path
    _make_expdir(d)
    base = ["--csv", str(d / "dev.csv"), "--stats", str(d / "stats.npz"), "--token-list", str(d / "tokens.txt"), "--token-column", "phonemes",
            "--checkpoint", str(d / "checkpoint-1steps.pkl"), "--verbose", "0", "--batch-size", "2"]
    tts_decode.main(base + ["--outdir", str(d / "out32")])
    tts_decode.main(base + ["--outdir", str(d / "outs"), "--precision", "fp32_split"])
    for i in range(3):
        with wave.open(str(d / "out32" / "wav" / f"utt{i}.wav")) as a, wave.open(str(d / "outs" / "wav" / f"utt{i}.wav")) as b:
            assert a.getnframes() == b.getnframes() > 0
            x = np.frombuffer(a.readframes(a.getnframes()), dtype="<i2").astype(np.int32)
            y = np.frombuffer(b.readframes(b.getnframes()), dtype="<i2").astype(np.int32)
            assert np.abs(x - y).max() <= 1


def test_recipe_option_parser(tmp_path):
    """egs/common/parse_options.sh: `--name value` / `--name=value` set declared variables (dashes or underscores), unknown
    options are refused -- the interface the reference recipes get from their utils/parse_options.sh."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "t.sh"
    script.write_text(f"stage=4\nstop_stage=4\ncheckpoint=\"\"\nn_gpus=1\n. {root}/egs/common/parse_options.sh\n"
                      "echo \"$stage|$stop_stage|$checkpoint|$n_gpus|$#\"\n")
    run = lambda *a: subprocess.run(["bash", str(script), *a], capture_output=True, text=True)  # noqa: E731
    assert run("--stage", "2", "--stop-stage=7", "--checkpoint", "a b.pkl", "--n_gpus", "8").stdout.strip() == "2|7|a b.pkl|8|0"
    assert run().stdout.strip() == "4|4||1|0"
    bad = run("--nope", "1")
    assert bad.returncode != 0 and "unknown option" in bad.stderr


def test_h5stats_converter_with_stub_h5py(tmp_path, monkeypatch):
    """tools/h5stats_to_npz.py copies every dataset of the reference's stats.h5 (compute_statistics.py:94-103) under its own key;
    h5py is absent here, so a recording stub stands in for it (the converter is meant for the machine that trained the model)."""
    import importlib.util
    import sys
    import types
    data = {"mel_mean": np.arange(80, dtype=np.float64), "mel_scale": np.ones(80), "pitch_mean": np.zeros(1)}
    h5 = types.ModuleType("h5py")

    class Dataset:
        def __init__(self, a):
            self.a = a

        def __getitem__(self, k):
            return self.a

    class File:
        def __init__(self, path, mode):
            assert mode == "r"

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def visititems(self, fn):
            for k, v in data.items():
                fn(k, Dataset(v))

    h5.Dataset, h5.File = Dataset, File
    monkeypatch.setitem(sys.modules, "h5py", h5)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("h5conv", os.path.join(root, "tools", "h5stats_to_npz.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    dst, keys = mod.convert(str(tmp_path / "stats.h5"))
    assert dst.endswith("stats.npz") and keys == sorted(data)
    from jatts_amd.bin.tts_decode import read_stats
    st = read_stats(dst, "mel")
    assert st["mean"].dtype == np.float32 and np.array_equal(st["mean"], np.arange(80, dtype=np.float32))


@pytest.mark.gpu
@pytest.mark.parametrize("recipe,sets,prec", [("jsut/tts1", ("test",), "fp32"), ("hificaptain_jp_female/tts1", ("dev_raw_feat", "test"), "fp32"),
                                              ("hificaptain_jp_female/tts2", ("dev_raw_feat", "test"), "fp32_bf16x3")])
def test_recipe_run_sh_stage4(cuda, lib, tmp_path, recipe, sets, prec):
    """egs/<corpus>/ttsN/run.sh --stage 4: the reference recipe's variables and directory layout, decoding on the HIP path.  The hificaptain
    recipes decode dev_raw_feat and the test set (reference egs/hificaptain_jp_female/tts1/run.sh:221-245); the checkpoint's config.yml names the
    model, so every recipe directory serves any of the model families (here the small FastSpeech2)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    work = tmp_path / "recipe"
    exp = work / "exp" / "train_phn_none_unit"
    os.makedirs(exp)
    os.makedirs(work / "data")
    _make_expdir(exp, "test.csv")
    for name in sets:
        shutil.copy(exp / "test.csv", work / "data" / f"{name}.csv")
    os.remove(exp / "test.csv")
    r = subprocess.run(["bash", os.path.join(root, "egs", *recipe.split("/"), "run.sh"), "--stage", "4", "--stop_stage", "4", "--tag", "unit",
                        "--verbose", "0", "--decode_batch_size", "2", "--precision", prec], cwd=work, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    for name in sets:
        out = exp / "results" / "checkpoint-1steps" / name
        assert (out / "decode.log").exists()
        for i in range(3):
            with wave.open(str(out / "wav" / f"utt{i}.wav")) as w:
                assert w.getframerate() == 24000 and w.getnframes() % 300 == 0 and w.getnframes() > 0


@pytest.mark.gpu
def test_stage4_cli_multispeaker_extracts_embeddings(cuda, lib, tmp_path):
    """feat_list has `spkemb` and the csv carries ref_wav_path (the reference's layout, tts_decode.py:209-212): the CLI runs the
    speaker-embedding front end on the GPU (jatts_amd.spkemb) and feeds mel-VITS with it."""
    from jatts_amd.bin import tts_decode
    from jatts_amd.models import VITS
    from jatts_amd.spkemb import ECAPA_TDNN
    from jatts_amd.synthetic import HIFIGAN_V1_24K, synth_hifigan_state, synth_state_dict
    d = tmp_path
    tokens = ["<blank>", "<unk>"] + [f"p{i}" for i in range(17)] + ["<sos/eos>"]
    (d / "tokens.txt").write_text("\n".join(tokens) + "\n")
    g = torch.Generator().manual_seed(0)
    refs = []
    for i in range(2):                                   # two reference speakers, 0.6 s of 16 kHz audio each
        p = d / f"spk{i}.wav"
        y = (torch.randn(9600, generator=g) * 0.1).clamp(-1, 1)
        with wave.open(str(p), "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
            w.writeframes((y * 32767).round().to(torch.int16).numpy().tobytes())
        refs.append(str(p))
    with open(d / "dev.csv", "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=["sample_id", "phonemes", "ref_wav_path"])
        w.writeheader()
        for i, n in enumerate((7, 12, 9)):
            w.writerow({"sample_id": f"utt{i}", "phonemes": " ".join(tokens[int(j)] for j in torch.randint(2, 19, (n,), generator=g)),
                        "ref_wav_path": refs[i % 2]})
    vcfg = dict(odim=80, adim=64, aheads=2, text_encoder_blocks=2, text_encoder_attention_heads=2, dlayers=2, dunits=128, flow_flows=2,
                flow_layers=2, posterior_encoder_layers=2, duration_predictor_chans=64, spk_embed_dim=192)
    m = VITS(idim=20, **vcfg)
    torch.save({"model": synth_state_dict(m.state_dict(), 2)}, d / "checkpoint-1steps.pkl")
    ecfg = dict(channels=[256, 256, 256, 256, 768], attention_channels=64, se_channels=64, lin_neurons=192, res2net_scale=4)
    torch.save(synth_state_dict(ECAPA_TDNN(**ecfg).state_dict(), 5), d / "ecapa.ckpt")
    vparams = dict(HIFIGAN_V1_24K, channels=512)
    v = d / "hfg"
    os.makedirs(v)
    torch.save({"model": {"generator": synth_hifigan_state(vparams, 0)}}, v / "voc.pkl")
    with open(v / "voc.yml", "w") as f:
        yaml.safe_dump({"sampling_rate": 24000, "generator_type": "HiFiGANGenerator",
                        "generator_params": {k: (list(x) if isinstance(x, tuple) else x) for k, x in vparams.items()}}, f)
    np.savez(d / "stats.npz", mel_mean=np.zeros(80, np.float32), mel_scale=np.ones(80, np.float32))
    np.savez(v / "vstats.npz", mean=np.zeros(80, np.float32), scale=np.ones(80, np.float32))
    with open(d / "config.yml", "w") as f:
        yaml.safe_dump({"model_type": "VITS", "model_params": dict(vcfg, idim=20), "out_feat_type": "mel", "feat_list": ["mel", "spkemb"],
                        "spkemb_checkpoint": str(d / "ecapa.ckpt"), "spkemb_params": ecfg,
                        "vocoder": {"checkpoint": str(v / "voc.pkl"), "config": str(v / "voc.yml"), "stats": str(v / "vstats.npz")}}, f)
    tts_decode.main(["--csv", str(d / "dev.csv"), "--stats", str(d / "stats.npz"), "--token-list", str(d / "tokens.txt"),
                     "--token-column", "phonemes", "--checkpoint", str(d / "checkpoint-1steps.pkl"), "--outdir", str(d / "out"),
                     "--verbose", "0", "--batch-size", "2"])
    for i in range(3):
        with wave.open(str(d / "out" / "wav" / f"utt{i}.wav")) as w:
            assert w.getframerate() == 24000 and w.getnframes() % 300 == 0 and w.getnframes() > 0


@pytest.mark.gpu
def test_trained_checkpoint_feeds_stage4(cuda, lib, tmp_path):
    """Train -> save -> decode: a checkpoint written by FastSpeech2Trainer.save_checkpoint (the reference's layout,
    trainers/base.py:85-105) is what stage 4 loads (`torch.load(ckpt)["model"]`, tts_decode.py:141) -- the CLI synthesises from it and
    the result differs from the untrained model's."""
    from jatts_amd.bin import tts_decode
    from jatts_amd.models import FastSpeech2
    from jatts_amd.synthetic import FS2_SMALL
    from jatts_amd.training import FastSpeech2Trainer
    d = tmp_path
    _make_expdir(d)
    sd0 = torch.load(d / "checkpoint-1steps.pkl")["model"]
    m = FastSpeech2(idim=20, **FS2_SMALL)
    m.load_state_dict(sd0)
    m = m.to(cuda)
    g = torch.Generator().manual_seed(3)
    il = torch.tensor([9, 14])
    xs, ds = torch.zeros(2, 14, dtype=torch.long), torch.zeros(2, 14, dtype=torch.long)
    for b in range(2):
        xs[b, : il[b]] = torch.randint(2, 19, (int(il[b]),), generator=g)
        ds[b, : il[b]] = torch.randint(1, 5, (int(il[b]),), generator=g)
    ol = ds.sum(1)
    mask = (torch.arange(14)[None, :] < il[:, None]).float().unsqueeze(-1)
    batch = dict(xs=xs, ilens=il, ys=torch.randn(2, int(ol.max()), 80, generator=g), olens=ol, durations=ds, duration_lens=il,
                 pitch=torch.randn(2, 14, 1, generator=g) * mask, pitch_lens=il, energys=torch.randn(2, 14, 1, generator=g) * mask, energy_lens=il)
    tr = FastSpeech2Trainer(m, lr=5e-3, warmup_steps=0)
    for _ in range(3):
        tr.train_step(batch)
    tr.save_checkpoint(str(d / "checkpoint-4steps.pkl"))
    args = ["--csv", str(d / "dev.csv"), "--stats", str(d / "stats.npz"), "--token-list", str(d / "tokens.txt"), "--token-column", "phonemes",
            "--verbose", "0", "--batch-size", "3"]
    tts_decode.main(args + ["--checkpoint", str(d / "checkpoint-4steps.pkl"), "--outdir", str(d / "out_trained")])
    tts_decode.main(args + ["--checkpoint", str(d / "checkpoint-1steps.pkl"), "--outdir", str(d / "out_init")])
    changed = 0
    for i in range(3):
        with wave.open(str(d / "out_trained" / "wav" / f"utt{i}.wav")) as w, wave.open(str(d / "out_init" / "wav" / f"utt{i}.wav")) as w0:
            assert w.getframerate() == 24000 and w.getnframes() > 0
            changed += int(w.getnframes() != w0.getnframes() or w.readframes(w.getnframes()) != w0.readframes(w0.getnframes()))
    assert changed == 3


@pytest.mark.gpu
@pytest.mark.parametrize("recipe,column", [("tts2", "spkemb_path"), ("tts1", "ref_wav_path")])
def test_jvs_recipe_run_sh_stage4_multispeaker(cuda, lib, tmp_path, recipe, column):
    """egs/jvs/tts{1,2}/run.sh --stage 4 (reference egs/jvs/tts2/run.sh:192-215): the multi-speaker layout -- test set
    `test_parallel_with_ref`, feat_list with `spkemb`, one speaker per csv row as a precomputed `spkemb_path` (.npy) or as a
    `ref_wav_path` the GPU front end embeds -- with the mel-VITS of BASELINE config 5 (192-d speaker embedding) at a small width."""
    import subprocess
    from jatts_amd.models import VITS
    from jatts_amd.spkemb import ECAPA_TDNN
    from jatts_amd.synthetic import HIFIGAN_V1_24K, synth_hifigan_state, synth_state_dict
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    work = tmp_path / "recipe"
    exp = work / "exp" / "train_phn_none_unit"
    os.makedirs(exp)
    os.makedirs(work / "data")
    tokens = ["<blank>", "<unk>"] + [f"p{i}" for i in range(17)] + ["<sos/eos>"]
    (exp / "tokens.txt").write_text("\n".join(tokens) + "\n")
    g = torch.Generator().manual_seed(0)
    spk = []
    for i in range(2):
        if column == "spkemb_path":
            p = work / f"spk{i}.npy"
            np.save(p, torch.randn(192, generator=g).numpy())
        else:
            p = work / f"spk{i}.wav"
            y = (torch.randn(9600, generator=g) * 0.1).clamp(-1, 1)
            with wave.open(str(p), "wb") as w:
                w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
                w.writeframes((y * 32767).round().to(torch.int16).numpy().tobytes())
        spk.append(str(p))
    with open(work / "data" / "test_parallel_with_ref.csv", "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=["sample_id", "phonemes", column])
        w.writeheader()
        for i, n in enumerate((7, 12, 9, 5)):
            w.writerow({"sample_id": f"jvs{i % 2 + 1:03d}_utt{i}", "phonemes": " ".join(tokens[int(j)] for j in torch.randint(2, 19, (n,), generator=g)),
                        column: spk[i % 2]})
    vcfg = dict(odim=80, adim=64, aheads=2, text_encoder_blocks=2, text_encoder_attention_heads=2, dlayers=2, dunits=128, flow_flows=2,
                flow_layers=2, posterior_encoder_layers=2, duration_predictor_chans=64, spk_embed_dim=192)
    torch.save({"model": synth_state_dict(VITS(idim=20, **vcfg).state_dict(), 2)}, exp / "checkpoint-1steps.pkl")
    ecfg = dict(channels=[256, 256, 256, 256, 768], attention_channels=64, se_channels=64, lin_neurons=192, res2net_scale=4)
    torch.save(synth_state_dict(ECAPA_TDNN(**ecfg).state_dict(), 5), exp / "ecapa.ckpt")
    vparams = dict(HIFIGAN_V1_24K, channels=512)
    v = exp / "hfg"
    os.makedirs(v)
    torch.save({"model": {"generator": synth_hifigan_state(vparams, 0)}}, v / "voc.pkl")
    with open(v / "voc.yml", "w") as f:
        yaml.safe_dump({"sampling_rate": 24000, "generator_type": "HiFiGANGenerator",
                        "generator_params": {k: (list(x) if isinstance(x, tuple) else x) for k, x in vparams.items()}}, f)
    np.savez(exp / "stats.npz", mel_mean=np.zeros(80, np.float32), mel_scale=np.ones(80, np.float32))
    np.savez(v / "vstats.npz", mean=np.zeros(80, np.float32), scale=np.ones(80, np.float32))
    with open(exp / "config.yml", "w") as f:
        yaml.safe_dump({"model_type": "VITS", "model_params": dict(vcfg, idim=20), "out_feat_type": "mel", "feat_list": ["mel", "spkemb"],
                        "spkemb_checkpoint": str(exp / "ecapa.ckpt"), "spkemb_params": ecfg,
                        "vocoder": {"checkpoint": str(v / "voc.pkl"), "config": str(v / "voc.yml"), "stats": str(v / "vstats.npz")}}, f)
    r = subprocess.run(["bash", os.path.join(root, "egs", "jvs", recipe, "run.sh"), "--stage", "4", "--stop_stage", "4", "--tag", "unit",
                        "--verbose", "0", "--decode_batch_size", "3"], cwd=work, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    out = exp / "results" / "checkpoint-1steps" / "test_parallel_with_ref"
    assert (out / "decode.log").exists()
    for i in range(4):
        with wave.open(str(out / "wav" / f"jvs{i % 2 + 1:03d}_utt{i}.wav")) as w:
            assert w.getframerate() == 24000 and w.getnframes() % 300 == 0 and w.getnframes() > 0
