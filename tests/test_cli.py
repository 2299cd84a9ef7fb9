"""Stage-4 CLI (jatts_amd.bin.tts_decode): host-side pieces on CPU, the whole recipe step on the GPU."""
import csv
import os
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


@pytest.mark.gpu
def test_stage4_cli_end_to_end(cuda, lib, tmp_path):
    from jatts_amd.bin import tts_decode
    from jatts_amd.models import FastSpeech2
    from jatts_amd.synthetic import FS2_SMALL, HIFIGAN_V1_24K, synth_hifigan_state, synth_state_dict
    d = tmp_path
    tokens = ["<blank>", "<unk>"] + [f"p{i}" for i in range(17)] + ["<sos/eos>"]
    (d / "tokens.txt").write_text("\n".join(tokens) + "\n")
    g = torch.Generator().manual_seed(0)
    with open(d / "dev.csv", "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=["sample_id", "phonemes"])
        w.writeheader()
        for i, n in enumerate((7, 15, 11)):
            w.writerow({"sample_id": f"utt{i}", "phonemes": " ".join(tokens[int(j)] for j in torch.randint(2, 19, (n,), generator=g))})
    m = FastSpeech2(idim=20, **FS2_SMALL)
    torch.save({"model": synth_state_dict(m.state_dict(), 0)}, d / "checkpoint-1steps.pkl")
    vparams = dict(HIFIGAN_V1_24K, channels=512)
    torch.save({"model": {"generator": synth_hifigan_state(vparams, 0)}}, d / "voc.pkl")
    with open(d / "voc.yml", "w") as f:
        yaml.safe_dump({"sampling_rate": 24000, "generator_type": "HiFiGANGenerator",
                        "generator_params": {k: (list(v) if isinstance(v, tuple) else v) for k, v in vparams.items()}}, f)
    np.savez(d / "stats.npz", mel_mean=np.zeros(80, np.float32), mel_scale=np.ones(80, np.float32))
    np.savez(d / "vstats.npz", mean=np.zeros(80, np.float32), scale=np.ones(80, np.float32))
    with open(d / "config.yml", "w") as f:
        yaml.safe_dump({"model_type": "FastSpeech2", "model_params": dict(FS2_SMALL, idim=20), "out_feat_type": "mel",
                        "feat_list": ["mel"], "vocoder": {"checkpoint": str(d / "voc.pkl"), "config": str(d / "voc.yml"),
                                                          "stats": str(d / "vstats.npz")}}, f)
    tts_decode.main(["--csv", str(d / "dev.csv"), "--stats", str(d / "stats.npz"), "--token-list", str(d / "tokens.txt"),
                     "--token-column", "phonemes", "--checkpoint", str(d / "checkpoint-1steps.pkl"),
                     "--outdir", str(d / "out"), "--verbose", "0", "--batch-size", "2"])
    for i in range(3):
        with wave.open(str(d / "out" / "wav" / f"utt{i}.wav")) as w:
            assert w.getframerate() == 24000 and w.getnframes() % 300 == 0 and w.getnframes() > 0
