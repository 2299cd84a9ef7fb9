"""CPU oracle (TEST INFRASTRUCTURE ONLY): the speaker-embedding front end of stage 4.

The reference delegates it to SpeechBrain (jatts/modules/feature_extract/spkemb_speechbrain.py:14-28:
``EncoderClassifier.from_hparams("speechbrain/spkrec-ecapa-voxceleb").encode_batch(signal)``, called per utterance at
jatts/bin/tts_decode.py:209-212).  speechbrain (setup.cfg, unpinned) is NOT under /root/reference and is not installed,
and the reference holds no test or fixture for it -> **PARITY UNPINNED**.  What follows restates the published recipe
[recalled: speechbrain.lobes.features.Fbank, speechbrain.processing.features.{STFT, Filterbank, InputNormalization},
speechbrain.lobes.models.ECAPA_TDNN (Desplanques et al., arXiv 2005.07143), hparams of spkrec-ecapa-voxceleb]:

  compute_features  Fbank(n_mels=80): STFT(16 kHz, win 25 ms = n_fft 400, hop 10 ms, Hamming, center, constant (zero) pad = SpeechBrain STFT default)
                    -> power spectrum -> triangular mel filterbank (0..8000 Hz) -> 10 log10(clamp 1e-10), top_db 80
  mean_var_norm     InputNormalization(norm_type="sentence", std_norm=False): subtract the utterance mean per channel
  embedding_model   ECAPA_TDNN(80, channels [1024, 1024, 1024, 1024, 3072], kernels [5, 3, 3, 3, 1], dilations
                    [1, 2, 3, 4, 1], attention_channels 128, res2net_scale 8, se_channels 128, lin_neurons 192);
                    every Conv1d pads "same" with padding_mode "reflect"; TDNNBlock = conv -> ReLU -> BatchNorm1d.
State-dict keys follow SpeechBrain's module tree (blocks.N.conv.conv.weight, ...norm.norm.running_mean, ...).
Nothing under jatts_amd/ imports this file.
"""
import math

import torch
import torch.nn.functional as F

BN_EPS = 1e-5


def mel_filterbank(n_mels=80, n_fft=400, sample_rate=16000, f_min=0.0, f_max=8000.0):
    """(n_fft // 2 + 1, n_mels) triangular filters as SpeechBrain's Filterbank builds them."""
    to_mel = lambda hz: 2595.0 * math.log10(1.0 + hz / 700.0)  # noqa: E731
    mel = torch.linspace(to_mel(f_min), to_mel(f_max), n_mels + 2)
    hz = 700.0 * (10.0 ** (mel / 2595.0) - 1.0)
    band = (hz[1:] - hz[:-1])[:-1]
    f_central = hz[1:-1]
    n_stft = n_fft // 2 + 1
    all_freqs = torch.linspace(0, sample_rate // 2, n_stft)
    slope = (all_freqs.unsqueeze(0) - f_central.unsqueeze(1)) / band.unsqueeze(1)     # (n_mels, n_stft)
    return torch.clamp(torch.minimum(slope + 1.0, -slope + 1.0), min=0.0).t().contiguous()


def fbank_features(wav, n_mels=80, n_fft=400, hop=160):
    """wav (n,) -> (T, n_mels) sentence-mean-normalised log-mel features."""
    window = torch.hamming_window(n_fft)
    spec = torch.stft(wav.unsqueeze(0), n_fft, hop, n_fft, window, center=True, pad_mode="constant", normalized=False,
                      onesided=True, return_complex=True)[0].t()                       # (T, 201)
    power = spec.real ** 2 + spec.imag ** 2
    fb = power @ mel_filterbank(n_mels, n_fft)
    db = 10.0 * torch.log10(torch.clamp(fb, min=1e-10))
    db = torch.maximum(db, db.max() - 80.0)
    return db - db.mean(dim=0, keepdim=True)


def _bn(x, sd, p):
    return F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"], False, 0.0, BN_EPS)


def _conv_same_reflect(x, w, b, dil=1):
    k = w.shape[-1]
    pad = dil * (k - 1) // 2
    if pad:
        x = F.pad(x, (pad, pad), mode="reflect")
    return F.conv1d(x, w, b, dilation=dil)


def tdnn_block(x, sd, p, dil=1):
    """TDNNBlock: Conv1d(same, reflect) -> ReLU -> BatchNorm1d."""
    return _bn(F.relu(_conv_same_reflect(x, sd[p + "conv.conv.weight"], sd[p + "conv.conv.bias"], dil)), sd, p + "norm.norm.")


def se_res2net_block(x, sd, p, dil, scale=8):
    res = x
    x = tdnn_block(x, sd, p + "tdnn1.")
    ys, prev = [], None
    for i, xi in enumerate(torch.chunk(x, scale, dim=1)):
        if i == 0:
            yi = xi
        elif i == 1:
            yi = tdnn_block(xi, sd, p + f"res2net_block.blocks.{i - 1}.", dil)
        else:
            yi = tdnn_block(xi + prev, sd, p + f"res2net_block.blocks.{i - 1}.", dil)
        ys.append(yi)
        prev = yi
    x = tdnn_block(torch.cat(ys, dim=1), sd, p + "tdnn2.")
    s = x.mean(dim=2, keepdim=True)
    s = F.relu(F.conv1d(s, sd[p + "se_block.conv1.conv.weight"], sd[p + "se_block.conv1.conv.bias"]))
    s = torch.sigmoid(F.conv1d(s, sd[p + "se_block.conv2.conv.weight"], sd[p + "se_block.conv2.conv.bias"]))
    return s * x + res


def ecapa_embedding(sd, feats, dilations=(1, 2, 3, 4)):
    """feats (T, n_mels) -> (lin_neurons,) embedding (ECAPA_TDNN.forward on one utterance)."""
    x = feats.t().unsqueeze(0)                                          # (1, 80, T)
    x = tdnn_block(x, sd, "blocks.0.", dilations[0])
    outs = []
    for i in (1, 2, 3):
        x = se_res2net_block(x, sd, f"blocks.{i}.", dilations[i])
        outs.append(x)
    x = tdnn_block(torch.cat(outs, dim=1), sd, "mfa.")
    # attentive statistics pooling with global context
    T = x.shape[2]
    mean = x.mean(dim=2, keepdim=True)
    std = torch.sqrt(((x - mean) ** 2).mean(dim=2, keepdim=True).clamp(min=1e-12))
    a = torch.cat([x, mean.expand(-1, -1, T), std.expand(-1, -1, T)], dim=1)
    a = torch.tanh(tdnn_block(a, sd, "asp.tdnn."))
    a = torch.softmax(F.conv1d(a, sd["asp.conv.conv.weight"], sd["asp.conv.conv.bias"]), dim=2)
    mean = (a * x).sum(dim=2, keepdim=True)
    std = torch.sqrt((a * (x - mean) ** 2).sum(dim=2, keepdim=True).clamp(min=1e-12))
    pooled = _bn(torch.cat([mean, std], dim=1), sd, "asp_bn.norm.")
    return F.conv1d(pooled, sd["fc.conv.weight"], sd["fc.conv.bias"]).reshape(-1)


def encode(sd, wav):
    """SpeechBrainSpkEmbExtractor.forward on a loaded waveform: (n,) float -> (192,) embedding."""
    return ecapa_embedding(sd, fbank_features(wav.float()))
