#!/usr/bin/env python3
"""HiFi-GAN generator cross-check vectors from an INDEPENDENT public implementation.

    python tests/golden/make_golden_hifigan_xcheck.py        (needs `transformers`; 5.15.0 in this image)

The reference's vocoder is `parallel_wavegan.models.HiFiGANGenerator` (/root/reference/jatts/vocoder/vocoder.py:13,41,43,64), an
un-vendored pip dependency that is absent here, so oracle/hifigan_oracle.py restates the published generator.  What IS installed is
Hugging Face `transformers`, whose `FastSpeech2ConformerHifiGan` (modeling_fastspeech2_conformer.py, "Copied from SpeechT5HifiGan", itself a
port of the HiFi-GAN authors' generator) implements the same published V1 network with its own code: conv_pre k7 -> [LeakyReLU(0.1) ->
ConvTranspose1d(k, s, padding (k - s) // 2) -> mean of the ResBlocks] x 4 -> LeakyReLU(0.01) -> conv_post k7 -> tanh.  For even strides
with k = 2 s (the 22.05 kHz V1 config of BASELINE configs[1]: 8, 8, 2, 2) its padding equals parallel_wavegan's `s // 2 + s % 2` with
output_padding `s % 2`, so both compute the same function of (weights, mel).  Odd strides (the 24 kHz config: 5, 5, 4, 3) differ by
that padding convention and stay restated-only.

hifigan_xcheck.npz holds, per case, a mel and the waveform the transformers module produced from it in fp32 and fp64, plus the
activations after conv_pre, the first upsampling conv and the first MRF stage.  Weights are NOT stored: the tests rebuild them with
jatts_amd.synthetic.synth_hifigan_state(params, seed) -- this script maps that state dict (parallel_wavegan key schema) onto the
transformers module.  Case `wn` loads the weights as weight-norm (g, v) pairs through torch's weight_norm parametrisation
(v = 3 w, g = ||w|| (1 + 0.5 u), u ~ U(0, 1) seeded) to pin oracle.fold_weight_norm.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from jatts_amd.synthetic import HIFIGAN_V1_22K, synth_hifigan_state  # noqa: E402

CASES = {  # name -> (generator params, weight seed, frames, mel seed)
    "v1": (dict(HIFIGAN_V1_22K), 5, 24, 11),                                   # the bench's vocoder at full width (512 channels)
    "w128": (dict(HIFIGAN_V1_22K, channels=128), 6, 37, 12),                   # narrow stages (64 / 32 / 16 / 8 channels)
    "two_blocks": (dict(HIFIGAN_V1_22K, channels=256, upsample_scales=(4, 4, 4), upsample_kernel_sizes=(8, 8, 8),
                        resblock_kernel_sizes=(3, 5), resblock_dilations=((1, 2), (2, 6, 3))), 7, 19, 13),
    "wn": (dict(HIFIGAN_V1_22K, channels=128), 8, 15, 14),
}


def wn_pairs(sd, seed):
    """(g, v) pairs whose fold is NOT the identity: v = 3 w, g = ||w|| (1 + 0.5 u); returns (state with weight_g / weight_v, folded)."""
    gen = torch.Generator().manual_seed(seed)
    out, folded = {}, {}
    for k, w in sd.items():
        if k.endswith(".weight"):
            norm = w.reshape(w.shape[0], -1).norm(dim=1).reshape(-1, *([1] * (w.dim() - 1)))
            g = norm * (1.0 + 0.5 * torch.rand(norm.shape, generator=gen))
            out[k[:-6] + "weight_g"], out[k[:-6] + "weight_v"] = g, 3.0 * w
            folded[k] = g * w / norm
        else:
            out[k] = folded[k] = w
    return out, folded


def hf_generator(params, sd, weight_norm=False):
    from transformers import FastSpeech2ConformerHifiGan, FastSpeech2ConformerHifiGanConfig
    cfg = FastSpeech2ConformerHifiGanConfig(
        model_in_dim=params["in_channels"], upsample_initial_channel=params["channels"], upsample_rates=list(params["upsample_scales"]),
        upsample_kernel_sizes=list(params["upsample_kernel_sizes"]), resblock_kernel_sizes=list(params["resblock_kernel_sizes"]),
        resblock_dilation_sizes=[list(d) for d in params["resblock_dilations"]], leaky_relu_slope=0.1, normalize_before=False)
    m = FastSpeech2ConformerHifiGan(cfg).eval()
    if weight_norm:
        m.apply_weight_norm()
    mods = {"input_conv": m.conv_pre, "output_conv.1": m.conv_post}
    for i, up in enumerate(m.upsampler):
        mods[f"upsamples.{i}.1"] = up
    for j, blk in enumerate(m.resblocks):
        for d, (c1, c2) in enumerate(zip(blk.convs1, blk.convs2)):
            mods[f"blocks.{j}.convs1.{d}.1"], mods[f"blocks.{j}.convs2.{d}.1"] = c1, c2
    used = set()
    with torch.no_grad():
        for stem, mod in mods.items():
            if weight_norm:
                p = mod.parametrizations.weight
                p.original0.copy_(sd[stem + ".weight_g"])
                p.original1.copy_(sd[stem + ".weight_v"])
                used |= {stem + ".weight_g", stem + ".weight_v"}
            else:
                mod.weight.copy_(sd[stem + ".weight"])
                used.add(stem + ".weight")
            mod.bias.copy_(sd[stem + ".bias"])
            used.add(stem + ".bias")
    assert used == set(sd), sorted(set(sd) ^ used)[:5]
    return m


def run(m, mel, dtype):
    """waveform + three taps (time-major) of the transformers module in `dtype`."""
    m = m.to(dtype)
    taps = {}
    hooks = [m.conv_pre.register_forward_hook(lambda _m, _i, o: taps.__setitem__("input_conv", o[0].t().clone())),
             m.upsampler[0].register_forward_hook(lambda _m, _i, o: taps.__setitem__("up0", o[0].t().clone())),
             m.upsampler[1].register_forward_pre_hook(lambda _m, i: taps.__setitem__("mrf0_lrelu", i[0][0].t().clone()))]
    with torch.no_grad():
        y = m(mel.to(dtype))
    for h in hooks:
        h.remove()
    return y, taps


def main():
    import transformers
    out = {"transformers_version": transformers.__version__, "cases": json.dumps({k: [v[0], v[1], v[2], v[3]] for k, v in CASES.items()})}
    from oracle.hifigan_oracle import hifigan_generate
    for name, (params, wseed, frames, mseed) in CASES.items():
        sd = synth_hifigan_state(params, seed=wseed)
        mel = torch.randn(frames, params["in_channels"], generator=torch.Generator().manual_seed(mseed))
        if name == "wn":
            sd_wn, folded = wn_pairs(sd, wseed)
            m = hf_generator(params, sd_wn, weight_norm=True)
            osd = sd_wn
        else:
            m = hf_generator(params, sd)
            osd = sd
        y32, t32 = run(m, mel, torch.float32)
        y64, t64 = run(m, mel, torch.float64)
        hop = int(np.prod(params["upsample_scales"]))
        assert y64.shape == (frames * hop,)
        out[f"{name}_mel"] = mel.numpy()
        out[f"{name}_wave_f32"] = y32.numpy()
        out[f"{name}_wave_f64"] = y64.numpy()
        for k in ("input_conv", "up0"):
            out[f"{name}_{k}"] = t64[k].float().numpy()
        otaps = {}
        yo = hifigan_generate({k: v.double() for k, v in osd.items()}, mel.double(), params["upsample_scales"], params["resblock_dilations"], taps=otaps)
        print(f"{name}: frames {frames} samples {y64.numel()} |y|max {float(y64.abs().max()):.3f}  oracle(f64)-vs-transformers(f64) max|d| = "
              f"{float((yo - y64).abs().max()):.3e}  f32-vs-f64 {float((y32.double() - y64).abs().max()):.3e}  "
              f"up0 {float((otaps['up0'] - t64['up0']).abs().max()):.3e}")
    np.savez_compressed(os.path.join(HERE, "hifigan_xcheck.npz"), **out)
    print("wrote hifigan_xcheck.npz", os.path.getsize(os.path.join(HERE, "hifigan_xcheck.npz")), "bytes")


if __name__ == "__main__":
    main()
