"""CPU oracle: HiFi-GAN v1 generator + Vocoder.decode normalisation.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

PARITY for the generator: the reference calls
``parallel_wavegan.utils.load_model(...)``, ``.remove_weight_norm()`` and
``.inference(c, normalize_before=False)`` (/root/reference/jatts/vocoder/vocoder.py:
13,41,43,64); `parallel-wavegan` is an un-vendored, unpinned pip dependency
(setup.cfg:17) that is not installed here, and the reference holds no test or
golden vector for it -- against parallel_wavegan ITSELF this file stays unpinned.
It IS pinned on an independent public implementation of the same published network
(HiFi-GAN, arXiv 2010.05646 section 2 / App. A, generator V1): Hugging Face transformers
5.15.0 `FastSpeech2ConformerHifiGan`, run in this container on the same weights and mels
(tests/golden/make_golden_hifigan_xcheck.py -> hifigan_xcheck.npz: the 22.05 kHz V1
config at full width, two narrower / differently blocked ones, and weight-norm (g, v)
loading).  In fp64 the two agree to rounding (<= 1e-15); tests/test_oracle_golden.py
checks the fixture and a live run, tests/test_hifigan_gpu.py holds the HIP path to the
same waveforms.  What that cannot pin: the odd-stride padding convention of the
transposed convs (`s // 2 + s % 2`, output_padding `s % 2`: the 24 kHz config 5, 5, 4, 3),
where transformers' `(k - s) // 2` differs and parallel_wavegan's is [recalled]; and the
state_dict key schema of parallel_wavegan.models.HiFiGANGenerator [recalled]:

  input_conv.{weight,bias}                      Conv1d(in, C, k7, pad 3)
  upsamples.<i>.1.{weight,bias}                 LeakyReLU(0.1) -> ConvTranspose1d(C_i, C_i/2, k_i, s_i,
                                                padding=s_i//2 + s_i%2, output_padding=s_i%2)
  blocks.<i*nb+j>.convs1.<d>.1.{weight,bias}    LeakyReLU(0.1) -> Conv1d(C, C, k_j, dilation=dil[j][d])
  blocks.<i*nb+j>.convs2.<d>.1.{weight,bias}    LeakyReLU(0.1) -> Conv1d(C, C, k_j, dilation=1)
  output_conv.1.{weight,bias}                   LeakyReLU(0.01 = torch default) -> Conv1d(C, 1, k7) -> tanh
  x = mean_j blocks[i,j](x) after each upsample; block: x = x + convs2(convs1(x)) per dilation.

``weight_g/weight_v`` pairs (checkpoints saved with weight norm) are folded by
``fold_weight_norm``.  Vocoder.decode's normalisation (vocoder.py:56-61) is in-repo
and pinned by tests/golden (golden made with the reference formula).
"""
import torch
import torch.nn.functional as F


def fold_weight_norm(sd):
    """weight = g * v / ||v||  (norm over all dims but 0), as torch weight_norm(dim=0)."""
    out = {}
    for k, v in sd.items():
        if k.endswith("weight_g"):
            stem = k[: -len("weight_g")]
            wv = sd[stem + "weight_v"]
            norm = wv.reshape(wv.shape[0], -1).norm(dim=1).reshape(-1, *([1] * (wv.dim() - 1)))
            out[stem + "weight"] = v * wv / norm
        elif k.endswith("weight_v"):
            continue
        else:
            out[k] = v
    return out


def infer_config(sd):
    """Recover upsample scales / kernel sizes / dilation count from tensor shapes."""
    n_up = 0
    while f"upsamples.{n_up}.1.weight" in sd:
        n_up += 1
    n_blocks_total = 0
    while f"blocks.{n_blocks_total}.convs1.0.1.weight" in sd:
        n_blocks_total += 1
    nb = n_blocks_total // n_up
    up_k = [sd[f"upsamples.{i}.1.weight"].shape[-1] for i in range(n_up)]
    rb_k = [sd[f"blocks.{j}.convs1.0.1.weight"].shape[-1] for j in range(nb)]
    return dict(n_up=n_up, n_blocks=nb, upsample_kernel_sizes=up_k, resblock_kernel_sizes=rb_k)


def hifigan_generate(sd, c, upsample_scales, resblock_dilations=((1, 3, 5),) * 3,
                     lrelu_slope=0.1, taps=None):
    """c (T, n_mels) normalised mel -> waveform (T*prod(scales),)."""
    sd = fold_weight_norm(sd)
    cfg = infer_config(sd)
    x = c.t().unsqueeze(0)  # (1, n_mels, T)
    w = sd["input_conv.weight"]
    x = F.conv1d(x, w, sd.get("input_conv.bias"), padding=(w.shape[-1] - 1) // 2)
    if taps is not None:
        taps["input_conv"] = x[0].t()
    nb = cfg["n_blocks"]
    for i, s in enumerate(upsample_scales):
        x = F.leaky_relu(x, lrelu_slope)
        x = F.conv_transpose1d(
            x, sd[f"upsamples.{i}.1.weight"], sd.get(f"upsamples.{i}.1.bias"),
            stride=s, padding=s // 2 + s % 2, output_padding=s % 2,
        )
        if taps is not None:
            taps[f"up{i}"] = x[0].t()
        acc = 0.0
        for j in range(nb):
            y = x
            blk = f"blocks.{i * nb + j}."
            has2 = (blk + "convs2.0.1.weight") in sd
            for di, d in enumerate(resblock_dilations[j]):
                w1 = sd[blk + f"convs1.{di}.1.weight"]
                k = w1.shape[-1]
                t = F.conv1d(F.leaky_relu(y, lrelu_slope), w1, sd.get(blk + f"convs1.{di}.1.bias"),
                             padding=(k - 1) // 2 * d, dilation=d)
                if has2:
                    w2 = sd[blk + f"convs2.{di}.1.weight"]
                    t = F.conv1d(F.leaky_relu(t, lrelu_slope), w2, sd.get(blk + f"convs2.{di}.1.bias"),
                                 padding=(w2.shape[-1] - 1) // 2)
                y = t + y
                if taps is not None and i == 0 and j == 0 and di == 0:
                    taps["unit000"] = y[0].t()
            acc = acc + y
        x = acc / nb
        if taps is not None:
            taps[f"mrf{i}"] = x[0].t()
    x = F.leaky_relu(x)  # default slope 0.01
    w = sd["output_conv.1.weight"]
    x = torch.tanh(F.conv1d(x, w, sd.get("output_conv.1.bias"), padding=(w.shape[-1] - 1) // 2))
    return x.reshape(-1)


def vocoder_normalize(c, trg_mean, trg_scale, voc_mean, voc_scale, take_norm_feat=True):
    """vocoder.py:56-61."""
    if take_norm_feat:
        c = c * trg_scale + trg_mean
    return (c - voc_mean) / voc_scale


def random_hifigan_state(in_channels=80, channels=512, kernel_size=7,
                         upsample_scales=(8, 8, 2, 2), upsample_kernel_sizes=(16, 16, 4, 4),
                         resblock_kernel_sizes=(3, 7, 11), n_dilations=3, seed=0, std=0.01,
                         dtype=torch.float32):
    """Synthetic generator weights: N(0, std) conv weights (the public recipe's
    reset_parameters), small random biases; weight norm already folded."""
    g = torch.Generator().manual_seed(seed)

    def rn(*shape, s=std):
        return (torch.randn(*shape, generator=g) * s).to(dtype)

    sd = {"input_conv.weight": rn(channels, in_channels, kernel_size),
          "input_conv.bias": rn(channels)}
    nb = len(resblock_kernel_sizes)
    c = channels
    for i, (s, k) in enumerate(zip(upsample_scales, upsample_kernel_sizes)):
        sd[f"upsamples.{i}.1.weight"] = rn(c, c // 2, k)
        sd[f"upsamples.{i}.1.bias"] = rn(c // 2)
        c //= 2
        for j, rk in enumerate(resblock_kernel_sizes):
            for d in range(n_dilations):
                for cv in ("convs1", "convs2"):
                    sd[f"blocks.{i * nb + j}.{cv}.{d}.1.weight"] = rn(c, c, rk)
                    sd[f"blocks.{i * nb + j}.{cv}.{d}.1.bias"] = rn(c)
    sd["output_conv.1.weight"] = rn(1, c, kernel_size)
    sd["output_conv.1.bias"] = rn(1)
    return sd
