"""Deterministic synthetic weights and inputs (no network: there are no checkpoints
or datasets here).  Used by bench.py, smoke(), the tests and the golden-vector
generator, so the SAME tensors can be rebuilt on the GPU box from (name, shape, seed).

Every tensor is a pure function of its state_dict key, its shape and a seed —
independent of construction order — so loading them into the real reference
(tests/golden/make_golden.py) and into this package gives identical models.
"""
import math
import zlib

import torch


def _gen(name, seed):
    return torch.Generator().manual_seed((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)


def synth_tensor(name, shape, seed=0, dtype=torch.float32):
    """Value for state_dict entry ``name`` (xavier-uniform-like for matrices, small
    non-zero biases, perturbed norm scales, plausible BatchNorm statistics)."""
    g = _gen(name, seed)
    shape = tuple(shape)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.long)
    if leaf == "running_mean":
        return (0.1 * torch.randn(shape, generator=g)).to(dtype)
    if leaf == "running_var":
        return (0.5 + torch.rand(shape, generator=g)).to(dtype)
    if len(shape) >= 2:
        if "embed.0.weight" in name and len(shape) == 2 and "encoder" in name:
            return torch.randn(shape, generator=g).to(dtype)  # nn.Embedding default N(0,1)
        rf = 1
        for s in shape[2:]:
            rf *= s
        fan_in, fan_out = shape[1] * rf, shape[0] * rf
        a = math.sqrt(6.0 / (fan_in + fan_out))
        return ((torch.rand(shape, generator=g) * 2 - 1) * a).to(dtype)
    if leaf == "weight":  # LayerNorm / BatchNorm scale
        return (1.0 + 0.1 * torch.randn(shape, generator=g)).to(dtype)
    return (0.02 * torch.randn(shape, generator=g)).to(dtype)  # biases


def synth_state_dict(shapes, seed=0):
    """shapes: mapping name -> shape (e.g. from ``module.state_dict()``)."""
    return {k: synth_tensor(k, tuple(v.shape) if hasattr(v, "shape") else tuple(v), seed)
            for k, v in shapes.items()}


def pin_duration_head(sd, frames_per_token, prefix="duration_predictor."):
    """linear.weight = 0, bias = ln(d+1) so every token predicts exactly
    ``d`` frames through the normal inference path (exp(ln(d+1)) - 1 = d, far from a
    rounding boundary)."""
    sd[prefix + "linear.weight"] = torch.zeros_like(sd[prefix + "linear.weight"])
    sd[prefix + "linear.bias"] = torch.full_like(sd[prefix + "linear.bias"], math.log(frames_per_token + 1.0))
    return sd


def matcha_golden_tweaks(sd):
    """Adjustments applied on top of synth_state_dict for the Matcha golden model (shared by
    tests/golden/make_golden.py and the tests): SnakeBeta's log-scale alpha/beta get a visible range
    and the duration head a +1.3 bias (about 3 frames per token) so the U-Net sees useful lengths."""
    for k in sd:
        if k.endswith(".alpha") or k.endswith(".beta"):
            sd[k] = sd[k] * 5.0
    sd["duration_predictor.linear.bias"] = sd["duration_predictor.linear.bias"] + 1.3
    return sd


def synth_texts(n_utts, t_text, vocab=45, seed=1, ragged_min=None):
    """Token id sequences ~ U{1..vocab-1} (0 is <blank>/pad).  ``ragged_min`` draws
    lengths ~ U{ragged_min..t_text}."""
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n_utts):
        n = t_text if ragged_min is None else int(torch.randint(ragged_min, t_text + 1, (1,), generator=g))
        out.append(torch.randint(1, vocab, (n,), generator=g, dtype=torch.long))
    return out


FS2_JSUT = dict(  # egs/jsut/tts1/conf/fastspeech2.v1.yaml:26-77 (model_params)
    odim=80, adim=384, aheads=2, elayers=4, eunits=1536, dlayers=4, dunits=1536,
    positionwise_layer_type="conv1d", positionwise_conv_kernel_size=3,
    duration_predictor_layers=2, duration_predictor_chans=256, duration_predictor_kernel_size=3,
    postnet_layers=5, postnet_filts=5, postnet_chans=256, use_masking=True,
    encoder_normalize_before=True, decoder_normalize_before=True, reduction_factor=1,
    encoder_type="conformer", decoder_type="conformer",
    conformer_pos_enc_layer_type="rel_pos", conformer_self_attn_layer_type="rel_selfattn",
    conformer_activation_type="swish", use_macaron_style_in_conformer=True,
    use_cnn_in_conformer=True, conformer_enc_kernel_size=7, conformer_dec_kernel_size=31,
    init_type="xavier_uniform",
    transformer_enc_dropout_rate=0.2, transformer_enc_positional_dropout_rate=0.2,
    transformer_enc_attn_dropout_rate=0.2, transformer_dec_dropout_rate=0.2,
    transformer_dec_positional_dropout_rate=0.2, transformer_dec_attn_dropout_rate=0.2,
    pitch_predictor_layers=5, pitch_predictor_chans=256, pitch_predictor_kernel_size=5,
    pitch_predictor_dropout=0.5, pitch_embed_kernel_size=1, pitch_embed_dropout=0.0,
    stop_gradient_from_pitch_predictor=True,
    energy_predictor_layers=2, energy_predictor_chans=256, energy_predictor_kernel_size=3,
    energy_predictor_dropout=0.5, energy_embed_kernel_size=1, energy_embed_dropout=0.0,
    stop_gradient_from_energy_predictor=False,
)

MATCHA_MAS_JSUT = dict(  # egs/jsut/tts2/conf/matcha_tts.mas.v1.yaml:22-63 (model_params; BASELINE config 3)
    odim=80, adim=384, aheads=2, elayers=4, eunits=1536, positionwise_layer_type="conv1d",
    positionwise_conv_kernel_size=3, duration_predictor_layers=2, duration_predictor_chans=256,
    duration_predictor_kernel_size=3, use_masking=True, encoder_normalize_before=True, reduction_factor=1,
    encoder_type="conformer", conformer_pos_enc_layer_type="rel_pos", conformer_self_attn_layer_type="rel_selfattn",
    conformer_activation_type="swish", use_macaron_style_in_conformer=True, use_cnn_in_conformer=True,
    conformer_enc_kernel_size=7, conformer_dec_kernel_size=31, init_type="xavier_uniform",
    transformer_enc_dropout_rate=0.2, transformer_enc_positional_dropout_rate=0.2, transformer_enc_attn_dropout_rate=0.2,
    decoder_channels=[512, 512], decoder_dropout=0.05, decoder_attention_head_dim=256, decoder_n_blocks=1,
    decoder_num_mid_blocks=2, decoder_num_heads=2, decoder_act_fn="snakebeta",
)

VITS_JSUT = dict(  # egs/jsut/tts2/conf/vits.v1.bs32.yaml:22-44 (model_params; BASELINE config 5 adds spk_embed_dim=192)
    odim=80, adim=384, aheads=2, dlayers=4, dunits=1536, decoder_positionwise_layer_type="conv1d",
    decoder_positionwise_conv_kernel_size=3, duration_predictor_layers=2, duration_predictor_chans=256,
    duration_predictor_kernel_size=3, use_masking=True, decoder_normalize_before=True, reduction_factor=1,
    use_macaron_style_in_conformer=True, use_cnn_in_conformer=True, conformer_dec_kernel_size=31,
    init_type="xavier_uniform", transformer_dec_dropout_rate=0.2, transformer_dec_positional_dropout_rate=0.2,
    transformer_dec_attn_dropout_rate=0.2,
)

FS2_SMALL = dict(  # reduced-width config for fast CPU golden vectors (same code paths)
    odim=80, adim=64, aheads=2, elayers=2, eunits=128, dlayers=2, dunits=128,
    positionwise_layer_type="conv1d", positionwise_conv_kernel_size=3,
    duration_predictor_layers=2, duration_predictor_chans=64, duration_predictor_kernel_size=3,
    postnet_layers=5, postnet_filts=5, postnet_chans=64,
    encoder_type="conformer", decoder_type="conformer",
    conformer_pos_enc_layer_type="rel_pos", conformer_self_attn_layer_type="rel_selfattn",
    use_macaron_style_in_conformer=True, use_cnn_in_conformer=True,
    conformer_enc_kernel_size=7, conformer_dec_kernel_size=31,
    pitch_predictor_layers=3, pitch_predictor_chans=64, pitch_predictor_kernel_size=5,
    pitch_embed_kernel_size=1, pitch_embed_dropout=0.0,
    energy_predictor_layers=2, energy_predictor_chans=64, energy_predictor_kernel_size=3,
    energy_embed_kernel_size=1, energy_embed_dropout=0.0,
)

HIFIGAN_V1_22K = dict(  # parallel_wavegan HiFiGANGenerator defaults [recalled]: hop 256, 22.05 kHz
    in_channels=80, out_channels=1, channels=512, kernel_size=7,
    upsample_scales=(8, 8, 2, 2), upsample_kernel_sizes=(16, 16, 4, 4),
    resblock_kernel_sizes=(3, 7, 11), resblock_dilations=((1, 3, 5), (1, 3, 5), (1, 3, 5)),
    use_additional_convs=True, bias=True,
    nonlinear_activation="LeakyReLU", nonlinear_activation_params={"negative_slope": 0.1},
    use_weight_norm=True,
)
HIFIGAN_V1_24K = dict(HIFIGAN_V1_22K, upsample_scales=(5, 5, 4, 3), upsample_kernel_sizes=(10, 10, 8, 6))


def synth_hifigan_state(params, seed=0, gain=1.0):
    """Generator weights, weight norm already folded.  Every conv is N(0, g/sqrt(fan_in)) so that
    activations stay O(1..10) through the 4 stages (a single global std either vanishes or
    explodes through ~40 convs): g = gain for input/upsample/convs1, 0.3*gain for convs2 (keeps
    the residual stream from doubling per unit), 0.5*gain for the output conv; biases N(0, 0.02)."""
    sd = {}
    ch, k = params["channels"], params["kernel_size"]

    def rn(name, shape, fan_in=None, g=1.0):
        std = 0.02 if fan_in is None else g * gain / math.sqrt(fan_in)
        sd[name] = torch.randn(shape, generator=_gen(name, seed)) * std

    rn("input_conv.weight", (ch, params["in_channels"], k), params["in_channels"] * k)
    rn("input_conv.bias", (ch,))
    nb = len(params["resblock_kernel_sizes"])
    c = ch
    for i, (us, uk) in enumerate(zip(params["upsample_scales"], params["upsample_kernel_sizes"])):
        rn(f"upsamples.{i}.1.weight", (c, c // 2, uk), c * uk / us)
        rn(f"upsamples.{i}.1.bias", (c // 2,))
        c //= 2
        for j, rk in enumerate(params["resblock_kernel_sizes"]):
            for d in range(len(params["resblock_dilations"][j])):
                rn(f"blocks.{i * nb + j}.convs1.{d}.1.weight", (c, c, rk), c * rk)
                rn(f"blocks.{i * nb + j}.convs1.{d}.1.bias", (c,))
                if params.get("use_additional_convs", True):
                    rn(f"blocks.{i * nb + j}.convs2.{d}.1.weight", (c, c, rk), c * rk, 0.3)
                    rn(f"blocks.{i * nb + j}.convs2.{d}.1.bias", (c,))
    rn("output_conv.1.weight", (params["out_channels"], c, k), c * k, 0.5)
    rn("output_conv.1.bias", (params["out_channels"],))
    return sd
