"""Parameter schema helpers: build nn.Module trees whose ``state_dict`` keys and
shapes equal the reference's, so ``load_state_dict(torch.load(ckpt)["model"])``
(/root/reference/jatts/bin/tts_decode.py:141) works unchanged.

The host classes own parameters only; all arithmetic happens in HIP kernels.
"""
from collections import OrderedDict

import torch


class _Node(torch.nn.Module):
    """Anonymous container (a stand-in for Sequential / ModuleList / Linear ...)."""


def attach(root, dotted, tensor, buffer=False):
    """Register ``tensor`` at ``root.<dotted>`` creating containers on the way."""
    parts = dotted.split(".")
    mod = root
    for p in parts[:-1]:
        if p not in mod._modules:
            mod.add_module(p, _Node())
        mod = mod._modules[p]
    if buffer:
        mod.register_buffer(parts[-1], tensor)
    else:
        mod.register_parameter(parts[-1], torch.nn.Parameter(tensor, requires_grad=False))


def build_from_spec(root, spec):
    """spec: OrderedDict name -> (shape, kind), kind in {"param","buffer","long_buffer"}."""
    for name, (shape, kind) in spec.items():
        if kind == "long_buffer":
            attach(root, name, torch.zeros(shape, dtype=torch.long), buffer=True)
        else:
            attach(root, name, torch.zeros(shape, dtype=torch.float32), buffer=(kind == "buffer"))


def _lin(spec, name, out_f, in_f, bias=True):
    spec[name + ".weight"] = ((out_f, in_f), "param")
    if bias:
        spec[name + ".bias"] = ((out_f,), "param")


def _conv(spec, name, out_c, in_c, k, bias=True):
    spec[name + ".weight"] = ((out_c, in_c, k), "param")
    if bias:
        spec[name + ".bias"] = ((out_c,), "param")


def _norm(spec, name, c):
    spec[name + ".weight"] = ((c,), "param")
    spec[name + ".bias"] = ((c,), "param")


def _bn(spec, name, c):
    _norm(spec, name, c)
    spec[name + ".running_mean"] = ((c,), "buffer")
    spec[name + ".running_var"] = ((c,), "buffer")
    spec[name + ".num_batches_tracked"] = ((), "long_buffer")


def _ffn(spec, name, adim, units, ff_type, k):
    """multi_layer_conv.py:26-50 / :82-92, positionwise_feed_forward (linear)."""
    if ff_type == "conv1d":
        _conv(spec, name + ".w_1", units, adim, k)
        _conv(spec, name + ".w_2", adim, units, k)
    elif ff_type == "conv1d-linear":
        _conv(spec, name + ".w_1", units, adim, k)
        _lin(spec, name + ".w_2", adim, units)
    elif ff_type == "linear":
        _lin(spec, name + ".w_1", units, adim)
        _lin(spec, name + ".w_2", adim, units)
    else:
        raise NotImplementedError("Support only linear or conv1d.")


def conformer_spec(spec, prefix, adim, heads, units, n_blocks, ff_type, ff_kernel,
                   macaron, use_cnn, cnn_kernel, attn_type="legacy_rel_selfattn",
                   normalize_before=True):
    """Keys of jatts.modules.conformer.encoder.Encoder (encoder.py:70-231)."""
    dk = adim // heads
    for i in range(n_blocks):
        p = f"{prefix}encoders.{i}."
        if attn_type in ("legacy_rel_selfattn", "rel_selfattn"):
            spec[p + "self_attn.pos_bias_u"] = ((heads, dk), "param")
            spec[p + "self_attn.pos_bias_v"] = ((heads, dk), "param")
        for n in ("linear_q", "linear_k", "linear_v", "linear_out"):
            _lin(spec, p + "self_attn." + n, adim, adim)
        if attn_type in ("legacy_rel_selfattn", "rel_selfattn"):
            _lin(spec, p + "self_attn.linear_pos", adim, adim, bias=False)
        _ffn(spec, p + "feed_forward", adim, units, ff_type, ff_kernel)
        if macaron:
            _ffn(spec, p + "feed_forward_macaron", adim, units, ff_type, ff_kernel)
        if use_cnn:
            _conv(spec, p + "conv_module.pointwise_conv1", 2 * adim, adim, 1)
            _conv(spec, p + "conv_module.depthwise_conv", adim, 1, cnn_kernel)
            _bn(spec, p + "conv_module.norm", adim)
            _conv(spec, p + "conv_module.pointwise_conv2", adim, adim, 1)
        _norm(spec, p + "norm_ff", adim)
        _norm(spec, p + "norm_mha", adim)
        if macaron:
            _norm(spec, p + "norm_ff_macaron", adim)
        if use_cnn:
            _norm(spec, p + "norm_conv", adim)
            _norm(spec, p + "norm_final", adim)
    if normalize_before:
        _norm(spec, prefix + "after_norm", adim)


def predictor_spec(spec, prefix, idim, n_layers, n_chans, k):
    """duration_predictor.py:60-76 / variance_predictor.py:47-63."""
    for i in range(n_layers):
        _conv(spec, f"{prefix}conv.{i}.0", n_chans, idim if i == 0 else n_chans, k)
        _norm(spec, f"{prefix}conv.{i}.2", n_chans)
    _lin(spec, prefix + "linear", 1, n_chans)


def postnet_spec(spec, prefix, odim, n_layers, n_chans, n_filts, use_bn=True):
    """pre_postnets.py:108-170."""
    for i in range(n_layers):
        ic = odim if i == 0 else n_chans
        oc = odim if i == n_layers - 1 else n_chans
        spec[f"{prefix}postnet.{i}.0.weight"] = ((oc, ic, n_filts), "param")
        if use_bn:
            _bn(spec, f"{prefix}postnet.{i}.1", oc)


def new_spec():
    return OrderedDict()
