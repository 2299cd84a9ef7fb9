"""CPU: the C-ABI library loads and exports every symbol include/jatts_hip.h declares; host-side
packing / polyphase / schema / sharding logic.  No compute calls (no GPU here)."""
import json
import os
import re

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import ROOT


def test_library_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "jatts_hip.h")).read()
    declared = set(re.findall(r"\b(jatts_[a-z0-9_]+)\s*\(", hdr))
    from jatts_amd import _abi
    assert declared == set(_abi.PROTOTYPES), declared ^ set(_abi.PROTOTYPES)
    for name in declared:
        assert hasattr(lib, name), name
    ver = int(re.search(r"#define JATTS_ABI_VERSION (\d+)", hdr).group(1))
    assert lib.jatts_abi_version() == ver == _abi.ABI_VERSION


def test_ctypes_structs_match_header_field_order():
    hdr = open(os.path.join(ROOT, "include", "jatts_hip.h")).read()
    from jatts_amd import _abi
    for cname, cls in (("jatts_ragged", _abi.Ragged), ("jatts_conv_desc", _abi.ConvDesc),
                       ("jatts_resunit_desc", _abi.ResUnitDesc), ("jatts_relattn_desc", _abi.RelAttnDesc),
                       ("jatts_resblock_desc", _abi.ResBlockDesc)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), hdr, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = re.findall(r"([a-z_0-9]+)(?:\[\d+\])?\s*;", body)
        assert names == [f[0] for f in cls._fields_], (cname, names)


def test_weight_packing_matches_c_index(lib):
    from jatts_amd import hip
    g = torch.Generator().manual_seed(0)
    for (n, c, k) in [(40, 48, 3), (32, 16, 1), (1, 80, 7), (700, 192, 1), (64, 64, 11)]:
        w = torch.randn(n, c, k, generator=g)
        wp = hip.pack_conv_weight(w, hip.F32)
        n_pad, c_pad = hip.round_up(n, 32), hip.round_up(c, 64)
        assert wp.numel() == n_pad * c_pad * k
        for _ in range(50):
            i, j, t = (int(torch.randint(0, m, (1,), generator=g)) for m in (n, c, k))
            assert wp[lib.jatts_conv_weight_index(i, t, j, n_pad, c_pad)] == w[i, j, t]
        assert float(wp.abs().sum()) == pytest.approx(float(w.abs().sum()), rel=1e-5)  # padding is zero


@pytest.mark.parametrize("s,K", [(8, 16), (2, 4), (5, 10), (4, 8), (3, 6)])
def test_polyphase_convtranspose_equals_torch(s, K):
    """ConvTranspose1d == stride-1 conv producing [stride][c_out] phases (the HIP path's formulation)."""
    from jatts_amd import hip
    g = torch.Generator().manual_seed(s)
    cin, cout, L = 6, 4, 11
    w, b, x = torch.randn(cin, cout, K, generator=g), torch.randn(cout, generator=g), torch.randn(1, cin, L, generator=g)
    p = s // 2 + s % 2
    ref = F.conv_transpose1d(x, w, b, stride=s, padding=p, output_padding=s % 2)
    wc, pad = hip.convtranspose_as_conv(w, s, p)
    taps = wc.shape[-1]
    y = F.conv1d(F.pad(x, (pad, taps - 1 - pad)), wc, b.repeat(s))[0].t().reshape(L * s, cout).t()
    assert ref.shape[-1] == L * s and float((y - ref[0]).abs().max()) < 1e-5


def test_state_dict_schema_equals_reference(golden_dir):
    """Key names, order and shapes of jatts.models.FastSpeech2.state_dict() (captured from the reference)."""
    from jatts_amd.models import FastSpeech2
    from jatts_amd.synthetic import FS2_JSUT, FS2_SMALL
    for name, cfg, idim, kw in (("fs2_jsut.npz", FS2_JSUT, 45, {}), ("fs2_small.npz", FS2_SMALL, 20, {}),
                                ("fs2_small_spk.npz", FS2_SMALL, 20, {"spk_embed_dim": 16})):
        keys = json.loads(str(np.load(os.path.join(golden_dir, name))["keys"]))
        sd = FastSpeech2(idim=idim, **cfg, **kw).state_dict()
        assert [k for k, _ in keys] == list(sd.keys())
        assert all(tuple(s) == tuple(sd[k].shape) for k, s in keys)


def test_matcha_and_vits_schemas_equal_reference(golden_dir):
    from jatts_amd.models import VITS, MatchaTTS_MAS
    for name, cls in (("matcha_small.npz", MatchaTTS_MAS), ("vits_small.npz", VITS)):
        z = np.load(os.path.join(golden_dir, name))
        keys = json.loads(str(z["keys"]))
        sd = cls(idim=20, **json.loads(str(z["config"]))).state_dict()
        assert [k for k, _ in keys] == list(sd.keys()), name
        assert all(tuple(s) == tuple(sd[k].shape) for k, s in keys), name


def test_hifigan_loads_weight_norm_checkpoints():
    from jatts_amd.synthetic import HIFIGAN_V1_24K, synth_hifigan_state
    from jatts_amd.vocoder import HiFiGANGenerator
    params = dict(HIFIGAN_V1_24K, channels=512)
    sd = synth_hifigan_state(params)
    wn = {}
    for k, v in sd.items():
        if k.endswith(".weight"):
            norm = v.reshape(v.shape[0], -1).norm(dim=1).reshape(-1, *([1] * (v.dim() - 1)))
            wn[k + "_g"], wn[k + "_v"] = norm, v * 0.5
        else:
            wn[k] = v
    g = HiFiGANGenerator(**params)
    g.load_state_dict(wn)
    for k, v in sd.items():
        assert torch.allclose(g.state_dict()[k], v, atol=1e-6), k
    assert g.hop == 300


def test_unsupported_configs_raise_like_the_reference():
    from jatts_amd.models import FastSpeech2
    with pytest.raises(ValueError):  # reference: NameError on the dead transformer branch
        FastSpeech2(idim=10, odim=80)
    import jatts_amd.models as M
    assert getattr(M, "FastSpeech2") is FastSpeech2  # registry contract of tts_decode.py:139


def test_shard_utterances_is_a_balanced_partition():
    from jatts_amd.hostlogic import shard_utterances
    lens = [5, 100, 7, 64, 64, 3, 90, 12, 1]
    parts = shard_utterances(lens, 4)
    assert sorted(i for p in parts for i in p) == list(range(len(lens)))
    assert max(len(p) for p in parts) <= -(-len(lens) // 4)          # never more than ceil(n / W) per rank: the exchange step's slot count
    load = [sum(lens[i] for i in p) for p in parts]
    order = sorted(range(len(lens)), key=lambda i: (-lens[i], i))
    assert max(load) <= max(sum(lens[i] for i in order[r::4]) for r in range(4))      # no worse than round-robin over the sorted list
    assert parts == shard_utterances(lens, 4)


def test_split_weight_packing_is_an_exact_scaled_hi_lo_pair():
    """hip.pack_conv_weight_split (the JATTS_F32S operand, CPU-checkable: pure torch): per output channel a power-of-two scale puts max |w| in
    [2^14, 2^15); hi + lo reconstructs w * 2^s to 2^-22 of the channel maximum or better; the inverse scales are exact powers of two; an all-zero
    channel takes scale 1; elements sit where jatts_conv_weight_index says, hi and lo halves of a lane side by side."""
    import math

    import torch

    from jatts_amd import hip
    g = torch.Generator().manual_seed(3)
    n, c, k = 40, 64, 3
    w = torch.randn(n, c, k, generator=g) * torch.pow(10.0, torch.rand(n, 1, 1, generator=g) * 6 - 4)
    w[7] = 0.0
    packed, inv = hip.pack_conv_weight_split(w, 64)
    n_pad = 64
    assert packed.dtype == torch.float16 and packed.numel() == 2 * n_pad * c * k and inv.shape == (n_pad,)
    m, e = torch.frexp(inv)
    assert torch.all(m == 0.5) and float(inv[7]) == 1.0 and torch.all(inv[n:] == 1.0)        # exact powers of two
    pk = packed.view(-1, 2, 8).float()              # [fragment lane slot][hi | lo][8]
    for (nn, tap, cc) in [(0, 0, 0), (5, 2, 17), (39, 1, 63), (7, 0, 3), (33, 2, 40)]:
        idx = ((tap * (c // 16) + cc // 16) * (n_pad // 32) + nn // 32) * 64 + 32 * ((cc % 16) // 8) + nn % 32
        hi, lo = float(pk[idx, 0, cc % 8]), float(pk[idx, 1, cc % 8])
        ws = float(w[nn, cc, tap]) / float(inv[nn])
        amax = float(w[nn].abs().max()) / float(inv[nn])
        assert amax == 0.0 or 2.0 ** 14 <= amax < 2.0 ** 15
        assert abs(hi + lo - ws) <= max(amax, 1.0) * 2.0 ** -22, (nn, tap, cc, hi, lo, ws)
        assert hi == float(torch.tensor(ws).half())                                            # hi = f16(ws), round to nearest
    assert math.isfinite(float(packed.float().abs().max())) and float(packed.float().abs().max()) < 65504.0


def test_bf16x3_weight_packing_is_exact():
    """hip.bf16x3_terms / hip.pack_conv_weight_bf16x3 (the JATTS_F32E / JATTS_F32E6 operand, CPU-checkable: pure torch): b0 + b1 + b2 == w EXACTLY
    for every f32 in 2^-110 <= |w| < 3.39e38 (no scale anywhere), |b1| <= 2^-8 |w|, |b2| <= 2^-16 |w| (what bounds the dropped partial products),
    zeros stay zeros, and elements sit where jatts_conv_weight_index says with the three planes of a lane side by side."""
    import torch

    from jatts_amd import hip
    g = torch.Generator().manual_seed(4)
    n, c, k = 40, 64, 3
    w = torch.randn(n, c, k, generator=g) * torch.pow(10.0, torch.rand(n, c, k, generator=g) * 60 - 30)    # 60 decades, element by element
    w[7] = 0.0
    b0, b1, b2 = hip.bf16x3_terms(w)
    assert torch.equal((b0.double() + b1.double() + b2.double()).float(), w)
    nz = w != 0
    assert float((b1.float().abs()[nz] / w.abs()[nz]).max()) <= 2.0 ** -8 and float((b2.float().abs()[nz] / w.abs()[nz]).max()) <= 2.0 ** -16
    m = torch.arange(1 << 23, 1 << 24, 4099, dtype=torch.int32).float() * 2.0 ** -23              # a comb of mantissas in one binade
    t0, t1, t2 = hip.bf16x3_terms(-m)
    assert torch.equal((t0.double() + t1.double() + t2.double()).float(), -m)
    packed = hip.pack_conv_weight_bf16x3(w, 64)
    n_pad = 64
    assert packed.dtype == torch.bfloat16 and packed.numel() == 3 * n_pad * c * k
    pk = packed.view(-1, 3, 8).double()             # [fragment lane slot][b0 | b1 | b2][8]
    for (nn, tap, cc) in [(0, 0, 0), (5, 2, 17), (39, 1, 63), (7, 0, 3), (33, 2, 40), (63, 1, 9)]:
        idx = ((tap * (c // 16) + cc // 16) * (n_pad // 32) + nn // 32) * 64 + 32 * ((cc % 16) // 8) + nn % 32
        want = float(w[nn, cc, tap]) if nn < n else 0.0
        assert float(pk[idx, :, cc % 8].sum()) == want, (nn, tap, cc)
        if nn < n:
            assert float(pk[idx, 0, cc % 8]) == float(torch.tensor(want).bfloat16())            # b0 = bf16(w), round to nearest even
