"""GPU parity for SURVEY 8(f).1: batched monotonic alignment search / viterbi_decode and AlignmentModule through the C ABI.
Integer results (path, durations) are bit-exact against the oracle and the reference golden; log-probabilities within 2e-5."""
import json

import numpy as np
import pytest
import torch

from helpers import maxdiff

pytestmark = pytest.mark.gpu


def _logsoftmax(rng, tm, ti, scale=2.0):
    z = rng.standard_normal((tm, ti)).astype(np.float32) * scale
    return (z - np.log(np.exp(z).sum(1, keepdims=True))).astype(np.float32)


def test_mas_matches_reference_golden(cuda, lib, golden_dir):
    from jatts_amd import hip
    z = np.load(golden_dir + "/mas_kat.npz")
    n = int(z["n_mas"])
    mats = [z[f"mas{i}_logp"] for i in range(n)]
    fl, tl = [m.shape[0] for m in mats], [m.shape[1] for m in mats]
    rb_f, rb_t = hip.RaggedBatch(fl, cuda), hip.RaggedBatch(tl, cuda)
    ld = hip.round_up(max(tl), 8)
    lp = torch.full((sum(fl), ld), float("nan"))
    o = 0
    for m in mats:
        lp[o:o + m.shape[0], : m.shape[1]] = torch.tensor(m)
        o += m.shape[0]
    path, dur, score = hip.mas_viterbi(rb_f, rb_t, lp.to(cuda))
    o = ot = 0
    for i, m in enumerate(mats):
        ref = z[f"mas{i}_path"]
        got = path[o:o + m.shape[0]].cpu().numpy()
        assert np.array_equal(got, ref), f"case {i} {m.shape}"
        assert np.array_equal(dur[ot:ot + m.shape[1]].cpu().numpy(), np.bincount(ref, minlength=m.shape[1]))
        want = float(m[np.arange(m.shape[0]), ref].astype(np.float64).sum())
        assert abs(float(score[i]) - want) <= 1e-9 * max(1.0, abs(want))
        o, ot = o + m.shape[0], ot + m.shape[1]


@pytest.mark.parametrize("shape", [(768, 128), (1500, 200), (64, 1024), (300, 333)])
def test_mas_large_against_oracle(cuda, lib, shape):
    """BASELINE-sized and limit-sized problems, 8 utterances per launch, against the float64 oracle."""
    from jatts_amd import hip
    from oracle.mas_oracle import monotonic_alignment_search
    rng = np.random.default_rng(shape[0] + shape[1])
    tm, ti = shape
    fl = [tm, max(1, tm // 2), tm, 1, tm - 3, tm, tm // 3 + 1, tm]
    tl = [ti, max(1, ti // 3), 1, 1, ti, ti - 1 if ti > 1 else 1, ti // 2 + 1, ti]
    mats = [_logsoftmax(rng, f, t) for f, t in zip(fl, tl)]
    rb_f, rb_t = hip.RaggedBatch(fl, cuda), hip.RaggedBatch(tl, cuda)
    ld = hip.round_up(max(tl), 8)
    lp = torch.zeros(sum(fl), ld)
    o = 0
    for m in mats:
        lp[o:o + m.shape[0], : m.shape[1]] = torch.tensor(m)
        o += m.shape[0]
    path, dur, _ = hip.mas_viterbi(rb_f, rb_t, lp.to(cuda))
    o = ot = 0
    for m in mats:
        ref = monotonic_alignment_search(m)
        assert np.array_equal(path[o:o + m.shape[0]].cpu().numpy(), ref), m.shape
        assert int(dur[ot:ot + m.shape[1]].sum()) == m.shape[0]
        o, ot = o + m.shape[0], ot + m.shape[1]


def test_alignment_module_and_viterbi_decode(cuda, lib, golden_dir):
    """Reference-signature classes on the golden batch: state_dict keys, padded tensors in, (ds, bin_loss) out."""
    from jatts_amd.alignments import AlignmentModule, viterbi_decode
    from jatts_amd.synthetic import synth_state_dict
    z = np.load(golden_dir + "/mas_kat.npz")
    keys = json.loads(str(z["keys"]))
    m = AlignmentModule(int(z["adim"]), int(z["odim"]))
    assert [[k, list(v.shape)] for k, v in m.state_dict().items()] == keys
    m.load_state_dict(synth_state_dict({k: tuple(s) for k, s in keys}, 7))
    m = m.to(cuda)
    tl, fl = z["text_lengths"], z["feats_lengths"]
    xm = torch.arange(int(tl.max()))[None, :] >= torch.tensor(tl)[:, None]
    lp = m(torch.tensor(z["text"]).to(cuda), torch.tensor(z["feats"]).to(cuda), xm.to(cuda), feats_lengths=fl)  # padded in, ragged inside
    ref = torch.tensor(z["log_p_attn"])
    for b in range(len(tl)):   # golden = the reference called per utterance (no pad leakage), see make_golden_mas.py
        assert maxdiff(lp[b, : fl[b], : tl[b]], ref[b, : fl[b], : tl[b]]) <= 2e-5
        assert torch.isinf(lp[b, : fl[b], tl[b]:]).all()
    ds, bin_loss = viterbi_decode(ref.to(cuda), torch.tensor(tl), torch.tensor(fl))
    assert np.array_equal(ds.cpu().numpy(), z["ds"])
    assert abs(float(bin_loss) - float(z["bin_loss"])) <= 1e-5
    ds2, _ = viterbi_decode(lp, torch.tensor(tl), torch.tensor(fl))   # end to end on the HIP log-probabilities
    assert np.array_equal(ds2.cpu().numpy(), z["ds"])


def test_mas_limits_fail_loudly_and_empty_sequences_are_handled(cuda, lib):
    from jatts_amd import hip
    from jatts_amd._abi import JattsHipError
    rng = np.random.default_rng(1)
    # an utterance with no frames in the middle of the batch: durations 0, score 0, neighbours unaffected
    fl, tl = [40, 0, 25], [7, 3, 5]
    mats = [_logsoftmax(rng, f, t) for f, t in zip(fl, tl)]
    rb_f, rb_t = hip.RaggedBatch(fl, cuda), hip.RaggedBatch(tl, cuda)
    lp = torch.zeros(sum(fl), 8)
    o = 0
    for m in mats:
        lp[o:o + m.shape[0], : m.shape[1]] = torch.tensor(m)
        o += m.shape[0]
    path, dur, score = hip.mas_viterbi(rb_f, rb_t, lp.to(cuda))
    assert dur.cpu().tolist()[7:10] == [0, 0, 0] and float(score[1]) == 0.0
    assert int(dur[:7].sum()) == 40 and int(dur[10:].sum()) == 25
    # more than 1024 tokens, or decision bits beyond LDS: refused, not silently wrong
    with pytest.raises(JattsHipError):
        hip.mas_viterbi(hip.RaggedBatch([8], cuda), hip.RaggedBatch([1025], cuda), torch.zeros(8, 1032, device=cuda))
    with pytest.raises(JattsHipError):
        hip.mas_viterbi(hip.RaggedBatch([4000], cuda), hip.RaggedBatch([1000], cuda), torch.zeros(4000, 1000, device=cuda))
