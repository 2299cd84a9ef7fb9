#!/usr/bin/env python3
"""Golden vectors for SURVEY 8(f).1 (AlignmentModule + monotonic alignment search + viterbi_decode) from the REAL reference
(/root/reference/jatts/modules/alignments.py), run in the build container only:
    python tests/golden/make_golden_mas.py  ->  tests/golden/mas_kat.npz
numba is absent: ``jit`` becomes the identity, so the reference's own statements run under numpy."""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from jatts_amd.synthetic import synth_state_dict  # noqa: E402


def main():
    class _T:
        def __getitem__(self, k):
            return self

        def __call__(self, *a, **k):
            return self
    nb = types.ModuleType("numba")
    nb.jit = lambda *a, **k: (lambda f: f)
    for n in ("float64", "float32", "int8", "int32", "int64", "boolean"):
        setattr(nb, n, _T())
    sys.modules["numba"] = nb
    from jatts.modules.alignments import AlignmentModule, _monotonic_alignment_search, viterbi_decode

    out = {}
    # ---- AlignmentModule + viterbi_decode on a padded batch
    adim, odim = 32, 20
    m = AlignmentModule(adim, odim).eval()
    ref_sd = m.state_dict()
    m.load_state_dict(synth_state_dict(ref_sd, 7))
    g = torch.Generator().manual_seed(8)
    tl, fl = [9, 6], [23, 15]
    text = torch.randn(2, max(tl), adim, generator=g)
    feats = torch.randn(2, max(fl), odim, generator=g)
    # The module is called per utterance on unpadded inputs (SURVEY 8a note N1: the reference's padded batch leaks pad
    # rows through the k=3 convolutions, so the parity target is the B=1 call); pad tokens are -inf, pad frames 0.
    lp = torch.zeros(2, max(fl), max(tl))
    lp[:, :, :] = float("-inf")
    with torch.no_grad():
        for b in range(2):
            lp[b, : fl[b], : tl[b]] = m(text[b:b + 1, : tl[b]], feats[b:b + 1, : fl[b]])[0]
            lp[b, fl[b]:, :] = 0.0
        ds, bin_loss = viterbi_decode(lp, torch.tensor(tl), torch.tensor(fl))
    out.update(keys=json.dumps([[k, list(v.shape)] for k, v in ref_sd.items()]), adim=adim, odim=odim,
               text=text.numpy(), feats=feats.numpy(), text_lengths=np.array(tl), feats_lengths=np.array(fl),
               log_p_attn=lp.numpy(), ds=ds.numpy(), bin_loss=float(bin_loss))
    # ---- bare MAS on random log-softmax matrices (float32, as viterbi_decode passes them), incl. T_inp > T_mel
    shapes = [(50, 12), (7, 7), (5, 9), (300, 100), (1, 1), (40, 1), (129, 64)]
    rng = np.random.default_rng(9)
    for n, (tm, ti) in enumerate(shapes):
        z = rng.standard_normal((tm, ti)).astype(np.float32) * 2.0
        z = z - np.log(np.exp(z).sum(1, keepdims=True))
        out[f"mas{n}_logp"] = z.astype(np.float32)
        out[f"mas{n}_path"] = _monotonic_alignment_search(z.astype(np.float32)).astype(np.int64)
    out["n_mas"] = len(shapes)
    np.savez_compressed(os.path.join(HERE, "mas_kat.npz"), **out)
    print("mas_kat.npz", os.path.getsize(os.path.join(HERE, "mas_kat.npz")), "bin_loss", float(bin_loss), "ds", ds.numpy())


if __name__ == "__main__":
    main()
