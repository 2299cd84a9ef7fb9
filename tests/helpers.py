"""Shared test helpers: rebuild the golden models' weights from (name, shape, seed)."""
import json
import os
import sys

import numpy as np
import torch

from jatts_amd.synthetic import synth_state_dict

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name))
    keys = json.loads(str(z["keys"])) if "keys" in z.files else None
    return z, keys


def golden_state(keys, seed):
    return synth_state_dict({k: tuple(s) for k, s in keys}, seed)


def maxdiff(a, b):
    a = torch.as_tensor(a).double().cpu()
    b = torch.as_tensor(b).double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max()) if a.numel() else 0.0


def relerr(a, b):
    a = torch.as_tensor(a).double().cpu()
    b = torch.as_tensor(b).double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).norm() / (b.norm() + 1e-30))


def hifigan_xcheck_case(name):
    """(params, state dict as the fixture's generator loaded it, mel, fixture) of one tests/golden/hifigan_xcheck.npz case: weights rebuilt
    from their seed (make_golden_hifigan_xcheck.py), `wn` as weight-norm (g, v) pairs."""
    from jatts_amd.synthetic import synth_hifigan_state
    z = np.load(os.path.join(GOLDEN, "hifigan_xcheck.npz"))
    params, wseed, frames, mseed = json.loads(str(z["cases"]))[name]
    params = {k: (tuple(tuple(d) if isinstance(d, list) else d for d in v) if isinstance(v, list) else v) for k, v in params.items()}
    sd = synth_hifigan_state(params, seed=wseed)
    if name == "wn":
        sys.path.insert(0, GOLDEN)
        from make_golden_hifigan_xcheck import wn_pairs
        sd, _ = wn_pairs(sd, wseed)
    mel = torch.tensor(z[f"{name}_mel"])
    assert mel.shape[0] == frames and torch.equal(mel, torch.randn(frames, params["in_channels"], generator=torch.Generator().manual_seed(mseed)))
    return params, sd, mel, z
