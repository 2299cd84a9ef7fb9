"""Shared test helpers: rebuild the golden models' weights from (name, shape, seed)."""
import json
import os

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
