#!/usr/bin/env python3
"""stats.h5 -> stats.npz, for hosts without h5py (this image has none; the reference writes its statistics with
h5py: jatts/bin/compute_statistics.py:94-103 -> datasets `<feat>_mean`, `<feat>_scale`; the vocoder's stats.h5 holds
`mean`, `scale`: vocoder.py:46-53).  Run it where h5py is installed (the machine that trained the model):
    python tools/h5stats_to_npz.py exp/train_phn_x/stats.h5 [out.npz]
Every 1-D/2-D float dataset of the file is copied under its own name, so the .npz answers the same keys."""
import sys

import numpy as np


def convert(src, dst=None):
    import h5py   # not importable here: the tool is for the reference's own environment

    dst = dst or (src[:-3] if src.endswith(".h5") else src) + ".npz"
    out = {}

    def visit(name, obj):
        if isinstance(obj, h5py.Dataset):
            out[name.replace("/", "_") if "/" in name else name] = np.asarray(obj[()], dtype=np.float32)

    with h5py.File(src, "r") as f:
        f.visititems(visit)
    if not out:
        raise SystemExit(f"{src}: no datasets found")
    np.savez(dst, **out)
    return dst, sorted(out)


if __name__ == "__main__":
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    path, keys = convert(*sys.argv[1:3])
    print(f"wrote {path}: {', '.join(keys)}")
