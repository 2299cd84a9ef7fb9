#!/usr/bin/env python3
"""Round-3 golden vectors from the REAL reference (/root/reference): the shapes bench.py times.

    python tests/golden/make_golden_r3.py          (build container only; the reference never travels)

fs2_bench768.npz: conf/fastspeech2.v1.yaml-width FastSpeech2 (jatts_amd.synthetic.FS2_JSUT), synthetic weights (seed 0)
with the duration head pinned to 6 frames per phoneme, utterances 0 and 37 of bench.py's batch
(synth_texts(64, 128, 45, seed=1)) through the reference's own B=1 `inference()` -> (768, 80) mel each.  Weights are
rebuilt from (name, shape, seed) by the tests, never stored.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import ROOT, import_reference, np_  # noqa: E402

sys.path.insert(0, ROOT)
from jatts_amd.synthetic import FS2_JSUT, pin_duration_head, synth_state_dict, synth_texts  # noqa: E402

UTTS = (0, 37)


def main():
    torch.set_num_threads(8)
    FastSpeech2 = import_reference()
    model = FastSpeech2(idim=45, **FS2_JSUT).eval()
    ref_sd = model.state_dict()
    sd = pin_duration_head(synth_state_dict(ref_sd, 0), 6)
    model.load_state_dict(sd)
    texts = synth_texts(64, 128, 45, seed=1)
    out = {"keys": json.dumps([[k, list(v.shape)] for k, v in ref_sd.items()]), "utts": np.array(UTTS, dtype=np.int64)}
    from oracle.fs2_oracle import fs2_inference
    for j, u in enumerate(UTTS):
        with torch.no_grad():
            r = model.inference(texts[u])
        assert r["feat_gen"].shape == (768, 80) and bool((r["duration"] == 6).all())
        out[f"u{j}_text"] = np_(texts[u])
        out[f"u{j}_feat_gen"] = np_(r["feat_gen"])
        out[f"u{j}_duration"] = np_(r["duration"])
        out[f"u{j}_pitch"] = np_(r["pitch"])
        out[f"u{j}_energy"] = np_(r["energy"])
        o = fs2_inference(sd, texts[u], 2)
        print(f"bench utt {u}: oracle-vs-ref mel max|d| =", float((o["feat_gen"] - r["feat_gen"]).abs().max()),
              "mel absmax", float(r["feat_gen"].abs().max()))
    path = os.path.join(HERE, "fs2_bench768.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path))


if __name__ == "__main__":
    main()
