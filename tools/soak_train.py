#!/usr/bin/env python3
"""Soak run of FastSpeech2Trainer at recipe size on one fixed synthetic batch (memorisation): the loss must fall steadily and stay
finite over a few hundred steps (dropout on, WarmupLR, gradient clipping).  python tools/soak_train.py [--steps 200] [--graph]"""
import argparse
import json
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from jatts_amd.models import FastSpeech2  # noqa: E402
from jatts_amd.synthetic import FS2_JSUT, synth_state_dict  # noqa: E402
from jatts_amd.training import FastSpeech2Trainer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--graph", action="store_true", help="FastSpeech2Trainer(capture_graph=True): the ragged batch replayed as one hipGraph")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    m = FastSpeech2(idim=45, **{**FS2_JSUT, "stop_gradient_from_pitch_predictor": True, "use_masking": True})
    sd = synth_state_dict(m.state_dict(), 0)
    m.load_state_dict(sd)
    m = m.to(dev)
    g = torch.Generator().manual_seed(5)
    B, T = a.batch, 128
    il = torch.randint(64, T + 1, (B,), generator=g)
    il[0] = T
    ds = torch.zeros(B, T, dtype=torch.long)
    for b in range(B):
        ds[b, : il[b]] = torch.randint(2, 9, (int(il[b]),), generator=g)
    ol = ds.sum(1)
    ys = torch.zeros(B, int(ol.max()), 80)
    for b in range(B):
        ys[b, : ol[b]] = torch.randn(int(ol[b]), 80, generator=g) * 0.5
    mask = (torch.arange(T)[None, :] < il[:, None]).float().unsqueeze(-1)
    batch = dict(xs=(torch.randint(1, 45, (B, T), generator=g) * mask.squeeze(-1).long()).to(dev), ilens=il, ys=ys.to(dev), olens=ol, durations=ds.to(dev),
                 duration_lens=il, pitch=(torch.randn(B, T, 1, generator=g) * mask).to(dev), pitch_lens=il,
                 energys=(torch.randn(B, T, 1, generator=g) * mask).to(dev), energy_lens=il)
    tr = FastSpeech2Trainer(m, lr=1.0e-3, grad_norm=1.0, warmup_steps=50, capture_graph=a.graph)
    hist, t0 = [], time.perf_counter()
    for s in range(a.steps):
        out = tr.train_step(batch)
        if s % 20 == 0 or s == a.steps - 1:
            hist.append((s, {k: round(float(v), 4) for k, v in out.items()}))
            print(hist[-1], flush=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    first, last = hist[0][1]["loss"], hist[-1][1]["loss"]
    print(json.dumps({"steps": a.steps, "ragged_batch": [int(il.min()), int(il.max()), int(ol.min()), int(ol.max())], "loss_first": first, "loss_last": last,
                      "finite": all(math.isfinite(h[1]["loss"]) for h in hist), "ms_per_step": dt / a.steps * 1e3}))


if __name__ == "__main__":
    main()
