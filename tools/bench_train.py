#!/usr/bin/env python3
"""One-GPU timing of FastSpeech2Trainer.train_step at the recipe's shape: conf/fastspeech2.v1.yaml model,
batch_size 32, 128 phonemes x 6 frames per utterance, synthetic weights / targets.  Prints one JSON line; --shapes adds the
per-kernel-family table from HIP-event records.
    python tools/bench_train.py [--steps 5] [--batch 32]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from jatts_amd.models import FastSpeech2, MatchaTTS  # noqa: E402
from jatts_amd.synthetic import FS2_JSUT, MATCHA_MAS_JSUT, matcha_golden_tweaks, synth_state_dict  # noqa: E402
from jatts_amd.training import FastSpeech2Trainer, MatchaTTSTrainer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--t-text", type=int, default=128)
    ap.add_argument("--frames", type=int, default=6)
    ap.add_argument("--model", default="fs2", choices=["fs2", "matcha"], help="matcha = tts1 MatchaTTS (matcha_tts.v1.prior.steplr.large.yaml)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    if a.model == "fs2":
        m = FastSpeech2(idim=45, **{**FS2_JSUT, "stop_gradient_from_pitch_predictor": True, "use_masking": True})
        m.load_state_dict(synth_state_dict(m.state_dict(), 0))
    else:
        m = MatchaTTS(idim=45, **MATCHA_MAS_JSUT)      # the tts1 yaml has the same model_params as the MAS recipe
        m.load_state_dict(matcha_golden_tweaks(synth_state_dict(m.state_dict(), 0)))
    m = m.to(dev)
    g = torch.Generator().manual_seed(5)
    B, T = a.batch, a.t_text
    il = torch.full((B,), T, dtype=torch.long)
    ds = torch.full((B, T), a.frames, dtype=torch.long)
    ol = ds.sum(1)
    batch = dict(xs=torch.randint(1, 45, (B, T), generator=g), ilens=il, ys=torch.randn(B, int(ol.max()), 80, generator=g), olens=ol,
                 durations=ds, duration_lens=il, pitch=torch.randn(B, T, 1, generator=g), pitch_lens=il,
                 energys=torch.randn(B, T, 1, generator=g), energy_lens=il)
    batch = {k: v.to(dev) if k in ("xs", "ys", "durations", "pitch", "energys") else v for k, v in batch.items()}
    tr = (FastSpeech2Trainer if a.model == "fs2" else MatchaTTSTrainer)(m, lr=1e-4, grad_norm=1.0, warmup_steps=0)
    l0 = float(tr.train_step(batch)["loss"])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = tr.train_step(batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    frames = int(ol.sum())
    print(json.dumps({"workload": f"{'FastSpeech2 v1' if a.model == 'fs2' else 'MatchaTTS (tts1)'} train step, batch {B} x {T} phonemes x {a.frames} frames", "ms_per_step": dt * 1e3,
                      "frames_per_s": frames / dt, "loss_first": l0, "loss_last": float(out["loss"]),
                      "peak_mem_gb": torch.cuda.max_memory_allocated() / 2 ** 30}))


if __name__ == "__main__":
    main()
