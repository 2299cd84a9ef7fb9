#!/usr/bin/env python3
"""Timing of BASELINE.json's other stage-4 configs on one GPU (not the headline bench line):
config 3 = MatchaTTS_MAS (10 Euler steps) + HiFi-GAN, 64 utts x 128 phonemes; config 5 = mel-VITS with 192-d speaker
embeddings + HiFi-GAN, 32 utts x 128 phonemes.  Synthetic weights / inputs; prints one JSON line per config.
    python tools/bench_models.py [--steps 3]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from jatts_amd.models import VITS, MatchaTTS_MAS  # noqa: E402
from jatts_amd.synthetic import (HIFIGAN_V1_24K, MATCHA_MAS_JSUT, VITS_JSUT, synth_hifigan_state, synth_state_dict,  # noqa: E402
                                 synth_texts)
from jatts_amd.vocoder import Vocoder  # noqa: E402


SHAPES = False


def timed(fn, steps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    if SHAPES:
        from jatts_amd import hip
        hip.profile_begin()
        fn()
        fam = {}
        for tag, meta, ms in hip.profile_end():
            fam.setdefault((tag, meta), []).append(ms)
        rows = []
        for (tag, meta), v in fam.items():
            fl = 2.0 * meta[0] * meta[1] * meta[2] * meta[3] * len(v) if tag == "conv1d" else 0.0
            rows.append((sum(v), tag, meta, len(v), fl / sum(v) / 1e9 if fl else 0.0))
        for tot, tag, meta, n, tf in sorted(rows, key=lambda r: -r[0])[:24]:
            print(f"  {tag:8s} {str(meta):34s} n={n:4d} total {tot:8.2f} ms  avg {tot / n * 1e3:8.1f} us  {tf:6.1f} TF", file=sys.stderr)
    return dt, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--precision", default="fp16", choices=["fp16", "fp32", "fp32_split", "fp32_bf16x3", "fp32_bf16x3_6p"])
    ap.add_argument("--model", default="both", choices=["both", "matcha", "vits"])
    ap.add_argument("--no-vocoder", action="store_true")
    ap.add_argument("--shapes", action="store_true", help="per-shape conv1d / attention table from HIP-event records")
    a = ap.parse_args()
    global SHAPES
    SHAPES = a.shapes
    dev = torch.device("cuda:0")
    ones, zeros = [1.0] * 80, [0.0] * 80
    voc = Vocoder(synth_hifigan_state(HIFIGAN_V1_24K, 0),
                  {"sampling_rate": 24000, "generator_type": "HiFiGANGenerator", "generator_params": HIFIGAN_V1_24K},
                  {"mean": zeros, "scale": ones}, dev, trg_stats={"mean": zeros, "scale": ones})
    voc.set_precision(a.precision)
    hop = voc.model.hop

    def report(name, dt, r, y, n_utts):
        frames = sum(r["olens"])
        print(json.dumps({"config": name, "utterances": n_utts, "frames": frames, "samples": frames * hop,
                          "ms_per_batch": dt * 1e3, "samples_per_s": frames * hop / dt, "rtf": dt / (frames * hop / 24000.0),
                          "finite": bool(torch.isfinite(y).all())}))

    if a.model in ("both", "matcha"):
        run_matcha_cfg(a, dev, voc, report)
    if a.model in ("both", "vits"):
        run_vits_cfg(a, dev, voc, report)


def run_matcha_cfg(a, dev, voc, report):
    m = MatchaTTS_MAS(idim=45, **MATCHA_MAS_JSUT)
    m.load_state_dict(synth_state_dict(m.state_dict(), 0))
    m = m.to(dev).set_precision(a.precision)
    texts = [t.to(dev) for t in synth_texts(64, 128, 45, seed=1)]
    dur = [torch.full((128,), 6, dtype=torch.int64, device=dev) for _ in texts]   # durations pinned to 6 frames / phoneme

    def run_matcha():
        r = m.inference_batch(texts, n_timesteps=10, temperature=0.667, durations=dur)
        return r, (r["feat_gen"] if a.no_vocoder else voc.decode_batch(r["feats_rb"], r["feat_gen"]))
    dt, (r, y) = timed(run_matcha, a.steps)
    report("3: MatchaTTS_MAS(10 Euler steps)+HiFi-GAN 24k, 64x128 phonemes x 6 frames", dt, r, y, 64)


def run_vits_cfg(a, dev, voc, report):
    v = VITS(idim=45, spk_embed_dim=192, **VITS_JSUT)
    v.load_state_dict(synth_state_dict(v.state_dict(), 0))
    v = v.to(dev).set_precision(a.precision)
    texts = [t.to(dev) for t in synth_texts(32, 128, 45, seed=3)]
    spk = torch.randn(32, 192, generator=torch.Generator().manual_seed(3)).to(dev)
    dur = [torch.full((128,), 6, dtype=torch.int64, device=dev) for _ in texts]

    def run_vits():
        r = v.inference_batch(texts, spk, noise_scale=0.667, durations=dur)
        return r, (r["feat_gen"] if a.no_vocoder else voc.decode_batch(r["feats_rb"], r["feat_gen"]))
    dt, (r, y) = timed(run_vits, a.steps)
    report("5: mel-VITS(spk 192)+HiFi-GAN 24k, 32x128 phonemes x 6 frames", dt, r, y, 32)


if __name__ == "__main__":
    main()
