#!/usr/bin/env python3
"""CPU-baseline thread sweep: the CPU oracle in the reference's B=1 stage-4 loop (tts_decode.py:203-255) over the
same utterances at 1 / 16 / 32 / 64 / 128 threads, text2mel and vocoder timed separately.  bench.py's `cpu_baseline` then runs on
the fastest setting found on ITS box (a short calibration over the same candidates), so the >= 100x claim is made against the best
CPU configuration.
    python tools/cpu_threads_sweep.py [--utts 16] [--out profiles/r03_cpu_threads.json]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def run(fs2_sd, voc_sd, vp, texts, threads):
    from oracle.fs2_oracle import fs2_inference
    from oracle.hifigan_oracle import hifigan_generate
    torch.set_num_threads(threads)
    t_fs2 = t_voc = 0.0
    samples = 0
    with torch.no_grad():
        for text in texts:
            t0 = time.time()
            r = fs2_inference(fs2_sd, text, 2)
            t1 = time.time()
            y = hifigan_generate(voc_sd, r["feat_gen"], vp["upsample_scales"], vp["resblock_dilations"])
            t2 = time.time()
            t_fs2, t_voc, samples = t_fs2 + t1 - t0, t_voc + t2 - t1, samples + int(y.numel())
    return dict(threads=threads, utterances=len(texts), samples=samples, text2mel_s=t_fs2, vocoder_s=t_voc,
                samples_per_s=samples / (t_fs2 + t_voc), rtf=(t_fs2 + t_voc) / (samples / 22050.0))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=16)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r03_cpu_threads.json"))
    a = ap.parse_args()
    from bench import cpu_model
    from jatts_amd import models
    from jatts_amd.synthetic import FS2_JSUT, HIFIGAN_V1_22K, pin_duration_head, synth_hifigan_state, synth_state_dict, synth_texts
    m = models.FastSpeech2(idim=45, **FS2_JSUT)
    sd = pin_duration_head(synth_state_dict(m.state_dict(), 0), 6)
    vsd = synth_hifigan_state(HIFIGAN_V1_22K, 0)
    texts = synth_texts(64, 128, 45, seed=1)
    cores = os.cpu_count() or 1
    run(sd, vsd, HIFIGAN_V1_22K, texts[:1], min(32, cores))      # warm-up (allocator, thread pool)
    rows = []
    for th in (1, 8, 16, 24, 32, 64, 128):
        if th > cores:
            continue
        n = 3 if th == 1 else a.utts
        rows.append(run(sd, vsd, HIFIGAN_V1_22K, texts[:n], th))
        print(json.dumps(rows[-1]), flush=True)
    best = max(rows, key=lambda r: r["samples_per_s"])
    out = {"cpu_model": cpu_model(), "host_logical_cores": cores, "workload": "bench.py's utterances (128 phonemes -> 768 frames -> 196 608 samples each), "
           "oracle/fs2_oracle + oracle/hifigan_oracle (torch CPU f32), one utterance at a time", "rows": rows, "best_threads": best["threads"],
           "best_samples_per_s": best["samples_per_s"]}
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1)
    print("best:", best["threads"], "threads", f"{best['samples_per_s']:.0f} samples/s")


if __name__ == "__main__":
    main()
