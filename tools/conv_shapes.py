#!/usr/bin/env python3
"""Per-shape jatts_conv1d time inside the real workloads (live HIP events, jatts_amd.hip profile hooks): one training step of a model
(--train fs2|matcha|matcha_mas|vits) or one inference step (--infer fs2|matcha|vits).  Run it once per JATTS_CONV_F32_TILE setting
(unset = the product heuristic, 1 / 2 = the LDS-staged tiles, 3 = register-streamed) and diff the tables: that is how the f32 kernel
heuristic is tuned.
    JATTS_CONV_F32_TILE=2 python tools/conv_shapes.py --train vits"""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--train", default=None)
    ap.add_argument("--infer", default=None)
    ap.add_argument("--top", type=int, default=40)
    a = ap.parse_args()
    import bench
    from jatts_amd import hip
    dev = torch.device("cuda:0")
    if a.train:
        import types
        # reuse bench.py's construction; time the profiled step ourselves
        orig = hip.flops_begin
        line = bench.train_step_line(dev, 1, a.train)
        print(f"# {a.train} train step: {line['ms_per_step']:.1f} ms (one timed step)")
        hip.profile_begin()
        bench.train_step_line(dev, 1, a.train)
        recs = hip.profile_end()
    else:
        ns = argparse.Namespace(vocoder="22k", t_text=128, frames_per_token=6, steps=1, warmup=1, batch=64)
        job = bench.Job(a.infer, ns, dev, 0, 64 if a.infer != "vits" else 32)
        job.set_precision("fp32")
        job.text2mel()
        torch.cuda.synchronize()
        hip.profile_begin()
        job.text2mel()
        recs = hip.profile_end()
    agg = collections.defaultdict(lambda: [0, 0.0])
    for tag, meta, ms in recs:
        if tag == "conv1d":
            agg[meta][0] += 1
            agg[meta][1] += ms
    tot = sum(v[1] for v in agg.values())
    print(f"# JATTS_CONV_F32_TILE={os.environ.get('JATTS_CONV_F32_TILE', '(unset)')}: {len(recs)} timed launches, conv1d total {tot:.2f} ms")
    for meta, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:a.top]:
        c, no, k, rows = meta
        print(f"{c:5d} -> {no:5d} k={k:2d} rows={rows:7d}  x{n:3d}  {ms:8.3f} ms  {2.0 * c * no * k * rows * n / ms / 1e9:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
