#!/usr/bin/env python3
"""Where does the HOST spend a training step?  cProfile over a few steps of one trainer (bench.py's construction), sorted by own time.
    python tools/profile_train_host.py --kind vits [--steps 3]"""
import argparse
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", default="vits", choices=["fs2", "matcha", "matcha_mas", "vits"])
    ap.add_argument("--steps", type=int, default=3)
    a = ap.parse_args()
    import bench
    dev = torch.device("cuda:0")
    holder = {}
    orig = bench.time.perf_counter
    # reuse bench.train_step_line to build model / batch / trainer, then profile extra steps through a hook on the trainer class
    from jatts_amd import training
    cls = {"fs2": training.FastSpeech2Trainer, "matcha": training.MatchaTTSTrainer, "matcha_mas": training.MatchaTTSTrainer, "vits": training.VITSTrainer}[a.kind]
    real = cls.train_step

    def spy(self, batch):
        holder["tr"], holder["b"] = self, batch
        return real(self, batch)
    cls.train_step = spy
    bench.train_step_line(dev, 1, a.kind)
    cls.train_step = real
    tr, b = holder["tr"], holder["b"]
    tr.capture_graph = False
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(a.steps):
        tr.train_step(b)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)
    st.print_callers("method 'to' of")


if __name__ == "__main__":
    main()
