#!/usr/bin/env python3
"""SURVEY 8(f).1 measurement: batched monotonic alignment search (jatts_mas_viterbi) at the bench geometry
(64 utterances x 768 frames x 128 tokens) against the reference-shaped host loop (numpy restatement, one utterance at a
time, as jatts/modules/alignments.py:299-305 does after a device->host copy).  Prints one JSON line."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from jatts_amd import hip  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    B, Tf, Tt = 64, 768, 128
    rng = np.random.default_rng(0)
    z = rng.standard_normal((B, Tf, Tt)).astype(np.float32) * 2
    lp = torch.log_softmax(torch.tensor(z), -1)
    rb_f, rb_t = hip.RaggedBatch([Tf] * B, dev), hip.RaggedBatch([Tt] * B, dev)
    lpd = lp.reshape(B * Tf, Tt).contiguous().to(dev)
    for _ in range(2):
        path, dur, score = hip.mas_viterbi(rb_f, rb_t, lpd)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        path, dur, score = hip.mas_viterbi(rb_f, rb_t, lpd)
    b.record()
    torch.cuda.synchronize()
    gpu_ms = a.elapsed_time(b) / 10
    from oracle.mas_oracle import monotonic_alignment_search
    t0 = time.time()
    n = 4
    ok = True
    for i in range(n):
        ref = monotonic_alignment_search(lp[i].numpy())
        ok &= bool(np.array_equal(ref, path[i * Tf:(i + 1) * Tf].cpu().numpy()))
    cpu_ms_per_utt = (time.time() - t0) / n * 1e3
    cells = B * Tf * Tt
    print(json.dumps({"what": "monotonic alignment search, 64 x 768 frames x 128 tokens", "gpu_ms_per_batch": gpu_ms,
                      "cells_per_s": cells / gpu_ms * 1e3, "bytes_read": cells * 4, "gbs": cells * 4 / gpu_ms / 1e6,
                      "cpu_numpy_ms_per_utterance": cpu_ms_per_utt, "cpu_ms_per_batch_est": cpu_ms_per_utt * B,
                      "paths_equal_oracle": ok}))


if __name__ == "__main__":
    main()
