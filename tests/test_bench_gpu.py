"""bench.py end to end at N = 1 on a reduced batch: the printed line is the last thing on stdout, parses, is small and
carries roofline + cpu_baseline (what the driver's BENCH record needs)."""
import json
import os
import subprocess
import sys

import pytest

from test_bench_line_cpu import check_line

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_line_n1(cuda, lib):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "4", "--no-pmc",
           "--no-configs", "--no-train", "--cpu-budget", "3"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    last = r.stdout.rstrip().splitlines()[-1]
    d = check_line(last)
    # round 6: the headline is the f32-equivalent emulated arithmetic (VERDICT r5's ruling), exact f32 rides beside it with its own roofline
    assert d["dtype"].startswith("f32 (emulated: 3 exact bf16 terms per operand, 7 MFMA products") and d["cpu_baseline"]["cores"] >= 1 and d["speedup_vs_cpu_rtf"] > 1.0
    assert d["fast_mode"]["max_abs_err_wave"] < 3e-2
    per_step = 4 * 128 * 6 * d["config"]["hop"]
    assert abs(d["value"] * d["ms_per_step"] / 1e3 - per_step) <= 1e-4 * per_step
    assert d["ragged"]["t_text"] == "U{64..128}" and 0.5 < d["ragged"]["per_sample_efficiency"] < 1.5 and d["ragged"]["value"] > 0
    ex = d["exact_f32_mode"]
    assert ex["dtype"] == "f32" and ex["max_abs_err_wave"] < 2e-4 and ex["value"] > 0 and 0 < ex["roofline"]["frac"] < 1      # the exact-f32 waveform tolerance
    rf = d["roofline"]
    assert rf["practical_peak"] > 0 and abs(rf["frac_of_practical"] - rf["achieved"] / rf["practical_peak"]) < 1e-3
    v = d["vocoder_24k"]
    assert v["hop"] == 300 and abs(v["value"] * v["ms_per_step"] / 1e3 - 4 * 128 * 6 * 300) <= 1e-4 * 4 * 128 * 6 * 300 and v["exact_f32"]["value"] > 0
    b1 = d["b1_latency"]
    assert b1["ms"] > 0 and b1["eager_ms"] > 0 and b1["utterance"].startswith("128 phonemes -> 768 frames")
    if b1.get("kernel_ms"):          # (rocprofv3 on the box: the kernel-trace child ran)
        assert 0.5 < b1["wall_over_kernel"] < 3.0
    detail = json.load(open(os.path.join(ROOT, "bench_detail.json")))
    assert "resunit_by_shape" in detail and "conv1d_by_shape" in detail and detail["cpu_baseline"]["sample"]
    for k in ("f32_emul6_mode", "f32_split_mode"):       # the ineligible arithmetics: detail file only
        assert detail[k]["max_abs_err_wave"] < 2e-4 and detail[k]["value"] > 0 and k not in d, k
    assert detail["mfma_ceilings"]["bf16"]["form"] in ("16x16x32", "32x32x16") and detail["mfma_ceilings"]["bf16"]["form_16x16x32"]["lds"]["tflops"] > 500
