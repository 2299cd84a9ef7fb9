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
    assert d["dtype"] == "f32" and d["cpu_baseline"]["cores"] >= 1 and d["speedup_vs_cpu_rtf"] > 1.0
    assert d["fast_mode"]["max_abs_err_wave"] < 3e-2
    per_step = 4 * 128 * 6 * d["config"]["hop"]
    assert abs(d["value"] * d["ms_per_step"] / 1e3 - per_step) <= 1e-4 * per_step
    # round 5: the ragged leg (T_text ~ U{64..128} on the same utterance budget) and the emulated arithmetics ride in the same line
    assert d["ragged"]["t_text"] == "U{64..128}" and 0.5 < d["ragged"]["per_sample_efficiency"] < 1.5 and d["ragged"]["value"] > 0
    for k in ("f32_emul_mode", "f32_emul6_mode", "f32_split_mode"):
        assert d[k]["max_abs_err_wave"] < 2e-4 and d[k]["value"] > 0, k          # the exact-f32 waveform tolerance
    detail = json.load(open(os.path.join(ROOT, "bench_detail.json")))
    assert "resunit_by_shape" in detail and "conv1d_by_shape" in detail and detail["cpu_baseline"]["sample"]
