"""The ONE JSON line bench.py prints must stay small enough for the driver to parse (round 2's grew to 30 KB and was
recorded as `parsed: null`) and must carry every graded key."""
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")
ROOFLINE = ("bound", "achieved", "peak", "unit", "frac", "practical_peak", "frac_of_practical", "traffic", "kernel", "avg_launch_ms", "alg_bytes", "alg_flops")
CPU = ("value", "unit", "cores", "kind", "sample", "rtf", "cpu_model")


def check_line(line, n_gpus=1):
    assert "\n" not in line and len(line) < 4096, len(line)
    d = json.loads(line)
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == n_gpus and d["unit"] == "samples/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and (d["dtype"] in ("f32", "f16") or d["dtype"].startswith("f32 (emulated: 3 exact bf16 terms per operand, 7 MFMA products"))
    assert isinstance(d["config"]["workload"], str) and "model" not in d["config"]
    for k in ROOFLINE:
        assert k in d["roofline"], k
    assert d["roofline"]["bound"] in ("hbm", "mfma") and 0.0 < d["roofline"]["frac"] < 1.0
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-3
    if n_gpus == 1 and d["cpu_baseline"] is not None:
        for k in CPU:
            assert k in d["cpu_baseline"], k
        assert d["cpu_baseline"]["kind"] in ("port", "reference")
    return d


def full_result():
    out = json.load(open(os.path.join(ROOT, "profiles", "r02_bench_n1.json")))      # a real full-size result (30 KB)
    for e, kind in zip(out["training"], ("fs2", "matcha", "matcha_mas", "vits")):
        e.setdefault("kind", kind)
    return out


def test_compact_line_from_a_full_result():
    out = full_result()
    assert len(json.dumps(out)) > 20000
    d = check_line(bench.compact_line(out, "bench_detail.json"))
    assert d["value"] == float(f"{out['value']:.5g}") and d["detail"] == "bench_detail.json"
    assert set(d["training"]) == {"fs2", "matcha", "matcha_mas", "vits"} and len(d["configs"]) == 2
    assert d["fast_mode"]["dtype"] == "f16" and d["speedup_vs_cpu_rtf"] > 1


def test_compact_line_never_exceeds_the_limit():
    out = full_result()
    out["config"]["workload"] = out["config"]["workload"] + " x" * 600       # something grows again: optional blocks go first
    out["cpu_baseline"]["sample_short"] = "s" * 900
    line = bench.compact_line(out, "bench_detail.json")
    assert len(line) <= bench.LINE_LIMIT
    d = json.loads(line)
    for k in REQUIRED:
        assert k in d, k


def test_compact_line_without_optional_blocks():
    out = full_result()
    for k in ("fast_mode", "configs", "training", "cpu_baseline", "speedup_vs_cpu_rtf"):
        out.pop(k, None)
    out["cpu_baseline"] = None
    d = json.loads(bench.compact_line(out))
    assert d["cpu_baseline"] is None and "detail" not in d and "training" not in d


def test_compact_line_carries_every_block_of_a_round6_result():
    """A real round-6 result (profiles/r06_bench_detail.json: the emulated headline with its live practical ceiling, exact f32 beside it, the 24 kHz
    recipe vocoder, the B = 1 latency block, the ragged leg, two configs, four training lines, the CPU baseline) must fit the line WITHOUT any optional
    block being dropped; the ineligible arithmetics (six products, split f16) stay in the detail file."""
    out = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_detail.json")))
    line = bench.compact_line(out, "bench_detail.json")
    assert len(line) <= bench.LINE_LIMIT
    d = check_line(line)
    for k in ("ragged", "exact_f32_mode", "vocoder_24k", "b1_latency", "fast_mode", "configs", "training", "roofline_conv1d"):
        assert k in d, k
    assert "f32_emul6_mode" not in d and "f32_split_mode" not in d and "f32_emul6_mode" in out and "f32_split_mode" in out
    assert set(d["ragged"]) >= {"t_text", "seed", "value", "ms_per_step", "per_sample_efficiency"}
    assert d["roofline"]["traffic"] and d["roofline"]["traffic_source"] in ("live", "committed")
    # the headline is the emulated arithmetic, priced against the spec peak / 7 AND the live ceiling / 7
    rf = d["roofline"]
    assert d["dtype"].startswith("f32 (emulated") and abs(rf["peak"] - 2500.0 / 7) < 0.01
    assert rf["frac"] < rf["frac_of_practical"] < 1.05 and abs(rf["frac_of_practical"] - rf["achieved"] / rf["practical_peak"]) < 1e-3
    ex = d["exact_f32_mode"]
    assert ex["dtype"] == "f32" and ex["roofline"]["peak"] == 157.3 and 0 < ex["roofline"]["frac"] < 1 and ex["ms_per_step"] > d["ms_per_step"]
    assert 0 <= ex["max_abs_err_wave"] < 1e-5 and 0 <= ex["max_abs_err_mel"] < 1e-4
    v = d["vocoder_24k"]
    assert v["sampling_rate"] == 24000 and v["hop"] == 300 and v["value"] > 0 and v["exact_f32"]["ms_per_step"] > v["ms_per_step"]
    b1 = d["b1_latency"]
    assert b1["ms"] > 0 and b1["kernel_ms"] > 0 and abs(b1["wall_over_kernel"] - b1["ms"] / b1["kernel_ms"]) < 1e-2


def test_traffic_lookup_answers_null_with_a_reason():
    """roofline.traffic never comes from a table that lacks the kernel asked about, and the committed table of this round
    holds the fused-unit kernels of every arithmetic."""
    table = json.load(open(os.path.join(ROOT, "profiles", f"{bench.PROFILE_ROUND}_traffic.json")))["kernels"]
    for prec in bench.PRECISIONS:
        v, why = bench.lookup_traffic(table, prec, 128)
        assert v and v > 1e9 and why is None, (prec, why)
    v, why = bench.lookup_traffic({"some_other_kernel": {"hbm_bytes": 1.0, "launches": 1}}, "fp32_bf16x3", 128)
    assert v is None and "resunit_emul16_kernel" in why
    assert bench.lookup_traffic(None, "fp32", 128) == (None, "no traffic table")
