#!/usr/bin/env python3
"""bench.py — stage-4 hot path throughput on MI355X: FastSpeech2 (JSUT config) + HiFi-GAN v1.

Workload (BASELINE.json configs[1]): random-init weights of the named architectures, 64 utterances
x 128 phonemes per GPU, duration head pinned to 6 frames/phoneme -> 768 mel frames/utt, HiFi-GAN v1
at 22.05 kHz / hop 256 (the metric's rate; --vocoder 24k gives the JSUT recipe's 24 kHz / hop 300).
A "step" = token ids resident on the GPU -> mel -> waveform resident on the GPU (+ one RCCL
all-gather of audio when N > 1).  Weak scaling: every rank synthesises its own 64 utterances.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel family (the fused HiFi-GAN
dilation unit), timed live with HIP events on the launch stream inside the timed region;
`cpu_baseline` times the CPU oracle (a port of the reference algorithm; the reference itself
cannot travel to the GPU box) on a bounded sample, rank 0, N == 1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_F16_PEAK_TF = 2500.0  # dense f16/bf16 MFMA
RIDGE = MFMA_F16_PEAK_TF * 1e12 / (HBM_PEAK_GBS * 1e9)  # FLOP/B


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="utterances per GPU")
    ap.add_argument("--t-text", type=int, default=128)
    ap.add_argument("--frames-per-token", type=int, default=6)
    ap.add_argument("--precision", default="fp16", choices=["fp16", "fp32"])
    ap.add_argument("--vocoder", default="22k", choices=["22k", "24k"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pipeline", action="store_true",
                    help="two-stream executor (jatts_amd.pipeline): text2mel of step k+1 overlaps the vocoder of step k; "
                         "per-kernel and per-stage timings then overlap too, so the default run stays sequential")
    ap.add_argument("--cpu-t-text", type=int, default=128, help="phonemes in the CPU-baseline sample utterance")
    return ap.parse_args()


def cpu_baseline(fs2_sd, voc_sd, voc_params, texts, heads, budget_s=12.0, threads=None):
    """Reference stage-4 loop shape (tts_decode.py:203-255): one utterance at a time (B=1) on the host cores with
    the CPU oracle, repeated over the bench's utterances until ~budget_s seconds of CPU work have been timed."""
    from oracle.fs2_oracle import fs2_inference
    from oracle.hifigan_oracle import hifigan_generate

    cores = threads or min(32, os.cpu_count() or 1)  # oversubscribing a 256-thread host makes torch CPU slower
    torch.set_num_threads(cores)
    n = samples = frames = 0
    t_fs2 = t_voc = 0.0
    with torch.no_grad():
        for text in texts:
            t0 = time.time()
            r = fs2_inference(fs2_sd, text, heads)
            t1 = time.time()
            y = hifigan_generate(voc_sd, r["feat_gen"], voc_params["upsample_scales"], voc_params["resblock_dilations"])
            t2 = time.time()
            n, samples, frames = n + 1, samples + int(y.numel()), frames + int(r["feat_gen"].shape[0])
            t_fs2, t_voc = t_fs2 + (t1 - t0), t_voc + (t2 - t1)
            if t_fs2 + t_voc >= budget_s and (n >= 2 or budget_s <= 0.0):
                break
    secs = t_fs2 + t_voc
    return dict(value=samples / secs, unit="samples/s", cores=cores, kind="port",
                sample=f"{n} utterances x {texts[0].numel()} phonemes, one at a time (the reference loop is B=1) -> {frames} frames "
                       f"-> {samples} samples, torch CPU fp32 oracle, {cores} threads, text2mel {t_fs2:.2f}s + vocoder {t_voc:.2f}s",
                seconds=secs, samples=samples)


def pmc_traffic(c, esz):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (tools/pmc_bench.sh ->
    tools/pmc_traffic.py; FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate passes, same workload).  PMC
    counters cannot be collected from inside this process, so the number is the last committed measurement."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_traffic.json")
    if not os.path.exists(path):
        return None, None
    key = f"resunit_kernelI{'DF16_' if esz == 2 else 'f'}Li{c}E"
    for name, v in json.load(open(path))["kernels"].items():
        if key in name:
            return v["hbm_bytes"], "profiles/r01_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this bench)"
    return None, None


def main():
    a = parse()
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:   # one process per GPU, RCCL over xGMI (backend "nccl" on ROCm); rendezvous from the torchrun env
        import torch.distributed as dist
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from jatts_amd import hip
    from jatts_amd.models import FastSpeech2
    from jatts_amd.synthetic import (FS2_JSUT, HIFIGAN_V1_22K, HIFIGAN_V1_24K, pin_duration_head,
                                     synth_hifigan_state, synth_state_dict, synth_texts)
    from jatts_amd.vocoder import Vocoder

    vocab = 45
    m = FastSpeech2(idim=vocab, **FS2_JSUT)
    fs2_sd = pin_duration_head(synth_state_dict(m.state_dict(), 0), a.frames_per_token)
    m.load_state_dict(fs2_sd)
    m = m.to(dev).set_precision(a.precision)
    vp = HIFIGAN_V1_22K if a.vocoder == "22k" else HIFIGAN_V1_24K
    sr = 22050 if a.vocoder == "22k" else 24000
    voc_sd = synth_hifigan_state(vp, 0)
    ones, zeros = [1.0] * 80, [0.0] * 80
    voc = Vocoder(voc_sd, {"sampling_rate": sr, "generator_type": "HiFiGANGenerator", "generator_params": vp},
                  {"mean": zeros, "scale": ones}, dev, trg_stats={"mean": zeros, "scale": ones})
    voc.set_precision(a.precision)
    hop = voc.model.hop
    texts = [t.to(dev) for t in synth_texts(a.batch, a.t_text, vocab, seed=1 + rank)]

    stage_ev = []   # per step: events at start / after text2mel / after vocoder / after the audio all-gather

    def step():
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ev[0].record()
        r = m.inference_batch(texts)
        ev[1].record()
        y = voc.decode_batch(r["feats_rb"], r["feat_gen"])
        ev[2].record()
        lens = [n * hop for n in r["olens"]]
        if world > 1:
            from jatts_amd.distributed import gather_audio
            gather_audio(y, lens)
        ev[3].record()
        stage_ev.append(ev)
        return y, lens

    for _ in range(a.warmup):
        y, lens = step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    hip.profile_begin()
    stage_ev.clear()
    t0 = time.perf_counter()
    if a.pipeline:
        from jatts_amd.pipeline import Stage4Pipeline
        for r, y in Stage4Pipeline(m, voc).run([texts] * a.steps):
            lens = [n * hop for n in r["olens"]]
            if world > 1:
                from jatts_amd.distributed import gather_audio
                gather_audio(y, lens)
    else:
        for _ in range(a.steps):
            y, lens = step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    recs = hip.profile_end()
    if dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    samples_rank = sum(lens)
    total_samples = samples_rank * world * a.steps
    value = total_samples / dt
    assert torch.isfinite(y).all() and float(y.abs().max()) <= 1.0

    # ---- per-kernel live timings (rank 0): aggregate by family / shape
    esz = 2 if a.precision == "fp16" else 4
    fam = {}
    for tag, meta, ms in recs:
        fam.setdefault((tag, meta), []).append(ms)
    units = []
    for (tag, meta), v in fam.items():
        if tag != "resunit":
            continue
        C, k, d, rows = meta
        avg = sum(v) / len(v)
        flops = 4.0 * C * C * k * rows          # 2 convs x 2 FLOP/MAC
        byts = 2.0 * rows * C * esz             # read x once + write y once
        units.append(dict(C=C, k=k, dil=d, rows=rows, launches=len(v), avg_ms=avg, total_ms=sum(v),
                          tflops=flops / avg / 1e9, gbs=byts / avg / 1e6, ai=flops / byts))
    tot_unit_ms = sum(u["total_ms"] for u in units)
    by_c = {}
    for u in units:
        by_c.setdefault(u["C"], []).append(u)
    dom_c = max(by_c, key=lambda c: sum(u["total_ms"] for u in by_c[c]))
    dom = by_c[dom_c]
    dom_ms = sum(u["total_ms"] for u in dom)
    dom_flops = sum(4.0 * u["C"] ** 2 * u["k"] * u["rows"] * u["launches"] for u in dom)
    dom_bytes = sum(2.0 * u["rows"] * u["C"] * esz * u["launches"] for u in dom)
    ai = dom_flops / dom_bytes
    if ai >= RIDGE and a.precision == "fp16":
        roof = dict(bound="mfma", achieved=dom_flops / dom_ms / 1e9, peak=MFMA_F16_PEAK_TF, unit="TFLOP/s")
    else:
        roof = dict(bound="hbm", achieved=dom_bytes / dom_ms / 1e6, peak=HBM_PEAK_GBS, unit="GB/s")
    roof["frac"] = roof["achieved"] / roof["peak"]
    roof["traffic"], roof["traffic_source"] = pmc_traffic(dom_c, esz)
    ceil_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_mfma_ceiling.json")
    if roof["bound"] == "mfma" and os.path.exists(ceil_path):
        c = json.load(open(ceil_path))
        roof["measured_ceiling"] = c["register_only_random_operands"]["tflops"]
        roof["frac_of_measured_ceiling"] = roof["achieved"] / roof["measured_ceiling"]
        roof["ceiling_note"] = c["note"]
    roof["kernel"] = f"resunit_kernel<{'f16' if esz == 2 else 'float'}, C={dom_c}> (fused HiFi-GAN dilation unit)"
    roof["avg_launch_ms"] = dom_ms / sum(u["launches"] for u in dom)
    roof["arith_intensity_flop_per_byte"] = ai
    roof["share_of_step"] = dom_ms / (dt * 1e3)
    other = {}
    for (tag, meta), v in fam.items():
        if tag != "resunit":
            other[tag] = other.get(tag, 0.0) + sum(v)

    out = {
        "metric": "audio samples/sec (22.05 kHz) + RTF, FastSpeech2+HiFi-GAN" if a.vocoder == "22k"
                  else "audio samples/sec (24 kHz) + RTF, FastSpeech2+HiFi-GAN",
        "value": value, "unit": "samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16 MFMA operands, f32 accumulate" if a.precision == "fp16" else "f32",
        "data": "synthetic (random-init weights, random phoneme ids, duration head pinned)",
        "config": {"workload": f"FastSpeech2(JSUT conformer 4+4, adim 384)+HiFi-GAN v1 {a.vocoder}, "
                               f"{a.batch} utts x {a.t_text} phonemes x {a.frames_per_token} frames per GPU",
                   "utterances_per_gpu": a.batch, "phonemes": a.t_text, "frames_per_utt": a.t_text * a.frames_per_token,
                   "hop": hop, "sampling_rate": sr, "parallelism": f"dp{world} (utterance sharding, audio all-gather)"},
        "rtf": dt / (total_samples / sr),
        "stage_ms_per_step": ({nme: sum(e[i].elapsed_time(e[i + 1]) for e in stage_ev) / len(stage_ev)
                               for i, nme in enumerate(["text2mel", "vocoder", "audio_all_gather"])} if stage_ev else None),
        "executor": "two-stream pipeline (jatts_amd.pipeline)" if a.pipeline else "sequential",
        "roofline": roof,
        "resunit_ms_per_step": tot_unit_ms / a.steps,
        "other_kernel_ms_per_step": {k: v / a.steps for k, v in other.items()},
        "resunit_by_shape": sorted(units, key=lambda u: -u["total_ms"]),
        "conv1d_by_shape": sorted(
            [dict(c_in=m[0], n_out=m[1], k=m[2], rows=m[3], launches_per_step=len(v) / a.steps,
                  ms_per_step=sum(v) / a.steps, tflops=2.0 * m[0] * m[1] * m[2] * m[3] * len(v) / sum(v) / 1e9)
             for (t, m), v in fam.items() if t == "conv1d"], key=lambda u: -u["ms_per_step"])[:14],
    }
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cb = cpu_baseline(fs2_sd, voc_sd, vp, synth_texts(64, a.cpu_t_text, vocab, seed=1), 2)
        out["cpu_baseline"] = cb
        out["cpu_baseline"]["rtf"] = cb["seconds"] / (cb["samples"] / sr)
        # SURVEY 8d also asks for the recipe default OMP_NUM_THREADS=1 (path.sh:15): one utterance, one thread
        c1 = cpu_baseline(fs2_sd, voc_sd, vp, synth_texts(1, a.cpu_t_text, vocab, seed=1), 2, budget_s=0.0, threads=1)
        out["cpu_baseline"]["single_thread"] = {k: c1[k] for k in ("value", "unit", "cores", "sample", "seconds")}
        out["cpu_baseline"]["single_thread"]["rtf"] = c1["seconds"] / (c1["samples"] / sr)
        out["speedup_vs_cpu_rtf"] = out["cpu_baseline"]["rtf"] / out["rtf"]
    else:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
