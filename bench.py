#!/usr/bin/env python3
"""bench.py — stage-4 hot path throughput on MI355X: FastSpeech2 (JSUT config) + HiFi-GAN v1.

Workload (BASELINE.json configs[1]): random-init weights of the named architectures, 64 utterances
x 128 phonemes per GPU, duration head pinned to 6 frames/phoneme -> 768 mel frames/utt, HiFi-GAN v1
at 22.05 kHz / hop 256 (the metric's rate; --vocoder 24k gives the JSUT recipe's 24 kHz / hop 300).
A "step" = token ids resident on the GPU -> mel -> waveform resident on the GPU (+ one RCCL
all-gather of int16 PCM when N > 1).  Weak scaling: every rank synthesises its own 64 utterances.

Prints ONE JSON line (rank 0):
  * headline (`value`, `ms_per_step`, `dtype`, `roofline`): the f32-EQUIVALENT emulated arithmetic `fp32_bf16x3` (VERDICT r5's ruling; DESIGN.md
    section 4): f32 tensors, every conv / fused-unit operand carried exactly as three bf16 terms, seven MFMA products per product, f32 accumulate;
    attention, normalisations and the duration predictor exact f32.  `roofline.peak` = the dense bf16 spec peak / 7; `roofline.practical_peak` =
    the matrix pipe's sustained rate on THIS part measured live before the timed region (jatts_mfma_probe: v_mfma_f32_32x32x16_bf16 fed from
    LDS on random operand bits) / 7, `frac_of_practical` beside `frac`;
  * `exact_f32_mode`: the same K steps in the reference's own arithmetic -- f32 operands on v_mfma_f32_32x32x2_f32 (exact f32 fma chains) --
    with its own `roofline` (157.3 TFLOP/s f32 MFMA peak) and the max mel / waveform difference between the two runs;
  * `vocoder_24k`: the same batch through the 24 kHz / hop-300 generator the JSUT / JVS recipes load (scales 5,5,4,3), headline arithmetic
    and exact f32;
  * `b1_latency`: ONE 128-phoneme utterance through the reference's own call shape -- model.inference(x) + vocoder.decode(mel),
    tts_decode.py:230,249 -- median wall ms (hipGraph replay, jatts_amd/graphs.py), the kernel-time sum of a rocprofv3 --kernel-trace child
    of the same calls, and the launch-gap fraction between them;
  * `fast_mode`: the same K steps with f16 MFMA operands (f32 accumulate, f32 residual streams) and its error against the exact-f32 run;
    the six-product and split-f16 arithmetics are timed too but live in bench_detail.json only (ruled ineligible as headlines);
  * `configs`: BASELINE configs[2] (Matcha-TTS MAS, 10 Euler steps, 64 utterances) and configs[4]'s per-GPU share
    (mel-VITS, 192-d speaker embeddings, 32 utterances), every arithmetic, N == 1 only;
  * `cpu_baseline`: the CPU oracle (a port of the reference algorithm; the reference cannot travel to the GPU box)
    in the reference's B=1 loop on the host cores, rank 0, N == 1 only.
`roofline.traffic` comes from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate child processes started
before this process touches the GPU); when rocprofv3 is unavailable the last committed measurement is used and
`traffic_source` says so.
"""
import argparse
import csv
import json
import os
import platform
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_F16_PEAK_TF = 2500.0   # dense f16/bf16 MFMA
MFMA_F32_PEAK_TF = 157.3    # v_mfma_f32_32x32x2_f32 (= the f32 vector rate)
PROFILE_ROUND = "r06"
# matrix-pipe peak per ALGORITHMIC FLOP of each arithmetic: split = 3 dense f16 MFMAs per product, bf16x3 emulation = 7 (6) dense bf16 MFMAs
ALG_PEAK_TF = {"fp16": MFMA_F16_PEAK_TF, "fp32": MFMA_F32_PEAK_TF, "fp32_split": MFMA_F16_PEAK_TF / 3.0, "fp32_bf16x3": MFMA_F16_PEAK_TF / 7.0,
               "fp32_bf16x3_6p": MFMA_F16_PEAK_TF / 6.0}
PRECISIONS = ("fp32", "fp32_bf16x3", "fp32_bf16x3_6p", "fp32_split", "fp16")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="utterances per GPU")
    ap.add_argument("--t-text", type=int, default=128)
    ap.add_argument("--frames-per-token", type=int, default=6)
    ap.add_argument("--precision", default="fp32_bf16x3", choices=["fp16", "fp32", "fp32_split", "fp32_bf16x3", "fp32_bf16x3_6p"],
                    help="arithmetic of the headline numbers: fp32_bf16x3 = f32-equivalent emulated operands (three exact bf16 terms, seven MFMA products; "
                         "VERDICT r5 ruled it eligible), fp32 = exact f32 MFMA (always reported beside it as exact_f32_mode)")
    ap.add_argument("--vocoder", default="22k", choices=["22k", "24k"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ragged", action="store_true", help="skip the ragged-length leg (T_text ~ U{64..t_text}, SURVEY 8d's optional variant)")
    ap.add_argument("--ragged-min", type=int, default=64, help="shortest utterance of the ragged leg, phonemes")
    ap.add_argument("--ragged-seed", type=int, default=7)
    ap.add_argument("--only-ragged", action="store_true", help="run ONLY the ragged leg at --precision and print nothing (the command under rocprofv3 for "
                                                              "profiles/rNN_bench_ragged_kernel_stats.csv)")
    ap.add_argument("--no-fast-mode", action="store_true", help="skip the f16 fast-mode leg")
    ap.add_argument("--no-train", action="store_true", help="skip the FastSpeech2 train-step line")
    ap.add_argument("--no-configs", action="store_true", help="skip the Matcha-TTS / VITS config lines")
    ap.add_argument("--no-pmc", action="store_true", help="do not spawn the rocprofv3 --pmc passes for roofline.traffic")
    ap.add_argument("--pmc-all", action="store_true", help="the --pmc child runs every arithmetic (default: the headline one and exact f32)")
    ap.add_argument("--no-24k", action="store_true", help="skip the 24 kHz / hop-300 recipe-vocoder block")
    ap.add_argument("--only-24k", action="store_true", help="run ONLY the 24 kHz vocoder leg at --precision and print nothing (the command under rocprofv3 "
                                                           "for profiles/rNN_bench_24k_kernel_stats.csv)")
    ap.add_argument("--no-b1", action="store_true", help="skip the B = 1 drop-in latency block")
    ap.add_argument("--b1-child", action="store_true", help=argparse.SUPPRESS)    # the rocprofv3 --kernel-trace child of the b1_latency block
    ap.add_argument("--b1-iters", type=int, default=30, help="timed utterances of the B = 1 latency block")
    ap.add_argument("--no-ceiling", action="store_true", help="skip the live MFMA ceiling probe (roofline.practical_peak = null)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)   # the profiled child: one step, no JSON
    ap.add_argument("--no-detail", action="store_true", help="do not (over)write bench_detail.json -- partial runs under a profiler (tools/profile_*.sh)")
    ap.add_argument("--pipeline", action="store_true",
                    help="two-stream executor (jatts_amd.pipeline): text2mel of step k+1 overlaps the vocoder of step k; "
                         "per-kernel and per-stage timings then overlap too, so the default run stays sequential")
    ap.add_argument("--profile-config", default=None, choices=["matcha", "vits"],
                    help="run ONLY that BASELINE config's leg (configs[2] / configs[4]'s share) at --precision for --warmup + --steps steps and print "
                         "nothing: the command tools/profile_models.sh puts under rocprofv3 (profiles/r04_infer_*_kernel_stats.csv)")
    ap.add_argument("--cpu-t-text", type=int, default=128, help="phonemes in the CPU-baseline sample utterance")
    ap.add_argument("--cpu-budget", type=float, default=45.0, help="seconds of N-thread CPU work before the sample is cut")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------- CPU baseline
def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def cpu_baseline(fs2_sd, voc_sd, voc_params, texts, heads, budget_s, threads):
    """Reference stage-4 loop shape (tts_decode.py:203-255): one utterance at a time (B=1) on the host cores with
    the CPU oracle, over the bench's utterances until they are all done or ~budget_s seconds have been timed."""
    import torch
    from oracle.fs2_oracle import fs2_inference
    from oracle.hifigan_oracle import hifigan_generate

    torch.set_num_threads(threads)
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
            if t_fs2 + t_voc >= budget_s:
                break
    secs = t_fs2 + t_voc
    return dict(value=samples / secs, unit="samples/s", cores=threads, kind="port",
                sample_short=f"{n}/{len(texts)} utts x {texts[0].numel()} phonemes, B=1 loop, torch CPU f32 oracle",
                sample=f"{n} of the bench's {len(texts)} utterances x {texts[0].numel()} phonemes, one at a time (the reference "
                       f"loop is B=1) -> {frames} frames -> {samples} samples, torch CPU fp32 oracle, {threads} threads, "
                       f"text2mel {t_fs2:.2f}s + vocoder {t_voc:.2f}s",
                seconds=secs, samples=samples, utterances=n)


# ------------------------------------------------------------------------------------------- PMC traffic
def pmc_passes(argv_tail, timeout_s=240):     # (main() scales timeout_s with the number of arithmetics the child runs: ADVICE r5)
    """HBM bytes per launch of every kernel of one f32 + one f16 bench step: two rocprofv3 passes (FETCH_SIZE, then
    WRITE_SIZE; kernel-trace only), each a CHILD process started before this process initialises the GPU.
    Corrections per MI355X_MICROARCH.md §HBM: counters are KiB; FETCH_SIZE x2 on gfx950 (wide streaming reads are
    tallied at half their bytes); WRITE_SIZE as is.  -> {kernel name: bytes per launch} or None."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="jatts_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    per = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [exe, "--kernel-trace", "--pmc", counter, "-d", out, "-o", "p", "--output-format", "csv", "--",
                   sys.executable, os.path.join(ROOT, "bench.py"), "--pmc-child"] + argv_tail
            p = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=timeout_s)
            path = None
            for dp, _, fs in os.walk(out):
                for f in fs:
                    if f.endswith("counter_collection.csv"):
                        path = os.path.join(dp, f)
            if p.returncode != 0 or path is None:
                return None, f"rocprofv3 --pmc {counter} failed (rc {p.returncode}): {p.stderr.decode(errors='replace')[-200:]}"
            agg = {}
            for r in csv.DictReader(open(path)):
                if r["Counter_Name"] == counter:
                    agg.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
            for k, v in agg.items():
                per.setdefault(k, {})[counter] = sum(v) / len(v) * 1024.0
                per[k]["launches"] = len(v)
        res = {k: dict(fetch_bytes=2.0 * v.get("FETCH_SIZE", 0.0), write_bytes=v.get("WRITE_SIZE", 0.0), launches=v.get("launches", 1))
               for k, v in per.items()}
        for v in res.values():
            v["hbm_bytes"] = v["fetch_bytes"] + v["write_bytes"]
        return res, "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate child passes of this run; FETCH_SIZE x2 gfx950 correction)"
    except (subprocess.TimeoutExpired, OSError) as e:
        return None, f"rocprofv3 passes failed: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def committed_traffic():
    """The committed PMC table of THIS round's code (profiles/<round>_traffic.json, tools/pmc_bench.sh + tools/pmc_traffic.py) or
    (None, reason).  Never an older round's table: its kernels are not this code's."""
    path = os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_traffic.json")
    if os.path.exists(path):
        return json.load(open(path))["kernels"], f"profiles/{PROFILE_ROUND}_traffic.json (committed rocprofv3 --pmc passes; not re-measured in this run)"
    return None, f"profiles/{PROFILE_ROUND}_traffic.json not found"



# ------------------------------------------------------------------------------------------- B = 1 drop-in latency
B1_MARK = "mfma_probe_kernel"     # the marker launches that bracket the timed utterances in the child's kernel trace


def b1_trace_pass(argv_tail, timeout_s=240):
    """Kernel-time sum of ONE utterance through model.inference(x) + vocoder.decode(mel): a rocprofv3 --kernel-trace CHILD of this file
    (`--b1-child`, started before this process touches the GPU) runs the timed utterances between two marker launches; the durations of every
    kernel between the markers / utterances = the GPU time the path needs when no launch gap is left.  -> (dict, None) or (None, why not)."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="jatts_b1_", dir="/tmp")
    try:
        cmd = [exe, "--kernel-trace", "-d", tmp, "-o", "b1", "--output-format", "csv", "--", sys.executable, os.path.join(ROOT, "bench.py"), "--b1-child"] + argv_tail
        p = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s)
        path = None
        for dp, _, fs in os.walk(tmp):
            for f in fs:
                if f.endswith("kernel_trace.csv"):
                    path = os.path.join(dp, f)
        if p.returncode != 0 or path is None:
            return None, f"rocprofv3 --kernel-trace child failed (rc {p.returncode}): {p.stderr.decode(errors='replace')[-200:]}"
        child = None
        for ln in p.stdout.decode(errors="replace").splitlines():
            if ln.startswith("{") and '"b1_child"' in ln:
                child = json.loads(ln)
        rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))), key=lambda t: t[0])
        marks = [i for i, r in enumerate(rows) if B1_MARK in r[2]]
        if len(marks) < 2 or child is None:
            return None, "the child's trace holds no marker pair"
        body = rows[marks[-2] + 1:marks[-1]]
        n = child["iters"]
        busy = sum(e - b for b, e, _ in body)
        span = body[-1][1] - body[0][0] if body else 0
        return dict(kernel_ms=busy / n / 1e6, launches_per_utt=len(body) / n, gpu_span_ms_profiled=span / n / 1e6, wall_ms_profiled=child["ms"],
                    source="rocprofv3 --kernel-trace child of this run (sum of kernel durations between two marker launches / utterances)"), None
    except (subprocess.TimeoutExpired, OSError, ValueError, KeyError) as e:
        return None, f"kernel-trace child failed: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def b1_run(a, dev, prec, iters, warm=4, graph=True, markers=False, job=None):
    """The reference's stage-4 loop body (tts_decode.py:230,249) on ONE synthetic utterance: model.inference(x) -> vocoder.decode(feat_gen),
    a host synchronisation per utterance (the loop writes a wav next).  -> dict(ms median, per-utterance list, frames, samples)."""
    import torch
    from jatts_amd import graphs, hip
    job = job or Job("fs2", a, dev, 0, 1)
    job.set_precision(prec)
    text = job.texts[0]
    on, graphs.ENABLED = graphs.ENABLED, bool(graph) and graphs.ENABLED

    def once():
        r = job.m.inference(text)
        y, _ = job.voc.decode(r["feat_gen"])
        return r, y

    def mark():
        if markers:
            hip.mfma_marker(dev)
    try:
        for _ in range(warm):          # first sight eager, second captures, then replays
            once()
        torch.cuda.synchronize()
        per = []
        mark()
        for _ in range(iters):
            t0 = time.perf_counter()
            r, y = once()
            torch.cuda.synchronize()
            per.append((time.perf_counter() - t0) * 1e3)
        mark()
        torch.cuda.synchronize()
    finally:
        graphs.ENABLED = on
    assert torch.isfinite(y).all()
    return dict(ms=sorted(per)[len(per) // 2], ms_min=min(per), ms_all=[round(v, 3) for v in per], frames=int(r["feat_gen"].shape[0]), samples=int(y.numel()),
                phonemes=int(text.numel()), job=job)

# fused-unit kernel of each arithmetic as rocprofv3 names it: (mangled, demangled) prefixes up to the channel count
UNIT_KERNEL = {"fp32_split": ("resunit_split_kernelILi{c}E", "resunit_split_kernel<{c},"),
               "fp32_bf16x3": ("resunit_emul16_kernelI4bf3pILi7EELi{c}E", "resunit_emul16_kernel<bf3p<7>, {c},"),
               "fp32_bf16x3_6p": ("resunit_emul16_kernelI4bf3pILi6EELi{c}E", "resunit_emul16_kernel<bf3p<6>, {c},")}


def lookup_traffic(table, prec, c):
    """-> (HBM bytes per launch of the fused-unit family at `c` channels, None) or (None, why not): a table that lacks the kernel it is
    asked about answers null with the reason, never another kernel's number."""
    if not table:
        return None, "no traffic table"
    if prec in UNIT_KERNEL:
        key, alt = (t.format(c=c) for t in UNIT_KERNEL[prec])
    else:
        key = f"resunit_kernelI{'DF16_' if prec == 'fp16' else 'f'}Li{c}E"
        alt = f"resunit_kernel<{'_Float16' if prec == 'fp16' else 'float'}, {c},"
    hits = [v for k, v in table.items() if key in k or alt in k]
    if not hits:
        return None, f"the traffic table holds no '{alt}...' kernel"
    n = sum(h.get("launches", 1) for h in hits)             # tile variants of one family: launch-weighted mean
    return sum(h["hbm_bytes"] * h.get("launches", 1) for h in hits) / n, None


# ------------------------------------------------------------------------------------------- workloads
class Job:
    """One BASELINE config on one GPU: text2mel model + HiFi-GAN, synthetic weights and inputs."""

    def __init__(self, kind, a, dev, rank, batch):
        import torch
        from jatts_amd import models
        from jatts_amd.synthetic import FS2_JSUT, MATCHA_MAS_JSUT, VITS_JSUT, pin_duration_head, synth_state_dict, synth_texts

        self.kind, self.dev, self.batch, self.vocab = kind, dev, batch, 45
        self._vocs, self.precision = {}, "fp32"
        self.use_vocoder(a.vocoder)
        fpt = a.frames_per_token
        if kind == "fs2":
            m = models.FastSpeech2(idim=self.vocab, **FS2_JSUT)
            self.sd = pin_duration_head(synth_state_dict(m.state_dict(), 0), fpt)
            self.texts = [t.to(dev) for t in synth_texts(batch, a.t_text, self.vocab, seed=1 + rank)]
            self.name = "FastSpeech2(JSUT conformer 4+4, adim 384)"
        elif kind == "matcha":
            m = models.MatchaTTS_MAS(idim=self.vocab, **MATCHA_MAS_JSUT)
            self.sd = synth_state_dict(m.state_dict(), 0)
            self.texts = [t.to(dev) for t in synth_texts(batch, a.t_text, self.vocab, seed=1 + rank)]
            self.name = "MatchaTTS_MAS(JSUT tts2: conformer encoder, U-Net 512/512 head dim 256, 10 Euler steps, temperature 0.667)"
        else:
            m = models.VITS(idim=self.vocab, spk_embed_dim=192, **VITS_JSUT)
            self.sd = synth_state_dict(m.state_dict(), 0)
            self.texts = [t.to(dev) for t in synth_texts(batch, a.t_text, self.vocab, seed=3 + rank)]
            self.spk = torch.randn(batch, 192, generator=torch.Generator().manual_seed(3)).to(dev)
            self.name = "mel-VITS(JSUT tts2 config + 192-d speaker embedding, noise_scale 0.667)"
        m.load_state_dict(self.sd)
        self.m = m.to(dev)
        frames = a.t_text * fpt
        self.dur = [torch.full((a.t_text,), fpt, dtype=torch.int64, device=dev) for _ in self.texts]
        g = torch.Generator().manual_seed(2)
        if kind == "matcha":    # the reference draws randn_like inside CFM.inference; fixed here so f16 and f32 see the same noise
            self.noise = [torch.randn(frames, 80, generator=g).to(dev) for _ in self.texts]
        elif kind == "vits":
            self.noise = [torch.randn(frames, 384, generator=g).to(dev) for _ in self.texts]

    def use_vocoder(self, which):
        """"22k": HiFi-GAN v1 at 22.05 kHz / hop 256 (scales 8,8,2,2 -- BASELINE's metric); "24k": 24 kHz / hop 300 (scales 5,5,4,3 -- what the JSUT / JVS
        recipes load, conf/fastspeech2.v1.yaml:4-6,96-99).  Synthetic weights (seed 0), identity feature statistics; built once per job."""
        from jatts_amd.synthetic import HIFIGAN_V1_22K, HIFIGAN_V1_24K, synth_hifigan_state
        from jatts_amd.vocoder import Vocoder
        if which not in self._vocs:
            vp = HIFIGAN_V1_22K if which == "22k" else HIFIGAN_V1_24K
            sr = 22050 if which == "22k" else 24000
            sd = synth_hifigan_state(vp, 0)
            ones, zeros = [1.0] * 80, [0.0] * 80
            voc = Vocoder(sd, {"sampling_rate": sr, "generator_type": "HiFiGANGenerator", "generator_params": vp},
                          {"mean": zeros, "scale": ones}, self.dev, trg_stats={"mean": zeros, "scale": ones})
            self._vocs[which] = (vp, sr, sd, voc)
        self.vp, self.sr, self.voc_sd, self.voc = self._vocs[which]
        self.voc.set_precision(self.precision)
        self.hop = self.voc.model.hop

    def set_precision(self, p):
        self.precision = p
        self.m.set_precision(p)      # fp32_split: every conv of the acoustic model but the duration predictor's, and the whole vocoder generator
        self.voc.set_precision(p)

    def text2mel(self):
        if self.kind == "fs2":
            return self.m.inference_batch(self.texts)
        if self.kind == "matcha":
            return self.m.inference_batch(self.texts, n_timesteps=10, temperature=0.667, durations=self.dur, noise=self.noise)
        return self.m.inference_batch(self.texts, self.spk, noise_scale=0.667, durations=self.dur, noise=self.noise)


def run_timed(job, a, world, dist, pipeline=False, record=True, job_samples=None):
    """W untimed + K timed steps, barrier + synchronize on both sides, max over ranks.  -> dict.  job_samples: samples ALL ranks produce per step when
    the ranks' shares differ (the sharded ragged leg); default: this rank's x world (weak scaling, equal shares)."""
    import torch
    from jatts_amd import hip

    stage_ev = []

    def step():
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ev[0].record()
        r = job.text2mel()
        ev[1].record()
        y = job.voc.decode_batch(r["feats_rb"], r["feat_gen"])
        ev[2].record()
        lens = [n * job.hop for n in r["olens"]]
        if world > 1:
            from jatts_amd.distributed import gather_audio
            gather_audio(y, lens, max_utts=job.batch)
        ev[3].record()
        stage_ev.append(ev)
        return r, y, lens

    for _ in range(a.warmup):
        r, y, lens = step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    if record:   # per-kernel HIP events (the roofline block); off for the many-small-launch configs, where they cost ~10 us each
        hip.profile_begin()
    stage_ev.clear()
    t0 = time.perf_counter()
    if pipeline and job.kind == "fs2":
        from jatts_amd.pipeline import Stage4Pipeline
        for r, y in Stage4Pipeline(job.m, job.voc).run([job.texts] * a.steps):
            lens = [n * job.hop for n in r["olens"]]
            if world > 1:
                from jatts_amd.distributed import gather_audio
                gather_audio(y, lens, max_utts=job.batch)
    else:
        for _ in range(a.steps):
            r, y, lens = step()
    torch.cuda.synchronize()
    busy = time.perf_counter() - t0
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    recs = hip.profile_end() if record else []
    rank_ms = None
    if dist:   # value uses the MAX over ranks; min / max of the ranks' own busy time (before the closing barrier) shows the imbalance
        t = torch.tensor([dt, busy, -busy], dtype=torch.float64, device=job.dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])
        rank_ms = {"min": -float(t[2]) / a.steps * 1e3, "max": float(t[1]) / a.steps * 1e3}
    total_samples = (job_samples if job_samples is not None else sum(lens) * world) * a.steps
    assert torch.isfinite(y).all() and float(y.abs().max()) <= 1.0
    stages = ({nme: sum(e[i].elapsed_time(e[i + 1]) for e in stage_ev) / len(stage_ev)
               for i, nme in enumerate(["text2mel", "vocoder", "audio_all_gather"])} if stage_ev else None)
    return dict(dt=dt, value=total_samples / dt, ms_per_step=dt / a.steps * 1e3, rtf=dt / (total_samples / job.sr),
                stages=stages, recs=recs, rank_ms=rank_ms, mel=r["feat_gen"], wave=y, frames=sum(r["olens"]), samples_per_step=total_samples // a.steps)


def ragged_leg(job, a, world, dist, rank, uniform_value, record=True):
    """SURVEY 8d's optional ragged variant (the reference's stage-4 loop is ragged by nature, tts_decode.py:203-255): the same number of
    utterances with T_text ~ U{ragged_min..t_text}, same frames per phoneme, in the job's current arithmetic.  per_sample_efficiency =
    ragged samples/s / uniform samples/s: 1.0 when a sample costs the same in a ragged batch (tile tails, early-exit workgroups of a grid
    sized for the longest utterance and, at N > 1, the ranks' unequal shares all push it below 1)."""
    from jatts_amd.hostlogic import shard_load, shard_utterances
    from jatts_amd.synthetic import synth_texts
    uniform = job.texts
    job_samples = imbalance = None
    if world > 1:
        # BASELINE configs[3]'s shape: ONE ragged batch of world x batch utterances (the same draw on every rank), dealt to the ranks by
        # shard_utterances -- what tts_decode --n_gpus N does with a csv -- instead of every rank drawing its own lengths (whose sums differ by
        # ~3 % between ranks at 64 x U{64..128}: that would measure the draw, not the path)
        every = synth_texts(job.batch * world, a.t_text, job.vocab, seed=a.ragged_seed, ragged_min=a.ragged_min)
        tl = [int(t.numel()) for t in every]
        parts = shard_utterances(tl, world)
        mx, mean = shard_load(tl, parts)
        imbalance = mx / mean
        job.texts = [every[i].to(job.dev) for i in parts[rank]]
        job_samples = sum(tl) * a.frames_per_token * job.hop
    else:
        job.texts = [t.to(job.dev) for t in synth_texts(job.batch, a.t_text, job.vocab, seed=a.ragged_seed + rank, ragged_min=a.ragged_min)]
    lens = [int(t.numel()) for t in job.texts]
    try:
        r = run_timed(job, a, world, dist, record=record, job_samples=job_samples)
    finally:
        job.texts = uniform
    blk = {"t_text": f"U{{{a.ragged_min}..{a.t_text}}}", "seed": a.ragged_seed, "utterances_per_gpu": job.batch,
           "phonemes_rank0": sum(lens), "min_max_phonemes_rank0": [min(lens), max(lens)],
           "value": r["value"], "unit": "samples/s", "ms_per_step": r["ms_per_step"], "samples_per_step": r["samples_per_step"],
           "stage_ms_per_step": r["stages"], "rank_ms_per_step": r["rank_ms"],
           "per_sample_efficiency": r["value"] / uniform_value}
    if imbalance is not None:
        blk["sharding"] = f"one batch of {job.batch * world} utterances dealt by jatts_amd.hostlogic.shard_utterances (longest first to the least-loaded rank)"
        blk["predicted_slowest_over_mean_load"] = imbalance
    if r["recs"]:
        units = [ms for tag, _, ms in r["recs"] if tag in ("resunit", "resblock")]
        blk["resunit_ms_per_step"] = sum(units) / a.steps
    return blk


# the MFMA each arithmetic issues and how many of them one algorithmic product costs: (jatts_mfma_probe dtype name, MFMAs per product)
PROBE_OF = {"fp32": ("f32", 1), "fp16": ("f16", 1), "fp32_split": ("f16", 3), "fp32_bf16x3": ("bf16", 7), "fp32_bf16x3_6p": ("bf16", 6)}


def measure_ceilings(dev, precisions):
    """Live matrix-pipe ceilings of this part (jatts_mfma_probe, ~60 ms per launch, before any timed region): for every MFMA type the given
    arithmetics issue, the sustained TFLOP/s with both operands re-read from LDS every K-step on N(0, 1) operand bits (`lds`: what a kernel that
    has to feed its operands can reach at the clock the power budget allows) and with the operands held in registers (`registers`: the issue
    rate alone).  -> {"bf16": {"lds": {...}, "registers": {...}}, ...}"""
    from jatts_amd import hip
    code = {"bf16": hip.F32E, "f16": hip.F16, "f32": hip.F32}
    out = {}
    for name in sorted({PROBE_OF[p][0] for p in precisions}):
        out[name] = {"lds": hip.mfma_ceiling(code[name], 1, device=dev), "registers": hip.mfma_ceiling(code[name], 0, device=dev), "form": "32x32x16" if name != "f32" else "32x32x2"}
        if name == "f16":
            # (round 6, late) the same question for the f16 pipe; the f16 / split kernels still issue 32 x 32 x 16, their ceiling is the better form's all the same
            alt = {"lds": hip.mfma_ceiling(16 + hip.F16, 1, device=dev), "registers": hip.mfma_ceiling(16 + hip.F16, 0, device=dev)}
            out[name]["form_32x32x16"] = {k: out[name][k] for k in ("lds", "registers")}
            out[name]["form_16x16x32"] = alt
            if alt["lds"]["tflops"] > out[name]["lds"]["tflops"]:
                out[name].update(lds=alt["lds"], registers=alt["registers"], form="16x16x32")
        if name == "bf16":
            # the bf16 pipe has two instruction forms and the power-limited part sustains MORE of v_mfma_f32_16x16x32_bf16 (half the accumulator traffic per
            # flop; the emulated units run on it since round 6): the ceiling a kernel is priced against is the better form's
            alt = {"lds": hip.mfma_ceiling(16 + hip.F32E, 1, device=dev), "registers": hip.mfma_ceiling(16 + hip.F32E, 0, device=dev)}
            out[name]["form_32x32x16"] = {k: out[name][k] for k in ("lds", "registers")}
            out[name]["form_16x16x32"] = alt
            if alt["lds"]["tflops"] > out[name]["lds"]["tflops"]:
                out[name].update(lds=alt["lds"], registers=alt["registers"], form="16x16x32")
    return out


def practical_peak(ceilings, prec):
    """TFLOP/s of ALGORITHMIC work the matrix pipe can sustain in arithmetic `prec` on this part: the LDS-fed probe of its MFMA type / MFMAs per product."""
    if not ceilings:
        return None
    name, div = PROBE_OF[prec]
    c = ceilings.get(name)
    return c["lds"]["tflops"] / div if c else None


def kernel_report(recs, steps, prec, dt, traffic_table, traffic_source, ceilings=None):
    """Per-kernel live timings (HIP events on the launch stream inside the timed region) -> roofline of the dominant
    fused-unit family + per-shape tables.  Algorithmic work per dilation unit: 4 C^2 k FLOP and
    2 C sizeof bytes per row (x in, y out); the last unit of a stage also reads the other ResBlocks' outputs for the
    MRF mean its epilogue carries (n_add more reads of C sizeof bytes per row)."""
    esz = 2 if prec == "fp16" else 4                      # bytes per activation element in HBM
    # matrix-pipe peak per ALGORITHMIC FLOP: the split mode spends three dense f16 MFMAs per product
    # ... the emulated modes seven / six dense bf16 MFMAs
    unit_peak = conv_peak = ALG_PEAK_TF[prec]
    fam = {}
    for tag, meta, ms in recs:
        fam.setdefault((tag, meta), []).append(ms)
    units = []
    for (tag, meta), v in fam.items():
        if tag not in ("resunit", "resblock"):
            continue
        C, k, d, rows, n_add = meta
        nu = len(d) if tag == "resblock" else 1     # a fused ResBlock launch = nu dilation units; x in + y out ONCE
        avg = sum(v) / len(v)
        flops, byts = 4.0 * C * C * k * rows * nu, (2.0 + n_add) * rows * C * esz
        peak_tf = unit_peak
        ridge = peak_tf * 1e12 / (HBM_PEAK_GBS * 1e9)
        u = dict(C=C, k=k, dil=d, rows=rows, launches=len(v), avg_ms=avg, total_ms=sum(v),
                 tflops=flops / avg / 1e9, gbs=byts / avg / 1e6, ai=flops / byts, units_per_launch=nu, mrf_addends=n_add)
        if nu > 1:   # what the same arithmetic costs as nu separate unit launches (SURVEY 8d's per-unit bytes)
            u["unit_equivalent_gbs"] = nu * byts / avg / 1e6
        u["bound"] = "mfma" if u["ai"] >= ridge else "hbm"
        u["frac"] = u["tflops"] / peak_tf if u["bound"] == "mfma" else u["gbs"] / HBM_PEAK_GBS
        units.append(u)
    by_c = {}
    for u in units:
        by_c.setdefault(u["C"], []).append(u)
    dom_c = max(by_c, key=lambda c: sum(u["total_ms"] for u in by_c[c]))
    dom = by_c[dom_c]
    dom_ms = sum(u["total_ms"] for u in dom)
    n_launch = sum(u["launches"] for u in dom)
    dom_flops = sum(4.0 * u["C"] ** 2 * u["k"] * u["rows"] * u["launches"] * u["units_per_launch"] for u in dom)
    dom_bytes = sum((2.0 + u["mrf_addends"]) * u["rows"] * u["C"] * esz * u["launches"] for u in dom)
    ai = dom_flops / dom_bytes
    peak_tf = unit_peak
    if ai >= peak_tf * 1e12 / (HBM_PEAK_GBS * 1e9):
        roof = dict(bound="mfma", achieved=dom_flops / dom_ms / 1e9, peak=peak_tf, unit="TFLOP/s")
    else:
        roof = dict(bound="hbm", achieved=dom_bytes / dom_ms / 1e6, peak=HBM_PEAK_GBS, unit="GB/s")
    roof["frac"] = roof["achieved"] / roof["peak"]
    pp = practical_peak(ceilings, prec) if roof["bound"] == "mfma" else None
    roof["practical_peak"] = pp
    roof["frac_of_practical"] = roof["achieved"] / pp if pp else None
    if pp:
        c = ceilings[PROBE_OF[prec][0]]
        roof["practical_peak_source"] = (f"jatts_mfma_probe, live: v_mfma {PROBE_OF[prec][0]} {c.get('form', '32x32x16')} fragments fed from LDS on N(0,1) operand bits, "
                                         f"{c['lds']['tflops']:.0f} TFLOP/s at {c['lds']['clock_ghz'] or 0:.2f} GHz (registers only: {c['registers']['tflops']:.0f}) / {PROBE_OF[prec][1]} MFMAs per product")
    roof["traffic"], why = lookup_traffic(traffic_table, prec, dom_c)
    roof["traffic_source"] = traffic_source if roof["traffic"] is not None else None
    if why:
        roof["traffic_note"] = why if traffic_table else (traffic_source or why)
    roof["algorithmic_bytes_per_launch"] = dom_bytes / n_launch
    roof["algorithmic_flops_per_launch"] = dom_flops / n_launch
    roof["kernel"] = (f"resunit_split_kernel<C={dom_c}> (fused HiFi-GAN dilation unit, f32 I/O, split f16 hi/lo MFMA operands: 3 MFMAs per product, "
                      f"peak = dense f16 / 3; 9 launches per step)" if prec == "fp32_split" else
                      f"resunit_emul16_kernel<C={dom_c}> (fused HiFi-GAN dilation unit, f32 I/O, three exact bf16 terms per operand: {7 if prec == 'fp32_bf16x3' else 6} "
                      f"v_mfma_f32_16x16x32_bf16 per product, peak = dense bf16 / {7 if prec == 'fp32_bf16x3' else 6}; 9 launches per step)" if prec.startswith("fp32_bf16x3") else
                      f"resunit_kernel<{'f16' if esz == 2 else 'float'}, C={dom_c}> (fused HiFi-GAN dilation unit, 9 launches per step)")
    roof["avg_launch_ms"] = dom_ms / n_launch
    roof["arith_intensity_flop_per_byte"] = ai
    roof["share_of_step"] = dom_ms / (dt * 1e3)
    other = {}
    for (tag, meta), v in fam.items():
        if tag not in ("resunit", "resblock"):
            other[tag] = other.get(tag, 0.0) + sum(v)
    conv = [(m, v) for (t, m), v in fam.items() if t == "conv1d"]
    conv_flops = sum(2.0 * m[0] * m[1] * m[2] * m[3] * len(v) for m, v in conv)
    conv_ms = sum(sum(v) for _, v in conv)
    roof_conv = None
    if conv_ms > 0:      # second family: every jatts_conv1d launch of the step (acoustic model + HiFi-GAN input / upsampling convs)
        roof_conv = dict(bound="mfma", achieved=conv_flops / conv_ms / 1e9, peak=conv_peak, unit="TFLOP/s", frac=conv_flops / conv_ms / 1e9 / conv_peak,
                         ms_per_step=conv_ms / steps, launches_per_step=sum(len(v) for _, v in conv) / steps)
        if practical_peak(ceilings, prec):
            roof_conv["frac_of_practical"] = roof_conv["achieved"] / practical_peak(ceilings, prec)
    return dict(
        roofline=roof,
        roofline_conv1d=roof_conv,
        resunit_ms_per_step=sum(u["total_ms"] for u in units) / steps,
        other_kernel_ms_per_step={k: v / steps for k, v in other.items()},
        resunit_by_shape=sorted(units, key=lambda u: -u["total_ms"]),
        conv1d_by_shape=sorted(
            [dict(c_in=m[0], n_out=m[1], k=m[2], rows=m[3], launches_per_step=len(v) / steps,
                  ms_per_step=sum(v) / steps, tflops=2.0 * m[0] * m[1] * m[2] * m[3] * len(v) / sum(v) / 1e9)
             for (t, m), v in fam.items() if t == "conv1d"], key=lambda u: -u["ms_per_step"])[:14],
    )


def family_report(recs, prec):
    """Per-kernel-family roofline of one recorded step of ANY config (BASELINE configs 3 / 5: VERDICT r3 item 4): algorithmic FLOPs from the
    launch metadata the C-ABI wrappers record (fused unit 4 C^2 k rows, conv 2 c_in n_out k rows, attention 4 d_k H sum T^2) over the
    family's HIP-event time.  -> {"dominant": {...}, "families": [...]}"""
    unit_peak = mm_peak = ALG_PEAK_TF[prec]
    fam = {}
    for tag, meta, ms in recs:
        if tag in ("resunit", "resblock"):
            C, k, d, rows, _ = meta
            fl, name, peak = 4.0 * C * C * k * rows * (len(d) if tag == "resblock" else 1), "resunit (fused HiFi-GAN dilation unit)", unit_peak
        elif tag == "conv1d":
            fl, name, peak = 2.0 * meta[0] * meta[1] * meta[2] * meta[3], "conv1d (every jatts_conv1d launch: acoustic model + vocoder input / upsampling convs)", mm_peak
        elif tag == "relattn":     # (exact f32 in the emulated mode too)
            fl, name, peak = 4.0 * meta[0] * meta[1] * meta[3], "relattn (fused attention)", MFMA_F16_PEAK_TF if prec == "fp16" else MFMA_F32_PEAK_TF
        else:
            continue
        f = fam.setdefault(name, dict(kernel=name, ms=0.0, flops=0.0, launches=0, peak=peak))
        f["ms"] += ms
        f["flops"] += fl
        f["launches"] += 1
    rows = []
    for f in fam.values():
        rows.append(dict(kernel=f["kernel"], bound="mfma", ms_per_step=f["ms"], launches_per_step=f["launches"], alg_tflop_per_step=f["flops"] / 1e12,
                         achieved=f["flops"] / f["ms"] / 1e9, peak=f["peak"], unit="TFLOP/s", frac=f["flops"] / f["ms"] / 1e9 / f["peak"]))
    rows.sort(key=lambda r: -r["ms_per_step"])
    return {"dominant": rows[0] if rows else None, "families": rows}


def train_setup(dev, kind="fs2", batch=32, t_text=128, frames=6):
    """The recipes' model (synthetic weights), one synthetic batch and the trainer class for a training leg
    -> (model, batch dict, trainer class, trainer kwargs, name, model TFLOP per utterance or None)."""
    import torch
    from jatts_amd.models import VITS, FastSpeech2, MatchaTTS, MatchaTTS_MAS
    from jatts_amd.synthetic import FS2_JSUT, MATCHA_MAS_JSUT, VITS_JSUT, matcha_golden_tweaks, synth_state_dict
    from jatts_amd.training import FastSpeech2Trainer, MatchaTTSTrainer, VITSTrainer
    extra = {}
    if kind == "fs2":
        m = FastSpeech2(idim=45, **{**FS2_JSUT, "stop_gradient_from_pitch_predictor": True, "use_masking": True})
        m.load_state_dict(synth_state_dict(m.state_dict(), 0))
        name, flop_per_utt = "FastSpeech2 (fastspeech2.v1.yaml)", 3 * 67.4e-3
    elif kind == "matcha":
        m = MatchaTTS(idim=45, **MATCHA_MAS_JSUT)      # the tts1 yaml's model_params equal the MAS recipe's
        m.load_state_dict(matcha_golden_tweaks(synth_state_dict(m.state_dict(), 0)))
        name, flop_per_utt = "MatchaTTS tts1 (matcha_tts.v1.prior.steplr.large.yaml)", None
    elif kind == "matcha_mas":
        m = MatchaTTS_MAS(idim=45, **MATCHA_MAS_JSUT)
        m.load_state_dict(matcha_golden_tweaks(synth_state_dict(m.state_dict(), 0)))
        name, flop_per_utt = "MatchaTTS_MAS tts2 (matcha_tts.mas.v1.yaml; forward-sum phase)", None
        extra = dict(dp_train_start_steps=10000, bin_loss_start_steps=15000)
    else:
        m = VITS(idim=45, spk_embed_dim=192, **VITS_JSUT)
        m.load_state_dict(synth_state_dict(m.state_dict(), 0))
        name, flop_per_utt = "mel-VITS (vits.v1.bs32.yaml + 192-d spkemb; forward-sum phase)", None
        extra = dict(dp_train_start_steps=10000, bin_loss_start_steps=15000)
    m = m.to(dev)
    g = torch.Generator().manual_seed(5)
    il = torch.full((batch,), t_text, dtype=torch.long)
    ds = torch.full((batch, t_text), frames, dtype=torch.long)
    ol = ds.sum(1)
    b = dict(xs=torch.randint(1, 45, (batch, t_text), generator=g).to(dev), ilens=il, ys=torch.randn(batch, int(ol.max()), 80, generator=g).to(dev),
             olens=ol, durations=ds.to(dev), duration_lens=il, pitch=torch.randn(batch, t_text, 1, generator=g).to(dev), pitch_lens=il,
             energys=torch.randn(batch, t_text, 1, generator=g).to(dev), energy_lens=il,
             spkembs=torch.randn(batch, 192, generator=g).to(dev) if kind == "vits" else None)
    cls = {"fs2": FastSpeech2Trainer, "matcha": MatchaTTSTrainer, "matcha_mas": MatchaTTSTrainer, "vits": VITSTrainer}[kind]
    return m, b, cls, extra, name, flop_per_utt


def train_step_line(dev, steps, kind="fs2", batch=32, t_text=128, frames=6, precision="fp32"):
    """One `_train_step` of the reference trainers on the recipes' models (jatts/trainers/fastspeech2.py:24-100 on
    conf/fastspeech2.v1.yaml; jatts/trainers/matchatts.py:23-120 on the tts1 conf/matcha_tts.v1.prior.steplr.large.yaml): forward in
    train mode, the losses, backward, clip + Adam -- f32, synthetic weights / targets, batch 32."""
    import torch
    m, b, cls, extra, name, flop_per_utt = train_setup(dev, kind, batch, t_text, frames)
    ol = b["olens"]
    # the whole step replayed as ONE captured hipGraph per batch signature (lengths + loss-schedule phase); a signature's first call runs
    # eagerly, its second captures, the timed ones replay (tts1 Matcha's signature changes once after step 1, when the duration loss joins)
    graph = True
    tr = cls(m, lr=1e-4, grad_norm=1.0, warmup_steps=0, capture_graph=graph, precision=precision, **extra)
    from jatts_amd import hip
    hip.flops_begin()            # dense work of ONE step as launched: 2 c_in n_out k rows per conv forward / dgrad / wgrad launch
    first = float(tr.train_step(b)["loss"])
    torch.cuda.synchronize()
    dense_tflop = hip.flops_end() / 1e12
    if graph:
        for _ in range(3):       # (eager first sight of the steady-state signature,) capture + first replay, one more replay
            tr.train_step(b)
        torch.cuda.synchronize()
        assert any(st.get("graph") is not None for st in tr._graphs.values()), "graph mode did not capture"
    per = []
    for _ in range(steps):       # every step timed on its own (a step ends in the optimiser kernels: nothing to overlap with the next one);
        t0 = time.perf_counter()  # the MEDIAN is reported: these steps launch 2 300-6 200 kernels each and a busy host shows up as outliers
        o = tr.train_step(b)
        torch.cuda.synchronize()
        per.append(time.perf_counter() - t0)
    dt = sorted(per)[len(per) // 2]
    n_frames = int(ol.sum())
    mean = sum(per) / len(per)
    line = {"kind": kind, "workload": f"{name} _train_step, batch {batch} x {t_text} phonemes x {frames} frames",
            "dtype": "f32" if precision == "fp32" else "f32 tensors; forward / data-gradient convs on split f16 hi/lo MFMA operands",
            "executor": "hipGraph replay (one graph per batch signature)" if graph and tr.capture_graph else "eager (one Python launch per kernel)",
            "steps": steps, "ms_per_step": dt * 1e3, "ms_per_step_mean": mean * 1e3, "ms_per_step_all": [round(v * 1e3, 2) for v in per],
            "frames_per_s": n_frames / dt, "loss_first": first, "loss_last": float(o["loss"]),
            "dense_tflops_per_step": dense_tflop, "achieved_tflops": dense_tflop / dt,
            "frac_of_f32_mfma_peak": dense_tflop / dt / MFMA_F32_PEAK_TF,
            "dense_work": "sum of 2 c_in n_out k rows over the conv forward / dgrad / wgrad launches of one step (attention products excluded)",
            "parity": "tests/test_training_gpu.py (one whole step vs the real reference: every parameter gradient)"}
    if flop_per_utt:
        line["model_tflops_per_step"] = flop_per_utt * batch
    return line


# ------------------------------------------------------------------------------------------- the report line
LINE_LIMIT = 3000           # bytes; the driver keeps a short tail of stdout and json.loads the last line of it


def _r(x, sig=5):
    """Round to `sig` significant digits (the detail file keeps full precision)."""
    if isinstance(x, float):
        return float(f"{x:.{sig}g}")
    if isinstance(x, dict):
        return {k: _r(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, sig) for v in x]
    return x


def _roof(rf):
    if not rf:
        return None
    keep = ("bound", "achieved", "peak", "unit", "frac", "practical_peak", "frac_of_practical", "traffic", "kernel", "avg_launch_ms")
    o = {k: rf.get(k) for k in keep}
    o["kernel"] = (o.get("kernel") or "").split(" (")[0]          # (the long description stays in the detail file)
    src = rf.get("traffic_source") or ""
    o["traffic_source"] = None if rf.get("traffic") is None else ("committed" if src.startswith("profiles/") or "committed" in src else "live")
    o["alg_bytes"] = rf.get("algorithmic_bytes_per_launch")
    o["alg_flops"] = rf.get("algorithmic_flops_per_launch")
    return o


def compact_line(out, detail_path=None):
    """The ONE line the driver parses: the graded keys only, numbers rounded to 5 significant digits, < LINE_LIMIT
    bytes.  Everything else (`resunit_by_shape`, `conv1d_by_shape`, per-step lists, long sample descriptions) lives
    in the detail file written next to it."""
    keys = ("metric", "value", "unit", "n_gpus", "n_ranks_seen", "backend", "steps", "warmup", "ms_per_step", "higher_is_better",
            "scaling", "vs_baseline", "dtype", "data", "config", "rtf", "executor")
    c = {k: out[k] for k in keys if k in out}
    if c.get("executor") == "sequential":
        del c["executor"]
    if out.get("stage_ms_per_step"):
        c["stage_ms"] = out["stage_ms_per_step"]
    if out.get("rank_ms_per_step"):
        c["rank_ms"] = out["rank_ms_per_step"]
    c["roofline"] = _roof(out.get("roofline"))
    if out.get("ragged"):
        rg = out["ragged"]
        c["ragged"] = {k: rg.get(k) for k in ("t_text", "seed", "value", "ms_per_step", "per_sample_efficiency")}
        if rg.get("rank_ms_per_step"):
            c["ragged"]["rank_ms"] = rg["rank_ms_per_step"]
        if rg.get("predicted_slowest_over_mean_load"):
            c["ragged"]["predicted_imbalance"] = rg["predicted_slowest_over_mean_load"]
    if out.get("roofline_conv1d"):
        c["roofline_conv1d"] = {k: out["roofline_conv1d"].get(k) for k in ("achieved", "peak", "frac", "frac_of_practical", "ms_per_step")}
    cb = out.get("cpu_baseline")
    if cb:
        c["cpu_baseline"] = {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "rtf", "cpu_model")}
        c["cpu_baseline"]["sample"] = cb.get("sample_short") or cb.get("sample")
        if cb.get("single_thread"):
            c["cpu_baseline"]["single_thread_rtf"] = cb["single_thread"]["rtf"]
        c["speedup_vs_cpu_rtf"] = out.get("speedup_vs_cpu_rtf")
    else:
        c["cpu_baseline"] = None
        if out.get("cpu_baseline_note"):
            c["cpu_baseline_note"] = out["cpu_baseline_note"]
    # (what each arithmetic is: DTYPE_NAME in the detail file; the six-product and split-f16 blocks are in the detail file only:
    # both are ruled ineligible as headlines)
    ex = out.get("f32_mode")
    if ex:       # the reference's own arithmetic beside the headline, with its own roofline and the distance between the two runs' outputs
        rf = ex.get("roofline") or {}
        c["exact_f32_mode"] = {"dtype": "f32", "value": ex["value"], "ms_per_step": ex["ms_per_step"],
                               "vocoder_ms": (ex.get("stage_ms_per_step") or {}).get("vocoder"),
                               "max_abs_err_mel": ex.get("max_abs_err_mel"), "max_abs_err_wave": ex.get("max_abs_err_wave"),
                               "roofline": {k: rf.get(k) for k in ("achieved", "peak", "frac", "practical_peak", "frac_of_practical", "avg_launch_ms", "traffic")},
                               "speedup_vs_cpu_rtf": ex.get("speedup_vs_cpu_rtf")}
    v24 = out.get("vocoder_24k")
    if v24:
        def leg(b):
            rf = b.get("roofline") or {}
            return {"value": b["value"], "ms_per_step": b["ms_per_step"], "vocoder_ms": (b.get("stage_ms_per_step") or {}).get("vocoder"),
                    "roofline_frac": rf.get("frac"), "frac_of_practical": rf.get("frac_of_practical")}
        c["vocoder_24k"] = {"sampling_rate": v24["sampling_rate"], "hop": v24["hop"], **leg(v24["headline"])}
        if v24.get("exact_f32"):
            c["vocoder_24k"]["exact_f32"] = leg(v24["exact_f32"])
    b1 = out.get("b1_latency")
    if b1:
        c["b1_latency"] = {k: b1.get(k) for k in ("ms", "kernel_ms", "wall_over_kernel", "launch_gap_frac", "eager_ms", "launches", "utterance")}
    fm = out.get("fast_mode")
    if fm:
        c["fast_mode"] = {"dtype": "f16", "value": fm["value"], "ms_per_step": fm["ms_per_step"],
                          "max_abs_err_wave": fm.get("max_abs_err_wave"),
                          "roofline_frac": (fm.get("roofline") or {}).get("frac")}
    if out.get("configs"):
        c["configs"] = {("matcha_mas_b64" if "Matcha" in e["config"] else "vits_spk192_b32"):
                        {"ms": (e.get("f32_emul_mode") or {}).get("ms_per_step"),
                         "text2mel_ms": ((e.get("f32_emul_mode") or {}).get("stage_ms_per_step") or {}).get("text2mel"),
                         "f32_ms": e["ms_per_step"], "f32_text2mel_ms": (e.get("stage_ms_per_step") or {}).get("text2mel"),
                         "f16_ms": (e.get("fast_mode") or {}).get("ms_per_step")} for e in out["configs"]}
    if out.get("training"):
        c["training"] = {e["kind"]: {"ms": e["ms_per_step"], "frac": e.get("frac_of_f32_mfma_peak")} for e in out["training"]}
    if detail_path:
        c["detail"] = detail_path
    c = _r(c)
    line = json.dumps(c, separators=(",", ":"))
    if len(line) > LINE_LIMIT:      # never let the line outgrow the driver's tail again: drop the optional blocks
        for k in ("executor", "training", "roofline_conv1d", "fast_mode", "configs", "stage_ms", "ragged", "b1_latency", "vocoder_24k", "exact_f32_mode"):
            c.pop(k, None)
            line = json.dumps(c, separators=(",", ":"))
            if len(line) <= LINE_LIMIT:
                break
    if len(line) > LINE_LIMIT:      # ... then clip the free-text fields
        c["config"]["workload"] = c["config"]["workload"][:160]
        if c.get("cpu_baseline"):
            c["cpu_baseline"]["sample"] = str(c["cpu_baseline"]["sample"])[:160]
        line = json.dumps(c, separators=(",", ":"))
    return line


def write_detail(out):
    """Full-precision, everything-included report: bench_detail.json beside bench.py (and under gpurun_out/ so a gpurun
    call brings it home).  -> the path written relative to the repo root, or None."""
    rel = None
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        try:
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, "bench_detail.json"), "w") as f:
                json.dump(out, f, indent=1)
            rel = rel or "bench_detail.json"
        except OSError:
            pass
    return rel


MODE_KEY = {"fp16": "fast_mode", "fp32": "f32_mode", "fp32_split": "f32_split_mode", "fp32_bf16x3": "f32_emul_mode", "fp32_bf16x3_6p": "f32_emul6_mode"}
DTYPE_NAME = {"fp32": "f32", "fp16": "f16 MFMA operands, f32 accumulate",
              "fp32_bf16x3": "f32 (emulated: 3 exact bf16 terms per operand, 7 MFMA products, f32 accumulate; attention / norms / duration predictor exact f32)",
              "fp32_bf16x3_6p": "as fp32_bf16x3 with six partial products per product (dropped terms <= 2^-23 of a product)",
              "fp32_split": "f32 tensors; every conv / fused HiFi-GAN unit on split f16 hi/lo MFMA operands (3 MFMAs per product, power-of-two scales), "
                            "f32 accumulate; attention, normalisations and the duration predictor exact f32"}


def self_launch(n, argv):
    """`python bench.py --gpus N` without a launcher: run the driver's own launch line
    (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...`) as a
    CHILD process -- never an exec, and before this process has touched the GPU -- relay everything the ranks print to
    stderr, print rank 0's JSON line as the LAST stdout line and return the child's exit code."""
    from jatts_amd.distributed import self_launch as launch
    found = []

    def relay(ln):
        t = ln.strip()
        if t.startswith("{") and '"metric"' in t:
            try:
                json.loads(t)
            except ValueError:
                return False
            found[:] = [t]
            return True
        return False
    rc = launch(n, os.path.abspath(__file__), argv, relay=relay)
    if found:
        print(found[0], flush=True)
    elif rc == 0:
        sys.stderr.write("bench.py: the ranks exited 0 without a JSON line\n")
        rc = 1
    return rc


def main():
    a = parse()
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (child processes; nothing here has touched the GPU yet)
        raise SystemExit(self_launch(a.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != a.gpus:
        raise SystemExit(f"bench.py --gpus {a.gpus} launched with WORLD_SIZE={world}: --nproc-per-node must equal --gpus")

    # roofline.traffic and b1_latency.kernel_ms: rocprofv3 passes run as children BEFORE this process touches the GPU
    traffic_table = traffic_source = None
    b1_trace = b1_trace_note = None
    special = a.pmc_child or a.b1_child or a.only_ragged or a.only_24k or a.profile_config
    tail = ["--vocoder", a.vocoder, "--batch", str(a.batch), "--t-text", str(a.t_text), "--frames-per-token", str(a.frames_per_token),
            "--precision", a.precision]
    if rank == 0 and world == 1 and not a.no_pmc and not special:
        n_prec = len(PRECISIONS) if a.pmc_all else len({a.precision, "fp32"})
        traffic_table, traffic_source = pmc_passes(tail + (["--pmc-all"] if a.pmc_all else []), timeout_s=120 + 60 * n_prec)
        if traffic_table is None:
            note = traffic_source
            traffic_table, traffic_source = committed_traffic()
            traffic_source = f"{traffic_source} [live passes unavailable: {note}]"
    elif rank == 0 and not special:
        traffic_table, traffic_source = committed_traffic()
    if rank == 0 and world == 1 and not a.no_b1 and not special:
        b1_trace, b1_trace_note = b1_trace_pass(tail + ["--b1-iters", str(a.b1_iters)])

    if world > 1:      # N launch-heavy host loops on one node: keep each rank's CPU-side torch work on one thread (torchrun's default too)
        os.environ.setdefault("OMP_NUM_THREADS", "1")
    import torch
    if world > 1:
        torch.set_num_threads(int(os.environ["OMP_NUM_THREADS"]))

    # JATTS_BENCH_SHARED_GPU=1 (tests/test_distributed_gpu.py only): every rank on cuda:0 with gloo collectives, to exercise the
    # N > 1 code path on a one-GPU test box (RCCL refuses two ranks per device).  Never set by the driver; the line says so.
    shared = os.environ.get("JATTS_BENCH_SHARED_GPU") == "1"
    dev = torch.device("cuda", 0 if shared else local)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:   # one process per GPU, RCCL over xGMI (backend "nccl" on ROCm); rendezvous from the torchrun env
        import torch.distributed as dist
        if shared:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    if a.profile_config:
        j = Job(a.profile_config, a, dev, rank, a.batch if a.profile_config == "matcha" else 32)
        j.set_precision(a.precision)
        run_timed(j, a, 1, None, record=False)
        return
    if a.b1_child:    # the rocprofv3 --kernel-trace child of the b1_latency block: the timed utterances between two marker launches
        r = b1_run(a, dev, a.precision, a.b1_iters, markers=True)
        print(json.dumps({"b1_child": 1, "ms": r["ms"], "iters": a.b1_iters, "frames": r["frames"]}), flush=True)
        return
    job = Job("fs2", a, dev, rank, a.batch)
    if a.pmc_child:   # profiled child: one step of the headline arithmetic and of exact f32 (--pmc-all: of every arithmetic), nothing printed
        a.steps, a.warmup = 1, 1
        for p in (PRECISIONS if a.pmc_all else dict.fromkeys((a.precision, "fp32"))):
            job.set_precision(p)
            run_timed(job, a, 1, None)
        return

    job.set_precision(a.precision)
    if a.only_ragged:
        ragged_leg(job, a, world, dist, rank, 1.0, record=False)
        return
    if a.only_24k:
        job.use_vocoder("24k")
        run_timed(job, a, world, dist, record=False)
        return
    # the matrix pipe's practical ceiling on THIS part, measured before any timed region (~1 s in all)
    ceilings = None
    if not a.no_ceiling:
        from jatts_amd import hip as _h
        ceilings = measure_ceilings(dev, PRECISIONS if world == 1 else (a.precision, "fp32"))
        del _h
    head = run_timed(job, a, world, dist, a.pipeline)
    rep = kernel_report(head["recs"], a.steps, a.precision, head["dt"], traffic_table, traffic_source, ceilings)

    out = {
        "metric": f"audio samples/sec ({'22.05' if a.vocoder == '22k' else '24'} kHz) + RTF, FastSpeech2+HiFi-GAN",
        "value": head["value"], "unit": "samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": DTYPE_NAME[a.precision],
        "data": "synthetic",
        "config": {"workload": f"{job.name}+HiFi-GAN v1 {a.vocoder}, {a.batch} utts x {a.t_text} phonemes x "
                               f"{a.frames_per_token} frames per GPU",
                   "hop": job.hop, "sampling_rate": job.sr,
                   "parallelism": f"dp{world} (utterance sharding, int16 PCM all-gather)"
                                  + (" [shared-GPU test mode: all ranks on cuda:0, gloo]" if shared else "")},
        "n_ranks_seen": dist.get_world_size() if dist else 1,
        "backend": (("gloo [shared-GPU test mode]" if shared else "nccl (RCCL)") if dist else None),
        "rtf": head["rtf"], "stage_ms_per_step": head["stages"], "rank_ms_per_step": head["rank_ms"],
        "executor": "two-stream pipeline (jatts_amd.pipeline)" if a.pipeline else "sequential",
        "mfma_ceilings": ceilings,
    }
    out.update(rep)

    # ---- the ragged-length leg, same arithmetic as the headline
    if not a.no_ragged:
        out["ragged"] = ragged_leg(job, a, world, dist, rank, out["value"])
        out["ragged"]["resunit_ms_per_step_uniform"] = rep["resunit_ms_per_step"]

    # ---- the other arithmetics on the same batch, and how far apart their outputs are from the headline's
    if not a.no_fast_mode:
        mel0, wav0 = head["mel"].float().clone(), head["wave"].float().clone()
        # exact f32 always rides along (exact_f32_mode); the other arithmetics at N == 1 only
        others = [p for p in PRECISIONS if p != a.precision and (world == 1 or p == "fp32")]
        for other in others:
            job.set_precision(other)
            alt = run_timed(job, a, world, dist, a.pipeline)
            arep = kernel_report(alt["recs"], a.steps, other, alt["dt"], traffic_table, traffic_source, ceilings)
            same = alt["mel"].shape == mel0.shape and alt["wave"].shape == wav0.shape
            dm = (alt["mel"].float() - mel0) if same else None
            dw = (alt["wave"].float() - wav0) if same else None
            blk = {"dtype": DTYPE_NAME[other], "value": alt["value"], "unit": "samples/s", "ms_per_step": alt["ms_per_step"],
                   "rtf": alt["rtf"], "stage_ms_per_step": alt["stages"],
                   "max_abs_err_mel": float(dm.abs().max()) if same else None,
                   "rms_err_mel": float(dm.pow(2).mean().sqrt()) if same else None,
                   "mel_abs_max": float(mel0.abs().max()),
                   "max_abs_err_wave": float(dw.abs().max()) if same else None,
                   "rms_err_wave": float(dw.pow(2).mean().sqrt()) if same else None,
                   "wave_abs_max": float(wav0.abs().max()), "wave_rms": float(wav0.pow(2).mean().sqrt()),
                   "error_reference": f"the {DTYPE_NAME[a.precision]} run of this process on the same {a.batch} x "
                                      f"{a.t_text * a.frames_per_token}-frame batch (same durations: {same})",
                   "speedup_vs_headline": head["ms_per_step"] / alt["ms_per_step"]}
            blk.update(arep)
            out[MODE_KEY[other]] = blk
            del dm, dw, alt
        del mel0, wav0
    del head
    job_fs2_sd, job_voc_sd, job_vp, job_sr = job.sd, job.voc_sd, job.vp, job.sr

    # ---- the 24 kHz / hop-300 generator every JSUT / JVS recipe loads (conf/fastspeech2.v1.yaml:4-6,96-99; SURVEY 8d "report both"): the same
    # batch and acoustic model, headline arithmetic and exact f32, N == 1 only
    if world == 1 and not a.no_24k and a.vocoder == "22k":
        job.use_vocoder("24k")
        blk = {"sampling_rate": job.sr, "hop": job.hop, "upsample_scales": list(job.vp["upsample_scales"]),
               "workload": f"{job.name}+HiFi-GAN v1 24k (scales 5,5,4,3), {a.batch} utts x {a.t_text} phonemes x {a.frames_per_token} frames"}
        for name, p in (("headline", a.precision), ("exact_f32", "fp32")):
            if name == "exact_f32" and p == a.precision:
                continue
            job.set_precision(p)
            r = run_timed(job, a, 1, None)
            e = {"dtype": DTYPE_NAME[p], "value": r["value"], "unit": "samples/s", "ms_per_step": r["ms_per_step"], "rtf": r["rtf"],
                 "stage_ms_per_step": r["stages"], "samples_per_step": r["samples_per_step"]}
            e.update(kernel_report(r["recs"], a.steps, p, r["dt"], None, None, ceilings))
            blk[name] = e
            del r
        out["vocoder_24k"] = blk
        job.use_vocoder(a.vocoder)

    # ---- the drop-in B = 1 path: one utterance through model.inference(x) + vocoder.decode(mel), hipGraph replay against eager launches
    if world == 1 and not a.no_b1:
        j1 = None
        g = b1_run(a, dev, a.precision, a.b1_iters, graph=True)
        j1 = g.pop("job")
        e = b1_run(a, dev, a.precision, max(5, a.b1_iters // 3), graph=False, job=j1)
        e.pop("job")
        b1 = {"utterance": f"{g['phonemes']} phonemes -> {g['frames']} frames -> {g['samples']} samples", "dtype": DTYPE_NAME[a.precision],
              "call": "model.inference(x) + vocoder.decode(feat_gen), a host synchronisation per utterance (tts_decode.py:230,249)",
              "ms": g["ms"], "ms_min": g["ms_min"], "ms_all": g["ms_all"], "eager_ms": e["ms"], "eager_ms_all": e["ms_all"],
              "executor": "hipGraph replay (jatts_amd/graphs.py: front keyed by T_text, back by T_feats, generator by T_feats)",
              "rtf": g["ms"] / 1e3 / (g["samples"] / job.sr)}
        if b1_trace:
            b1.update(kernel_ms=b1_trace["kernel_ms"], launches=b1_trace["launches_per_utt"], wall_over_kernel=g["ms"] / b1_trace["kernel_ms"],
                      launch_gap_frac=1.0 - b1_trace["kernel_ms"] / g["ms"], eager_launch_gap_frac=1.0 - b1_trace["kernel_ms"] / e["ms"], trace=b1_trace)
        else:
            b1["kernel_ms"], b1["kernel_ms_note"] = None, b1_trace_note
        out["b1_latency"] = b1
        del j1, g, e
    del job
    torch.cuda.empty_cache()

    # ---- BASELINE configs 3 and 5 (per-GPU share), f32 and f16: N == 1 only
    if world == 1 and not a.no_configs:
        cfgs = []
        aa = argparse.Namespace(**vars(a))
        aa.steps, aa.warmup = max(2, min(3, a.steps)), 1
        for kind, nb, label in (("matcha", a.batch, "BASELINE configs[2]: Matcha-TTS (tts2 MAS) + HiFi-GAN, batch 64, ODE steps 10"),
                                ("vits", 32, "BASELINE configs[4] per-GPU share: JVS-style mel-VITS, 192-d spkemb, 32 utterances per GPU")):
            j = Job(kind, a, dev, rank, nb)
            line = {"config": label, "workload": f"{j.name}+HiFi-GAN v1 {a.vocoder}, {nb} utts x {a.t_text} phonemes x "
                                                 f"{a.frames_per_token} frames", "steps": aa.steps, "warmup": aa.warmup}
            ref = None
            for p in PRECISIONS:
                j.set_precision(p)
                r = run_timed(j, aa, 1, None, record=False)
                e = {"value": r["value"], "unit": "samples/s", "ms_per_step": r["ms_per_step"], "rtf": r["rtf"],
                     "stage_ms_per_step": r["stages"]}
                if p == "fp32":
                    ref = (r["mel"].float().clone(), r["wave"].float().clone())
                    line.update(dtype="f32", **e)
                    from jatts_amd import hip as _hip          # one more step with per-launch HIP events: the config's own roofline block
                    _hip.profile_begin()
                    rr = j.text2mel()
                    j.voc.decode_batch(rr["feats_rb"], rr["feat_gen"])
                    line["roofline"] = family_report(_hip.profile_end(), "fp32")
                    del rr
                else:
                    e["dtype"] = DTYPE_NAME[p]
                    e["max_abs_err_mel"] = float((r["mel"].float() - ref[0]).abs().max())
                    e["max_abs_err_wave"] = float((r["wave"].float() - ref[1]).abs().max())
                    e["mel_abs_max"], e["wave_abs_max"] = float(ref[0].abs().max()), float(ref[1].abs().max())
                    line[MODE_KEY[p]] = e
                del r
            cfgs.append(line)
            del j, ref
            import gc
            gc.collect()
            torch.cuda.empty_cache()
        out["configs"] = cfgs
        out["configs_note"] = ("configs[0] is the reference's own CPU case (see cpu_baseline); configs[3] = this line's workload on "
                               "8 GPUs (bench.py --gpus 8); configs[4] = 8 x the VITS line's per-GPU share")

    # ---- one FastSpeech2 `_train_step` at the recipe's batch size (not part of `value`): N == 1 only
    if world == 1 and not a.no_train:
        import gc
        out["training"] = []
        for kind in ("fs2", "matcha", "matcha_mas", "vits"):
            gc.collect()               # (the inference jobs above hold reference cycles; a live 10+ GB job slows the step by 20 %)
            torch.cuda.empty_cache()
            ln = train_step_line(dev, 5, kind)
            gc.collect()
            torch.cuda.empty_cache()
            sp = train_step_line(dev, 5, kind, precision="fp32_split")      # the same step with precision="fp32_split" (weight gradients stay exact f32)
            ln["fp32_split"] = {k: sp[k] for k in ("ms_per_step", "ms_per_step_mean", "loss_first", "loss_last", "dtype")}
            out["training"].append(ln)

    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        from jatts_amd.synthetic import synth_texts
        total = os.cpu_count() or 1
        texts = synth_texts(a.batch, a.cpu_t_text, 45, seed=1)
        # the CPU side gets its best thread count on THIS box: two utterances per candidate (tools/cpu_threads_sweep.py is the full
        # sweep, profiles/r03_cpu_threads.json), then the whole batch on the winner
        cands = [t for t in (8, 16, 24, 32, 64) if t <= total] or [total]   # profiles/r03_cpu_threads.json: 16 wins on the EPYC 9575F box, 128 is 5x slower
        cpu_baseline(job_fs2_sd, job_voc_sd, job_vp, texts[:1], 2, 1e9, cands[-1])        # warm-up
        trial = {t: cpu_baseline(job_fs2_sd, job_voc_sd, job_vp, texts[:2], 2, 1e9, t)["value"] for t in cands}
        threads = max(trial, key=trial.get)
        cb = cpu_baseline(job_fs2_sd, job_voc_sd, job_vp, texts, 2, a.cpu_budget, threads)
        cb["thread_calibration"] = {str(t): v for t, v in trial.items()}
        cb["rtf"] = cb["seconds"] / (cb["samples"] / job_sr)
        cb["cpu_model"], cb["host_logical_cores"] = cpu_model(), total
        # SURVEY 8d also asks for the recipe default OMP_NUM_THREADS=1 (path.sh:15): one thread, bounded sample
        c1 = cpu_baseline(job_fs2_sd, job_voc_sd, job_vp, texts, 2, min(20.0, a.cpu_budget), 1)
        cb["single_thread"] = {k: c1[k] for k in ("value", "unit", "cores", "sample", "seconds", "utterances")}
        cb["single_thread"]["rtf"] = c1["seconds"] / (c1["samples"] / job_sr)
        out["cpu_baseline"] = cb
        out["speedup_vs_cpu_rtf"] = cb["rtf"] / out["rtf"]
        for k in ("fast_mode", "f32_split_mode", "f32_emul_mode", "f32_emul6_mode", "f32_mode"):     # (f32_mode -> the line's exact_f32_mode)
            if k in out:
                out[k]["speedup_vs_cpu_rtf"] = cb["rtf"] / out[k]["rtf"]
    else:
        out["cpu_baseline"] = None
        out["cpu_baseline_note"] = ("N=1 only (rank 0 at N=1 times the CPU leg; the N>1 line also carries no live PMC traffic: "
                                    "roofline.traffic is the committed table)" if world > 1 else "--no-cpu-baseline")
    if rank == 0:
        sys.stdout.flush()
        sys.stderr.flush()
        print(compact_line(out, None if a.no_detail else write_detail(out)), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
