#!/usr/bin/env python3
"""Stage-4 decoding CLI — drop-in for ``jatts/bin/tts_decode.py`` (reference :30-274) on the MI355X path.

Same flags (``--csv --stats --token-list --token-column --outdir --checkpoint [--config] [--verbose]``),
same outputs (``<outdir>/wav/<sample_id>.wav`` as PCM_16 at the vocoder's sampling rate).  Differences,
all additive: utterances are synthesised in ragged batches (``--batch-size``) instead of one by one; the
per-utterance PNG (reference :240-244) is written only with ``--plot``; under ``torch.distributed.run``
every rank decodes its own shard of the csv (no collective: the outputs are files).

    python -m jatts_amd.bin.tts_decode --csv data/dev.csv --stats exp/stats.h5 --token-list exp/tokens.txt \\
        --token-column phonemes --checkpoint exp/checkpoint-100000steps.pkl --outdir exp/results/dev
"""
import argparse
import csv
import logging
import os
import time
import wave

import numpy as np
import torch
import yaml

import jatts_amd.models
from jatts_amd.hostlogic import shard_utterances
from jatts_amd.vocoder import Vocoder


class TokenIDConverter:
    """Token list file: one symbol per line, id = line number (reference utils/token_id_converter.py:12-60)."""

    def __init__(self, token_list, unk_symbol="<unk>"):
        with open(token_list, "r", encoding="utf-8") as f:
            self.token_list = [line.rstrip() for line in f]
        self.token2id = {}
        for i, t in enumerate(self.token_list):
            if t in self.token2id:
                raise RuntimeError(f"token list {token_list}: line {i + 1} repeats the symbol {t!r}")
            self.token2id[t] = i
        if unk_symbol not in self.token2id:
            raise RuntimeError(f"token list {token_list} has no {unk_symbol!r} entry to map out-of-vocabulary tokens to")
        self.unk_id = self.token2id[unk_symbol]

    def tokens2ids(self, tokens):
        return [self.token2id.get(t, self.unk_id) for t in tokens]


def read_stats(path, feat):
    """``<feat>_mean`` / ``<feat>_scale`` (reference compute_statistics.py:94-103); .h5 needs h5py, .npz works everywhere."""
    if str(path).endswith(".npz"):
        z = np.load(path)
        return {"mean": z[f"{feat}_mean"].astype(np.float32), "scale": z[f"{feat}_scale"].astype(np.float32)}
    try:
        import h5py
    except ImportError as e:
        raise ImportError("reading .h5 stats needs h5py; convert to .npz (<feat>_mean, <feat>_scale)") from e
    with h5py.File(path, "r") as f:
        return {"mean": f[f"{feat}_mean"][()].astype(np.float32), "scale": f[f"{feat}_scale"][()].astype(np.float32)}


def to_pcm16(y):
    """float [-1, 1] -> int16 exactly as libsndfile's PCM_16 writer converts float input (float multiply by 32767,
    lrintf), the host twin of jatts_pcm16."""
    y = np.clip(np.asarray(y, dtype=np.float32), np.float32(-1.0), np.float32(1.0))
    return np.rint(y * np.float32(32767.0)).astype("<i2")


def write_wav_pcm16(path, y, sr):
    """Write mono 16-bit PCM; ``y`` is float in [-1, 1] or already int16 PCM (from jatts_pcm16)."""
    pcm = np.asarray(y)
    if pcm.dtype != np.dtype("<i2"):
        pcm = to_pcm16(pcm)
    with wave.open(path, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(int(sr))
        w.writeframes(pcm.tobytes())


class OutputPipeline:
    """the stage-4 output side overlapped with synthesis.  Per batch: jatts_pcm16 on the compute stream,
    an asynchronous device->host copy of the int16 samples (half the bytes of the float waveform) into one of two pinned
    buffers on a side stream, and the wav files written by a worker thread while the GPU already runs the next batch.
    The reference converts and writes utterance by utterance inside the synthesis loop (tts_decode.py:240-255)."""

    def __init__(self, device, sr, n_buffers=2):
        from concurrent.futures import ThreadPoolExecutor
        self.device, self.sr = device, int(sr)
        self.copy_stream = torch.cuda.Stream(device=device)
        self.slots = [dict(buf=None, fut=None) for _ in range(n_buffers)]
        self.k = 0
        self.pool = ThreadPoolExecutor(max_workers=1)

    def submit(self, y, jobs):
        """y: packed float waveform on the GPU; jobs: list of (path, first sample, n samples)."""
        from jatts_amd import hip
        slot = self.slots[self.k % len(self.slots)]
        self.k += 1
        if slot["fut"] is not None:
            slot["fut"].result()                      # this buffer's previous batch is on disk
        pcm = hip.pcm16(y)
        if slot["buf"] is None or slot["buf"].numel() < pcm.numel():
            slot["buf"] = torch.empty(max(pcm.numel(), 1), dtype=torch.int16).pin_memory()
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.copy_stream):
            self.copy_stream.wait_event(ready)
            slot["buf"][: pcm.numel()].copy_(pcm, non_blocking=True)
            pcm.record_stream(self.copy_stream)
            done = torch.cuda.Event()
            done.record(self.copy_stream)
        host = slot["buf"]

        def write():
            done.synchronize()
            a = host.numpy()
            for path, o, n in jobs:
                write_wav_pcm16(path, a[o:o + n], self.sr)
        slot["fut"] = self.pool.submit(write)

    def close(self):
        for s in self.slots:
            if s["fut"] is not None:
                s["fut"].result()
        self.pool.shutdown()


_SPKEMB_CACHE = {}


def load_spkemb(path):
    """Per-file cache of precomputed speaker embeddings (JVS decodes 100 speakers x many utterances; the
    reference re-runs the SpeechBrain extractor per utterance, tts_decode.py:209-212)."""
    v = _SPKEMB_CACHE.get(path)
    if v is None:
        v = _SPKEMB_CACHE[path] = np.load(path).astype(np.float32)
    return v


def read_items(csv_path, token_column, converter):
    items = []
    with open(csv_path, newline="") as f:
        for row in csv.DictReader(f):
            tokens = [p for p in row[token_column].split(" ") if p != ""]   # tts_dataset.py:108-116
            row["token_indices"] = np.array(converter.tokens2ids(tokens), dtype=np.int64)
            items.append(row)
    return items


def get_parser():
    p = argparse.ArgumentParser(description="Decode with trained TTS model (MI355X HIP path).")
    p.add_argument("--csv", required=True, type=str)
    p.add_argument("--stats", required=True, type=str)
    p.add_argument("--token-list", required=True, type=str)
    p.add_argument("--token-column", required=True, type=str)
    p.add_argument("--outdir", required=True, type=str)
    p.add_argument("--checkpoint", required=True, type=str)
    p.add_argument("--config", default=None, type=str)
    p.add_argument("--verbose", type=int, default=1)
    p.add_argument("--batch-size", type=int, default=64, help="utterances per ragged batch (extension)")
    p.add_argument("--precision", default="fp32", choices=["fp16", "fp32", "fp32_split", "fp32_bf16x3", "fp32_bf16x3_6p"],
                   help="fp32 = the reference's arithmetic (default); fp16 = fast mode: f16 MFMA operands, f32 accumulate; fp32_split = f32 "
                        "tensors, convs and fused vocoder units on error-corrected split f16 hi/lo MFMA operands; fp32_bf16x3 = f32 tensors, the "
                        "same kernels on three exact bf16 terms per operand and seven MFMA products (a one-term contraction within 2 x an f32 FMA's error "
                        "bound for every input); fp32_bf16x3_6p = six products (extensions)")
    p.add_argument("--plot", action="store_true", help="also write <outdir>/outs/<id>.png like the reference")
    p.add_argument("--n_gpus", "--n-gpus", dest="n_gpus", type=int, default=1,
                   help="one process per GPU, every rank decodes its own shard of the csv (extension; the recipes' n_gpus). "
                        "Without a launcher in the environment the ranks are started here as child processes")
    return p


def main(argv=None):
    args = get_parser().parse_args(argv)
    if args.n_gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python -m jatts_amd.bin.tts_decode --n_gpus N`: start the N ranks as children before anything touches the GPU
        import sys
        from jatts_amd.distributed import self_launch
        raise SystemExit(self_launch(args.n_gpus, "jatts_amd.bin.tts_decode", sys.argv[1:] if argv is None else list(argv), module=True))
    level = logging.DEBUG if args.verbose > 1 else logging.INFO if args.verbose > 0 else logging.WARN
    logging.basicConfig(level=level, format="%(asctime)s (%(module)s:%(lineno)d) %(levelname)s: %(message)s")
    os.makedirs(os.path.join(args.outdir, "wav"), exist_ok=True)
    if args.config is None:
        args.config = os.path.join(os.path.dirname(args.checkpoint), "config.yml")
    with open(args.config) as f:
        config = yaml.load(f, Loader=yaml.Loader)
    config.update(vars(args))

    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    if args.n_gpus > 1 and world != args.n_gpus:
        raise RuntimeError(f"--n_gpus {args.n_gpus} under a launcher with WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise RuntimeError("jatts_amd needs an MI355X: there is no CPU fallback")
    # JATTS_SHARED_GPU=1 (tests only): every rank on cuda:0, to run the world > 1 branch on a one-GPU box
    shared = os.environ.get("JATTS_SHARED_GPU") == "1"
    device = torch.device("cuda", 0 if shared else int(os.environ.get("LOCAL_RANK", 0)))
    torch.cuda.set_device(device)

    converter = TokenIDConverter(args.token_list)
    items = read_items(args.csv, args.token_column, converter)
    logging.info(f"Dataset size = {len(items)}.")
    if world > 1:
        from jatts_amd.distributed import pin_host_threads
        pin_host_threads(1)
        mine = shard_utterances([len(it["token_indices"]) for it in items], world)[rank]
        items = [items[i] for i in mine]

    model_class = getattr(jatts_amd.models, config["model_type"])          # tts_decode.py:139
    model = model_class(**config["model_params"])
    model.load_state_dict(torch.load(args.checkpoint, map_location="cpu")["model"])
    model = model.eval().to(device).set_precision(args.precision)
    logging.info(f"Loaded model parameters from {args.checkpoint}.")

    stats = read_stats(args.stats, config["out_feat_type"])                # tts_decode.py:160-164
    if not config.get("vocoder", False):
        raise NotImplementedError("Griffin-Lim fallback (no vocoder) is outside the HIP path")
    vcfg = config["vocoder"]
    if vcfg.get("vocoder_type", "") not in ("", "hifigan", "parallel_wavegan"):
        raise NotImplementedError(f"vocoder_type {vcfg.get('vocoder_type')} is not on the HIP path")
    vocoder = Vocoder(vcfg["checkpoint"], vcfg["config"], vcfg["stats"], device, trg_stats=stats)
    vocoder.set_precision(args.precision)
    hop = vocoder.model.hop
    uses_spk = "spkemb" in config.get("feat_list", [])
    kw = {}
    if config["model_type"] in ("MatchaTTS", "MatchaTTS_MAS"):              # tts_decode.py:217-226
        kw = {"temperature": config["temperature"], "n_timesteps": config["ode_steps"]}

    order = sorted(range(len(items)), key=lambda i: -len(items[i]["token_indices"]))  # similar lengths together
    n_frames, t0 = 0, time.time()
    out_pipe = OutputPipeline(device, vocoder.config["sampling_rate"])
    spk_extractor = None
    for s in range(0, len(order), args.batch_size):
        batch = [items[i] for i in order[s:s + args.batch_size]]
        texts = [torch.from_numpy(it["token_indices"]).to(device) for it in batch]
        if uses_spk:
            if "spkemb_path" in batch[0]:          # precomputed embeddings (.npy), cached per file
                spk = torch.from_numpy(np.stack([load_spkemb(it["spkemb_path"]) for it in batch])).float().to(device)
            else:                                  # the reference's path: extract from ref_wav_path (tts_decode.py:209-212)
                if spk_extractor is None:
                    from jatts_amd.spkemb import SpkEmbExtractor
                    ckpt = config.get("spkemb_checkpoint") or os.environ.get("JATTS_SPKEMB_CHECKPOINT")
                    if not ckpt:
                        raise RuntimeError("feat_list has 'spkemb' and the csv has no spkemb_path column: set spkemb_checkpoint (config) or "
                                           "JATTS_SPKEMB_CHECKPOINT to SpeechBrain's ECAPA embedding_model.ckpt")
                    spk_extractor = SpkEmbExtractor(device, checkpoint=ckpt, **config.get("spkemb_params", {}))   # ECAPA_TDNN kwargs
                spk = torch.from_numpy(spk_extractor.forward_many([it["ref_wav_path"] for it in batch])).float().to(device)
            r = model.inference_batch(texts, spembs=spk, **kw)
        else:
            r = model.inference_batch(texts, **kw)
        y = vocoder.decode_batch(r["feats_rb"], r["feat_gen"])
        jobs, o = [], 0
        for it, nf in zip(batch, r["olens"]):
            sid = it.get("sample_id", it.get("id", str(o)))
            jobs.append((os.path.join(args.outdir, "wav", f"{sid}.wav"), o * hop, nf * hop))
            o += nf
        out_pipe.submit(y, jobs)                                            # PCM conversion, async D2H, wav writer thread
        mel_host = r["feat_gen"].cpu().numpy() if args.plot else None
        o = 0
        for it, nf in zip(batch, r["olens"]):
            sid = it.get("sample_id", it.get("id", str(o)))
            if args.plot:
                import matplotlib
                matplotlib.use("Agg")
                import matplotlib.pyplot as plt
                os.makedirs(os.path.join(args.outdir, "outs"), exist_ok=True)
                plt.figure(figsize=(8, 3))
                plt.imshow(mel_host[o:o + nf].T, origin="lower", aspect="auto")
                plt.savefig(os.path.join(args.outdir, "outs", f"{sid}.png"))
                plt.close()
            o += nf
        n_frames += sum(r["olens"])
    out_pipe.close()
    dt = time.time() - t0
    logging.info("inference speed = %.1f frames / sec. (RTF = %.5f)" % (
        n_frames / max(dt, 1e-9), dt / max(n_frames * hop / vocoder.config["sampling_rate"], 1e-9)))


if __name__ == "__main__":
    main()
