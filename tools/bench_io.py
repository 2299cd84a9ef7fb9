#!/usr/bin/env python3
"""SURVEY 8(f).2 measurement: stage-4 with its output side (PCM_16 wav files on local disk), 64 utterances x 128 phonemes
per batch: the reference-shaped serial output (float D2H, host conversion, write inside the loop) against
jatts_amd.bin.tts_decode.OutputPipeline (GPU PCM conversion, async int16 D2H into pinned buffers, writer thread)."""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from jatts_amd.bin.tts_decode import OutputPipeline, write_wav_pcm16  # noqa: E402
from jatts_amd.models import FastSpeech2  # noqa: E402
from jatts_amd.synthetic import (FS2_JSUT, HIFIGAN_V1_22K, pin_duration_head, synth_hifigan_state, synth_state_dict,  # noqa: E402
                                 synth_texts)
from jatts_amd.vocoder import Vocoder  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    m = FastSpeech2(idim=45, **FS2_JSUT)
    m.load_state_dict(pin_duration_head(synth_state_dict(m.state_dict(), 0), 6))
    m = m.to(dev).set_precision("fp16")
    ones, zeros = [1.0] * 80, [0.0] * 80
    voc = Vocoder(synth_hifigan_state(HIFIGAN_V1_22K, 0),
                  {"sampling_rate": 22050, "generator_type": "HiFiGANGenerator", "generator_params": HIFIGAN_V1_22K},
                  {"mean": zeros, "scale": ones}, dev, trg_stats={"mean": zeros, "scale": ones})
    voc.set_precision("fp16")
    hop = voc.model.hop
    texts = [t.to(dev) for t in synth_texts(64, 128, 45, seed=1)]
    n_batches = 8

    def synth():
        r = m.inference_batch(texts)
        return r, voc.decode_batch(r["feats_rb"], r["feat_gen"])

    synth()
    torch.cuda.synchronize()
    res = {}
    with tempfile.TemporaryDirectory() as td:
        def jobs_for(r, bi):
            o, jobs = 0, []
            for u, nf in enumerate(r["olens"]):
                jobs.append((os.path.join(td, f"b{bi}_u{u}.wav"), o * hop, nf * hop))
                o += nf
            return jobs
        t0 = time.perf_counter()
        for bi in range(n_batches):
            r, y = synth()
            yh = y.cpu().numpy()
            for path, o, n in jobs_for(r, bi):
                write_wav_pcm16(path, yh[o:o + n], 22050)
        res["serial_ms_per_batch"] = (time.perf_counter() - t0) / n_batches * 1e3
        pipe = OutputPipeline(dev, 22050)
        t0 = time.perf_counter()
        for bi in range(n_batches):
            r, y = synth()
            pipe.submit(y, jobs_for(r, bi))
        pipe.close()
        torch.cuda.synchronize()
        res["overlapped_ms_per_batch"] = (time.perf_counter() - t0) / n_batches * 1e3
        t0 = time.perf_counter()
        for bi in range(n_batches):
            synth()
        torch.cuda.synchronize()
        res["synthesis_only_ms_per_batch"] = (time.perf_counter() - t0) / n_batches * 1e3
    res["samples_per_batch"] = sum(r["olens"]) * hop
    print(json.dumps(res))


if __name__ == "__main__":
    main()
