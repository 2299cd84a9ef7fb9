"""Two-stream stage-4 executor: the text2mel of batch k+1 runs on one HIP stream while the vocoder of batch k runs on
another.  The acoustic model's many small, latency-bound kernels (and its one host sync for the frame counts) then hide
under the MFMA-bound vocoder instead of idling the GPU between batches.  Results are bit-identical to the sequential
loop (tests/test_fullsize_gpu.py); measured 47.1 -> 45.0 ms per batch of 64 x 128 phonemes (profiles/r01_notes.md).
The reference has no counterpart (tts_decode.py:203-255 is one utterance at a time on one stream).
"""
import torch


class Stage4Pipeline:
    def __init__(self, model, vocoder):
        self.model, self.vocoder = model, vocoder
        self.s_mel, self.s_voc = torch.cuda.Stream(), torch.cuda.Stream()

    @torch.no_grad()
    def run(self, batches, **infer_kw):
        """batches: iterable of lists of token-id tensors.  Yields (model result dict, packed waveform) per batch, in
        order; the waveform of a yielded batch is complete only after ``torch.cuda.current_stream().wait_stream(...)``
        or a device synchronize -- ``run`` makes the caller's stream wait for it before yielding."""
        caller = torch.cuda.current_stream()
        self.s_mel.wait_stream(caller)
        self.s_voc.wait_stream(caller)
        it = iter(batches)
        try:
            cur = next(it)
        except StopIteration:
            return
        with torch.cuda.stream(self.s_mel):
            r = self.model.inference_batch(cur, **infer_kw)
            ready = torch.cuda.Event()
            ready.record(self.s_mel)
        while True:
            with torch.cuda.stream(self.s_voc):
                self.s_voc.wait_event(ready)
                for t in (r["feat_gen"], r["feats_rb"].cu):
                    t.record_stream(self.s_voc)
                y = self.vocoder.decode_batch(r["feats_rb"], r["feat_gen"])
                done = torch.cuda.Event()
                done.record(self.s_voc)
            out = (r, y)
            try:
                nxt = next(it)
            except StopIteration:
                nxt = None
            if nxt is not None:   # enqueued AFTER this batch's vocoder, so the two overlap on the GPU
                with torch.cuda.stream(self.s_mel):
                    r = self.model.inference_batch(nxt, **infer_kw)
                    ready = torch.cuda.Event()
                    ready.record(self.s_mel)
            caller.wait_event(done)
            y.record_stream(caller)
            for t in out[0].values():   # every tensor of the result was allocated on s_mel and is read on the caller's stream
                if isinstance(t, torch.Tensor) and t.is_cuda:
                    t.record_stream(caller)
            yield out
            if nxt is None:
                return
