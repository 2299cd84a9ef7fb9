"""hipGraph replay of the B = 1 drop-in path (`model.inference(x)` / `vocoder.decode(mel)`, the calls of the reference's stage-4 loop,
jatts/bin/tts_decode.py:203-255).

One 128-phoneme utterance is ~600 kernel launches of a few microseconds each: issued one by one from Python the GPU waits for the host
between most of them.  A segment of the path whose launch sequence depends only on its shape signature (T_text for everything up to the
length regulator, T_feats behind it) is captured ONCE as a hipGraph -- through torch.cuda.graph, which supplies the capture stream and
the graph's private memory pool: plumbing, like the stream and the allocator everywhere else -- and replayed for every later utterance
with that signature: the same kernels in the same order on the same arithmetic, bit-identical to the eager launches
(tests/test_graph_gpu.py).

Rules (the trainer's graph mode follows the same ones, jatts_amd/training.py):
  * a signature's FIRST sight runs eagerly -- it fills the bounded host-to-device caches (ragged geometry, positional tables) that a
    capture may not upload into -- its SECOND sight captures, later ones replay;
  * every cached device tensor handed out during a capture is pinned by the graph's record (hip.keep_begin), the caches may evict;
  * inputs are copied into the graph's static buffers (stream-ordered), outputs are handed out as stream-ordered CLONES: a caller who
    keeps the mel of utterance n while utterance n + 1 replays the same graph sees its own values;
  * a failed capture marks the signature eager-only (logged): it never retries, and the call is answered by eager launches;
  * least-recently-used graphs are dropped beyond `max_graphs` (each owns one utterance's activations).
JATTS_INFER_GRAPH=0 switches the whole mechanism off (every call eager)."""
import collections
import logging
import os

import torch

from . import hip

ENABLED = os.environ.get("JATTS_INFER_GRAPH", "1") != "0"


class GraphCache:
    def __init__(self, max_graphs=24, max_seen=4096):
        self.max_graphs, self.max_seen = int(max_graphs), int(max_seen)
        self._g = collections.OrderedDict()       # key -> state dict
        self.stats = dict(eager=0, captured=0, replayed=0, failed=0)

    def clear(self):
        self._g.clear()

    def __len__(self):
        return sum(1 for v in self._g.values() if v.get("graph") is not None)

    def run(self, key, fn, inputs):
        """outputs = fn(*inputs) for a launch sequence that depends on `key` only.  inputs: tuple of device tensors (contiguous); fn returns a
        tensor or a tuple of tensors / None.  -> the same structure, eagerly computed or replayed (clones of the graph's static outputs)."""
        if not ENABLED or hip._PROF is not None or torch.cuda.is_current_stream_capturing():
            self.stats["eager"] += 1
            return fn(*inputs)
        st = self._g.get(key)
        if st is None:                                  # first sight: eager (fills the upload caches), remember the signature
            if len(self._g) >= self.max_seen:
                for k in [k for k, v in self._g.items() if v.get("graph") is None and not v.get("eager_only")][: self.max_seen // 2]:
                    del self._g[k]
            self._g[key] = {"graph": None}
            self.stats["eager"] += 1
            return fn(*inputs)
        if st.get("eager_only"):
            self.stats["eager"] += 1
            return fn(*inputs)
        if st["graph"] is None:                         # second sight: capture
            live = [k for k, v in self._g.items() if v.get("graph") is not None]
            while len(live) >= self.max_graphs:         # least recently used first (move_to_end below)
                self._g[live.pop(0)] = {"graph": None}
            st["in"] = tuple(t.clone() for t in inputs)
            g = torch.cuda.CUDAGraph()
            st["keep"] = hip.keep_begin()
            try:
                torch.cuda.current_stream().synchronize()
                with torch.cuda.graph(g):
                    st["out"] = fn(*st["in"])
            except Exception as e:      # a capture that cannot be made (an upload the caches had evicted, a launch the runtime refuses to record):
                hip.keep_end()          # this signature runs eagerly from now on -- the same kernels, launched one by one; never a wrong result
                self._g[key] = {"graph": None, "eager_only": True}
                self.stats["failed"] += 1
                logging.warning("jatts_amd.graphs: capture of %r failed (%s: %s); this signature stays on eager launches", key, type(e).__name__, e)
                torch.cuda.synchronize()
                return fn(*inputs)
            hip.keep_end()
            st["graph"] = g
            self.stats["captured"] += 1
        else:
            self.stats["replayed"] += 1
        self._g.move_to_end(key)
        for dst, src in zip(st["in"], inputs):
            if dst.shape != src.shape or dst.dtype != src.dtype:
                raise ValueError(f"graph signature {key!r}: input {tuple(src.shape)} {src.dtype} does not match the captured {tuple(dst.shape)} {dst.dtype}")
            dst.copy_(src, non_blocking=True)
        st["graph"].replay()
        out = st["out"]
        if torch.is_tensor(out):
            return out.clone()
        return tuple(None if o is None else o.clone() for o in out)
