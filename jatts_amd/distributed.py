"""Utterance-batch data parallelism for stage 4.

One process per GPU.  Utterances are independent (eval-mode BatchNorm, no cross-utterance
state), so each rank synthesises its own shard with NO collective on the data path; the only
exchange step is at the end: an all-gather of per-utterance sample counts followed by ONE
all-gather of the generated audio (RCCL over xGMI: backend "nccl" on ROCm).  The reference has
no inference-time collective to mirror (tts_decode.py:203-255 is a single-process loop).
"""
import torch
import torch.distributed as dist

from .hostlogic import shard_utterances  # noqa: F401  (re-exported)


def gather_audio(wave, lens, group=None):
    """All-gather variable-length audio.

    wave: (sum(lens),) float tensor of this rank's packed waveforms; lens: list[int] samples per
    local utterance.  Returns (list over ranks of packed tensors, list over ranks of lens).
    Two collectives: lengths (tiny) then ONE flat padded payload.
    """
    world = dist.get_world_size(group)
    dev = wave.device
    n_local = torch.tensor([len(lens), int(wave.numel())], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    max_utts = max(int(c[0]) for c in counts)
    max_samples = max(int(c[1]) for c in counts)
    lens_t = torch.zeros(max_utts, dtype=torch.int64, device=dev)
    lens_t[: len(lens)] = torch.tensor(lens, dtype=torch.int64, device=dev)
    all_lens = torch.empty(world * max_utts, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(all_lens, lens_t, group=group)
    payload = wave if wave.numel() == max_samples else torch.cat(
        [wave, wave.new_zeros(max_samples - wave.numel())])
    out = torch.empty(world * max_samples, dtype=wave.dtype, device=dev)
    dist.all_gather_into_tensor(out, payload.contiguous(), group=group)
    waves, lens_out = [], []
    for r in range(world):
        n_u, n_s = int(counts[r][0]), int(counts[r][1])
        waves.append(out[r * max_samples: r * max_samples + n_s])
        lens_out.append(all_lens[r * max_utts: r * max_utts + n_u].tolist())
    return waves, lens_out


def unshard(waves, lens_per_rank, parts):
    """Put gathered utterances back into the original order given shard_utterances()' parts."""
    n = sum(len(p) for p in parts)
    out = [None] * n
    for r, idxs in enumerate(parts):
        o = 0
        for j, i in enumerate(idxs):
            m = lens_per_rank[r][j]
            out[i] = waves[r][o:o + m]
            o += m
    return out
