"""Host-side helpers of the stage-4 path (pure index logic, no arithmetic on features)."""
import torch


def make_pad_mask(lengths, maxlen=None):
    """True at padded positions (reference modules/utils.py:9-126, basic form)."""
    if not isinstance(lengths, list):
        lengths = torch.as_tensor(lengths).long().tolist()
    maxlen = int(max(lengths)) if maxlen is None else maxlen
    rng = torch.arange(0, maxlen, dtype=torch.int64).unsqueeze(0)
    return rng >= torch.tensor(lengths, dtype=torch.int64).unsqueeze(-1)


def make_non_pad_mask(lengths, maxlen=None):
    """modules/utils.py:129-215."""
    return ~make_pad_mask(lengths, maxlen)


def unpack(packed, lens, mul=1):
    """Split a packed (sum(lens)*mul, ...) tensor into per-utterance views."""
    out, o = [], 0
    for n in lens:
        out.append(packed[o:o + n * mul])
        o += n * mul
    return out


def shard_utterances(lengths, world_size):
    """Deal utterances to ranks: sort by length (longest first), round-robin.
    Returns per-rank lists of original indices; deterministic."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    return [order[r::world_size] for r in range(world_size)]
