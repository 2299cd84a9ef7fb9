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
    """Deal utterances to ranks: longest first, each to the rank with the smallest load so far among those that still have room
    (at most ceil(n / W) utterances per rank, so the exchange step's per-rank slot count stays the batch size); ties go to the lowest rank.
    Plain round-robin over the sorted list hands rank 0 every round's longest utterance and rank W-1 every round's shortest -- a systematic
    0.5 % surplus at 512 utterances of 64..128 phonemes on 8 ranks; this deal leaves 0.006 % (tests/test_distributed_cpu.py).
    Returns per-rank lists of original indices; deterministic."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    cap = -(-len(order) // world_size)
    parts, load = [[] for _ in range(world_size)], [0] * world_size
    for i in order:
        r = min((r for r in range(world_size) if len(parts[r]) < cap), key=lambda r: (load[r], r))
        parts[r].append(i)
        load[r] += int(lengths[i])
    return parts


def shard_load(lengths, parts):
    """-> (max over ranks, mean over ranks) of the summed lengths of a sharding: the predicted imbalance of the data-parallel step
    (its time follows the slowest rank; per-utterance cost is linear in frames to within the attention's 7 % share)."""
    loads = [sum(int(lengths[i]) for i in p) for p in parts]
    return max(loads), sum(loads) / max(1, len(loads))
