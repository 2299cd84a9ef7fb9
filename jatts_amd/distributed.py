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


def pin_host_threads(n=1):
    """One process per GPU: every rank is a launch-heavy host loop, and eight of them each spinning up torch's default intra-op
    pool (all cores) fight for the same cores.  Call before any torch CPU work; honours an OMP_NUM_THREADS the user set."""
    import os
    if "OMP_NUM_THREADS" not in os.environ:
        os.environ["OMP_NUM_THREADS"] = str(n)
        torch.set_num_threads(n)


def gather_audio(wave, lens, max_utts=None, group=None, pcm16=None):
    """All-gather of variable-length audio: one dense collective, ragged shards padded to the longest rank.

    wave: (sum(lens),) packed waveforms of this rank (f32 in [-1, 1], or int16 PCM already); lens: samples per local
    utterance.  Float audio on the GPU is first converted to int16 PCM by the jatts_pcm16 kernel -- the format
    tts_decode.py:250-255 stores anyway (sf.write(..., "PCM_16")) and half the bytes on xGMI.  ``pcm16=False`` keeps
    the input dtype.  ``max_utts``: an upper bound on utterances per rank known to every rank (the batch size); when
    omitted it is agreed on with one extra tiny all-reduce.
    Returns (list over ranks of packed tensors -- views of one (world, slot) buffer --, list over ranks of lens).

    Two collectives: (1) one fixed-size header all-gather [n_utts, n_samples, lens...]; (2) the payload as ONE
    all_gather_into_tensor of `slot = max over ranks of n_samples` elements per rank (the same call on RCCL and gloo).
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = wave.device
    if pcm16 is None:
        pcm16 = wave.is_cuda and wave.is_floating_point()
    if pcm16 and wave.is_floating_point():
        from . import hip
        wave = hip.pcm16(wave.float().contiguous())
    wave = wave.contiguous()
    if max_utts is None:
        m = torch.tensor([len(lens)], dtype=torch.int64, device=dev)
        dist.all_reduce(m, op=dist.ReduceOp.MAX, group=group)
        max_utts = int(m)
    if len(lens) > max_utts:
        raise ValueError("gather_audio: more local utterances than max_utts")
    head = torch.zeros(max_utts + 2, dtype=torch.int64)
    head[0], head[1] = len(lens), int(wave.numel())
    if lens:
        head[2:2 + len(lens)] = torch.tensor(lens, dtype=torch.int64)
    heads = torch.empty(world * (max_utts + 2), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(heads, head.to(dev), group=group)
    heads = heads.view(world, max_utts + 2).cpu()
    n_utts, n_samp = heads[:, 0].tolist(), heads[:, 1].tolist()
    # ONE dense all-gather for every backend and every shape of shard: each rank contributes max(n_samp) samples (ragged shards
    # are padded -- int16 PCM of a 64-utterance shard is 25 MB, the padding costs microseconds on xGMI), so the ragged case runs
    # the very collective the equal-shard bench exercises; no uneven all_gather / grouped broadcasts that only a multi-GPU ragged
    # run would ever reach.
    slot = max(n_samp)
    esz = wave.element_size()
    as_bytes = (lambda t: t.view(torch.uint8)) if esz != 1 and wave.dtype == torch.int16 else (lambda t: t)  # RCCL has no int16 type
    buf = torch.empty(world, slot, dtype=wave.dtype, device=dev)
    if slot > 0:
        if wave.numel() == slot:
            send = wave
        else:
            send = torch.zeros(slot, dtype=wave.dtype, device=dev)
            send[:wave.numel()].copy_(wave)
        dist.all_gather_into_tensor(as_bytes(buf.view(-1)), as_bytes(send), group=group)
    parts = [buf[r, :n_samp[r]] for r in range(world)]
    lens_out = [heads[r, 2:2 + n_utts[r]].tolist() for r in range(world)]
    return parts, lens_out


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


def self_launch(n, target, argv, module=False, relay=None):
    """Start `n` ranks of `target` (a script path, or a module name with module=True) the way the recipes / the driver do --
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node n --master-addr 127.0.0.1 --master-port <free> ...` -- as a CHILD
    process and return its exit code.  To be called BEFORE the calling process touches the GPU (a process that has initialised HIP
    must never be replaced by an exec on this pool, and does not need to be: the parent only waits).  `relay(line) -> bool` may claim
    stdout lines (they are then not echoed to stderr)."""
    import os
    import socket
    import subprocess
    import sys
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port)]
    cmd += (["-m", target] if module else [target]) + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this pool (RCCL across processes)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    for ln in proc.stdout:
        if relay is None or not relay(ln):
            sys.stderr.write(ln)
    rc = proc.wait()
    sys.stderr.flush()
    return rc
