"""CPU, world_size 2 over gloo: the N>1 exchange step (lengths + one audio all-gather) and the
shard/unshard bookkeeping.  The synthesis itself needs no collective."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from jatts_amd.distributed import gather_audio, shard_utterances, unshard

    lens_all = [700, 30, 512, 64, 5]           # samples per utterance (global order)
    parts = shard_utterances(lens_all, world)
    mine = parts[rank]
    # "synthesise": utterance i is a ramp tagged with its id
    waves = [torch.arange(lens_all[i], dtype=torch.float32) + 1000.0 * i for i in mine]
    packed = torch.cat(waves) if waves else torch.zeros(0)
    got, lens = gather_audio(packed, [lens_all[i] for i in mine])
    full = unshard(got, lens, parts)
    ok = all(torch.equal(full[i], torch.arange(lens_all[i], dtype=torch.float32) + 1000.0 * i)
             for i in range(len(lens_all)))
    q.put((rank, ok, [len(p) for p in parts]))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_audio_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert res[0][2] == [3, 2]
