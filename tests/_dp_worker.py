"""Worker of tests/test_distributed_gpu.py::test_data_parallel_training_world2: one data-parallel rank of a FastSpeech2 training job.
Both ranks share the test box's single GPU; the gradient all-reduce runs over gloo (RCCL refuses two ranks on one device), which
exercises the same FastSpeech2Trainer / allreduce_flat code path the RCCL job takes."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    overlap = len(sys.argv) < 5 or sys.argv[4] != "0"
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from helpers import golden_state, load_golden
    from jatts_amd.models import FastSpeech2
    from jatts_amd.synthetic import FS2_SMALL
    from jatts_amd.training import FastSpeech2Trainer
    z, keys = load_golden("fs2_train_small.npz")
    zi, _ = load_golden("fs2_forward_small.npz")
    dev = torch.device("cuda:0")
    m = FastSpeech2(idim=20, **{**FS2_SMALL, **json.loads(str(z["config"]))})
    m.load_state_dict(golden_state(keys, 0))
    m = m.to(dev)
    t = lambda k: torch.tensor(zi[k])  # noqa: E731
    il = t("text_lengths")
    feats = t("feats") + 0.25 * rank                                   # every rank trains on different data
    batch = dict(xs=t("text"), ilens=il, ys=feats, olens=t("feats_lengths"), durations=t("durations"), duration_lens=il, pitch=t("pitch") * (1 + rank),
                 pitch_lens=il, energys=t("energy"), energy_lens=il)
    tr = FastSpeech2Trainer(m, lr=1e-3, grad_norm=1.0, warmup_steps=0, bucket_bytes=256 << 10, overlap=overlap)     # small buckets: several collectives
    losses = [float(tr.train_step(batch)["loss"]) for _ in range(3)]
    p = tr.flat_p.detach().cpu()
    hi, lo = p.clone(), p.clone()
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    ls = [None] * world
    dist.all_gather_object(ls, losses)
    print(json.dumps({"rank": rank, "replica_spread": float((hi - lo).abs().max()), "losses": ls, "finite": bool(np.isfinite(losses).all()),
                      "checksum": float(p.double().abs().sum()), "buckets": len(tr._buckets) if tr._buckets else 0}),
          flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
