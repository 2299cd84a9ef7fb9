"""The RCCL side of the exchange step on the one GPU a test box has: a single-rank "nccl" process group exercises the same
calls (header all_gather_into_tensor, uint8 payload collectives, jatts_pcm16 conversion) that N ranks issue; the multi-rank
bookkeeping is covered on CPU (tests/test_distributed_cpu.py)."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_gather_audio_on_rccl_world1(cuda, lib):
    import torch.distributed as dist
    from jatts_amd.bin.tts_decode import to_pcm16
    from jatts_amd.distributed import gather_audio
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(cuda)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=cuda)
    try:
        g = torch.Generator().manual_seed(0)
        lens = [4096, 300, 77777]
        y = torch.tanh(torch.randn(sum(lens), generator=g)).to(cuda)
        got, lens_out = gather_audio(y, lens, max_utts=8)
        assert lens_out == [lens] and got[0].dtype == torch.int16 and got[0].numel() == sum(lens)
        assert torch.equal(got[0].cpu(), torch.from_numpy(to_pcm16(y.cpu().numpy())))      # jatts_pcm16 == the host conversion
        got, lens_out = gather_audio(y, lens, pcm16=False)                                 # f32 payload, max_utts by all-reduce
        assert torch.equal(got[0], y) and lens_out == [lens]
    finally:
        dist.destroy_process_group()


def test_data_parallel_training_world2(cuda, lib):
    """Two data-parallel FastSpeech2Trainer ranks (child processes sharing this box's GPU, gradient all-reduce over gloo) train on
    different data for three steps: the replicas stay bit-identical (same averaged gradients, same clip + Adam), the per-rank losses
    differ, everything is finite.  The N > 1 training path of SURVEY §8(e)/(f.4) end to end on the HIP kernels."""
    import json
    import subprocess
    import sys
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_dp_worker.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")

    def run(overlap):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        procs = [subprocess.Popen([sys.executable, worker, str(r), "2", str(port), "1" if overlap else "0"], stdout=subprocess.PIPE,
                                  stderr=subprocess.PIPE, env=env, text=True) for r in range(2)]
        outs = []
        for p in procs:
            o, e = p.communicate(timeout=300)
            assert p.returncode == 0, e[-2000:]
            outs.append(json.loads([ln for ln in o.splitlines() if ln.startswith("{")][-1]))
        for o in outs:
            assert o["finite"] and o["replica_spread"] == 0.0, o
        l0, l1 = outs[0]["losses"]
        assert l0 != l1 and outs[0]["losses"] == outs[1]["losses"]
        return outs[0]
    a = run(True)       # all-reduce issued per bucket from the post-accumulate hooks, overlapped with backward
    b = run(False)      # one pass over the flat gradient buffer after backward
    assert a["buckets"] >= 3 and b["buckets"] == 0
    # same sums, same averages -> the SAME training, bit for bit (round 4: the parameter-gradient reductions are fixed-order, csrc/det_reduce.h;
    # a two-rank sum is commutative, so the order the buckets travel in does not matter)
    assert a["losses"] == b["losses"], (a["losses"], b["losses"])
    assert a["checksum"] == b["checksum"]


@pytest.mark.parametrize("plain", [False, True], ids=["torchrun", "plain-python"])
def test_bench_two_ranks_on_the_shared_gpu(cuda, lib, plain):
    """bench.py's N > 1 path (the driver's `torch.distributed.run --nproc-per-node N bench.py --gpus N` launch: sharding by rank,
    barrier-bracketed timing, max over ranks, int16 PCM all-gather inside the step) with two ranks on this box's one GPU
    (JATTS_BENCH_SHARED_GPU=1: gloo instead of RCCL).  The line must report the whole job: n_gpus 2, twice one rank's samples."""
    import json
    import subprocess
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, JATTS_BENCH_SHARED_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    tail = [os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--batch", "8", "--no-fast-mode"]
    if plain:    # `python bench.py --gpus 2` with no launcher in the environment: bench.py starts its own two ranks as children
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        cmd = [sys.executable] + tail
    else:        # the driver's launch line
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port)] + tail
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                       # rank 0 prints ONE line
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["steps"] == 1 and j["cpu_baseline"] is None
    assert j["n_ranks_seen"] == 2 and j["backend"].startswith("gloo") and "N=1 only" in j["cpu_baseline_note"]
    per_rank = 8 * 128 * 6 * j["config"]["hop"]
    assert abs(j["value"] * j["ms_per_step"] / 1e3 - 2 * per_rank) <= 1e-4 * per_rank      # the line rounds to 5 significant digits
    assert j["stage_ms"]["audio_all_gather"] > 0.0 and "shared-GPU test mode" in j["config"]["parallelism"]
    assert 0.0 < j["rank_ms"]["min"] <= j["rank_ms"]["max"] <= j["ms_per_step"] * 1.001
    from test_bench_line_cpu import check_line
    check_line(lines[0], n_gpus=2)                                   # small enough for the driver, every graded key present
    assert r.stdout.rstrip().endswith(lines[0])                      # nothing printed after it
    detail = json.load(open(os.path.join(root, "bench_detail.json")))
    assert detail["n_gpus"] == 2 and "resunit_by_shape" in detail


def test_cli_two_ranks_equal_one_rank(cuda, lib, tmp_path):
    """`python -m jatts_amd.bin.tts_decode --n_gpus 2` with no launcher: the CLI starts its two ranks itself (children), each decodes
    its shard of a RAGGED csv (JATTS_SHARED_GPU=1: both on this box's one GPU; there is no collective, the outputs are files) --
    every wav exists exactly once and is bit-identical to the one-process run."""
    import subprocess
    import sys
    from test_cli import _make_expdir
    import csv as _csv
    d = tmp_path
    _make_expdir(d)
    tokens = (d / "tokens.txt").read_text().split("\n")[:-1]
    g = torch.Generator().manual_seed(5)
    lens = (7, 15, 11, 3, 22, 9, 1)
    with open(d / "ragged.csv", "w", newline="") as f:
        w = _csv.DictWriter(f, fieldnames=["sample_id", "phonemes"])
        w.writeheader()
        for i, n in enumerate(lens):
            w.writerow({"sample_id": f"r{i}", "phonemes": " ".join(tokens[int(j)] for j in torch.randint(2, 19, (n,), generator=g))})
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [sys.executable, "-m", "jatts_amd.bin.tts_decode", "--csv", str(d / "ragged.csv"), "--stats", str(d / "stats.npz"),
            "--token-list", str(d / "tokens.txt"), "--token-column", "phonemes", "--checkpoint", str(d / "checkpoint-1steps.pkl"),
            "--verbose", "0", "--batch-size", "3"]
    env = dict(os.environ, JATTS_SHARED_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    for n, out in ((1, "out1"), (2, "out2")):
        r = subprocess.run(base + ["--outdir", str(d / out), "--n_gpus", str(n)], capture_output=True, text=True, timeout=600, env=env, cwd=root)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    names = sorted(os.listdir(d / "out1" / "wav"))
    assert names == sorted(f"r{i}.wav" for i in range(len(lens))) == sorted(os.listdir(d / "out2" / "wav"))
    for nm in names:
        a, b = (d / "out1" / "wav" / nm).read_bytes(), (d / "out2" / "wav" / nm).read_bytes()
        assert len(a) > 44 and a == b, nm
