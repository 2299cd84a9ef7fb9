set -x
O=gpurun_out
python -m pytest tests/test_emul_gpu.py tests/test_spk_concat_gpu.py -q 2>&1 | tail -25 > $O/r05_t_emul.txt
python -m pytest tests/test_cli.py tests/test_fullsize_gpu.py -q -k "recipe or full_batch" 2>&1 | tail -15 > $O/r05_t_misc.txt
python tools/emul_sweep.py --units 400 --convs 700 --out $O/r05_emul_sweep.json > $O/r05_emul_sweep.txt 2>&1
# the slow 2048 -> 512 k = 1 shape: which part
python - > $O/r05_conv_diag.txt 2>&1 <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
from jatts_amd import hip
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
def run(c, n, k, T, res, dt, B=64):
    rb = hip.RaggedBatch([T] * B, dev); rows = rb.total
    x = (torch.randn(rows, c, generator=g) * 0.5).to(dev)
    wf = (torch.randn(n, c, k, generator=g) / (c * k) ** 0.5).to(dev)
    w = hip.pack_conv_weight_bf16x3(wf, 64) if dt == hip.F32E else hip.pack_conv_weight(wf, dt)
    b = torch.zeros(n, device=dev); r = torch.zeros(rows, n, device=dev) if res else None
    f = lambda: hip.conv1d(rb, x, w, c, n, k, dtype=dt, bias=b, resid=r, out=r, out_f32=True)
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{'emul' if dt == hip.F32E else 'f32'} {c}->{n} k{k} rows={rows} res={res}: {ms*1e3:8.1f} us {2.0*c*n*k*rows/ms/1e9:7.1f} TFLOP/s", flush=True)
for c, n, res in ((2048, 512, 1), (2048, 512, 0), (2048, 1024, 0), (1024, 512, 1), (1536, 512, 1), (2048, 384, 1), (2048 + 64, 512, 1), (4096, 512, 0), (2048, 256, 1), (2048, 128, 1)):
    run(c, n, 1, 768, res, hip.F32E)
run(2048, 512, 1, 768, 1, hip.F32E, B=32)
run(2048, 512, 3, 768, 1, hip.F32E)
PY
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train --no-configs --no-pmc --no-fast-mode > $O/r05_bench_quick2.json 2> $O/r05_bench_quick2.err
JATTS_RAGGED_1D=0 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train --no-configs --no-pmc --no-fast-mode --no-detail > $O/r05_bench_quick2_rect.json 2>> $O/r05_bench_quick2.err
tail -n 3 $O/r05_t_emul.txt; tail -n 3 $O/r05_t_misc.txt
