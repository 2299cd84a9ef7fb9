"""Tensor-level wrappers over the C ABI (jatts_amd/_abi.py).

PyTorch is used here only as plumbing: device memory (torch tensors), the current
HIP stream and dtype bookkeeping.  Every wrapper validates shapes, fills the POD
descriptor and launches asynchronously on ``torch.cuda.current_stream()``.
"""
import collections
import contextlib
import ctypes as C
import os

import torch

from . import _abi
from ._abi import ACT_MISH, ACT_NONE, ACT_RELU, ACT_SWISH, ACT_TANH, F16, F32, F32E, F32E6, F32S  # noqa: F401

_TORCH = {F32: torch.float32, F16: torch.float16, F32S: torch.float32, F32E: torch.float32, F32E6: torch.float32}
EMUL = (F32E, F32E6)      # f32 tensors, three exact bf16 terms per operand: seven / six partial products per product (include/jatts_hip.h)


def torch_dtype(code):
    return _TORCH[code]


def code_of(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.float16:
        return F16
    raise TypeError(f"unsupported dtype {t.dtype}")


# ---- optional live kernel timing (HIP events on the launch stream); used by bench.py only
_PROF = None


def profile_begin():
    global _PROF
    _PROF = []


def profile_end():
    """-> list of (tag, meta, milliseconds); synchronises the device."""
    global _PROF
    recs, _PROF = _PROF or [], None
    torch.cuda.synchronize()
    return [(tag, meta, a.elapsed_time(b)) for tag, meta, a, b in recs]


# ---- optional dense-FLOP tally of the MFMA launches (conv forward / dgrad / wgrad, attention); bench.py's training lines
_FLOPS = None


def flops_begin():
    global _FLOPS
    _FLOPS = 0.0


def flops_end():
    global _FLOPS
    v, _FLOPS = _FLOPS or 0.0, None
    return v


def _count(flops):
    global _FLOPS
    if _FLOPS is not None:
        _FLOPS += flops


# ---- zero pool for the training step: the small accumulators the reduction kernels add into (column sums, LayerNorm / GroupNorm
# parameter gradients, per-sequence sums ...) used to be one torch.zeros() each = one fill launch each (~550 of a FastSpeech2 step's
# 2 960 launches).  A trainer opens the pool at the top of a step: ONE fill zeroes the region the previous step used, and every
# accumulator is a 64-byte-aligned slice of it; slices are never handed out twice within a step, and nothing taken from the pool
# outlives the step (gradients are accumulated into the flat gradient buffer, saved statistics die with the graph).
class _ZeroPool:
    def __init__(self, device, floats=1 << 20):
        self.buf = torch.zeros(floats, dtype=torch.float32, device=device)
        self.off, self.high, self.misses = 0, 0, 0
        self.active = False
        self._retired = []        # outgrown buffers stay alive: captured graphs may still write into them

    def begin(self):
        """Zero everything any step has used so far (the high-water mark: a few hundred KB) and start handing out from the front.  Never
        allocates: safe inside a graph capture."""
        self.high = max(self.high, self.off)
        if self.high:
            self.buf[:self.high].zero_()
        self.off = 0

    def end(self):
        self.high = max(self.high, self.off)
        if self.misses:           # the step ran out (its overflow used torch.zeros): grow now, outside any capture
            self._retired.append(self.buf)
            self.buf = torch.zeros(max(2 * self.buf.numel(), 2 * (self.high + self.misses)), dtype=torch.float32, device=self.buf.device)
            self.misses = 0

    def take(self, numel):
        n = (numel + 15) & ~15
        if self.off + n > self.buf.numel():
            self.misses += n
            return None
        t = self.buf[self.off:self.off + numel]
        self.off += n
        return t


_ZPOOL = None


def zero_pool_begin(device):
    """Open (or re-arm) the device's zero pool: call at the top of a training step.  Pair with zero_pool_end()."""
    global _ZPOOL
    dev = torch.device(device)
    if dev.index is None:
        dev = torch.device(dev.type, torch.cuda.current_device())
    if _ZPOOL is None or _ZPOOL.buf.device != dev:
        _ZPOOL = _ZeroPool(dev)
    _ZPOOL.begin()
    _ZPOOL.active = True


def zero_pool_end():
    if _ZPOOL is not None:
        _ZPOOL.active = False
        if not torch.cuda.is_current_stream_capturing():
            _ZPOOL.end()


def _zeros(shape, device):
    """Zero-initialised f32 accumulator: a slice of the step's zero pool when one is open on ``device``, else torch.zeros."""
    shape = (shape,) if isinstance(shape, int) else tuple(shape)
    if _ZPOOL is not None and getattr(_ZPOOL, "active", False) and _ZPOOL.buf.device == device:
        n = 1
        for v in shape:
            n *= v
        t = _ZPOOL.take(n)
        if t is not None:
            return t.view(shape)
    return torch.zeros(shape, dtype=torch.float32, device=device)


class _Timed:
    def __init__(self, tag, meta):
        self.tag, self.meta = tag, meta

    def __enter__(self):
        if _PROF is not None:
            self.a = torch.cuda.Event(enable_timing=True)
            self.a.record()

    def __exit__(self, *exc):
        if _PROF is not None:
            b = torch.cuda.Event(enable_timing=True)
            b.record()
            _PROF.append((self.tag, self.meta, self.a, b))


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


_WS = {}               # device -> the zero-initialised scratch of the deterministic reductions (include/jatts_hip.h: jatts_set_workspace)
_WS_CURRENT = [None]
_WS_OWNER = {}         # device -> (stream handle, event recorded behind that stream's last reduction launch)
WS_BYTES = 64 * 1024 + 96 * 1024 * 1024


def _ws(device):
    """(Graph replays: a captured step bakes the scratch in and is ordered against other users only on the stream it was CAPTURED on -- replay a trainer's
    graph on its capture stream, which is what jatts_amd.training does; ownership is not tracked for replays on other streams.)
    Register this device's scratch with the library before a kernel that reduces across workgroups (LayerNorm / GroupNorm / SnakeBeta
    parameter gradients, column sums, depthwise-conv weights, the gradient norm): allocated once, outside any graph capture (the trainers'
    first step of a signature runs eagerly), kept alive for the life of the process -- a captured step bakes its address in."""
    key = str(device)
    # ONE scratch (tickets + slabs) per device: two streams reducing concurrently would corrupt each other's tickets silently.
    # The stream that used it last owns it; another stream first waits for that stream's last reduction (an event, no host sync).  During a
    # graph capture the scratch belongs to the capturing stream for the graph's whole life (the trainers capture on one stream).
    if not torch.cuda.is_current_stream_capturing():
        cur = torch.cuda.current_stream(device)
        own = _WS_OWNER.get(key)
        if own is not None and own[0] != cur.cuda_stream and not own[1].query():
            cur.wait_event(own[1])
        ev = own[1] if (own is not None and own[0] == cur.cuda_stream) else torch.cuda.Event()
        _WS_OWNER[key] = (cur.cuda_stream, ev)
        _WS_PENDING[0] = (cur, ev)
    if _WS_CURRENT[0] == key:
        return
    buf = _WS.get(key)
    if buf is None:
        if torch.cuda.is_current_stream_capturing():
            raise _abi.JattsHipError("the reduction scratch must exist before a graph capture: run one eager step first")
        buf = _WS[key] = torch.zeros(WS_BYTES, dtype=torch.uint8, device=device)
        torch.cuda.current_stream(device).synchronize()
    _abi.check(_abi.load().jatts_set_workspace(buf.data_ptr(), buf.numel()), "jatts_set_workspace")
    _WS_CURRENT[0] = key


_WS_PENDING = [None]


def _ws_done():
    """Behind a reduction launch: mark how far the owning stream has got with the scratch (what another stream would wait for)."""
    p, _WS_PENDING[0] = _WS_PENDING[0], None
    if p is not None:
        p[1].record(p[0])


def _ws_check(rc, name):
    try:
        _abi.check(rc, name)
    finally:          # a failed launch must not leave the pending marker behind: the next reduction would record its event on the wrong stream
        _ws_done()


def _ptr(t, col0=0):
    if t is None:
        return None
    return t.data_ptr() + col0 * t.element_size()


def _dev(t):
    if not t.is_cuda:
        raise _abi.JattsHipError("jatts_amd kernels need device tensors (no CPU fallback)")
    return t


_KEEP = [None]        # while a training step is being captured: every cached device tensor handed out (the graph bakes their addresses in)


def keep_begin():
    """Start recording the cached device tensors handed out from here on (hip.h2d, RaggedBatch geometry, the models' per-length tables):
    a captured graph reads them at their addresses for as long as it is replayed, while the caches are bounded and evict -- the capture's
    owner keeps the returned list alive next to the graph (jatts_amd.training)."""
    _KEEP[0] = []
    return _KEEP[0]


def keep_end():
    _KEEP[0] = None


def keep(t):
    """Called by every cache that hands out a device tensor (hit or miss); returns t."""
    if _KEEP[0] is not None and t is not None:
        _KEEP[0].append(t)
    return t


def h2d(values, dtype, device):
    """Small host list -> device tensor WITHOUT stalling the host: staged through pinned memory and copied asynchronously on the
    current stream.  torch.tensor(values, device=...) copies from pageable memory, which blocks the host until everything already
    queued on the stream has run -- one pipeline drain per call in the middle of a forward pass."""
    dev = torch.device(device)
    if dev.type != "cuda":
        return torch.tensor(values, dtype=dtype, device=dev)
    # small length-derived arrays repeat from step to step (same bucket of lengths): cached per (values, dtype, device), which also makes
    # a training step capturable -- a cache hit issues no copy, so a captured graph never holds a memcpy from a host buffer that is gone.
    # LRU, bounded by BYTES (device bytes + the key tuple's ~8 x that on the host): a long decode run produces a new key per batch.
    key = None
    if isinstance(values, (list, tuple)) and len(values) <= 65536 and all(isinstance(v, (int, float, bool)) for v in values[:1] + values[-1:]):
        key = (tuple(values), dtype, dev.index if dev.index is not None else torch.cuda.current_device())
        hit = _H2D_CACHE.get(key)
        if hit is not None:
            _H2D_CACHE.move_to_end(key)
            if hit[2] != torch.cuda.current_stream().cuda_stream and not torch.cuda.is_current_stream_capturing() and not hit[1].query():
                torch.cuda.current_stream().wait_event(hit[1])
            return keep(hit[0])
    if torch.cuda.is_current_stream_capturing():
        raise _abi.JattsHipError("h2d of new values while a graph is being captured: run the step once outside the capture first")
    t = torch.tensor(values, dtype=dtype).pin_memory().to(dev, non_blocking=True)
    if key is not None:
        ev = torch.cuda.Event()
        ev.record()
        _H2D_CACHE[key] = (t, ev, torch.cuda.current_stream().cuda_stream)
        _H2D_BYTES[0] += t.numel() * t.element_size()
        while _H2D_BYTES[0] > _H2D_CACHE_MAX_BYTES and len(_H2D_CACHE) > 1:
            _, old = _H2D_CACHE.popitem(last=False)
            _H2D_BYTES[0] -= old[0].numel() * old[0].element_size()
    return keep(t)


_H2D_CACHE = collections.OrderedDict()       # (values, dtype, device) -> (tensor, upload-complete event, stream); least recently used first
_H2D_BYTES = [0]
_H2D_CACHE_MAX_BYTES = 8 << 20               # device bytes; the host-side key tuples cost about 8x that
_GEOM_CACHE = {}      # (lens, device) -> (cu tensor, upload-complete event); insertion-ordered, oldest evicted
_GEOM_CACHE_MAX = 512


class RaggedBatch:
    """Packed ragged batch geometry: sequence b owns rows cu[b]..cu[b+1]-1.

    The device copy of ``cu`` is cached per (lengths, device): a forward / training step builds the same handful of geometries
    (``[T] * B`` padded forms, the valid-length forms) dozens of times, and every upload is a launch plus a pinned staging copy."""

    def __init__(self, lens, device):
        self.lens = [int(v) for v in lens]
        cu = [0]
        for v in self.lens:
            cu.append(cu[-1] + v)
        self.cu_host = cu
        self.total = cu[-1]
        self.n_seq = len(self.lens)
        self.max_len = max(self.lens) if self.lens else 0
        self.device = device
        dev = torch.device(device)
        if dev.type != "cuda":
            self.cu = torch.tensor(cu, dtype=torch.int32, device=dev)
            return
        key = (tuple(self.lens), dev.index if dev.index is not None else torch.cuda.current_device())
        hit = _GEOM_CACHE.get(key)
        if hit is None:
            t = h2d(cu, torch.int32, dev)
            ev = torch.cuda.Event()
            ev.record()
            if len(_GEOM_CACHE) >= _GEOM_CACHE_MAX:
                _GEOM_CACHE.pop(next(iter(_GEOM_CACHE)))
            _GEOM_CACHE[key] = hit = (t, ev, torch.cuda.current_stream().cuda_stream)
        elif hit[2] != torch.cuda.current_stream().cuda_stream and not torch.cuda.is_current_stream_capturing() and not hit[1].query():
            torch.cuda.current_stream().wait_event(hit[1])     # uploaded on another stream and still in flight
        self.cu = keep(hit[0])

    def struct(self, len_mul=1):
        if _RAGGED_1D and self.n_seq > 1 and min(self.lens) != self.max_len:     # a ragged batch: 1-D grids over its real tiles (jatts_ragged.host_lens)
            if not hasattr(self, "_lens_c"):
                self._lens_c = (C.c_int32 * self.n_seq)(*self.lens)               # (kept alive with the batch; read by the launchers only)
            return _abi.Ragged(self.cu.data_ptr(), self.n_seq, self.max_len, len_mul, self.total, C.addressof(self._lens_c))
        return _abi.Ragged(self.cu.data_ptr(), self.n_seq, self.max_len, len_mul, 0, None)

    def vt_layout(self):
        """(col0 int32 device tensor, ld): V^T column layout with every sequence starting on a multiple of 8 columns
        and 8 columns of slack at the end (jatts_relattn_desc.vt_col0)."""
        if not hasattr(self, "_vt"):
            col, c = [], 0
            for v in self.lens:
                col.append(c)
                c += round_up(v, 8)
            self._vt = (h2d(col or [0], torch.int32, self.device), c + 8)
        return self._vt


# 1-D grids over the real tiles of a ragged batch (jatts_ragged.total_rows); JATTS_RAGGED_1D=0: the rectangular grids (A/B runs)
_RAGGED_1D = os.environ.get("JATTS_RAGGED_1D", "1") != "0"


def round_up(v, m):
    return (v + m - 1) // m * m


def pack_conv_weight(w, dtype_code, c_mult=64):
    """(n_out, c_in, k) -> MFMA fragment order [tap][c/16][n/32][lane][8] (include/jatts_hip.h).

    c_in is zero-padded to a multiple of ``c_mult`` (64 for jatts_conv1d, whose LDS chunks are 64
    channels; 32 for the fused HiFi-GAN unit, which takes c_in == channels) and n_out to a
    multiple of 32.  Pure permutation + cast: done once per checkpoint at prepare time.
    """
    n, c, k = w.shape
    n_pad, c_pad = round_up(n, 32), round_up(c, c_mult)
    wp = torch.zeros(n_pad, c_pad, k, dtype=torch.float32, device=w.device)
    wp[:n, :c] = w.float()
    wp = wp.permute(2, 0, 1).reshape(k, n_pad // 32, 32, c_pad // 16, 2, 8)
    wp = wp.permute(0, 3, 1, 4, 2, 5).contiguous().reshape(-1)
    return wp.to(torch_dtype(dtype_code))


def pack_conv_weight_split(w, c_mult=32):
    """(n_out, c_in, k) f32 -> the JATTS_F32S operand of the fused HiFi-GAN unit: per output channel n a power-of-two scale
    2^s[n] puts max |w[n]| in [2^14, 2^15) (so that the lo halves stay normal f16 numbers over 18 binary orders of magnitude
    below the channel's largest weight); ws = w * 2^s[n] travels as hi = f16(ws), lo = f16(ws - hi) in the fragment order of
    pack_conv_weight with the two halves of a lane's 8 elements side by side: [tap][c/16][n/32][lane][hi x8 | lo x8].
    -> (packed f16 tensor, inverse scales 2^-s[n] as f32 (n_pad,)).  Pure scaling by powers of two, one rounding per half."""
    n, c, k = w.shape
    w = w.detach().float()
    n_pad = round_up(n, 32)
    amax = w.abs().reshape(n, -1).amax(dim=1)
    e = torch.frexp(amax)[1]                               # amax = m 2^e, m in [0.5, 1)
    s = torch.where(amax > 0, 15 - e, torch.zeros_like(e)).clamp(-60, 60).to(torch.int32)
    # 2^s and 2^-s assembled from their bit patterns: torch.ldexp goes through pow() on the GPU and is an ulp off for some exponents -- the
    # scales must be EXACT powers of two (tests/test_training_gpu.py::test_split_weight_packer_on_the_device_equals_the_host_packing)
    scale = ((127 + s) << 23).view(torch.float32)
    ws = w * scale.view(-1, 1, 1)
    hi = ws.half()
    lo = (ws - hi.float()).half()
    ph = pack_conv_weight(hi.float(), F16, c_mult).view(-1, 8)
    pl = pack_conv_weight(lo.float(), F16, c_mult).view(-1, 8)
    inv = torch.ones(n_pad, dtype=torch.float32, device=w.device)
    inv[:n] = ((127 - s) << 23).view(torch.float32)
    return torch.stack([ph, pl], dim=1).reshape(-1).contiguous(), inv.contiguous()


def bf16x3_terms(w):
    """f32 tensor -> (b0, b1, b2) bfloat16 tensors with b0 + b1 + b2 == w exactly (round-to-nearest-even at every step; both
    differences are exact in f32): 3 x 8 significand bits = f32's 24, f32's exponent range.  Holds for every finite value whose last
    bit lies at or above bf16's smallest subnormal (|w| >= 2^-110)."""
    w = w.detach().float()
    b0 = w.bfloat16()
    r1 = w - b0.float()
    b1 = r1.bfloat16()
    b2 = (r1 - b1.float()).bfloat16()
    return b0, b1, b2


def pack_conv_weight_bf16x3(w, c_mult=32):
    """(n_out, c_in, k) f32 -> the JATTS_F32E / JATTS_F32E6 operand: the three bf16 terms of every weight (bf16x3_terms: exact, no scales) in the
    fragment order of pack_conv_weight with a lane's 8 elements of each term side by side: [tap][c/16][n/32][lane][b0 x8 | b1 x8 | b2 x8]."""
    planes = [pack_conv_weight(t.float(), F32, c_mult).view(-1, 8).bfloat16() for t in bf16x3_terms(w)]   # bf16 -> f32 -> bf16 is exact
    return torch.stack(planes, dim=1).reshape(-1).contiguous()


def pack_unit_weight_bf16x3_k32(w):
    """(C, C, k) f32 -> the JATTS_F32E / JATTS_F32E6 fused-unit operand in the fragment order of the v_mfma_f32_16x16x32_bf16 kernels
    (jatts_resunit_desc.w_layout = 1, csrc/resunit_emul16_impl.h): [tap][c / 32][n / 16][lane = 16 ((c % 32) / 8) + n % 16][b0 x8 | b1 x8 | b2 x8] over c % 8."""
    n_out, c_in, k = w.shape
    if n_out % 16 or c_in % 32:
        raise ValueError("pack_unit_weight_bf16x3_k32: channels must be a multiple of 32")
    planes = []
    for t in bf16x3_terms(w):
        # (n, c, k) -> [k][c / 32][c % 32 / 8][c % 8][n / 16][n % 16] -> [k][c / 32][n / 16][c % 32 / 8][n % 16][c % 8]
        v = t.permute(2, 1, 0).reshape(k, c_in // 32, 4, 8, n_out // 16, 16).permute(0, 1, 4, 2, 5, 3)
        planes.append(v.reshape(-1, 8))
    return torch.stack(planes, dim=1).reshape(-1).contiguous()


def pack_conv_weight_bf16x3_k32(w, c_mult=64):
    """(n_out, c_in, k) f32 -> the JATTS_F32E / JATTS_F32E6 conv operand in the fragment order of the v_mfma_f32_16x16x32_bf16 kernels (jatts_conv_desc.w_layout = 1,
    csrc/conv1d_emul16.h): output channels zero-padded to 32, input channels to c_mult, then pack_unit_weight_bf16x3_k32's order."""
    n, c, k = w.shape
    n_pad, c_pad = round_up(n, 32), round_up(c, c_mult)
    wp = torch.zeros(n_pad, c_pad, k, dtype=torch.float32, device=w.device)
    wp[:n, :c] = w.detach().float()
    return pack_unit_weight_bf16x3_k32(wp)


# fragment order of the emulated convs' weights: "16" = the v_mfma_f32_16x16x32_bf16 kernels (round 6), "32" = the round-5 32 x 32 x 16 kernels (A/B runs)
CONV_EMUL_FORM = os.environ.get("JATTS_CONV_EMUL_FORM", "16")


class SplitWeight:
    """A conv weight prepared for JATTS_F32S (the pair of pack_conv_weight_split).  hip.conv1d recognises it in place of a packed f32
    weight -- call sites stay `dtype=hip.F32` -- and takes the split kernel; shapes the split kernel does not cover (a halo beyond 32
    rows) fall back, loudly never silently wrong, to the exact-f32 kernel on a lazily packed f32 copy of the same weight."""

    def __init__(self, w, c_mult=64):
        self.packed, self.inv = pack_conv_weight_split(w, c_mult)
        self._src, self._c_mult, self._f32 = w.detach(), c_mult, None

    def f32(self):
        if self._f32 is None:
            self._f32 = pack_conv_weight(self._src, F32, self._c_mult)
        return self._f32


class EmulWeight:
    """A conv weight prepared for JATTS_F32E / JATTS_F32E6 (pack_conv_weight_bf16x3: the three bf16 terms of every weight, exact; ``code`` picks
    seven or six partial products).  hip.conv1d recognises it in place of a packed f32 weight -- call sites stay `dtype=hip.F32` -- and takes the
    emulated kernel; a halo beyond its staging registers (32 rows) takes the exact-f32 kernel on a lazily packed f32 copy of the same weight."""

    def __init__(self, w, c_mult=64, code=F32E, layout=None):
        self.code = code
        self.layout = (1 if CONV_EMUL_FORM == "16" and c_mult % 64 == 0 else 0) if layout is None else layout      # jatts_conv_desc.w_layout
        self.packed = pack_conv_weight_bf16x3_k32(w, c_mult) if self.layout else pack_conv_weight_bf16x3(w, c_mult)
        self._src, self._c_mult, self._f32 = w.detach(), c_mult, None

    def f32(self):
        if self._f32 is None:
            self._f32 = pack_conv_weight(self._src, F32, self._c_mult)
        return self._f32


_SPLIT_WEIGHTS = [0]     # 0: packed f32 weights; 1: SplitWeight (fp32_split); 2 / 3: EmulWeight, seven / six products (fp32_bf16x3 / fp32_bf16x3_6p)
WEIGHT_MODE = {"fp16": 0, "fp32": 0, "fp32_split": 1, "fp32_bf16x3": 2, "fp32_bf16x3_6p": 3}
EMUL_CODE = {2: F32E, 3: F32E6}
PRECISIONS = tuple(WEIGHT_MODE)


@contextlib.contextmanager
def split_weights(on=True):
    """Inside: PackedConv(..., dtype=F32) packs SplitWeight operands (the models' set_precision("fp32_split")) or, with on == 2 / 3
    ("fp32_bf16x3" / "fp32_bf16x3_6p"), EmulWeight operands.  Accepts a bool, a mode number or a precision name.  Nests."""
    mode = WEIGHT_MODE[on] if isinstance(on, str) else int(on)
    prev, _SPLIT_WEIGHTS[0] = _SPLIT_WEIGHTS[0], mode
    try:
        yield
    finally:
        _SPLIT_WEIGHTS[0] = prev


def f32_operand(w, c_mult=64):
    """A conv weight for `dtype=F32` call sites in the current weight mode: packed exact f32, SplitWeight or EmulWeight."""
    m = _SPLIT_WEIGHTS[0]
    return SplitWeight(w, c_mult) if m == 1 else EmulWeight(w, c_mult, EMUL_CODE[m]) if m in EMUL_CODE else pack_conv_weight(w, F32, c_mult)


def pack_conv_weight_dev(w, dtype_code, c_mult=64, dgrad=False):
    """pack_conv_weight as ONE HIP launch on a device f32 weight (n_out, c_in, k); dgrad=True packs the data-gradient operand
    W'[c][n][k-1-tap] directly (no permute / flip copies).  -> (packed, padded c_in of the packed conv)."""
    lib = _abi.load()
    w = _dev(w)
    if w.dtype != torch.float32 or not w.is_contiguous():
        w = w.float().contiguous()
    n, c, k = w.shape
    pn, pc = (c, n) if dgrad else (n, c)
    n_pad, c_pad = round_up(pn, 32), round_up(pc, c_mult)
    out = torch.empty(k * n_pad * c_pad, dtype=torch_dtype(dtype_code), device=w.device)
    _abi.check(lib.jatts_pack_conv_weight(w.data_ptr(), n, c, k, c_mult, int(dgrad), dtype_code, out.data_ptr(), _stream()),
               "jatts_pack_conv_weight")
    return out, c_pad


def pack_conv_weight_split_dev(w, c_mult=64, dgrad=False):
    """pack_conv_weight_split on the device (two launches: per-row maxima, pack), for weights that change every step (training).
    -> (packed f16, inverse scales f32 (n_pad,), padded c_in of the packed conv)."""
    lib = _abi.load()
    w = _dev(w)
    if w.dtype != torch.float32 or not w.is_contiguous():
        w = w.float().contiguous()
    n, c, k = w.shape
    pn, pc = (c, n) if dgrad else (n, c)
    n_pad, c_pad = round_up(pn, 32), round_up(pc, c_mult)
    out = torch.empty(2 * k * n_pad * c_pad, dtype=torch.float16, device=w.device)
    inv = torch.empty(n_pad, dtype=torch.float32, device=w.device)
    _abi.check(lib.jatts_pack_conv_weight_split(w.data_ptr(), n, c, k, c_mult, int(dgrad), out.data_ptr(), inv.data_ptr(), _stream()),
               "jatts_pack_conv_weight_split")
    return out, inv, c_pad


def convtranspose_as_conv(w, stride, padding):
    """ConvTranspose1d weight (c_in, c_out, K) -> polyphase Conv1d weight
    (stride*c_out, c_in, taps) + input offset `pad`, such that the conv output row j,
    viewed as [stride][c_out], equals output steps j*stride .. j*stride+stride-1."""
    c_in, c_out, K = w.shape
    qs = [(r, q) for r in range(stride) for q in range(-K, K + 1) if 0 <= stride * q + r + padding < K]
    qmin, qmax = min(q for _, q in qs), max(q for _, q in qs)
    taps = qmax - qmin + 1
    wc = torch.zeros(stride * c_out, c_in, taps, dtype=w.dtype, device=w.device)
    for r, q in qs:
        kk = stride * q + r + padding
        wc[r * c_out:(r + 1) * c_out, :, qmax - q] = w[:, :, kk].t()
    return wc, qmax


def conv1d(rb, xs, w_packed, c_in, n_out, k_w, *, dtype, dil=1, pad=None, bias=None, act=ACT_NONE,
           alpha=1.0, resid=None, out=None, out_f32=False, transposed=False, pre_lrelu=None,
           in_scale=1.0, ldx=None, x_col0=0, len_mul=1, out_ld=None, out_col0=0, out_rows=None, resid_col0=0,
           y_seq_col0=None, reflect=False, variant=0, w_inv=None, snake=None, split=None, w_layout=0):
    """See jatts_conv1d in include/jatts_hip.h.  ``xs`` is a tensor or list of <=3 tensors.  dtype F32S: f32 tensors, ``w_packed`` / ``w_inv``
    from pack_conv_weight_split (c_mult 64).  ``snake`` = (exp(alpha), 1 / (exp(beta) + 1e-9)) f32 vectors of n_out: the SnakeBeta
    activation (the ``snakebeta`` op) applied in the epilogue instead of ``act``.  ``split`` = (n_split, ld2, seq_col0): TWO outputs from one launch
    (the Q | K | V projection): channels < n_split row-major as usual, channels >= n_split transposed into a (n_out - n_split, ld2) matrix with the V^T
    column layout ``seq_col0`` (RaggedBatch.vt_layout); returns (out, out2)."""
    lib = _abi.load()
    if isinstance(xs, torch.Tensor):
        xs = [xs]
    if isinstance(w_packed, SplitWeight):
        if dtype != F32:
            raise ValueError("conv1d: a SplitWeight goes with dtype F32 tensors")
        if (k_w - 1) * dil <= 32:
            dtype, w_inv, w_packed, out_f32 = F32S, w_packed.inv, w_packed.packed, True
        else:                                  # outside the split kernel's tiles: the exact-f32 kernel on the same weight
            w_packed = w_packed.f32()
    elif isinstance(w_packed, EmulWeight):
        if dtype != F32:
            raise ValueError("conv1d: an EmulWeight goes with dtype F32 tensors")
        if (k_w - 1) * dil <= 32:
            dtype, w_layout, w_packed, out_f32 = w_packed.code, w_packed.layout, w_packed.packed, True
        else:
            w_packed = w_packed.f32()
    x0 = _dev(xs[0])
    rows = rb.total * len_mul
    tdt = torch_dtype(dtype)
    for x in xs:
        if x.dtype != tdt or not x.is_contiguous():
            raise ValueError(f"conv1d: inputs must be contiguous {tdt}")
    ldx = ldx if ldx is not None else x0.shape[-1]
    if x0.numel() < rows * ldx or max(x_col0 if isinstance(x_col0, (list, tuple)) else [x_col0]) + c_in > ldx:
        raise ValueError("conv1d: input too small for the ragged geometry")
    if pad is None:
        pad = (k_w - 1) // 2 * dil
    n_pad = round_up(n_out, 32)
    if dtype == F32S:
        if w_packed.dtype != torch.float16 or w_packed.numel() != 2 * n_pad * c_in * k_w or w_inv is None or w_inv.numel() != n_pad:
            raise ValueError("conv1d: F32S takes the (packed, inverse scales) pair of pack_conv_weight_split")
    elif dtype in EMUL:
        if w_packed.dtype != torch.bfloat16 or w_packed.numel() != 3 * n_pad * c_in * k_w:
            raise ValueError("conv1d: F32E / F32E6 take the packed bf16 terms of pack_conv_weight_bf16x3")
        out_f32 = True
    elif w_packed.dtype != tdt or w_packed.numel() != n_pad * c_in * k_w:
        raise ValueError("conv1d: packed weight has wrong dtype/size")
    odt = torch.float32 if out_f32 else tdt
    out2 = None
    if split is not None:
        n_split, ld2, col2 = split
        if transposed or out is not None or resid is not None or n_split % 256 or not 0 < n_split < n_out:
            raise ValueError("conv1d: split takes a fresh row-major output, no residual, n_split a multiple of 256 below n_out")
        out = torch.empty(rows, out_ld or n_split, dtype=odt, device=x0.device)
        out2 = torch.empty(n_out - n_split, ld2, dtype=odt, device=x0.device)
    if out is None:
        if transposed:
            # slack columns of an aligned V^T layout stay uninitialised: the attention kernel masks every element
            # outside [0, T) after loading it (tests fill them with NaN)
            out = torch.empty(n_out, rows if out_ld is None else out_ld, dtype=odt, device=x0.device)
        else:
            out = torch.empty(rows if out_rows is None else out_rows, out_ld or n_out, dtype=odt, device=x0.device)
    if out.dtype != odt:
        raise ValueError("conv1d: output dtype mismatch")
    ldy = out.shape[-1] if out_ld is None else out_ld
    d = _abi.ConvDesc()
    d.rg = rb.struct(len_mul)
    d.dtype, d.n_in = dtype, len(xs)
    cols = x_col0 if isinstance(x_col0, (list, tuple)) else [x_col0] * len(xs)   # per-input first column (views into wider rows)
    for i, x in enumerate(xs):
        d.x[i] = _ptr(x, cols[i])
    d.pad_mode = _abi.PAD_REFLECT if reflect else _abi.PAD_ZERO
    d.variant = variant
    d.ldx, d.in_scale = ldx, in_scale
    d.pre_act = _abi.PRE_LRELU if pre_lrelu is not None else _abi.PRE_NONE
    d.pre_slope = pre_lrelu or 0.0
    d.w, d.c_in, d.n_out, d.k_w, d.dil, d.pad = w_packed.data_ptr(), c_in, n_out, k_w, dil, pad
    d.bias = _ptr(bias)
    d.act, d.alpha = act, alpha
    if snake is not None:
        if act != ACT_NONE or any(v.dtype != torch.float32 or v.numel() != n_out or not v.is_contiguous() for v in snake):
            raise ValueError("conv1d: snake takes two contiguous f32 vectors of n_out and replaces act")
        d.act, d.act_a, d.act_b = _abi.ACT_SNAKEBETA, snake[0].data_ptr(), snake[1].data_ptr()
    if resid is not None:
        if resid.dtype != torch.float32:
            raise ValueError("conv1d: residual must be f32")
        d.resid, d.ldr = _ptr(resid, resid_col0), resid.shape[-1]
    d.y, d.ldy = _ptr(out, out_col0), ldy
    d.y_is_f32, d.y_transposed = int(odt == torch.float32), int(transposed)
    d.y_seq_col0 = _ptr(y_seq_col0) if transposed else None
    d.w_inv = _ptr(w_inv) if dtype == F32S else None
    d.w_layout = w_layout if dtype in EMUL else 0
    if out2 is not None:
        d.n_split, d.ldy2, d.y2, d.y2_seq_col0 = split[0], split[1], out2.data_ptr(), _ptr(split[2])
    _count(2.0 * c_in * n_out * k_w * rows)
    with _Timed("conv1d", (c_in, n_out, k_w, rows)):
        _abi.check(lib.jatts_conv1d(C.byref(d), _stream()), "jatts_conv1d")
    return out if out2 is None else (out, out2)


def _check_unit_weights(who, dtype, channels, k_w, *ws):
    """The packed operand a fused-unit dtype expects, element for element: the C entry point sees only a pointer, and an f32-packed weight (4 B per
    element) handed to an emulated kernel (three bf16 planes: 6 B) would be read out of bounds on the device."""
    n = channels * channels * k_w
    want = {F32: (torch.float32, n), F16: (torch.float16, n), F32S: (torch.float16, 2 * n), F32E: (torch.bfloat16, 3 * n), F32E6: (torch.bfloat16, 3 * n)}[dtype]
    for w in ws:
        if w.dtype != want[0] or w.numel() != want[1] or not w.is_cuda:
            raise ValueError(f"{who}: dtype code {dtype} needs packed weights of {want[1]} x {want[0]} on the GPU (pack_conv_weight / pack_conv_weight_split / "
                             f"pack_conv_weight_bf16x3 with c_mult=32), got {w.numel()} x {w.dtype}")


def hifigan_resunit(rb, len_mul, x, y, w1, b1, w2, b2, channels, k_w, dil, slope, dtype, add=None, out_scale=1.0, ws=None, w_layout=0):
    """jatts_hifigan_resunit.  w_layout (F32E / F32E6): 0 = weights from pack_conv_weight_bf16x3(w, 32) (32 x 32 x 16 kernels), 1 = from
    pack_unit_weight_bf16x3_k32 (16 x 16 x 32 kernels)."""
    lib = _abi.load()
    d = _abi.ResUnitDesc()
    d.w_layout = w_layout
    d.rg = rb.struct(len_mul)
    d.dtype, d.channels, d.k_w, d.dil, d.slope = dtype, channels, k_w, dil, slope
    rows = rb.total * len_mul
    if x.numel() != rows * channels or y.numel() != rows * channels or x.dtype != torch_dtype(dtype):
        raise ValueError("hifigan_resunit: bad buffer size/dtype")
    d.x, d.y = _dev(x).data_ptr(), y.data_ptr()
    _check_unit_weights("hifigan_resunit", dtype, channels, k_w, w1, w2)
    d.w1, d.b1, d.w2, d.b2 = w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr()
    if dtype == F32S:
        if ws is None or ws[0].numel() < channels or ws[1].numel() < channels or ws[0].dtype != torch.float32:
            raise ValueError("hifigan_resunit: F32S needs the inverse weight scales (pack_conv_weight_split)")
        d.ws1, d.ws2 = ws[0].data_ptr(), ws[1].data_ptr()
    if add:
        for a in add:
            if a.numel() != rows * channels or a.dtype != x.dtype:
                raise ValueError("hifigan_resunit: bad mix buffer")
        d.add0 = add[0].data_ptr()
        d.add1 = add[1].data_ptr() if len(add) > 1 else None
    d.out_scale = out_scale
    with _Timed("resunit", (channels, k_w, dil, rows, len(add) if add else 0)):
        _abi.check(lib.jatts_hifigan_resunit(C.byref(d), _stream()), "jatts_hifigan_resunit")
    return y


def hifigan_resblock(rb, len_mul, x, y, units, channels, k_w, slope, dtype, add=None, out_scale=1.0, ws=None):
    """jatts_hifigan_resblock: ``units`` = [(w1, b1, w2, b2, dil), ...] (<= 3), all dilation units of one ResBlock in one launch.
    dtype F32S: ``ws`` = [(inverse scales of w1, of w2), ...] per unit (pack_conv_weight_split)."""
    lib = _abi.load()
    d = _abi.ResBlockDesc()
    d.rg = rb.struct(len_mul)
    d.dtype, d.channels, d.k_w, d.n_units, d.slope = dtype, channels, k_w, len(units), slope
    rows = rb.total * len_mul
    if x.numel() != rows * channels or y.numel() != rows * channels or x.dtype != torch_dtype(dtype):
        raise ValueError("hifigan_resblock: bad buffer size/dtype")
    d.x, d.y = _dev(x).data_ptr(), y.data_ptr()
    for i, (w1, b1, w2, b2, dil) in enumerate(units):
        _check_unit_weights("hifigan_resblock", dtype, channels, k_w, w1, w2)
        d.w1[i], d.b1[i], d.w2[i], d.b2[i], d.dil[i] = w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), dil
    if dtype == F32S:
        if ws is None or len(ws) != len(units):
            raise ValueError("hifigan_resblock: F32S needs the inverse weight scales of every unit")
        for i, (a, b) in enumerate(ws):
            d.ws1[i], d.ws2[i] = a.data_ptr(), b.data_ptr()
    if add:
        for a in add:
            if a.numel() != rows * channels or a.dtype != x.dtype:
                raise ValueError("hifigan_resblock: bad mix buffer")
        d.add0 = add[0].data_ptr()
        d.add1 = add[1].data_ptr() if len(add) > 1 else None
    d.out_scale = out_scale
    with _Timed("resblock", (channels, k_w, tuple(u[4] for u in units), rows, len(add) if add else 0)):
        _abi.check(lib.jatts_hifigan_resblock(C.byref(d), _stream()), "jatts_hifigan_resblock")
    return y


def hifigan_output(rb, len_mul, xs, in_scale, slope, c_in, k_w, w, bias, dtype):
    lib = _abi.load()
    rows = rb.total * len_mul
    y = torch.empty(rows, dtype=torch.float32, device=xs[0].device)
    arr = (C.c_void_p * len(xs))(*[_dev(x).data_ptr() for x in xs])
    rg = rb.struct(len_mul)
    _abi.check(lib.jatts_hifigan_output(C.byref(rg), dtype, arr, len(xs), in_scale, slope, c_in, k_w,
                                        w.data_ptr(), float(bias), y.data_ptr(), _stream()),
               "jatts_hifigan_output")
    return y


def relpos_attention(rb, q, ldq, k, ldk, vt, ldvt, g, ldg, ku, scale, n_heads, d_k, dtype,
                     q_col0=0, k_col0=0, rel_mode=1, rel_center=0, vt_col0=None, kv_len=None):
    lib = _abi.load()
    out = torch.empty(rb.total, n_heads * d_k, dtype=torch_dtype(dtype), device=q.device)
    d = _abi.RelAttnDesc()
    d.rg = rb.struct()
    d.dtype, d.n_heads, d.d_k = dtype, n_heads, d_k
    d.q, d.ldq = _ptr(_dev(q), q_col0), ldq
    d.k, d.ldk = _ptr(k, k_col0), ldk
    d.vt, d.ldvt = vt.data_ptr(), ldvt
    d.g, d.ldg = _ptr(g), ldg
    d.ku = _ptr(ku)
    d.scale = scale
    d.out, d.ldo = out.data_ptr(), n_heads * d_k
    d.rel_mode, d.rel_center = rel_mode, rel_center
    d.vt_col0 = _ptr(vt_col0)
    d.kv_len = _ptr(kv_len)
    with _Timed("relattn", (n_heads, d_k, rb.total, sum(v * v for v in rb.lens))):
        _abi.check(lib.jatts_relpos_attention(C.byref(d), _stream()), "jatts_relpos_attention")
    return out


def rowdot(x, ldx, rows, n_heads, d_k, vec, col0=0):
    lib = _abi.load()
    out = torch.empty(rows, n_heads, dtype=torch.float32, device=x.device)
    _abi.check(lib.jatts_rowdot(code_of(x), _ptr(_dev(x), col0), ldx, rows, n_heads, d_k, vec.data_ptr(),
                                out.data_ptr(), _stream()), "jatts_rowdot")
    return out


_BAD_IDS = {}      # device index -> int64 (1,) counter of out-of-range embedding ids seen by launches that passed no counter of their own


def _bad_counter(dev):
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    t = _BAD_IDS.get(key)
    if t is None:
        t = _BAD_IDS[key] = torch.zeros(1, dtype=torch.int64, device=dev)
    return t


def embed_scale(ids, table, scale, n_bad=None):
    """rows = table[ids] * scale.  The kernel ALWAYS bounds-checks: an id outside the table yields a zero row (never a read outside
    the table) and is counted -- in ``n_bad`` (int64 (1,) device counter zeroed by the caller, read at the caller's next host
    synchronisation: lr_sizes(..., check=n_bad)) or, when none is passed, in the device's shared counter, which
    check_bad_ids() / bad_ids_async() turn into the IndexError torch.nn.Embedding would have raised."""
    lib = _abi.load()
    if n_bad is None:
        n_bad = _bad_counter(_dev(ids).device)
    out = torch.empty(ids.numel(), table.shape[1], dtype=torch.float32, device=ids.device)
    _abi.check(lib.jatts_embed_scale(_dev(ids).data_ptr(), ids.numel(), table.data_ptr(), table.shape[1],
                                     float(scale), out.data_ptr(), table.shape[0], _ptr(n_bad), _stream()), "jatts_embed_scale")
    return out


def check_bad_ids(device):
    """Read the device's shared bad-id counter (a host synchronisation: call it right after one the caller needs anyway) and raise
    IndexError like torch.nn.Embedding if a launch since the last check saw an id outside its table."""
    t = _BAD_IDS.get(torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device())
    if t is None:
        return
    n = int(t)
    if n:
        t.zero_()
        raise IndexError(f"index out of range in self ({n} token ids outside the embedding table)")


def bad_ids_async(device):
    """Stream-ordered snapshot of the shared counter into pinned memory, no host wait.  -> resolve() which raises IndexError once the
    copy has landed and the count is non-zero (returns False while the copy is still in flight)."""
    dev = torch.device(device)
    t = _BAD_IDS.get(dev.index if dev.index is not None else torch.cuda.current_device())
    if t is None:
        return lambda wait=False: True
    host = torch.zeros(1, dtype=torch.int64).pin_memory()
    host.copy_(t, non_blocking=True)
    t.zero_()
    ev = torch.cuda.Event()
    ev.record()

    def resolve(wait=False):
        if wait:
            ev.synchronize()
        elif not ev.query():
            return False
        if int(host):
            n = int(host)
            host.zero_()
            raise IndexError(f"index out of range in self ({n} token ids outside the embedding table in an earlier training step)")
        return True
    return resolve


def layernorm(x, gamma, beta, out_dtype, eps=1e-12, out=None):
    lib = _abi.load()
    rows, dim = x.shape
    if out is None:
        out = torch.empty(rows, dim, dtype=torch_dtype(out_dtype), device=x.device)
    _abi.check(lib.jatts_layernorm(_dev(x).data_ptr(), code_of(x), dim, out.data_ptr(), out_dtype, dim, rows,
                                   dim, gamma.data_ptr(), beta.data_ptr(), eps, _stream()), "jatts_layernorm")
    return out


def affine_cast(x, out_dtype, scale=None, shift=None, ldy=None, x_col0=0, dim=None, out=None, out_col0=0):
    """y = x * scale + shift -> out_dtype.  ``x_col0`` / ``dim`` select a column slice of x; ``out`` / ``out_col0`` write into
    a column slice of an existing matrix (row stride out.shape[1]) instead of a fresh (rows, ldy) one."""
    lib = _abi.load()
    rows, ldx = x.shape
    dim = dim if dim is not None else ldx - x_col0
    fresh = out is None
    if fresh:
        ldy = ldy or dim
        out = torch.empty(rows, ldy, dtype=torch_dtype(out_dtype), device=x.device)
        dim_w = dim
    else:
        ldy, dim_w = out.shape[1], dim
        if out.dtype != torch_dtype(out_dtype) or out.shape[0] != rows or out_col0 + dim > ldy:
            raise ValueError("affine_cast: bad output slice")
    fn = lib.jatts_affine_cast if fresh else lib.jatts_affine_slice   # a slice writes exactly dim columns, no zero fill
    _abi.check(fn(_ptr(_dev(x), x_col0), ldx, _ptr(out, out_col0), out_dtype, ldy, rows, dim_w,
                  _ptr(scale), _ptr(shift), _stream()), "jatts_affine_cast")
    return out


def glu_dwconv_bn_swish(rb, x, channels, k_w, w_dw, bn_scale, bn_shift, dtype):
    lib = _abi.load()
    y = torch.empty(rb.total, channels, dtype=torch_dtype(dtype), device=x.device)
    rg = rb.struct()
    _abi.check(lib.jatts_glu_dwconv_bn_swish(C.byref(rg), dtype, _dev(x).data_ptr(), y.data_ptr(), channels,
                                             k_w, w_dw.data_ptr(), bn_scale.data_ptr(), bn_shift.data_ptr(),
                                             _stream()), "jatts_glu_dwconv_bn_swish")
    return y


def predictor_head(x, w, b, want_duration=False, offset=1.0):
    lib = _abi.load()
    rows, dim = x.shape
    v = torch.empty(rows, dtype=torch.float32, device=x.device)
    dur = torch.empty(rows, dtype=torch.int64, device=x.device) if want_duration else None
    _abi.check(lib.jatts_predictor_head(code_of(x), _dev(x).data_ptr(), dim, rows, dim, w.data_ptr(), float(b),
                                        v.data_ptr(), _ptr(dur), offset, _stream()), "jatts_predictor_head")
    return (v, dur) if want_duration else v


def variance_embed_add(rb, hs, p, wp, bp, e, we, be):
    lib = _abi.load()
    rg = rb.struct()
    _abi.check(lib.jatts_variance_embed_add(C.byref(rg), _dev(hs).data_ptr(), hs.shape[1], p.data_ptr(),
                                            wp.data_ptr(), bp.data_ptr(), wp.shape[-1], e.data_ptr(),
                                            we.data_ptr(), be.data_ptr(), we.shape[-1], _stream()),
               "jatts_variance_embed_add")
    return hs


def gated_tanh_sigmoid(rb, x, gseq, channels, dtype):
    lib = _abi.load()
    y = torch.empty(rb.total, channels, dtype=torch_dtype(dtype), device=x.device)
    rg = rb.struct()
    _abi.check(lib.jatts_gated_tanh_sigmoid(C.byref(rg), dtype, _dev(x).data_ptr(), _ptr(gseq), y.data_ptr(),
                                            channels, _stream()), "jatts_gated_tanh_sigmoid")
    return y


def groupnorm_mish(rb, x, channels, groups, gamma, beta, out_dtype, eps=1e-5, addvec=None, time_split=True):
    lib = _abi.load()
    y = torch.empty(rb.total, channels, dtype=torch_dtype(out_dtype), device=x.device)
    rg = rb.struct()
    ws = None
    if time_split:  # chunk statistics of the two-launch form (include/jatts_hip.h)
        ws = torch.empty(rb.n_seq * groups * ((rb.max_len + 63) // 64) * 3, dtype=torch.float32, device=x.device)
    _abi.check(lib.jatts_groupnorm_mish(C.byref(rg), _dev(x).data_ptr(), code_of(x), y.data_ptr(), out_dtype, channels,
                                        groups, gamma.data_ptr(), beta.data_ptr(), eps, _ptr(addvec), _ptr(ws), _stream()),
               "jatts_groupnorm_mish")
    return y


def snakebeta(x, alpha, inv_beta):
    lib = _abi.load()
    y = torch.empty_like(x)
    _abi.check(lib.jatts_snakebeta(code_of(x), _dev(x).data_ptr(), y.data_ptr(), x.shape[0], x.shape[1],
                                   alpha.data_ptr(), inv_beta.data_ptr(), _stream()), "jatts_snakebeta")
    return y


def l2_normalize(x, out_dtype, ldy=None, eps=1e-12):
    """F.normalize(x, dim=1) (eps 1e-12) -> out_dtype, zero-padded to ldy columns."""
    lib = _abi.load()
    rows, dim = x.shape
    ldy = ldy or dim
    y = torch.empty(rows, ldy, dtype=torch_dtype(out_dtype), device=x.device)
    _abi.check(lib.jatts_l2_normalize(_dev(x).data_ptr(), dim, y.data_ptr(), out_dtype, ldy, rows, dim, eps,
                                      _stream()), "jatts_l2_normalize")
    return y


def gaussian_sample(stats, noise, noise_scale):
    lib = _abi.load()
    rows, c2 = stats.shape
    z = torch.empty(rows, c2 // 2, dtype=torch.float32, device=stats.device)
    _abi.check(lib.jatts_gaussian_sample(_dev(stats).data_ptr(), noise.data_ptr(), z.data_ptr(), rows, c2 // 2,
                                         float(noise_scale), _stream()), "jatts_gaussian_sample")
    return z


def flip_channels(x):
    lib = _abi.load()
    y = torch.empty_like(x)
    _abi.check(lib.jatts_flip_channels(_dev(x).data_ptr(), y.data_ptr(), x.shape[0], x.shape[1], _stream()),
               "jatts_flip_channels")
    return y


def add_seq_vector(rb, hs, vec):
    lib = _abi.load()
    rg = rb.struct()
    _abi.check(lib.jatts_add_seq_vector(C.byref(rg), _dev(hs).data_ptr(), hs.shape[1], vec.data_ptr(),
                                        _stream()), "jatts_add_seq_vector")
    return hs


def lr_durations(rb, d, alpha=1.0, zero_rule=2):
    """-> (d_eff, cum, olens int64 (n_seq,), fallback int64 (n_seq,) or None).  zero_rule 2 (default): an utterance whose
    durations sum to 0 gets all ones, as the reference's B=1 inference() does; ``fallback`` marks those utterances."""
    lib = _abi.load()
    d_eff = torch.empty_like(d)
    cum = torch.empty_like(d)
    buf = torch.empty(2 * rb.n_seq, dtype=torch.int64, device=d.device)
    rg = rb.struct()
    _abi.check(lib.jatts_lr_durations(C.byref(rg), _dev(d).data_ptr(), float(alpha), int(zero_rule),
                                      d_eff.data_ptr(), cum.data_ptr(), buf.data_ptr(), _stream()),
               "jatts_lr_durations")
    return d_eff, cum, buf[:rb.n_seq], (buf[rb.n_seq:] if zero_rule == 2 else None)


def lr_sizes_dev(rb, d, alpha=1.0, check=None):
    """The device half of lr_sizes (capturable: no host access): -> (d_eff, cum, sizes) with sizes = int64 [olens (n_seq) | all-zero
    fallback flags (n_seq) | the bad-id counter `check` (1, when given)], read by lr_sizes_host."""
    d_eff, cum, olens, fb = lr_durations(rb, d, alpha)
    return d_eff, cum, torch.cat([olens, fb] + ([check.view(-1)] if check is not None else []))


def lr_sizes_host(rb, sizes, checked):
    """The one host sync of the path: sizes (lr_sizes_dev) -> olens list.  Logs the reference's warning (length_regulator.py:87-90) for
    utterances that took the all-zero fallback; a non-zero bad-id counter raises IndexError like torch.nn.Embedding."""
    host = sizes.tolist()
    if checked and host[2 * rb.n_seq]:
        raise IndexError("token id out of range")
    olens_h, fb_h = host[:rb.n_seq], host[rb.n_seq:2 * rb.n_seq]
    if any(fb_h):
        import logging
        logging.warning("predicted durations includes all 0 sequences. fill the first element with 1.")
    return olens_h


def lr_sizes(rb, d, alpha=1.0, check=None):
    """lr_durations + the one host sync of the path: -> (d_eff, cum, olens list).  ``check``: the embed_scale bad-id counter, read in
    the same transfer."""
    d_eff, cum, sizes = lr_sizes_dev(rb, d, alpha, check)
    return d_eff, cum, lr_sizes_host(rb, sizes, check is not None)


def zero_pad_rows(rb, x, valid_len, len_mul=1):
    """x: f32 / f16 (rb.total, dim) or (rb.total,); zero the rows t >= valid_len[b] of every sequence, in place."""
    lib = _abi.load()
    dim = x.shape[1] if x.dim() == 2 else 1
    rg = rb.struct(len_mul)
    _abi.check(lib.jatts_zero_pad_rows(C.byref(rg), _dev(x).data_ptr(), code_of(x), dim, dim, valid_len.data_ptr(), _stream()),
               "jatts_zero_pad_rows")
    return x


def cfm_mix(rb, x1, z, t, sigma_min):
    lib = _abi.load()
    y, u = torch.empty_like(x1), torch.empty_like(x1)
    rg = rb.struct()
    _abi.check(lib.jatts_cfm_mix(C.byref(rg), _dev(x1).data_ptr(), z.data_ptr(), t.data_ptr(), float(sigma_min), x1.shape[1],
                                 y.data_ptr(), u.data_ptr(), _stream()), "jatts_cfm_mix")
    return y, u


def sq_err_sum(a, b, scale=1.0):
    """scale * sum((a - b)^2) -> f32 scalar tensor on the GPU (double accumulation, deterministic order)."""
    lib = _abi.load()
    out = torch.empty((), dtype=torch.float32, device=a.device)
    ws = torch.empty(256, dtype=torch.float64, device=a.device)
    _abi.check(lib.jatts_sq_err_sum(_dev(a).data_ptr(), b.data_ptr(), a.numel(), float(scale), out.data_ptr(), ws.data_ptr(),
                                    _stream()), "jatts_sq_err_sum")
    return out


def lr_gather(rb_in, cum, rb_out, x, want_index=False):
    lib = _abi.load()
    dim = x.shape[1]
    out = torch.empty(rb_out.total, dim, dtype=torch.float32, device=x.device)
    idx = torch.empty(rb_out.total, dtype=torch.int64, device=x.device) if want_index else None
    rg = rb_in.struct()
    _abi.check(lib.jatts_lr_gather(C.byref(rg), cum.data_ptr(), rb_out.cu.data_ptr(), rb_out.max_len,
                                   _dev(x).data_ptr(), dim, out.data_ptr(), _ptr(idx), _stream()),
               "jatts_lr_gather")
    return (out, idx) if want_index else out


def gaussian_upsample(rb_in, d, rb_out, hs, delta=0.1):
    lib = _abi.load()
    dim = hs.shape[1]
    out = torch.empty(rb_out.total, dim, dtype=torch.float32, device=hs.device)
    rg = rb_in.struct()
    _abi.check(lib.jatts_gaussian_upsample(C.byref(rg), _dev(d).data_ptr(), rb_out.cu.data_ptr(),
                                           rb_out.max_len, hs.data_ptr(), dim, delta, out.data_ptr(),
                                           _stream()), "jatts_gaussian_upsample")
    return out


def alignment_logp(rb_f, rb_t, feats, text, adim):
    """jatts_alignment_logp: f32 (frame rows, ld) log-probabilities, ld = round_up(max text length, 8)."""
    lib = _abi.load()
    ld = round_up(max(rb_t.max_len, 1), 8)
    out = torch.empty(rb_f.total, ld, dtype=torch.float32, device=feats.device)
    rg = rb_f.struct()
    _abi.check(lib.jatts_alignment_logp(C.byref(rg), rb_t.cu.data_ptr(), rb_t.max_len, _dev(feats).data_ptr(),
                                        _dev(text).data_ptr(), adim, out.data_ptr(), ld, _stream()), "jatts_alignment_logp")
    return out


def mas_viterbi(rb_f, rb_t, log_p):
    """jatts_mas_viterbi -> (path int64 (frame rows,), dur int64 (token rows,), score f64 (n_seq,))."""
    lib = _abi.load()
    dev = log_p.device
    path = torch.empty(rb_f.total, dtype=torch.int64, device=dev)
    dur = torch.empty(rb_t.total, dtype=torch.int64, device=dev)
    score = torch.empty(rb_f.n_seq, dtype=torch.float64, device=dev)
    rg = rb_f.struct()
    if log_p.dtype != torch.float32 or not log_p.is_contiguous():
        raise ValueError("mas_viterbi: log_p must be contiguous float32")
    _abi.check(lib.jatts_mas_viterbi(C.byref(rg), rb_t.cu.data_ptr(), _dev(log_p).data_ptr(), log_p.shape[1], rb_t.max_len,
                                     path.data_ptr(), dur.data_ptr(), score.data_ptr(), _stream()), "jatts_mas_viterbi")
    return path, dur, score


def pcm16(y, out=None):
    """jatts_pcm16: f32 waveform (n,) on the GPU -> int16 PCM (n,) on the GPU."""
    lib = _abi.load()
    y = _dev(y)
    if y.dtype != torch.float32 or not y.is_contiguous():
        raise ValueError("pcm16: contiguous float32 expected")
    if out is None:
        out = torch.empty(y.numel(), dtype=torch.int16, device=y.device)
    _abi.check(lib.jatts_pcm16(y.data_ptr(), y.numel(), out.data_ptr(), _stream()), "jatts_pcm16")
    return out


# ---- speaker-embedding front end (csrc/spkemb.hip)
def frame_signal(rb_frames, cu_samples, x, window, n_fft, hop, ldo):
    lib = _abi.load()
    out = torch.empty(rb_frames.total, ldo, dtype=torch.float32, device=x.device)
    rg = rb_frames.struct()
    _abi.check(lib.jatts_frame_signal(C.byref(rg), cu_samples.data_ptr(), _dev(x).data_ptr(), window.data_ptr(), n_fft, hop,
                                      out.data_ptr(), ldo, _stream()), "jatts_frame_signal")
    return out


def power_spectrum(x, n_bins, ldo):
    lib = _abi.load()
    out = torch.empty(x.shape[0], ldo, dtype=torch.float32, device=x.device)
    _abi.check(lib.jatts_power_spectrum(_dev(x).data_ptr(), x.shape[1], n_bins, x.shape[0], out.data_ptr(), ldo, _stream()),
               "jatts_power_spectrum")
    return out


def fbank_post(rb, p, n_mels, ldo, amin=1e-10, top_db=80.0):
    lib = _abi.load()
    out = torch.empty(rb.total, ldo, dtype=torch.float32, device=p.device)
    rg = rb.struct()
    _abi.check(lib.jatts_fbank_post(C.byref(rg), _dev(p).data_ptr(), p.shape[1], n_mels, amin, top_db, out.data_ptr(), ldo,
                                    _stream()), "jatts_fbank_post")
    return out


def seq_mean_std(rb, x, dim, logits=None, want_std=True, eps=1e-12):
    """-> (n_seq, 2*dim) f32 [mean | std] (or (n_seq, dim) means when want_std is False)."""
    lib = _abi.load()
    ld = 2 * dim if want_std else dim
    out = torch.empty(rb.n_seq, ld, dtype=torch.float32, device=x.device)
    rg = rb.struct()
    _abi.check(lib.jatts_seq_mean_std(C.byref(rg), _dev(x).data_ptr(), x.shape[1], dim, _ptr(logits),
                                      logits.shape[1] if logits is not None else 0, out.data_ptr(),
                                      _ptr(out, dim) if want_std else None, ld, eps, _stream()), "jatts_seq_mean_std")
    return out


def seq_affine_act(rb, x, dim, out_dtype, seq_vec=None, pre_act=ACT_NONE, scale=None, shift=None, post_act=ACT_NONE, ldy=None):
    lib = _abi.load()
    ldy = ldy or dim
    y = torch.empty(rb.total, ldy, dtype=torch_dtype(out_dtype), device=x.device)
    rg = rb.struct()
    _abi.check(lib.jatts_seq_affine_act(C.byref(rg), _dev(x).data_ptr(), x.shape[1], dim, _ptr(seq_vec), pre_act, _ptr(scale),
                                        _ptr(shift), post_act, y.data_ptr(), out_dtype, ldy, _stream()), "jatts_seq_affine_act")
    return y


def se_scale_add(rb, x, s, resid=None, out=None, out_col0=0):
    """y = x * sigmoid(s[b]) + resid; ``out`` / ``out_col0``: write into a column slice of a wider matrix."""
    lib = _abi.load()
    y = torch.empty_like(x) if out is None else out
    rg = rb.struct()
    _abi.check(lib.jatts_se_scale_add(C.byref(rg), _dev(x).data_ptr(), x.shape[1], s.data_ptr(), _ptr(resid), _ptr(y, out_col0),
                                      y.shape[1], _stream()), "jatts_se_scale_add")
    return y


# ---- training side (csrc/training.hip)
def masked_loss(rb, a, b, valid_len, kind, scale, log_offset=-1.0):
    """scale * sum over valid rows of |a - b| (kind 0) or (a - b)^2 (kind 1); b -> log(b + log_offset) when log_offset >= 0."""
    lib = _abi.load()
    a2 = a if a.dim() == 2 else a.reshape(-1, 1)
    b2 = b if b.dim() == 2 else b.reshape(-1, 1)
    out = torch.empty((), dtype=torch.float32, device=a.device)
    ws = torch.empty(4 * rb.n_seq, dtype=torch.float64, device=a.device)
    rg = rb.struct()
    _abi.check(lib.jatts_masked_loss(C.byref(rg), _dev(a2).data_ptr(), a2.shape[1], b2.data_ptr(), b2.shape[1], a2.shape[1],
                                     _ptr(valid_len), kind, float(log_offset), float(scale), out.data_ptr(), ws.data_ptr(), _stream()),
               "jatts_masked_loss")
    return out


def conv1d_wgrad(rb, x, dy, c_in, n_out, k_w, dil, pad, len_mul=1, want_db=False):
    """-> dw (n_out, c_in, k_w); with want_db -> (dw, db): the bias gradient comes out of the same launch (the kernel's dy tiles)."""
    lib = _abi.load()
    dw = torch.empty(n_out, c_in, k_w, dtype=torch.float32, device=x.device)
    db = torch.empty(n_out, dtype=torch.float32, device=x.device) if want_db else None
    ws = torch.empty(rb.n_seq * (k_w * round_up(n_out, 64) * round_up(c_in, 64) + round_up(n_out, 64)), dtype=torch.float32,
                     device=x.device)                                                                                   # split-K partials
    rg = rb.struct(len_mul)
    _count(2.0 * c_in * n_out * k_w * rb.total * len_mul)
    _ws(x.device)
    _ws_check(lib.jatts_conv1d_wgrad(C.byref(rg), _dev(x).data_ptr(), x.shape[1], dy.data_ptr(), dy.shape[1], c_in, n_out, k_w, dil,
                                      pad, dw.data_ptr(), _ptr(db), ws.data_ptr(), _stream()), "jatts_conv1d_wgrad")
    return (dw, db) if want_db else dw


def col_sum(x, dim=None):
    lib = _abi.load()
    dim = dim or x.shape[1]
    out = _zeros((dim), x.device)
    _ws(x.device)
    _ws_check(lib.jatts_col_sum(_dev(x).data_ptr(), x.shape[1], x.shape[0], dim, out.data_ptr(), _stream()), "jatts_col_sum")
    return out


# ------------------------------------------------------------------------------------------ training ops (train_ops.hip)
ACT_MODE = {"relu": 1, "tanh": 2, "swish": 3, "mish": 4}


def _f32c(t):
    t = _dev(t)
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise ValueError("training ops take contiguous f32 device tensors")
    return t


def layernorm_bwd(x, dy, gamma, eps, need_dx=True, need_dparam=True):
    lib = _abi.load()
    x, dy = _f32c(x), _f32c(dy)
    rows, dim = x.shape
    dx = torch.empty_like(x) if need_dx else None
    dg = _zeros((dim), x.device) if need_dparam else None
    db = _zeros((dim), x.device) if need_dparam else None
    _ws(x.device)
    _ws_check(lib.jatts_layernorm_bwd(x.data_ptr(), dim, dy.data_ptr(), dim, _f32c(gamma).data_ptr(), rows, dim, float(eps), _ptr(dx), dim,
                                       _ptr(dg), _ptr(db), _stream()), "jatts_layernorm_bwd")
    return dx, dg, db


def act_fwd(x, mode):
    lib = _abi.load()
    x = _f32c(x)
    y = torch.empty_like(x)
    _abi.check(lib.jatts_act_fwd(ACT_MODE[mode], x.data_ptr(), y.data_ptr(), x.numel(), _stream()), "jatts_act_fwd")
    return y


def act_bwd(x, dy, mode):
    lib = _abi.load()
    x, dy = _f32c(x), _f32c(dy)
    dx = torch.empty_like(x)
    _abi.check(lib.jatts_act_bwd(ACT_MODE[mode], x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.numel(), _stream()), "jatts_act_bwd")
    return dx


def glu_fwd(x):
    lib = _abi.load()
    x = _f32c(x)
    rows, c2 = x.shape
    y = torch.empty(rows, c2 // 2, dtype=torch.float32, device=x.device)
    _abi.check(lib.jatts_glu_fwd(x.data_ptr(), y.data_ptr(), rows, c2 // 2, _stream()), "jatts_glu_fwd")
    return y


def glu_bwd(x, dy):
    lib = _abi.load()
    x, dy = _f32c(x), _f32c(dy)
    dx = torch.empty_like(x)
    _abi.check(lib.jatts_glu_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.shape[0], x.shape[1] // 2, _stream()), "jatts_glu_bwd")
    return dx


def dwconv(rb, x, w, bias, pad, flip=False):
    """depthwise conv on packed rows: w (C, K); flip=True is the data gradient (call with pad' = K - 1 - pad)."""
    lib = _abi.load()
    x, w = _f32c(x), _f32c(w)
    y = torch.empty_like(x)
    rg = rb.struct()
    _abi.check(lib.jatts_dwconv(C.byref(rg), x.data_ptr(), w.data_ptr(), _ptr(bias), y.data_ptr(), x.shape[1], w.shape[1], pad, int(flip),
                                _stream()), "jatts_dwconv")
    return y


def dwconv_wgrad(rb, x, dy, k_w, pad):
    lib = _abi.load()
    x, dy = _f32c(x), _f32c(dy)
    dw = _zeros((x.shape[1], k_w), x.device)
    rg = rb.struct()
    _ws(x.device)
    _ws_check(lib.jatts_dwconv_wgrad(C.byref(rg), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), x.shape[1], k_w, pad, _stream()),
               "jatts_dwconv_wgrad")
    return dw


def col_stats(x, y2=None, shift=None, mul=None):
    """y2 None -> (sum (x - shift), sum (x - shift)^2); else (sum y2, sum y2 (x - shift) mul), per column."""
    lib = _abi.load()
    x = _f32c(x)
    rows, dim = x.shape
    o0 = _zeros((dim), x.device)
    o1 = _zeros((dim), x.device)
    _ws(x.device)
    _ws_check(lib.jatts_col_stats(x.data_ptr(), _ptr(y2), dim, rows, dim, _ptr(shift), _ptr(mul), 0 if y2 is None else 1, o0.data_ptr(),
                                   o1.data_ptr(), _stream()), "jatts_col_stats")
    return o0, o1


def bn_bwd_apply(x, dy, mean, rstd, gamma, s_dy, s_dyx):
    lib = _abi.load()
    x, dy = _f32c(x), _f32c(dy)
    dx = torch.empty_like(x)
    _abi.check(lib.jatts_bn_bwd_apply(x.data_ptr(), dy.data_ptr(), x.shape[0], x.shape[1], mean.data_ptr(), rstd.data_ptr(),
                                      _f32c(gamma).data_ptr(), s_dy.data_ptr(), s_dyx.data_ptr(), dx.data_ptr(), _stream()), "jatts_bn_bwd_apply")
    return dx


def index_add_rows(src, idx, n_dst, scale=1.0, skip=-1):
    lib = _abi.load()
    src = _f32c(src)
    dst = _zeros((n_dst, src.shape[1]), src.device)
    _abi.check(lib.jatts_index_add_rows(src.data_ptr(), src.shape[1], _dev(idx).data_ptr(), src.shape[0], src.shape[1], float(scale), int(skip),
                                        n_dst, dst.data_ptr(), _stream()), "jatts_index_add_rows")
    return dst


def lr_segment_sum(rb_in, cum, rb_out, dy):
    lib = _abi.load()
    dy = _f32c(dy)
    dhs = torch.empty(rb_in.total, dy.shape[1], dtype=torch.float32, device=dy.device)
    rg = rb_in.struct()
    _abi.check(lib.jatts_lr_segment_sum(C.byref(rg), _dev(cum).data_ptr(), rb_out.cu.data_ptr(), dy.data_ptr(), dy.shape[1], dhs.data_ptr(),
                                        _stream()), "jatts_lr_segment_sum")
    return dhs


def shift_softmax_fwd(ac, bd, lens, scale, mode=1):
    """ac (B, H, T, T), bd (B, H, T, T) [mode 1, legacy rel_shift] or (B, H, T, 2T-1) [mode 2, new rel_shift] -> attention
    probabilities (bd None = plain attention)."""
    lib = _abi.load()
    ac = _f32c(ac)
    B, H, T, _ = ac.shape
    p = torch.empty_like(ac)
    _abi.check(lib.jatts_shift_softmax_fwd(ac.data_ptr(), _ptr(bd), B, H, T, _ptr(lens), float(scale), mode, p.data_ptr(), _stream()),
               "jatts_shift_softmax_fwd")
    return p


def shift_softmax_bwd(p, dp, scale, need_dbd=True, mode=1):
    lib = _abi.load()
    p, dp = _f32c(p), _f32c(dp)
    B, H, T, _ = p.shape
    ds = torch.empty_like(p)
    dbd = torch.empty(B, H, T, 2 * T - 1 if mode == 2 else T, dtype=torch.float32, device=p.device) if need_dbd else None
    _abi.check(lib.jatts_shift_softmax_bwd(p.data_ptr(), dp.data_ptr(), B, H, T, float(scale), mode, ds.data_ptr(), _ptr(dbd), _stream()),
               "jatts_shift_softmax_bwd")
    return ds, dbd


def gate_bwd(x, dy):
    lib = _abi.load()
    x, dy = _f32c(x), _f32c(dy)
    dx = torch.empty_like(x)
    _abi.check(lib.jatts_gate_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.shape[0], x.shape[1] // 2, _stream()), "jatts_gate_bwd")
    return dx


def weight_norm_fwd(v, g):
    """-> (w, inv_norm): w = g v / ||v|| per output channel (v (n_out, ...), g (n_out, 1, ...))."""
    lib = _abi.load()
    v, g = _f32c(v), _f32c(g)
    n_out = v.shape[0]
    w = torch.empty_like(v)
    inv = torch.empty(n_out, dtype=torch.float32, device=v.device)
    _abi.check(lib.jatts_weight_norm_fwd(v.data_ptr(), g.data_ptr(), n_out, v.numel() // n_out, w.data_ptr(), inv.data_ptr(), _stream()),
               "jatts_weight_norm_fwd")
    return w, inv


def weight_norm_bwd(v, g, inv, dw):
    lib = _abi.load()
    v, g, dw = _f32c(v), _f32c(g), _f32c(dw)
    n_out = v.shape[0]
    dv, dg = torch.empty_like(v), torch.empty_like(g)
    _abi.check(lib.jatts_weight_norm_bwd(v.data_ptr(), g.data_ptr(), inv.data_ptr(), dw.data_ptr(), n_out, v.numel() // n_out,
                                         dv.data_ptr(), dg.data_ptr(), _stream()), "jatts_weight_norm_bwd")
    return dv, dg


def split_add(o, h, skip=None):
    """o (rows, 2 dim) = [res | skip part]; -> (h + res, skip + skip part)."""
    lib = _abi.load()
    o, h = _f32c(o), _f32c(h)
    rows, dim = h.shape
    if o.shape != (rows, 2 * dim):
        raise ValueError("split_add: o must be (rows, 2 dim)")
    h_out, s_out = torch.empty_like(h), torch.empty_like(h)
    _abi.check(lib.jatts_split_add(o.data_ptr(), h.data_ptr(), _ptr(_f32c(skip) if skip is not None else None), rows, dim,
                                   h_out.data_ptr(), s_out.data_ptr(), _stream()), "jatts_split_add")
    return h_out, s_out


def concat2(a, b, rows, dim, device):
    """-> (rows, 2 dim) = [a | b]; None = zeros."""
    lib = _abi.load()
    out = torch.empty(rows, 2 * dim, dtype=torch.float32, device=device)
    _abi.check(lib.jatts_concat2(_ptr(_f32c(a) if a is not None else None), _ptr(_f32c(b) if b is not None else None), rows, dim,
                                 out.data_ptr(), _stream()), "jatts_concat2")
    return out


def outer_rows(v, w, bias=None, out=None):
    lib = _abi.load()
    v, w = _f32c(v), _f32c(w)
    acc = out is not None
    if out is None:
        out = torch.empty(v.numel(), w.numel(), dtype=torch.float32, device=v.device)
    _abi.check(lib.jatts_outer_rows(v.data_ptr(), w.data_ptr(), _ptr(bias), v.numel(), w.numel(), int(acc), out.data_ptr(), _stream()),
               "jatts_outer_rows")
    return out


def col_wsum(x, v):
    lib = _abi.load()
    x, v = _f32c(x), _f32c(v)
    out = _zeros((x.shape[1]), x.device)
    _ws(x.device)
    _ws_check(lib.jatts_col_wsum(x.data_ptr(), x.shape[1], v.data_ptr(), x.shape[0], x.shape[1], out.data_ptr(), _stream()), "jatts_col_wsum")
    return out


def row_dot(x, w, bias=None):
    lib = _abi.load()
    x, w = _f32c(x), _f32c(w)
    y = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    _abi.check(lib.jatts_row_dot(x.data_ptr(), x.shape[1], w.data_ptr(), _ptr(bias), x.shape[0], x.shape[1], y.data_ptr(), _stream()),
               "jatts_row_dot")
    return y


def masked_loss_bwd(rb, a, b, valid_len, kind, scale, upstream=None, log_offset=-1.0):
    lib = _abi.load()
    a2 = _f32c(a if a.dim() == 2 else a.reshape(-1, 1))
    b2 = _f32c(b if b.dim() == 2 else b.reshape(-1, 1))
    da = torch.empty_like(a2)
    rg = rb.struct()
    _abi.check(lib.jatts_masked_loss_bwd(C.byref(rg), a2.data_ptr(), a2.shape[1], b2.data_ptr(), b2.shape[1], a2.shape[1], _ptr(valid_len), kind,
                                         float(log_offset), float(scale), _ptr(upstream), da.data_ptr(), a2.shape[1], _stream()),
               "jatts_masked_loss_bwd")
    return da.view_as(a)


def dropout(x, p, seed, seed_dev=None):
    """seed_dev: optional int64 (1,) device tensor -- the mask seed is then *seed_dev + seed (graph mode: per-step base on the device)."""
    lib = _abi.load()
    x = _f32c(x)
    y = torch.empty_like(x)
    _abi.check(lib.jatts_dropout(x.data_ptr(), y.data_ptr(), x.numel(), float(p), int(seed) & 0xFFFFFFFFFFFFFFFF, _ptr(seed_dev), _stream()),
               "jatts_dropout")
    return y


def qkv_split(qkv, u, v, B, T, H):
    """qkv (B T, 3 A) -> (q + u, q + v, k, v) each (B, H, T, d_k) contiguous; u = v = None: (q, k, v)."""
    lib = _abi.load()
    qkv = _f32c(qkv)
    A3 = qkv.shape[1]
    dk = A3 // 3 // H
    if qkv.shape[0] != B * T or dk * H * 3 != A3 or (u is None) != (v is None):
        raise ValueError("qkv_split: shapes")
    if u is not None:
        u, v = _f32c(u), _f32c(v)
        if u.numel() != H * dk or v.numel() != H * dk:
            raise ValueError("qkv_split: bias shapes")
    outs = [torch.empty(B, H, T, dk, dtype=torch.float32, device=qkv.device) for _ in range(4 if u is not None else 3)]
    qu, qv, k, vv = outs if u is not None else (outs[0], None, outs[1], outs[2])
    _abi.check(lib.jatts_qkv_split(qkv.data_ptr(), _ptr(u), _ptr(v), B, T, H, dk, qu.data_ptr(), _ptr(qv), k.data_ptr(), vv.data_ptr(), _stream()),
               "jatts_qkv_split")
    return outs


def qkv_split_bwd(dqu, dqv, dk_, dvv):
    """-> (dqkv (B T, 3 A), du (A,), dv (A,)); dqv None (no position biases): du = dv = None."""
    lib = _abi.load()
    dqu, dk_, dvv = _f32c(dqu), _f32c(dk_), _f32c(dvv)
    B, H, T, dk = dqu.shape
    dqkv = torch.empty(B * T, 3 * H * dk, dtype=torch.float32, device=dqu.device)
    du = dv = None
    if dqv is not None:
        dqv = _f32c(dqv)
        du, dv = _zeros((H * dk), dqu.device), _zeros((H * dk), dqu.device)
    _ws(dqu.device)
    _ws_check(lib.jatts_qkv_split_bwd(dqu.data_ptr(), _ptr(dqv), dk_.data_ptr(), dvv.data_ptr(), B, T, H, dk, dqkv.data_ptr(), _ptr(du), _ptr(dv),
                                       _stream()), "jatts_qkv_split_bwd")
    return dqkv, du, dv


def act_dropout(x, mode, p, seed, seed_dev=None, dy=None):
    """dropout(act(x)) in one pass; with dy: its backward dy * act'(x) * mask / (1 - p).  Same mask as dropout()."""
    lib = _abi.load()
    x = _f32c(x)
    out = torch.empty_like(x)
    _abi.check(lib.jatts_act_dropout(ACT_MODE[mode], x.data_ptr(), _ptr(_f32c(dy) if dy is not None else None), out.data_ptr(), x.numel(),
                                     float(p), int(seed) & 0xFFFFFFFFFFFFFFFF, _ptr(seed_dev), _stream()), "jatts_act_dropout")
    return out


def dropout_add(x, resid, p, alpha, seed, seed_dev=None):
    """resid + alpha * dropout(x) (resid may be None); see jatts_dropout_add."""
    lib = _abi.load()
    x = _f32c(x)
    if resid is not None:
        resid = _f32c(resid)
        if resid.shape != x.shape:
            raise ValueError("dropout_add: shape mismatch")
    y = torch.empty_like(x)
    _abi.check(lib.jatts_dropout_add(x.data_ptr(), _ptr(resid), y.data_ptr(), x.numel(), float(p), float(alpha),
                                     int(seed) & 0xFFFFFFFFFFFFFFFF, _ptr(seed_dev), _stream()), "jatts_dropout_add")
    return y


def sumsq(x, out):
    """out (f64 scalar tensor on the device) += sum x^2"""
    lib = _abi.load()
    x = _f32c(x)
    _ws(x.device)
    _ws_check(lib.jatts_sumsq(x.data_ptr(), x.numel(), out.data_ptr(), _stream()), "jatts_sumsq")
    return out


def gather_grads(grads, offsets, flat, accumulate=False):
    """grads: list of contiguous f32 device tensors, offsets: their element offsets in ``flat``; flat[off : off + numel] (+)= grad."""
    lib = _abi.load()
    n = len(grads)
    if n == 0:
        return
    for g in grads:
        if g.dtype != torch.float32 or not g.is_contiguous() or g.device != flat.device:
            raise ValueError("gather_grads takes contiguous f32 tensors on the flat buffer's device")
    src = (C.c_void_p * n)(*[g.data_ptr() for g in grads])
    num = (C.c_int64 * n)(*[g.numel() for g in grads])
    off = (C.c_int64 * n)(*[int(o) for o in offsets])
    _abi.check(lib.jatts_gather_grads(src, num, off, n, flat.data_ptr(), 1 if accumulate else 0, _stream()), "jatts_gather_grads")


def adam_hyper(lr, beta1, beta2, eps, weight_decay, step):
    """The 7 per-step scalars of jatts_adam_step's hyper_dev, computed in double like torch: [lr / bc1, 1 - b1, b2, 1 - b2, eps, wd, sqrt(bc2)]."""
    bc1, bc2 = 1.0 - beta1 ** step, 1.0 - beta2 ** step
    return [lr / bc1, 1.0 - beta1, beta2, 1.0 - beta2, eps, weight_decay, bc2 ** 0.5]


def adam_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_sumsq=None, max_norm=0.0, hyper_dev=None):
    lib = _abi.load()
    _abi.check(lib.jatts_adam_step(_f32c(p).data_ptr(), _f32c(g).data_ptr(), _f32c(m).data_ptr(), _f32c(v).data_ptr(), p.numel(), float(lr),
                                   float(beta1), float(beta2), float(eps), float(weight_decay), int(step), _ptr(grad_sumsq), float(max_norm),
                                   _ptr(hyper_dev), _stream()), "jatts_adam_step")


def groupnorm_fwd(rb, x, groups, gamma, beta, eps):
    """-> (y, mean, rstd); mean / rstd (n_seq * groups,) for groupnorm_bwd."""
    lib = _abi.load()
    x = _f32c(x)
    y = torch.empty_like(x)
    mean = torch.empty(rb.n_seq * groups, dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    rg = rb.struct()
    _abi.check(lib.jatts_groupnorm_fwd(C.byref(rg), x.data_ptr(), x.shape[1], groups, _f32c(gamma).data_ptr(), _f32c(beta).data_ptr(), float(eps),
                                       y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _stream()), "jatts_groupnorm_fwd")
    return y, mean, rstd


def groupnorm_bwd(rb, x, dy, groups, gamma, mean, rstd, need_dx=True, need_dparam=True):
    lib = _abi.load()
    x, dy = _f32c(x), _f32c(dy)
    dim = x.shape[1]
    dx = torch.empty_like(x) if need_dx else None
    dg = _zeros((dim), x.device) if need_dparam else None
    db = _zeros((dim), x.device) if need_dparam else None
    rg = rb.struct()
    _ws(x.device)
    _ws_check(lib.jatts_groupnorm_bwd(C.byref(rg), x.data_ptr(), dy.data_ptr(), dim, groups, _f32c(gamma).data_ptr(), mean.data_ptr(),
                                       rstd.data_ptr(), _ptr(dx), _ptr(dg), _ptr(db), _stream()), "jatts_groupnorm_bwd")
    return dx, dg, db


def snakebeta_fwd(x, alpha, beta):
    lib = _abi.load()
    x = _f32c(x)
    y = torch.empty_like(x)
    _abi.check(lib.jatts_snakebeta_fwd(x.data_ptr(), y.data_ptr(), x.shape[0], x.shape[1], _f32c(alpha).data_ptr(), _f32c(beta).data_ptr(),
                                       _stream()), "jatts_snakebeta_fwd")
    return y


def snakebeta_bwd(x, dy, alpha, beta):
    lib = _abi.load()
    x, dy = _f32c(x), _f32c(dy)
    dx = torch.empty_like(x)
    da = _zeros((x.shape[1]), x.device)
    db = _zeros((x.shape[1]), x.device)
    _ws(x.device)
    _ws_check(lib.jatts_snakebeta_bwd(x.data_ptr(), dy.data_ptr(), x.shape[0], x.shape[1], _f32c(alpha).data_ptr(), _f32c(beta).data_ptr(),
                                       dx.data_ptr(), da.data_ptr(), db.data_ptr(), _stream()), "jatts_snakebeta_bwd")
    return dx, da, db


def ctc_forward_sum(log_p, ilens, olens, log_blank, want_grad=True, grad_scale=1.0):
    """log_p: (B, T, N) f32 (prior already added; entries beyond ilens / olens are ignored).  -> (nll (B,) = -log p / ilens, grad or None)."""
    lib = _abi.load()
    log_p = _f32c(log_p)
    B, T, ld = log_p.shape
    max_i = int(ilens.max())
    ws = torch.empty(2 * B * T * (2 * max_i + 1), dtype=torch.float32, device=log_p.device)
    nll = torch.empty(B, dtype=torch.float32, device=log_p.device)
    grad = torch.empty_like(log_p) if want_grad else None
    # (cached pinned uploads: .to(device) of a pageable CPU tensor blocks the host until everything queued so far -- the whole forward --
    # has run: 50 ms per VITS step)
    _up = lambda t: t.to(torch.int32).contiguous() if t.is_cuda else h2d([int(v) for v in t.tolist()], torch.int32, log_p.device)  # noqa: E731
    il, ol = _up(ilens), _up(olens)
    _abi.check(lib.jatts_ctc_forward_sum(log_p.data_ptr(), B, T, ld, il.data_ptr(), ol.data_ptr(), max_i, float(log_blank), ws.data_ptr(),
                                         nll.data_ptr(), _ptr(grad), float(grad_scale), _stream()), "jatts_ctc_forward_sum")
    return nll, grad


def seq_sum(rb, x):
    """-> (n_seq, dim) f32: per-sequence column sums."""
    lib = _abi.load()
    x = _f32c(x)
    out = _zeros((rb.n_seq, x.shape[1]), x.device)
    rg = rb.struct()
    _ws(x.device)
    _ws_check(lib.jatts_seq_sum(C.byref(rg), x.data_ptr(), x.shape[1], out.data_ptr(), _stream()), "jatts_seq_sum")
    return out


def bgemm(a, b, trans_a=False, trans_b=False, alpha=1.0, out=None):
    """jatts_bgemm: C[o][i] = alpha * op(a[o][i]) @ op(b[o][i]) on the exact-f32 matrix pipe (the training step's attention products).
    a, b: f32 device tensors of 4 dims (O, I, rows, cols) -- any strides on the two batch dims, rows contiguous -- or 3 dims (I, rows, cols)
    = shared over O.  trans_a: a holds (k x m); trans_b: b holds (n x k).  -> (O, I, m, n) contiguous."""
    lib = _abi.load()
    a, b = _dev(a), _dev(b)

    def geom(t):
        if t.dtype != torch.float32 or t.dim() not in (3, 4) or t.stride(-1) != 1:
            raise ValueError("bgemm: f32 tensors of 3 / 4 dims with contiguous rows")
        if t.dim() == 3:
            return None, t.shape[0], 0, t.stride(0), t.stride(1)
        return t.shape[0], t.shape[1], t.stride(0), t.stride(1), t.stride(2)
    ao, ai, sao, sai, lda = geom(a)
    bo, bi, sbo, sbi, ldb = geom(b)
    O = ao if ao is not None else bo
    if O is None:
        O = 1
    if ai != bi or (ao is not None and ao != O) or (bo is not None and bo != O):
        raise ValueError("bgemm: batch dims differ")
    m, k = (a.shape[-1], a.shape[-2]) if trans_a else (a.shape[-2], a.shape[-1])
    n, kb = (b.shape[-2], b.shape[-1]) if trans_b else (b.shape[-1], b.shape[-2])
    if k != kb:
        raise ValueError("bgemm: contraction sizes differ")
    if out is None:
        out = torch.empty(O, ai, m, n, dtype=torch.float32, device=a.device)
    elif (not out.is_cuda or out.dtype != torch.float32 or out.dim() != 4 or tuple(out.shape) != (O, ai, m, n) or out.stride(3) != 1
          or out.device != a.device):
        raise ValueError(f"bgemm: out must be an f32 device tensor of shape {(O, ai, m, n)} with contiguous rows")
    _count(2.0 * O * ai * m * n * k)
    with _Timed("bgemm", (O * ai, m, n, k)):
        _abi.check(lib.jatts_bgemm(a.data_ptr(), sao, sai, lda, int(trans_a), b.data_ptr(), sbo, sbi, ldb, int(trans_b), out.data_ptr(), out.stride(0),
                                   out.stride(1), out.stride(2), O, ai, m, n, k, float(alpha), 0, _stream()), "jatts_bgemm")
    return out


def mfma_ceiling(dtype=F32E, feed=1, target_ms=60.0, device=None, seed=0):
    """jatts_mfma_probe: the matrix pipe's sustained rate on this part for `dtype` -- nothing but MFMAs (2 x 2 fragments per wave, two 4-wave workgroups per
    CU), operands re-read from LDS every K-step (feed=1) or held in registers (feed=0), on N(0, 1) operand bits in the form the kernels see them (bf16: the
    three exact terms of the emulated arithmetic in equal parts).  A calibration launch sizes the timed one to ~target_ms: long enough for the clock to
    settle at the power limit, short enough to sit in front of a bench run.  -> dict(tflops, ms, clock_ghz, iters, workgroups)."""
    lib = _abi.load()
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    g = torch.Generator().manual_seed(seed)
    v = torch.randn(1 << 16, generator=g)
    if dtype in EMUL or dtype == 16 + F32E:
        ops = torch.cat([t.reshape(-1) for t in bf16x3_terms(v)])
    elif dtype in (F16, F32S, 16 + F16):
        ops = v.half()
    else:
        ops = v
    ops = ops.to(dev).contiguous()
    clocks = torch.zeros(2, dtype=torch.int64, device=dev)
    sink = torch.zeros(1, dtype=torch.float32, device=dev)
    wgs = 2 * torch.cuda.get_device_properties(dev).multi_processor_count

    def run(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _abi.check(lib.jatts_mfma_probe(dtype, feed, ops.data_ptr(), ops.numel() * ops.element_size(), iters, wgs, clocks.data_ptr(), sink.data_ptr(),
                                        _stream()), "jatts_mfma_probe")
        b.record()
        b.synchronize()
        return a.elapsed_time(b)

    it = 2000
    ms = run(it)
    it = max(2000, min(int(it * target_ms / max(ms, 1e-3)) & ~1, 1 << 24))
    run(it)                              # settles the clock at the sustained level
    ms = run(it)
    c = clocks.tolist()
    return dict(tflops=lib.jatts_mfma_probe_flops(dtype, it, wgs) / ms / 1e9, ms=ms, clock_ghz=c[0] / (c[1] * 10.0) if c[1] else None, iters=it, workgroups=wgs,
                feed="lds" if feed else "registers")


_MARKER = {}


def mfma_marker(device=None):
    """One 1-workgroup, 2-K-step jatts_mfma_probe launch: a marker that tools (bench.py's B = 1 kernel-trace child) find by name in a rocprofv3 trace."""
    lib = _abi.load()
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    key = str(dev)
    if key not in _MARKER:
        _MARKER[key] = (torch.zeros(1 << 15, dtype=torch.bfloat16, device=dev), torch.zeros(1, dtype=torch.float32, device=dev))
    ops, sink = _MARKER[key]
    _abi.check(lib.jatts_mfma_probe(F32E, 0, ops.data_ptr(), ops.numel() * 2, 2, 1, None, sink.data_ptr(), _stream()), "jatts_mfma_probe")
