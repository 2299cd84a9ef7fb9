// Generic Conv1d / Linear / polyphase ConvTranspose1d kernel (MFMA implicit GEMM over chunked LDS activation tiles).
// One translation unit per dtype instantiates it (conv1d_f16.hip, conv1d_f32.hip) so the builds run in parallel.
#pragma once
#include "conv_tiles.h"

namespace {

// ------------------------------------------------------------------ XCD-aware work order (shared by both conv kernels)
// The dispatcher places workgroup id on XCD id % 8, each XCD with its own 4 MiB L2.  In the natural (time tile, sequence, n-block) grid
// the n-blocks that share an activation tile are a whole (tiles x sequences) plane of ids apart, land on every XCD and re-read the
// tile through L2 misses: 512 -> 2048 k1 moved 2.2 GB from the fabric per launch for a 100 MB input.  Here XCD x owns the contiguous
// tile range [x * tpx, (x + 1) * tpx) (tile = (time tile, sequence)) and walks it as  for n-block chunk: for tile: for n-block in
// chunk  with chunks of <= ~2 MiB of weights (they stay in that L2), evened out so that almost no workgroup finds an empty slot
// (a workgroup that exits at once still perturbs the placement of the others).  4.6x less fabric traffic, +3-12 % on the wide f32
// shapes -- on a power-limited kernel fabric bytes are clock (profiles/r03_notes.md).
struct XcdOrder {
  int gx, n_tiles, tpx, gz, zc;   // gx == 0: the tiles are the REAL tiles of a ragged batch (1-D, jatts_ragged.host_lens), located from cu_rows
  int64_t plan(int gx_, int n_seq, int gz_, int64_t slice_bytes, int64_t tiles_1d = 0) {
    gx = tiles_1d > 0 ? 0 : gx_; gz = gz_;
    n_tiles = tiles_1d > 0 ? (int)tiles_1d : gx_ * n_seq;
    tpx = (n_tiles + 7) / 8;
    zc = (int)((2 << 20) / (slice_bytes > 0 ? slice_bytes : 1));
    zc = zc < 1 ? 1 : (zc > gz ? gz : zc);
    const int n_chunks = (gz + zc - 1) / zc;
    zc = (gz + n_chunks - 1) / n_chunks;
    return 8 * (int64_t)tpx * zc * n_chunks;     // workgroups of the 1-D grid
  }
  // -> (time tile bx of sequence by, n-block bz); `bt` = rows per time tile.  All 64 lanes active (ragged_locate).
  __device__ __forceinline__ bool decode(unsigned id, int& bx, int& by, int& bz, const jatts_ragged& rg, int bt) const {
    const int xcd = (int)(id & 7u), m = (int)(id >> 3);
    const int per_chunk = tpx * zc;
    const int c = m / per_chunk, rem = m - c * per_chunk;
    const int tl = rem / zc;
    const int tile = xcd * tpx + tl;
    bz = c * zc + (rem - tl * zc);
    if (tile >= n_tiles || bz >= gz) return false;
    if (gx == 0) return ragged_locate(rg, bt, (unsigned)tile, by, bx);
    by = tile / gx;
    bx = tile - by * gx;
    return true;
  }
};

// Two outputs from one launch (jatts_conv_desc.n_split > 0: the Q | K | V projection): a workgroup whose channel slab starts at or beyond n_split
// writes the SECOND output -- transposed, through the generic fragment-order epilogue -- by pointing its private copy of the descriptor at it, shifted
// so that the absolute channel index n addresses row n - n_split.  n_split is a multiple of every kernel's slab (256), so a slab never straddles it.
__device__ __forceinline__ void conv_second_output(jatts_conv_desc& d, int n_first) {
  if (d.n_split > 0 && n_first >= d.n_split) {
    const size_t esz = (d.y_is_f32 || d.dtype != JATTS_F16) ? 4 : 2;
    d.y = (char*)d.y2 - (size_t)d.n_split * (size_t)d.ldy2 * esz;
    d.ldy = d.ldy2;
    d.y_transposed = 1;
    d.y_seq_col0 = d.y2_seq_col0;
  } else if (d.n_split > 0) {
    d.n_out = d.n_split;      // the first output's rows are n_split wide: the row-major epilogues bound their stores (and the buffer range check that drops
  }                           // the rows past a sequence's end) by n_out.  Callers have taken the packed-weight geometry (NFR) from the full n_out BEFORE this.
}

// ------------------------------------------------------------------ generic conv kernel
constexpr int KCH = 64;  // channels staged per LDS chunk

// Conv epilogue: y = act(acc + bias) * alpha + resid.  Lane owns column (lane&31) and the channel quads
// n0 + {0..3}; quads fully inside n_out take the 16-byte bias / residual / store path, the ragged last quad
// (n_out % 4 != 0) a scalar loop.
template <typename T, int ACT, int NF, int NT>
__device__ __forceinline__ void conv_epilogue(const jatts_conv_desc& d, f32x16 (&acc)[NF][NT], int t0, int col0, int nf0,
                                              int lane, int L, int64_t seq_row0, int seq) {
  const int g = lane >> 5;
  const bool vec_r = d.resid && (d.ldr & 3) == 0, vec_y = (d.ldy & 3) == 0 && !d.y_transposed;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int pos = t0 + col0 + t * 32 + (lane & 31);
    if (pos >= L) continue;
    const int64_t row = seq_row0 + pos;
    const int64_t trow = d.y_seq_col0 ? (int64_t)d.y_seq_col0[seq] * d.rg.len_mul + pos : row;  // transposed output column
#pragma unroll
    for (int f = 0; f < NF; ++f) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n0 = (nf0 + f) * 32 + 8 * q + 4 * g;
        if (n0 >= d.n_out) continue;
        const bool full = n0 + 3 < d.n_out;
        f32x4 rq = {0.f, 0.f, 0.f, 0.f};   // (the bias is already in the accumulators: conv1d_kernel's init)
        if (full && vec_r) rq = *reinterpret_cast<const f32x4*>(d.resid + row * d.ldr + n0);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = act_c<ACT>(acc[f][t][4 * q + e]) * d.alpha + rq[e];
        if (full && vec_y && (vec_r || !d.resid)) {
          const int64_t o = row * d.ldy + n0;
          if (d.y_is_f32 || sizeof(T) == 4) *reinterpret_cast<f32x4*>((float*)d.y + o) = v;
          else *reinterpret_cast<f16x4*>((f16*)d.y + o) = f16x4{(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int n = n0 + e;
            if (n >= d.n_out) break;
            float s = v[e];
            if (!full) s = act_c<ACT>(acc[f][t][4 * q + e]) * d.alpha;
            if (d.resid && !(full && vec_r)) s += d.resid[row * d.ldr + n];
            const int64_t o = d.y_transposed ? (int64_t)n * d.ldy + trow : row * d.ldy + n;
            if (d.y_is_f32) ((float*)d.y)[o] = s; else ((T*)d.y)[o] = from_f32<T>(s);
          }
        }
      }
      // keep the epilogue's live ranges short: without this hipcc hoists every bias / residual
      // load of the tile to the top and the kernel loses a wave of occupancy
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// Coalesced epilogue for row-major outputs (projections, FFN, the polyphase upsampling convs): act(acc + bias) * alpha
// is assembled as a [BT][BN] tile of the OUTPUT type in LDS (the activation buffers are dead), then the f32 residual
// is added and the tile written with row-contiguous 16-byte accesses.  In fragment order every 128-byte output line
// is otherwise hit by 8 separate 8-byte stores (and residual loads) from lanes 32 rows apart.
template <typename T, typename TO, int ACT, int NF, int NT, int BN>
__device__ __forceinline__ void conv_epilogue_lds(const jatts_conv_desc& d, f32x16 (&acc)[NF][NT], char* smem, int t0,
                                                  int col0, int nf_local0, int n_base, int lane, int L,
                                                  int64_t seq_row0, int BT) {
  constexpr int opitch = BN * (int)sizeof(TO) + 16;
  const int g = lane >> 5;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = col0 + t * 32 + (lane & 31);
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int nl = (nf_local0 + f) * 32 + 8 * q + 4 * g;   // channel inside the workgroup's BN slab
        if (n_base + nl >= d.n_out) continue;                   // n_out % 8 == 0: quads are all-or-nothing
        f32x4 o;   // the bias is already in the accumulators
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = act_c<ACT>(acc[f][t][4 * q + e]) * d.alpha;
        char* p = smem + (size_t)col * opitch + (size_t)nl * sizeof(TO);
        if (sizeof(TO) == 2) *reinterpret_cast<f16x4*>(p) = f16x4{(f16)o[0], (f16)o[1], (f16)o[2], (f16)o[3]};
        else *reinterpret_cast<f32x4*>(p) = o;
      }
  }
  __syncthreads();
  const int vrows = min(BT, L - t0);
  const int upr = min(BN, d.n_out - n_base) / 8;   // 8-element units per row
  const int total = vrows * upr;
  TO* yg = (TO*)d.y + (seq_row0 + t0) * (int64_t)d.ldy + n_base;
  const float* rg = d.resid ? d.resid + (seq_row0 + t0) * (int64_t)d.ldr + n_base : nullptr;
  for (int u = threadIdx.x; u < total; u += blockDim.x) {
    const int r = u / upr, cu = u - r * upr;
    const char* src = smem + (size_t)r * opitch + (size_t)cu * 8 * sizeof(TO);
    TO* dst = yg + (int64_t)r * d.ldy + cu * 8;
    if (sizeof(TO) == 2) {
      *reinterpret_cast<f16x8*>(dst) = *reinterpret_cast<const f16x8*>(src);   // (no residual on this path)
    } else {
      f32x4 lo = *reinterpret_cast<const f32x4*>(src), hi = *reinterpret_cast<const f32x4*>(src + 16);
      if (rg) {
        const float* rp = rg + (int64_t)r * d.ldr + cu * 8;
        lo += *reinterpret_cast<const f32x4*>(rp);
        hi += *reinterpret_cast<const f32x4*>(rp + 4);
      }
      *reinterpret_cast<f32x4*>(dst) = lo;
      *reinterpret_cast<f32x4*>((float*)dst + 4) = hi;
    }
  }
}

template <typename T, int NF, int NT, int WN, int WT, int NIN, bool ASYNC, int KCHT = KCH>
// two workgroups per CU (<= 256 registers) for the single-input pipelines: one workgroup's LDS commits, barriers and epilogue
// then run under the other's MFMAs (the f32 kernel at 268 registers had the CU to itself and sat at ~55 % MFMA utilisation)
__global__ __launch_bounds__(WN* WT * 64, (KCHT == 128 || (sizeof(T) == 4 && NIN == 1 && ASYNC && WT * NT * 32 <= 128)) ? 2 : 1) void conv1d_kernel(jatts_conv_desc d, int f32_tile, XcdOrder xo) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BT = WT * NT * 32;
  int bx, b, bz;
  if (!xo.decode(blockIdx.x, bx, b, bz, d.rg, BT)) return;
  const int row_b = d.rg.cu_rows[b];
  const int L = (d.rg.cu_rows[b + 1] - row_b) * d.rg.len_mul;
  const int t0 = bx * BT;
  if (t0 >= L) return;
  const int64_t seq_row0 = (int64_t)row_b * d.rg.len_mul;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wn = wave / WT, wt = wave % WT;
  const int pitch = KCHT * (int)sizeof(T) + 16;
  const int rows = BT + (d.k_w - 1) * d.dil;
  const int KC16 = d.c_in >> 4;
  const int n_pad = (d.n_out + 31) & ~31;
  const int NFR = n_pad >> 5;
  const int nf0 = (bz * WN + wn) * NF;
  const int col0 = wt * NT * 32;
  conv_second_output(d, bz * WN * NF * 32);

  const T* xin[3] = {(const T*)d.x[0], (const T*)d.x[1], (const T*)d.x[2]};
  const bool reflect = d.pad_mode == JATTS_PAD_REFLECT;
  f32x16 acc[NF][NT];
  zero_acc<NF, NT>(acc);
  if (d.bias) {   // accumulators start at the bias: its loads overlap the first staging round trip instead of the epilogue
    const int gq = lane >> 5;
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n0 = (nf0 + f) * 32 + 8 * q + 4 * gq;
        f32x4 bq = {0.f, 0.f, 0.f, 0.f};
        if (n0 + 3 < d.n_out) bq = *reinterpret_cast<const f32x4*>(d.bias + n0);
        else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n0 + e < d.n_out) bq[e] = d.bias[n0 + e];
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[f][t][4 * q + e] = bq[e];
      }
  }

  // staged 8-element units per thread in the async pipeline (halo <= 32 rows; larger halos take
  // the synchronous single-buffer path)
  constexpr int UPRC = KCHT / 8;
  constexpr int MAXU = ((BT + 32) * UPRC + WN * WT * 64 - 1) / (WN * WT * 64);
  // ring depth: k_w * (KCHT/16) is always a multiple of it.  f32: a step is 32 x 64-cycle MFMAs, two steps of look-ahead
  // cover any L2 round trip and free 32 registers
  constexpr int RD = sizeof(T) == 4 ? 2 : KCHT / 16;
  WRing<T, NF, RD> ring;
  const int n_chunks = d.c_in / KCHT;
  ring.init((const T*)d.w, KC16, NFR, nf0, d.k_w, KCHT / 16, n_chunks, lane);
  if constexpr (ASYNC) {
    const size_t buf_bytes = (size_t)rows * pitch;
    StageRegs<T, MAXU, NIN> sr;
    stage_issue<T, MAXU, NIN, UPRC, WN * WT * 64>(sr, rows, t0 - d.pad, L, seq_row0, xin, d.n_in, d.ldx, 0, reflect);
    stage_commit<T, MAXU, NIN, UPRC, WN * WT * 64>(sr, smem, pitch, rows, d.n_in, d.in_scale, d.pre_act, d.pre_slope);
    __syncthreads();
    for (int ci = 0; ci < n_chunks; ++ci) {
      const bool more = ci + 1 < n_chunks;
      if (more) stage_issue<T, MAXU, NIN, UPRC, WN * WT * 64>(sr, rows, t0 - d.pad, L, seq_row0, xin, d.n_in, d.ldx, (ci + 1) * KCHT, reflect);
      conv_stage<T, NF, NT, RD>(acc, ring, KCHT / 16, d.k_w, d.dil, smem + (size_t)(ci & 1) * buf_bytes, pitch, col0,
                                lane);
      if (more) stage_commit<T, MAXU, NIN, UPRC, WN * WT * 64>(sr, smem + (size_t)((ci + 1) & 1) * buf_bytes, pitch, rows, d.n_in,
                                           d.in_scale, d.pre_act, d.pre_slope);
      __syncthreads();
    }
  } else {
    for (int ci = 0; ci < n_chunks; ++ci) {
      stage_rows<T>(smem, pitch, rows, KCHT, t0 - d.pad, L, seq_row0, xin, d.n_in, d.ldx, ci * KCHT, d.in_scale,
                    d.pre_act, d.pre_slope, reflect);
      __syncthreads();
      conv_stage<T, NF, NT, RD>(acc, ring, KCHT / 16, d.k_w, d.dil, smem, pitch, col0, lane);
      __syncthreads();
    }
  }

  // epilogue, specialised per activation by ONE uniform branch: a runtime switch inside the 64-element
  // unrolled body inlined tanh/mish 128 times, the unroller gave up and the accumulators went to scratch
  // (1.8x slower conv, profiles/r01_notes.md).
  if (d.act == JATTS_ACT_SNAKEBETA) snake_acc<NF, NT>(acc, d.act_a, d.act_b, nf0, d.n_out, lane);
  {
    // the loop's last barrier has retired every read of the activation buffers: reuse them as the output tile
    constexpr int BN = WN * NF * 32;
    const int n_base = bz * BN;
    const bool rowmajor = !d.y_transposed && (d.n_out & 7) == 0 && (reinterpret_cast<uintptr_t>(d.y) & 15) == 0;
#define JATTS_EPI(TO)                                                                                                  \
  switch (d.act) {                                                                                                     \
    case JATTS_ACT_RELU: conv_epilogue_lds<T, TO, JATTS_ACT_RELU, NF, NT, BN>(d, acc, smem, t0, col0, wn * NF, n_base, lane, L, seq_row0, BT); break;  \
    case JATTS_ACT_TANH: conv_epilogue_lds<T, TO, JATTS_ACT_TANH, NF, NT, BN>(d, acc, smem, t0, col0, wn * NF, n_base, lane, L, seq_row0, BT); break;  \
    case JATTS_ACT_SWISH: conv_epilogue_lds<T, TO, JATTS_ACT_SWISH, NF, NT, BN>(d, acc, smem, t0, col0, wn * NF, n_base, lane, L, seq_row0, BT); break; \
    case JATTS_ACT_MISH: conv_epilogue_lds<T, TO, JATTS_ACT_MISH, NF, NT, BN>(d, acc, smem, t0, col0, wn * NF, n_base, lane, L, seq_row0, BT); break;  \
    default: conv_epilogue_lds<T, TO, JATTS_ACT_NONE, NF, NT, BN>(d, acc, smem, t0, col0, wn * NF, n_base, lane, L, seq_row0, BT); break;              \
  }
    if constexpr (sizeof(T) == 2) {
      if (rowmajor && !d.y_is_f32 && !d.resid && (d.ldy & 7) == 0) {
        JATTS_EPI(T)
        return;
      }
    }
    // f32 rows (f16 kernels writing the f32 residual streams, and every row-major output of the f32 kernels): in fragment
    // order a lane's 16-byte store lands 32 rows from its neighbour's, so each output row gets 32-byte pieces
    if (rowmajor && (d.y_is_f32 || sizeof(T) == 4) && f32_tile && (d.ldy & 3) == 0 &&
        (!d.resid || ((d.ldr & 3) == 0 && (reinterpret_cast<uintptr_t>(d.resid) & 15) == 0))) {
      JATTS_EPI(float)
      return;
    }
#undef JATTS_EPI
  }
  switch (d.act) {
    case JATTS_ACT_RELU: conv_epilogue<T, JATTS_ACT_RELU, NF, NT>(d, acc, t0, col0, nf0, lane, L, seq_row0, b); break;
    case JATTS_ACT_TANH: conv_epilogue<T, JATTS_ACT_TANH, NF, NT>(d, acc, t0, col0, nf0, lane, L, seq_row0, b); break;
    case JATTS_ACT_SWISH: conv_epilogue<T, JATTS_ACT_SWISH, NF, NT>(d, acc, t0, col0, nf0, lane, L, seq_row0, b); break;
    case JATTS_ACT_MISH: conv_epilogue<T, JATTS_ACT_MISH, NF, NT>(d, acc, t0, col0, nf0, lane, L, seq_row0, b); break;
    default: conv_epilogue<T, JATTS_ACT_NONE, NF, NT>(d, acc, t0, col0, nf0, lane, L, seq_row0, b); break;
  }
}

template <typename T, int NF, int NT, int WN, int WT, int NIN, bool ASYNC, int KCHT = KCH>
int launch_conv_k(const jatts_conv_desc& d, hipStream_t s) {
  constexpr int BT = WT * NT * 32, BN = WN * NF * 32;
  const int64_t maxL = (int64_t)d.rg.max_len * d.rg.len_mul;
  dim3 grid((unsigned)((maxL + BT - 1) / BT), (unsigned)d.rg.n_seq, (unsigned)((d.n_out + BN - 1) / BN));
  const size_t rows = (size_t)BT + (size_t)(d.k_w - 1) * d.dil;
  size_t lds = (ASYNC ? 2 : 1) * rows * (KCHT * sizeof(T) + 16);
  // output tile of the coalesced epilogues (f16 kernels): T-typed always, f32 (row-major f32 outputs, e.g. the
  // in-place residual-stream updates) when the conv is long enough that the bigger LDS footprint does not matter
  int f32_tile = 0;
  if (sizeof(T) == 4) {   // f32 kernels: coalesced f32 output tile whenever it fits beside nothing else (the staging buffers are dead by then)
    if (!d.y_transposed && (size_t)BT * (BN * 4 + 16) <= 160 * 1024) {
      f32_tile = 1;
      if (lds < (size_t)BT * (BN * 4 + 16)) lds = (size_t)BT * (BN * 4 + 16);
    }
  }
  if (sizeof(T) == 2) {
    if (lds < (size_t)BT * (BN * sizeof(T) + 16)) lds = (size_t)BT * (BN * sizeof(T) + 16);
    if (d.y_is_f32 && !d.y_transposed) {
      f32_tile = 1;
      if (lds < (size_t)BT * (BN * 4 + 16)) lds = (size_t)BT * (BN * 4 + 16);
    }
  }
  if (lds > 160 * 1024) return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "conv1d: halo too large for LDS");
  auto kern = conv1d_kernel<T, NF, NT, WN, WT, NIN, ASYNC, KCHT>;
  // the dynamic-LDS limit ONCE per kernel (function-local static of this template instantiation), not per launch: the per-launch call was seen to stall the
  // host for ~20 ms now and then (profiles/r06_notes.md section 8)
  static const hipError_t lds_attr = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (lds_attr != hipSuccess) return jatts_set_error(lds_attr, __FILE__, __LINE__);
  XcdOrder xo;
  const int64_t total = xo.plan((int)grid.x, (int)grid.y, (int)grid.z, (int64_t)BN * d.c_in * d.k_w * (int64_t)sizeof(T), ragged_tiles_1d(d.rg, BT));
  if (total >= (int64_t)1 << 31) return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "conv1d: launch too large");
  hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(WN * WT * 64), lds, s, d, f32_tile, xo);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

template <typename T, int NF, int NT, int WN, int WT>
int launch_conv(const jatts_conv_desc& d, hipStream_t s) {
  const bool small_halo = (d.k_w - 1) * d.dil <= 32;
  const bool multi_chunk = d.c_in > KCH;  // a single chunk has nothing to overlap with
  if constexpr (sizeof(T) == 2 && NF == 2 && NT == 2 && WN == 2) {
    // 128-channel chunks for deep-K convs: half as many stage -> barrier -> MFMA round trips per workgroup (a k=1
    // projection with K=384 is otherwise 6 latency-bound chunk iterations around 2 us of MFMA work)
    static const int kch = [] { const char* e = getenv("JATTS_CONV_KCH"); return e ? atoi(e) : 2; }();  // 0: never, 1: k=1 only, 2: all (default)
    if (small_halo && d.n_in == 1 && d.c_in % 128 == 0 && d.c_in >= 256 && (kch == 2 || (kch == 1 && d.k_w == 1)))
      return launch_conv_k<T, NF, NT, WN, WT, 1, true, 128>(d, s);
  }
  if (small_halo && multi_chunk && d.n_in == 1) return launch_conv_k<T, NF, NT, WN, WT, 1, true>(d, s);
  if constexpr (sizeof(T) == 2) {
    if (small_halo && multi_chunk) return launch_conv_k<T, NF, NT, WN, WT, 3, true>(d, s);
  }
  return launch_conv_k<T, NF, NT, WN, WT, 3, false>(d, s);
}

}  // namespace
