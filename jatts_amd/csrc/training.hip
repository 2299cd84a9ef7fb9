// Training-side kernels (first slice): the loss sums of the FastSpeech2 criterion and the weight / bias
// gradients of jatts_conv1d.  The data gradient of a conv is jatts_conv1d itself on flipped, transposed weights.
// Reference: jatts/trainers/fastspeech2.py:24-100 (_train_step), jatts/losses/{l1l2_loss,duration_predictor_loss,
// variance_predictor_loss}.py, torch.nn.Conv1d's autograd.
#include <stdlib.h>

#include "common.h"
#include "det_reduce.h"

namespace {

// sum over valid rows (t < valid_len[b]) and the first `dim` columns of |a - b'| (kind 0) or (a - b')^2 (kind 1);
// b' = b, or log(b + log_offset) when log_offset >= 0 (DurationPredictorLoss: targets in the log domain)
__global__ __launch_bounds__(256) void masked_loss_partial_kernel(jatts_ragged rg, const float* a, int lda, const float* b, int ldb,
                                                                 int dim, const int32_t* valid_len, int kind, float log_offset,
                                                                 double* part) {
  __shared__ double red[256];
  const int s = blockIdx.y;
  const int row0 = rg.cu_rows[s], L = rg.cu_rows[s + 1] - row0;
  const int v = valid_len ? min(max(valid_len[s], 0), L) : L;
  const int64_t n = (int64_t)v * dim;
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t t = i / dim;
    const int c = (int)(i - t * dim);
    float y = b[(row0 + t) * ldb + c];
    if (log_offset >= 0.f) y = logf(y + log_offset);
    const float dlt = a[(row0 + t) * lda + c] - y;
    acc += kind == 0 ? (double)fabsf(dlt) : (double)dlt * (double)dlt;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.y * gridDim.x + blockIdx.x] = red[0];
}
__global__ void fold_loss_kernel(const double* part, int n, double scale, float* out) {
  double s = 0.0;
  for (int i = 0; i < n; ++i) s += part[i];
  *out = (float)(s * scale);
}

// dW[n][c][tap] += sum over sequences, t of dy[t][n] * x[t + tap*dil - pad][c] (x outside its sequence = 0).
// One workgroup = a 64(n) x 64(c) tile of one tap over a slice of the sequences; a thread owns 4 x 4 outputs; dy / x tiles of
// 32 time steps go through LDS; per-group partial tiles go to the workspace and are summed in a fixed order by wgrad_reduce_kernel
// (f32 FMA at the f32-MFMA rate on gfx950).
constexpr int WG_TT = 32;
__global__ __launch_bounds__(256) void zero_fill_kernel(float* __restrict__ p, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = 0.f;
}

__global__ __launch_bounds__(256) void conv_wgrad_kernel(jatts_ragged rg, const float* x, int ldx, const float* dy, int ldy, int c_in,
                                                         int n_out, int k_w, int dil, int pad, int seq_groups, float* dw, int overwrite,
                                                         float* __restrict__ slabs, unsigned* __restrict__ tickets) {
  __shared__ float dys[WG_TT][64 + 4];
  __shared__ __attribute__((aligned(16))) float xs[WG_TT][64 + 4];
  const int n0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tap = blockIdx.z % k_w, grp = blockIdx.z / k_w;
  const int tn = threadIdx.x >> 4, tc = threadIdx.x & 15;   // 16 x 16 threads, 4 x 4 outputs each
  float acc[4][4] = {};
  for (int s = grp; s < rg.n_seq; s += seq_groups) {
    const int row0 = rg.cu_rows[s] * rg.len_mul, L = (rg.cu_rows[s + 1] - rg.cu_rows[s]) * rg.len_mul;
    for (int t0 = 0; t0 < L; t0 += WG_TT) {
      for (int i = threadIdx.x; i < WG_TT * 64; i += 256) {
        const int t = i >> 6, j = i & 63;
        const int tt = t0 + t, p = tt + tap * dil - pad;
        dys[t][j] = (tt < L && n0 + j < n_out) ? dy[(int64_t)(row0 + tt) * ldy + n0 + j] : 0.f;
        xs[t][j] = (tt < L && p >= 0 && p < L && c0 + j < c_in) ? x[(int64_t)(row0 + p) * ldx + c0 + j] : 0.f;
      }
      __syncthreads();
#pragma unroll 4
      for (int t = 0; t < WG_TT; ++t) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(&dys[t][4 * tn]);
        const f32x4 b = *reinterpret_cast<const f32x4*>(&xs[t][4 * tc]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] += a[i] * b[j];
      }
      __syncthreads();
    }
  }
  // deterministic split-K (det_reduce.h): group = (n tile, c tile, tap), parts = the sequence groups, slab = the 64 x 64 partial tile
  const int tile = (tap * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  float* gslab = slabs + (int64_t)tile * seq_groups * 4096;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    *reinterpret_cast<f32x4*>(gslab + (int64_t)grp * 4096 + (4 * tn + i) * 64 + 4 * tc) = f32x4{acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
  if (det_arrive(tickets + tile, seq_groups, reinterpret_cast<unsigned*>(&dys[0][0])))
    det_sum_slabs<4>(gslab, seq_groups, 4096, &xs[0][0], WG_TT * 68, [&](int e, float t) {
      const int n = n0 + (e >> 6), c = c0 + (e & 63);
      if (n < n_out && c < c_in) {
        float* o = &dw[((int64_t)n * c_in + c) * k_w + tap];
        *o = overwrite ? t : *o + t;
      }
    });
}

// The same reduction on the f32 MFMA pipe (v_mfma_f32_32x32x2_f32: A = dy^T fragment [32 n][2 t], B = x fragment [2 t][32 c], the
// contraction runs over TIME).  In the time-major layout both operands of a 2-step MFMA are two coalesced 128-byte row pieces, so the
// fragments come straight out of row-major LDS tiles with one ds_read_b32 each.  A workgroup (4 waves, one 32 x 32 fragment each)
// owns a 64(n) x 64(c) tile for ALL taps (accumulators [KW] x 16 registers): dy and x are staged once per 32-step chunk instead of
// once per tap.  Split over the sequences like the VALU kernel; partial tiles go to the workspace (wgrad_reduce_kernel sums them, no atomics).
#ifndef JATTS_WGRAD_UNROLL
#define JATTS_WGRAD_UNROLL 8   // (4 / 8 / 16 measured: 848 / 833 / 814 us at 1536 x 384 k3, 360 / 350 / 376 us at 512 x 512 k3)
#endif
template <int KW>
__global__ __launch_bounds__(256) void conv_wgrad_mfma_kernel(jatts_ragged rg, const float* __restrict__ x, int ldx, const float* __restrict__ dy,
                                                              int ldy, int c_in, int n_out, int dil, int pad, int seq_groups,
                                                              float* __restrict__ dw, float* __restrict__ ws, float* __restrict__ bws) {
  constexpr int TT = 32, P = 68;
  extern __shared__ float sm[];
  const int halo = (KW - 1) * dil;
  const int n0 = blockIdx.x * 64, c0 = blockIdx.y * 64, grp = blockIdx.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wn = wave >> 1, wc = wave & 1, hi = lane >> 5, lo = lane & 31;
  const bool vec_dy = (ldy & 3) == 0 && n0 + 64 <= n_out && (reinterpret_cast<uintptr_t>(dy) & 15) == 0;
  const bool vec_x = (ldx & 3) == 0 && c0 + 64 <= c_in && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
  f32x16 acc[KW];
#pragma unroll
  for (int k = 0; k < KW; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
  // Software pipeline over the (sequence, 32-step chunk) pairs of this group (round 3): the global loads of chunk i + 1 are issued into
  // registers BEFORE the MFMAs of chunk i and committed to the other LDS buffer after them -- one barrier per chunk and no exposed load
  // latency (the synchronous load -> store -> barrier -> MFMA -> barrier form left a k = 1 chunk of 1 024 MFMA cycles waiting ~2 us
  // for its operands: 43-67 TFLOP/s).
  constexpr int NDY = TT * 16 / 256, NXV = (TT + 32) * 16 / 256;      // f32x4 per thread: dy tile, x tile (halo <= 32)
  f32x4 rdy[NDY], rx[NXV];
  const int nx_rows = TT + halo;
  // bias gradient (column sums of dy) for free: the c-tile-0 workgroups already hold every dy tile of their n-tile in registers on its
  // way to LDS; a thread's NDY vectors share one 4-column group, so it keeps a running f32x4 and the group's 16 threads are folded at
  // the end.  Per-group partials go to bws[grp][n], summed with the weight partials by wgrad_reduce_kernel (deterministic, no atomics).
  const bool do_b = bws != nullptr && blockIdx.y == 0;
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
  auto issue = [&](int s, int t0) {
    const int64_t row0 = (int64_t)rg.cu_rows[s] * rg.len_mul;
    const int L = (rg.cu_rows[s + 1] - rg.cu_rows[s]) * rg.len_mul;
#pragma unroll
    for (int j = 0; j < NDY; ++j) {
      const int i = threadIdx.x + 256 * j;
      const int r = i >> 4, q = (i & 15) * 4;
      const int tt = t0 + r;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (tt < L) {
        const float* src = dy + (row0 + tt) * ldy + n0 + q;
        if (vec_dy) v = *reinterpret_cast<const f32x4*>(src);
        else
#pragma unroll
          for (int e = 0; e < 4; ++e) if (n0 + q + e < n_out) v[e] = src[e];
      }
      rdy[j] = v;
    }
#pragma unroll
    for (int j = 0; j < NXV; ++j) {
      const int i = threadIdx.x + 256 * j;
      const int r = i >> 4, q = (i & 15) * 4;
      const int p = t0 - pad + r;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (r < nx_rows && p >= 0 && p < L) {
        const float* src = x + (row0 + p) * ldx + c0 + q;
        if (vec_x) v = *reinterpret_cast<const f32x4*>(src);
        else
#pragma unroll
          for (int e = 0; e < 4; ++e) if (c0 + q + e < c_in) v[e] = src[e];
      }
      rx[j] = v;
    }
  };
  auto commit = [&](float* dst_dy, float* dst_x) {
#pragma unroll
    for (int j = 0; j < NDY; ++j) {
      const int i = threadIdx.x + 256 * j;
      *reinterpret_cast<f32x4*>(&dst_dy[(i >> 4) * P + (i & 15) * 4]) = rdy[j];
      if (do_b) bsum += rdy[j];
    }
#pragma unroll
    for (int j = 0; j < NXV; ++j) {
      const int i = threadIdx.x + 256 * j;
      if ((i >> 4) < nx_rows) *reinterpret_cast<f32x4*>(&dst_x[(i >> 4) * P + (i & 15) * 4]) = rx[j];
    }
  };
  const int buf_floats = (2 * TT + halo) * P;       // one buffer: dy tile then x tile
  int cs = grp, ct0 = 0;                            // current chunk
  if (cs < rg.n_seq) {
    // (sequences of length 0 contribute one all-zero chunk: harmless)
    issue(cs, ct0);
    commit(sm, sm + TT * P);
  }
  __syncthreads();
  int it = 0;
  while (cs < rg.n_seq) {
    const int Lc = (rg.cu_rows[cs + 1] - rg.cu_rows[cs]) * rg.len_mul;
    int ns = cs, nt0 = ct0 + TT;                    // next chunk
    if (nt0 >= Lc) { ns = cs + seq_groups; nt0 = 0; }
    const bool more = ns < rg.n_seq;
    if (more) issue(ns, nt0);
    const float* cur = sm + (it & 1) * buf_floats;
    const float* ap = cur + hi * P + wn * 32 + lo;
    const float* bp = cur + TT * P + hi * P + wc * 32 + lo;
#pragma unroll JATTS_WGRAD_UNROLL
    for (int tp = 0; tp < TT / 2; ++tp) {
      const float a = ap[2 * tp * P];
#pragma unroll
      for (int k = 0; k < KW; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bp[(2 * tp + k * dil) * P], acc[k], 0, 0, 0);
    }
    if (more) {
      float* nxt = sm + ((it + 1) & 1) * buf_floats;
      commit(nxt, nxt + TT * P);
    }
    __syncthreads();
    cs = ns; ct0 = nt0; ++it;
  }
  if (do_b) {   // (uniform per workgroup; the loop above ended on a barrier, LDS is free)
    *reinterpret_cast<f32x4*>(&sm[(threadIdx.x >> 4) * 64 + (threadIdx.x & 15) * 4]) = bsum;
    __syncthreads();
    if (threadIdx.x < 64) {
      float a = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) a += sm[r * 64 + threadIdx.x];
      bws[(int64_t)grp * gridDim.x * 64 + n0 + threadIdx.x] = a;
    }
  }
  // C/D map: column (lane & 31) = c, row (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) = n
  const int c = c0 + wc * 32 + lo;
  if (ws) {   // split-K partial of this sequence group: ws[grp][tap][n][c] (c fastest: 128-byte row stores), summed by wgrad_reduce_kernel
    const int n64 = gridDim.x * 64, c64 = gridDim.y * 64;
    float* o = ws + (int64_t)grp * KW * n64 * c64;
#pragma unroll
    for (int k = 0; k < KW; ++k)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        o[((int64_t)k * n64 + n) * c64 + c] = acc[k][r];
      }
    return;
  }
  // (no workspace: refused by the launcher -- the f32-atomic accumulation this kernel once had was not reproducible run to run)
}
// dw[n][c][k] = sum over groups of ws[g][k][n][c]  (overwrites dw: no zero fill, no atomics -- 32-way contended f32 atomics on a
// 368 k-element gradient were 85 % of the k = 5 launches' time)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, int groups, int K, int n64, int c64, int n_out, int c_in,
                                                           float* __restrict__ dw, const float* __restrict__ bws, float* __restrict__ db) {
  const int64_t total = (int64_t)n_out * c_in * K;
  const int64_t gstride = (int64_t)K * n64 * c64;
  if (db)
    for (int n = blockIdx.x * 256 + threadIdx.x; n < n_out; n += gridDim.x * 256) {
      float a = 0.f;
      for (int g = 0; g < groups; ++g) a += bws[(int64_t)g * n64 + n];
      db[n] = a;
    }
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int k = (int)(i % K);
    const int64_t nc = i / K;
    const int c = (int)(nc % c_in), n = (int)(nc / c_in);
    const float* p = ws + ((int64_t)k * n64 + n) * c64 + c;
    float a = 0.f;
    for (int g = 0; g < groups; ++g) a += p[g * gstride];
    dw[i] = a;
  }
}

// Weight packing for jatts_conv1d ([tap][c/16][n/32][g][n%32][8], include/jatts_hip.h) in ONE launch: zero padding of n to 32 and c
// to c_mult, the permutation, the cast -- and for the data gradient (mode 1) the transposition / tap flip of
// W'[c][n][k'] = W[n][c][K-1-k'] on the fly.  The training step re-packs every weight twice per step (forward, dgrad); as torch
// ops that was a zero fill + two copies (+ permute / flip copies) per use, a third of all launches of a step.
template <typename TO>
__global__ __launch_bounds__(256) void pack_conv_weight_kernel(const float* __restrict__ w, int n_out, int c_in, int K, int n_pad, int c_pad,
                                                               int mode, TO* __restrict__ out) {
  const int64_t total = (int64_t)K * n_pad * c_pad;
  const int KC16 = c_pad / 16, NFR = n_pad / 32;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int e = (int)(i & 7), nl = (int)((i >> 3) & 31), g = (int)((i >> 8) & 1);
    int64_t r = i >> 9;
    const int nf = (int)(r % NFR);
    r /= NFR;
    const int c16 = (int)(r % KC16), tap = (int)(r / KC16);
    const int n = nf * 32 + nl, c = c16 * 16 + g * 8 + e;
    float v = 0.f;
    if (mode == 0) {
      if (n < n_out && c < c_in) v = w[((int64_t)n * c_in + c) * K + tap];
    } else {          // packed "n" runs over the original input channels, packed "c" over the original output channels
      if (n < c_in && c < n_out) v = w[((int64_t)c * c_in + n) * K + (K - 1 - tap)];
    }
    out[i] = from_f32<TO>(v);
  }
}

// The JATTS_F32S operand of a weight, on the device (round 4: the training step re-packs every weight twice per step): packed row n of the
// operand (mode 0: output channel n of W; mode 1, the data-gradient operand: INPUT channel n of W) gets the power-of-two scale that puts its
// largest magnitude in [2^14, 2^15) -- exactly hip.pack_conv_weight_split's rule -- and inv[n] = 2^-s[n]; rows of zero padding take 1.
__global__ __launch_bounds__(256) void wscale_kernel(const float* __restrict__ w, int n_out, int c_in, int K, int mode, int n_rows, float* __restrict__ inv) {
  __shared__ float red[4];
  const int n = blockIdx.x;
  float m = 0.f;
  if (n < n_rows) {
    if (mode == 0) {
      const float* p = w + (int64_t)n * c_in * K;
      for (int i = threadIdx.x; i < c_in * K; i += 256) m = fmaxf(m, fabsf(p[i]));
    } else {
      for (int i = threadIdx.x; i < n_out * K; i += 256) m = fmaxf(m, fabsf(w[((int64_t)(i / K) * c_in + n) * K + i % K]));
    }
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const int bexp = (int)((__float_as_uint(m) >> 23) & 0xff);
    int s = 15 - (bexp - 126);
    s = m > 0.f ? (s > 60 ? 60 : (s < -60 ? -60 : s)) : 0;
    inv[n] = __uint_as_float((unsigned)(127 - s) << 23);
  }
}
// out: [tap][c/16][n/32][lane][hi x8 | lo x8] f16 (pack_conv_weight_split's layout)
__global__ __launch_bounds__(256) void pack_conv_weight_split_kernel(const float* __restrict__ w, int n_out, int c_in, int K, int n_pad, int c_pad,
                                                                     int mode, const float* __restrict__ inv, f16* __restrict__ out) {
  const int64_t total = (int64_t)K * n_pad * c_pad;
  const int KC16 = c_pad / 16, NFR = n_pad / 32;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int e = (int)(i & 7), nl = (int)((i >> 3) & 31), g = (int)((i >> 8) & 1);
    int64_t r = i >> 9;
    const int nf = (int)(r % NFR);
    r /= NFR;
    const int c16 = (int)(r % KC16), tap = (int)(r / KC16);
    const int n = nf * 32 + nl, c = c16 * 16 + g * 8 + e;
    float v = 0.f;
    if (mode == 0) {
      if (n < n_out && c < c_in) v = w[((int64_t)n * c_in + c) * K + tap];
    } else {
      if (n < c_in && c < n_out) v = w[((int64_t)c * c_in + n) * K + (K - 1 - tap)];
    }
    const float sv = v / inv[n];                 // exact: inv is a power of two
    const f16 hi = (f16)sv;
    out[(i >> 3) * 16 + e] = hi;
    out[(i >> 3) * 16 + 8 + e] = (f16)(sv - (float)hi);
  }
}

// out[c] += sum over rows of x[row][c]
__global__ __launch_bounds__(256) void col_sum_kernel(const float* x, int ld, int64_t rows, int dim, float* out, int overwrite,
                                                      float* __restrict__ slabs, unsigned* __restrict__ tickets) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int part = threadIdx.x >> 6;
  __shared__ __attribute__((aligned(16))) float red[16][64];
  float s = 0.f;
  if (c < dim)
    for (int64_t r = (int64_t)blockIdx.y * 4 + part; r < rows; r += (int64_t)gridDim.y * 4) s += x[r * ld + c];
  red[part][threadIdx.x & 63] = s;
  __syncthreads();
  float* gslab = slabs + (int64_t)blockIdx.x * gridDim.y * 64;      // group = channel tile, parts = the row splits
  if (part == 0) gslab[blockIdx.y * 64 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
  if (det_arrive(tickets + blockIdx.x, gridDim.y, reinterpret_cast<unsigned*>(&red[0][0])))
    det_sum_slabs<4>(gslab, (int)gridDim.y, 64, &red[0][0], 1024, [&](int i, float t) {
      const int cc = blockIdx.x * 64 + i;
      if (cc < dim) out[cc] = overwrite ? t : out[cc] + t;
    });
}

// the same over all rows of a ragged batch (row count read from the device-side offsets), into a zero-filled out: the bias gradient of
// the convolutions whose weight gradient takes the VALU path
__global__ __launch_bounds__(256) void col_sum_ragged_kernel(jatts_ragged rg, const float* x, int ld, int dim, float* out, int overwrite,
                                                             float* __restrict__ slabs, unsigned* __restrict__ tickets) {
  const int64_t rows = (int64_t)rg.cu_rows[rg.n_seq] * rg.len_mul;
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int part = threadIdx.x >> 6;
  __shared__ __attribute__((aligned(16))) float red[16][64];
  float s = 0.f;
  if (c < dim)
    for (int64_t r = (int64_t)blockIdx.y * 4 + part; r < rows; r += (int64_t)gridDim.y * 4) s += x[r * ld + c];
  red[part][threadIdx.x & 63] = s;
  __syncthreads();
  float* gslab = slabs + (int64_t)blockIdx.x * gridDim.y * 64;      // group = channel tile, parts = the row splits
  if (part == 0) gslab[blockIdx.y * 64 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
  if (det_arrive(tickets + blockIdx.x, gridDim.y, reinterpret_cast<unsigned*>(&red[0][0])))
    det_sum_slabs<4>(gslab, (int)gridDim.y, 64, &red[0][0], 1024, [&](int i, float t) {
      const int cc = blockIdx.x * 64 + i;
      if (cc < dim) out[cc] = overwrite ? t : out[cc] + t;
    });
}

}  // namespace

#define S_ ((hipStream_t)stream)

static int bias_grad_plain(const jatts_ragged* rg, const float* dy, int ldy, int n_out, float* db, void* stream) {
  const unsigned gx = (unsigned)((n_out + 63) / 64);
  const int rc = jatts_ws_need(gx, (int64_t)gx * 128 * 64);
  if (rc != JATTS_OK) return rc;
  hipLaunchKernelGGL(col_sum_ragged_kernel, dim3(gx, 128), dim3(256), 0, S_, *rg, dy, ldy, n_out, db, 1, jatts_g_ws.slabs, jatts_g_ws.tickets);
  return JATTS_OK;
}

extern "C" int jatts_masked_loss(const jatts_ragged* rg, const float* a, int32_t lda, const float* b, int32_t ldb, int32_t dim,
                                 const int32_t* valid_len, int32_t kind, float log_offset, double scale, float* out, double* workspace,
                                 void* stream) {
  if (!rg || !a || !b || !out || !workspace) return jatts_set_error_msg(JATTS_ERR_ARG, "masked_loss: null pointer");
  if (kind != 0 && kind != 1) return jatts_set_error_msg(JATTS_ERR_ARG, "masked_loss: kind must be 0 (L1) or 1 (L2)");
  if (rg->n_seq <= 0 || rg->n_seq > 4096) return jatts_set_error_msg(JATTS_ERR_ARG, "masked_loss: 1..4096 sequences");
  const int bx = 4;
  hipLaunchKernelGGL(masked_loss_partial_kernel, dim3(bx, (unsigned)rg->n_seq), dim3(256), 0, S_, *rg, a, lda, b, ldb, dim, valid_len, kind,
                     log_offset, workspace);
  hipLaunchKernelGGL(fold_loss_kernel, dim3(1), dim3(1), 0, S_, workspace, bx * rg->n_seq, scale, out);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

extern "C" int jatts_conv1d_wgrad(const jatts_ragged* rg, const float* x, int32_t ldx, const float* dy, int32_t ldy, int32_t c_in,
                                  int32_t n_out, int32_t k_w, int32_t dil, int32_t pad, float* dw, float* db, float* workspace, void* stream) {
  if (!rg || !x || !dy || !dw) return jatts_set_error_msg(JATTS_ERR_ARG, "conv1d_wgrad: null pointer");
  if (c_in < 1 || n_out < 1 || k_w < 1 || dil < 1) return jatts_set_error_msg(JATTS_ERR_ARG, "conv1d_wgrad: bad geometry");
  if (rg->n_seq <= 0 || rg->max_len <= 0) return JATTS_OK;
  const int tiles = ((n_out + 63) / 64) * ((c_in + 63) / 64) * k_w;
  int groups = (1024 + tiles - 1) / tiles;            // enough workgroups to fill the chip; each group strides over the sequences
  if (groups > rg->n_seq) groups = rg->n_seq;
  if (groups < 1) groups = 1;
  static const int use_mfma = [] { const char* e = getenv("JATTS_WGRAD_MFMA"); return e ? atoi(e) : 1; }();
  if (use_mfma && (k_w == 1 || k_w == 3 || k_w == 5) && (k_w - 1) * dil <= 32) {
    const int tiles_m = ((n_out + 63) / 64) * ((c_in + 63) / 64);
    // Sequence groups (split-K factor): the launch ends when the fullest CU has walked its workgroups, each over ceil(n_seq / g) sequences
    // -- minimise ceil(tiles g / 256) ceil(n_seq / g) (the old "at least 1 536 workgroups" rule gave 1 584 for 1536 x 384: 7 on some CUs, 6
    // on others, each 3 sequences long = 21 units against 18 for g = 16: 840 -> 782 us; tools/sweep_wgrad_groups.sh), preferring four resident
    // workgroups per CU, with a small tax per group for the reduction pass.  Deterministic: a function of the shapes only.
    static const int g_env = [] { const char* e = getenv("JATTS_WGRAD_GROUPS"); return e ? atoi(e) : 0; }();   // (tools/ A/B only)
    int g = 1;
    {
      double best = 1e300;
      for (int c = 1; c <= rg->n_seq; ++c) {
        const int64_t wgs = (int64_t)tiles_m * c;
        double cost = (double)((wgs + 255) / 256) * (double)((rg->n_seq + c - 1) / c);
        const int64_t per_cu = (wgs + 255) / 256;
        if (wgs < 512) cost *= 512.0 / (double)wgs;
        if (per_cu < 4) cost *= 1.0 + 0.06 * (double)(4 - per_cu);   // four resident workgroups per CU hide the staging (2048 x 512 k1: g = 2 638 us, g = 4 544 us)
        cost *= 1.0 + 0.002 * c;
        if (cost < best) { best = cost; g = c; }
      }
    }
    if (g_env > 0) g = g_env;
    if (g > rg->n_seq) g = rg->n_seq;
    if (g < 1) g = 1;
    const dim3 grid((unsigned)((n_out + 63) / 64), (unsigned)((c_in + 63) / 64), (unsigned)g);
    const size_t lds = 2 * (size_t)(64 + (k_w - 1) * dil) * 68 * sizeof(float);   // two buffers of (dy tile | x tile + halo)
    // bias partials live behind the weight partials: workspace[g k n64 c64 ..][g][n64]
    if (!workspace) return jatts_set_error_msg(JATTS_ERR_ARG, "conv1d_wgrad: the MFMA path needs its split-K workspace (see include/jatts_hip.h)");
    float* bws = db ? workspace + (int64_t)g * k_w * grid.x * 64 * grid.y * 64 : nullptr;
    if (k_w == 1) hipLaunchKernelGGL(conv_wgrad_mfma_kernel<1>, grid, dim3(256), lds, S_, *rg, x, ldx, dy, ldy, c_in, n_out, dil, pad, g, dw, workspace, bws);
    else if (k_w == 3) hipLaunchKernelGGL(conv_wgrad_mfma_kernel<3>, grid, dim3(256), lds, S_, *rg, x, ldx, dy, ldy, c_in, n_out, dil, pad, g, dw, workspace, bws);
    else hipLaunchKernelGGL(conv_wgrad_mfma_kernel<5>, grid, dim3(256), lds, S_, *rg, x, ldx, dy, ldy, c_in, n_out, dil, pad, g, dw, workspace, bws);
    {
      const int64_t total = (int64_t)n_out * c_in * k_w;
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048)), dim3(256), 0, S_, workspace, g, k_w,
                         (int)grid.x * 64, (int)grid.y * 64, n_out, c_in, dw, bws, bws ? db : nullptr);
    }
    JATTS_CHECK_LAUNCH();
    return JATTS_OK;
  }
  if (db) {
    const int rc = bias_grad_plain(rg, dy, ldy, n_out, db, stream);
    if (rc != JATTS_OK) return rc;
  }
  // VALU fallback (other kernel widths): deterministic split-K through the library scratch (det_reduce.h); with a workspace argument dw is
  // OVERWRITTEN (the contract of the MFMA path), without one the result is accumulated into dw
  {
    const int rc = jatts_ws_need((int64_t)tiles, (int64_t)tiles * groups * 4096);
    if (rc != JATTS_OK) return rc;
  }
  hipLaunchKernelGGL(conv_wgrad_kernel, dim3((unsigned)((n_out + 63) / 64), (unsigned)((c_in + 63) / 64), (unsigned)(k_w * groups)), dim3(256), 0, S_,
                     *rg, x, ldx, dy, ldy, c_in, n_out, k_w, dil, pad, groups, dw, workspace ? 1 : 0, jatts_g_ws.slabs, jatts_g_ws.tickets);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

extern "C" int jatts_col_sum(const float* x, int32_t ld, int64_t rows, int32_t dim, float* out, void* stream) {
  if (!x || !out) return jatts_set_error_msg(JATTS_ERR_ARG, "col_sum: null pointer");
  if (rows <= 0 || dim <= 0) return JATTS_OK;
  const int64_t gy0 = (rows + 255) / 256;
  const unsigned gx = (unsigned)((dim + 63) / 64), gy = (unsigned)(gy0 < 256 ? gy0 : 256);
  {
    const int rc = jatts_ws_need(gx, (int64_t)gx * gy * 64);
    if (rc != JATTS_OK) return rc;
  }
  hipLaunchKernelGGL(col_sum_kernel, dim3(gx, gy), dim3(256), 0, S_, x, ld, rows, dim, out, 0, jatts_g_ws.slabs, jatts_g_ws.tickets);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

extern "C" int jatts_pack_conv_weight(const float* w, int32_t n_out, int32_t c_in, int32_t k_w, int32_t c_mult, int32_t mode, int32_t dtype,
                                      void* out, void* stream) {
  if (!w || !out) return jatts_set_error_msg(JATTS_ERR_ARG, "pack_conv_weight: null pointer");
  if (n_out < 1 || c_in < 1 || k_w < 1 || c_mult < 16 || c_mult % 16 != 0 || (mode != 0 && mode != 1))
    return jatts_set_error_msg(JATTS_ERR_ARG, "pack_conv_weight: bad geometry");
  const int pn = mode == 0 ? n_out : c_in, pc = mode == 0 ? c_in : n_out;
  const int n_pad = (pn + 31) / 32 * 32, c_pad = (pc + c_mult - 1) / c_mult * c_mult;
  const int64_t total = (int64_t)k_w * n_pad * c_pad;
  const unsigned blocks = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  if (dtype == JATTS_F32)
    hipLaunchKernelGGL(pack_conv_weight_kernel<float>, dim3(blocks), dim3(256), 0, S_, w, n_out, c_in, k_w, n_pad, c_pad, mode, (float*)out);
  else if (dtype == JATTS_F16)
    hipLaunchKernelGGL(pack_conv_weight_kernel<f16>, dim3(blocks), dim3(256), 0, S_, w, n_out, c_in, k_w, n_pad, c_pad, mode, (f16*)out);
  else
    return jatts_set_error_msg(JATTS_ERR_ARG, "pack_conv_weight: dtype");
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

extern "C" int jatts_pack_conv_weight_split(const float* w, int32_t n_out, int32_t c_in, int32_t k_w, int32_t c_mult, int32_t mode, void* out,
                                            float* inv, void* stream) {
  if (!w || !out || !inv) return jatts_set_error_msg(JATTS_ERR_ARG, "pack_conv_weight_split: null pointer");
  if (n_out < 1 || c_in < 1 || k_w < 1 || c_mult < 16 || c_mult % 16 != 0 || (mode != 0 && mode != 1))
    return jatts_set_error_msg(JATTS_ERR_ARG, "pack_conv_weight_split: bad geometry");
  const int pn = mode == 0 ? n_out : c_in, pc = mode == 0 ? c_in : n_out;
  const int n_pad = (pn + 31) / 32 * 32, c_pad = (pc + c_mult - 1) / c_mult * c_mult;
  const int64_t total = (int64_t)k_w * n_pad * c_pad;
  hipLaunchKernelGGL(wscale_kernel, dim3((unsigned)n_pad), dim3(256), 0, S_, w, n_out, c_in, k_w, mode, pn, inv);
  hipLaunchKernelGGL(pack_conv_weight_split_kernel, dim3((unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096)), dim3(256), 0, S_, w, n_out,
                     c_in, k_w, n_pad, c_pad, mode, inv, (f16*)out);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}
