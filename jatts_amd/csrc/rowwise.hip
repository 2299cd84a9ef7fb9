// HBM-bound row-wise kernels of the jatts hot path (gfx950): embedding, LayerNorm,
// affine/cast, conformer GLU+depthwise+BN+Swish, predictor heads, variance embeddings,
// length regulator (bit-exact integer path), Gaussian upsampling, HiFi-GAN output conv.
#include <stdlib.h>

#include "common.h"

namespace {

// 8 consecutive elements as f32 (16-byte load for f16, 2 x 16-byte for f32); p must be 16-byte aligned.
template <typename T>
__device__ __forceinline__ void load8f(const T* p, float (&o)[8]) {
  if (sizeof(T) == 2) {
    const f16x8 v = *reinterpret_cast<const f16x8*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (float)v[e];
  } else {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>((const float*)p + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e] = a[e]; o[e + 4] = b[e]; }
  }
}

// ------------------------------------------------------------------------- embedding
__global__ void embed_scale_kernel(const int64_t* ids, int64_t rows, const float* table, int dim,
                                   float scale, float* out, int64_t vocab, int64_t* n_bad) {
  const int64_t row = blockIdx.x;
  const int64_t id = ids[row];
  const bool bad = vocab > 0 && (id < 0 || id >= vocab);   // torch.nn.Embedding raises IndexError: counted here, raised by the host
  if (bad && threadIdx.x == 0 && n_bad) atomicAdd(reinterpret_cast<unsigned long long*>(n_bad), 1ull);
  const float* src = table + (bad ? 0 : id) * (int64_t)dim;
  float* dst = out + row * (int64_t)dim;
  for (int c = threadIdx.x; c < dim; c += blockDim.x) dst[c] = bad ? 0.f : src[c] * scale;
}

// ------------------------------------------------------------------------- layernorm
// One wave per row; two-pass mean/variance on register-resident values (torch semantics:
// biased variance, y = (x - mean) * rsqrt(var + eps) * gamma + beta).
constexpr int LN_MAXV = 32;  // dim <= 64 * 32 = 2048

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void layernorm_kernel(const TI* x, int ldx, TO* y, int ldy, int64_t rows,
                                                        int dim, const float* gamma, const float* beta,
                                                        float eps) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const TI* xr = x + row * (int64_t)ldx;
  float v[LN_MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = lane + 64 * i;
    v[i] = c < dim ? to_f32(xr[c]) : 0.f;
    s += v[i];
  }
  const float mean = wave_sum(s) / (float)dim;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = lane + 64 * i;
    const float t = c < dim ? v[i] - mean : 0.f;
    q += t * t;
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)dim + eps);
  TO* yr = y + row * (int64_t)ldy;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = lane + 64 * i;
    if (c < dim) yr[c] = from_f32<TO>((v[i] - mean) * rstd * gamma[c] + beta[c]);
  }
}

// ----------------------------------------------------------------------- affine + cast
// W = columns written per row: ldy (zero-padded rows) or dim (a column slice of a wider matrix: jatts_affine_slice)
template <typename TO>
__global__ void affine_cast_kernel(const float* x, int ldx, TO* y, int ldy, int64_t rows, int dim,
                                   const float* scale, const float* shift, int W) {
  const int64_t total = rows * (int64_t)W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / W;
    const int c = (int)(i - r * W);
    float v = 0.f;
    if (c < dim) {
      v = x[r * ldx + c];
      if (scale) v = v * scale[c];
      if (shift) v = v + shift[c];
    }
    y[r * ldy + c] = from_f32<TO>(v);
  }
}

// ------------------------------------------------ GLU -> depthwise conv -> BN -> Swish
constexpr int DW_TT = 64;  // time steps per block
constexpr int DW_CB = 64;  // channels per block

template <typename T, int KMAX>
__global__ __launch_bounds__(256) void glu_dw_kernel(jatts_ragged rg, const T* x, T* y, int C, int K,
                                                     const float* w, const float* sc, const float* sh) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* hs = reinterpret_cast<float*>(smem);  // [(DW_TT + K - 1)][DW_CB + 1]
  const int b = blockIdx.y;
  const int row0 = rg.cu_rows[b];
  const int L = rg.cu_rows[b + 1] - row0;
  const int t0 = blockIdx.x * DW_TT;
  if (t0 >= L) return;
  const int c0 = blockIdx.z * DW_CB;
  const int pad = (K - 1) / 2;
  const int rows = DW_TT + K - 1;
  const int P = DW_CB + 1;
  // depthwise taps of this block's 64 channels, transposed to [k][channel] in LDS: the 64*K floats are contiguous
  // in w (coalesced), whereas w[c*K + k] read per lane inside the tap loop is a 64-line gather per instruction
  // (the kernel spent 88 % of its wave cycles waiting on those; profiles/r01_notes.md)
  float* ws = hs + rows * P;
  for (int u = threadIdx.x; u < DW_CB * K; u += 256) {
    const int cc = u / K, k = u - cc * K;
    ws[k * P + cc] = c0 + cc < C ? w[(int64_t)c0 * K + u] : 0.f;
  }
  if ((C & 7) == 0) {
    // 16-byte loads of 8 channels of a and of g, 4 units per thread in flight before any is consumed (the
    // one-element-per-iteration form serialised a memory round trip per 2 bytes: 157 us per launch).
    constexpr int UPR = DW_CB / 8, UB = 4;
    const int total = rows * UPR;
    for (int u0 = threadIdx.x; u0 < total; u0 += UB * 256) {
      float a[UB][8], g[UB][8];
      bool ok[UB];
#pragma unroll
      for (int i = 0; i < UB; ++i) {
        const int u = u0 + i * 256, r = u / UPR, cu = u - r * UPR;
        const int pos = t0 - pad + r, c = c0 + cu * 8;
        ok[i] = u < total && pos >= 0 && pos < L && c < C;
        if (ok[i]) {
          const T* xr = x + (int64_t)(row0 + pos) * (2 * C) + c;
          load8f<T>(xr, a[i]);
          load8f<T>(xr + C, g[i]);
        }
      }
#pragma unroll
      for (int i = 0; i < UB; ++i) {
        const int u = u0 + i * 256;
        if (u >= total) continue;
        const int r = u / UPR, cu = u - r * UPR;
#pragma unroll
        for (int e = 0; e < 8; ++e) hs[r * P + cu * 8 + e] = ok[i] ? a[i][e] / (1.f + __expf(-g[i][e])) : 0.f;  // a * sigmoid(g)
      }
    }
  } else {
  for (int u = threadIdx.x; u < rows * DW_CB; u += 256) {
    const int r = u / DW_CB, cc = u - r * DW_CB;
    const int pos = t0 - pad + r, c = c0 + cc;
    float h = 0.f;
    if (pos >= 0 && pos < L && c < C) {
      const T* xr = x + (int64_t)(row0 + pos) * (2 * C);
      const float a = to_f32(xr[c]), g = to_f32(xr[C + c]);
      h = a / (1.f + __expf(-g));  // a * sigmoid(g)
    }
    hs[r * P + cc] = h;
  }
  }
  __syncthreads();
  const int cc = threadIdx.x & (DW_CB - 1), tg = threadIdx.x / DW_CB;  // 4 time groups of 16
  const int c = c0 + cc;
  if (c >= C) return;
  const float s = sc[c], t = sh[c];
  constexpr int OPT = DW_TT / 4;  // outputs per thread: a register window of OPT + K - 1 inputs, each read from LDS once
  if constexpr (KMAX > 0) {
    float wreg[KMAX], win[OPT + KMAX - 1];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) wreg[k] = k < K ? ws[k * P + cc] : 0.f;
#pragma unroll
    for (int j = 0; j < OPT + KMAX - 1; ++j) win[j] = j < OPT + K - 1 ? hs[(tg * OPT + j) * P + cc] : 0.f;
#pragma unroll
    for (int i = 0; i < OPT; ++i) {
      const int tl = tg * OPT + i;
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < KMAX; ++k) acc += wreg[k] * win[i + k];   // taps >= K carry weight 0
      const float v = acc * s + t;
      if (t0 + tl < L) y[(int64_t)(row0 + t0 + tl) * C + c] = from_f32<T>(v / (1.f + __expf(-v)));
    }
  } else {
    for (int i = 0; i < OPT; ++i) {
      const int tl = tg * OPT + i;
      if (t0 + tl >= L) break;
      float acc = 0.f;
      for (int k = 0; k < K; ++k) acc += ws[k * P + cc] * hs[(tl + k) * P + cc];
      const float v = acc * s + t;
      y[(int64_t)(row0 + t0 + tl) * C + c] = from_f32<T>(v / (1.f + __expf(-v)));
    }
  }
}

// --------------------------------------------------------------------- predictor head
template <typename T>
__global__ __launch_bounds__(256) void predictor_head_kernel(const T* x, int ldx, int64_t rows, int dim,
                                                             const float* w, float b, float* v_out,
                                                             int64_t* dur_out, float offset) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  for (int c = lane; c < dim; c += 64) s += to_f32(x[row * ldx + c]) * w[c];
  s = wave_sum(s) + b;
  if (lane == 0) {
    if (v_out) v_out[row] = s;
    if (dur_out) {
      // duration_predictor.py:87-90: clamp(round(exp(x) - offset), min=0).long()
      float dlin = rintf(expf(s) - offset);
      dlin = dlin > 0.f ? dlin : 0.f;
      dur_out[row] = (int64_t)dlin;
    }
  }
}

// ------------------------------------------------------- pitch / energy embedding add
__global__ void variance_embed_kernel(jatts_ragged rg, float* hs, int dim, const float* p, const float* wp,
                                      const float* bp, int kp, const float* e, const float* we,
                                      const float* be, int ke) {
  const int b = blockIdx.y;
  const int row0 = rg.cu_rows[b];
  const int L = rg.cu_rows[b + 1] - row0;
  const int t = blockIdx.x;
  if (t >= L) return;
  const int pp = (kp - 1) / 2, pe = (ke - 1) / 2;
  for (int c = threadIdx.x; c < dim; c += blockDim.x) {
    float acc = bp[c] + be[c];
    for (int k = 0; k < kp; ++k) {
      const int q = t + k - pp;
      if (q >= 0 && q < L) acc += p[row0 + q] * wp[c * kp + k];
    }
    for (int k = 0; k < ke; ++k) {
      const int q = t + k - pe;
      if (q >= 0 && q < L) acc += e[row0 + q] * we[c * ke + k];
    }
    hs[(int64_t)(row0 + t) * dim + c] += acc;
  }
}

__global__ void add_seq_vector_kernel(jatts_ragged rg, float* hs, int dim, const float* vec) {
  const int b = blockIdx.y;
  const int row0 = rg.cu_rows[b];
  const int L = rg.cu_rows[b + 1] - row0;
  const int t = blockIdx.x;
  if (t >= L) return;
  for (int c = threadIdx.x; c < dim; c += blockDim.x) hs[(int64_t)(row0 + t) * dim + c] += vec[(int64_t)b * dim + c];
}

// ------------------------------------------------------------ WaveNet gate, channel flip
template <typename T>
__global__ void gated_kernel(jatts_ragged rg, const T* x, const float* gseq, T* y, int C) {
  const int b = blockIdx.y;
  const int row0 = rg.cu_rows[b];
  const int L = rg.cu_rows[b + 1] - row0;
  const int t = blockIdx.x;
  if (t >= L) return;
  const T* xr = x + (int64_t)(row0 + t) * (2 * C);
  const float* g = gseq ? gseq + (int64_t)b * (2 * C) : nullptr;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float a = to_f32(xr[c]), s = to_f32(xr[C + c]);
    if (g) { a += g[c]; s += g[C + c]; }
    y[(int64_t)(row0 + t) * C + c] = from_f32<T>(tanhf(a) / (1.f + expf(-s)));
  }
}

// ------------------------------------------------------------ GroupNorm + Mish, SnakeBeta
__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

template <typename T, typename TO>
__global__ __launch_bounds__(256) void groupnorm_mish_kernel(jatts_ragged rg, const T* x, TO* y, int C, int groups,
                                                            const float* gamma, const float* beta, float eps,
                                                            const float* addvec) {
  __shared__ float red[4];
  const int b = blockIdx.y, g = blockIdx.x;
  const int row0 = rg.cu_rows[b];
  const int L = rg.cu_rows[b + 1] - row0;
  if (L <= 0) return;
  const int gc = C / groups;
  const int c0 = g * gc;
  const int n = L * gc;
  const T* xb = x + (int64_t)row0 * C + c0;
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += to_f32(xb[(int64_t)(i / gc) * C + (i % gc)]);
  const float mean = block_sum_256(s, red) / (float)n;
  float q = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float d = to_f32(xb[(int64_t)(i / gc) * C + (i % gc)]) - mean;
    q += d * d;
  }
  const float rstd = rsqrtf(block_sum_256(q, red) / (float)n + eps);
  for (int i = threadIdx.x; i < n; i += 256) {
    const int r = i / gc, c = c0 + i % gc;
    float v = (to_f32(xb[(int64_t)r * C + (i % gc)]) - mean) * rstd * gamma[c] + beta[c];
    v = mish_f(v);
    if (addvec) v += addvec[(int64_t)b * C + c];
    y[(int64_t)(row0 + r) * C + c] = from_f32<TO>(v);
  }
}

// Time-split GroupNorm: (groups x n_seq x chunks) workgroups instead of (groups x n_seq), two launches.
//   pass 1: each workgroup holds its GN_TCH-row chunk of one (utterance, group) in registers and writes the chunk's
//           (count, mean, M2) -- two-pass inside the chunk;
//   pass 2: every workgroup merges the chunk statistics of its (utterance, group) with Chan's parallel formula (a few
//           dozen values, done redundantly), then normalises + Mish + per-utterance vector from the registers it
//           re-loads with 16-byte accesses.
// The single-workgroup kernel above streams 196 KB three times with 2-byte loads from 512 workgroups (~10x the HBM time).
constexpr int GN_TCH = 64;  // rows per chunk
constexpr int GN_MAXU = 4;  // 8-element units per thread: chunk rows * (C/groups) <= 256 * 4 * 8 elements

template <typename T>
__global__ __launch_bounds__(256) void gn_partial_kernel(jatts_ragged rg, const T* x, int C, int groups, float* ws, int n_chunks) {
  __shared__ float red[4];
  const int g = blockIdx.x, b = blockIdx.y, ch = blockIdx.z;
  const int row0 = rg.cu_rows[b];
  const int L = rg.cu_rows[b + 1] - row0;
  float* out = ws + (((int64_t)b * groups + g) * n_chunks + ch) * 3;
  const int r0 = ch * GN_TCH;
  if (r0 >= L) {
    if (threadIdx.x == 0) { out[0] = 0.f; out[1] = 0.f; out[2] = 0.f; }
    return;
  }
  const int rows = min(GN_TCH, L - r0);
  const int gc = C / groups, upr = gc >> 3, total = rows * upr;
  const T* xb = x + (int64_t)(row0 + r0) * C + g * gc;
  float v[GN_MAXU][8];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < GN_MAXU; ++j) {
    const int u = threadIdx.x + j * 256;
    if (u < total) {
      const int r = u / upr, cu = u - r * upr;
      load8f<T>(xb + (int64_t)r * C + cu * 8, v[j]);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[j][e];
    }
  }
  const float n = (float)(rows * gc);
  const float mean = block_sum_256(s, red) / n;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < GN_MAXU; ++j) {
    const int u = threadIdx.x + j * 256;
    if (u < total) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float dlt = v[j][e] - mean; q += dlt * dlt; }
    }
  }
  const float m2 = block_sum_256(q, red);
  if (threadIdx.x == 0) { out[0] = n; out[1] = mean; out[2] = m2; }
}

template <typename T, typename TO>
__global__ __launch_bounds__(256) void gn_apply_kernel(jatts_ragged rg, const T* x, TO* y, int C, int groups, const float* gamma,
                                                       const float* beta, float eps, const float* addvec, const float* ws,
                                                       int n_chunks) {
  const int g = blockIdx.x, b = blockIdx.y, ch = blockIdx.z;
  const int row0 = rg.cu_rows[b];
  const int L = rg.cu_rows[b + 1] - row0;
  const int r0 = ch * GN_TCH;
  if (r0 >= L) return;
  const float* st = ws + ((int64_t)b * groups + g) * n_chunks * 3;
  float na = 0.f, ma = 0.f, m2a = 0.f;   // Chan et al.: merge (n, mean, M2) pairs
  for (int i = 0; i < n_chunks; ++i) {
    const float nb = st[3 * i], mb = st[3 * i + 1], m2b = st[3 * i + 2];
    if (nb > 0.f) {
      const float nn = na + nb, dlt = mb - ma;
      ma += dlt * (nb / nn);
      m2a += m2b + dlt * dlt * (na * nb / nn);
      na = nn;
    }
  }
  const float mean = ma, rstd = rsqrtf(m2a / na + eps);
  const int rows = min(GN_TCH, L - r0);
  const int gc = C / groups, upr = gc >> 3, total = rows * upr;
  const T* xb = x + (int64_t)(row0 + r0) * C + g * gc;
  float v[GN_MAXU][8];
#pragma unroll
  for (int j = 0; j < GN_MAXU; ++j) {
    const int u = threadIdx.x + j * 256;
    if (u < total) {
      const int r = u / upr, cu = u - r * upr;
      load8f<T>(xb + (int64_t)r * C + cu * 8, v[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < GN_MAXU; ++j) {
    const int u = threadIdx.x + j * 256;
    if (u >= total) continue;
    const int r = u / upr, cu = u - r * upr;
    const int c = g * gc + cu * 8;
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = (v[j][e] - mean) * rstd * gamma[c + e] + beta[c + e];
      t = mish_f(t);
      if (addvec) t += addvec[(int64_t)b * C + c + e];
      o[e] = t;
    }
    TO* dst = y + (int64_t)(row0 + r0 + r) * C + c;
    if (sizeof(TO) == 2) *reinterpret_cast<f16x8*>(dst) = f16x8{(f16)o[0], (f16)o[1], (f16)o[2], (f16)o[3], (f16)o[4], (f16)o[5], (f16)o[6], (f16)o[7]};
    else {
      *reinterpret_cast<f32x4*>(dst) = f32x4{o[0], o[1], o[2], o[3]};
      *reinterpret_cast<f32x4*>((float*)dst + 4) = f32x4{o[4], o[5], o[6], o[7]};
    }
  }
}

// Row-streaming form of the apply pass: a workgroup takes GN_AROWS full rows (ALL groups: contiguous C-element rows instead of one
// group's 256-byte pieces 2 KB apart), merges the chunk statistics of every group of its utterance once into LDS, then streams.
// GN_AROWS is independent of the statistics chunking: at 64 rows the bench shapes gave 768 workgroups of 8 dependent load -> Mish ->
// store rounds each, 24 KB in flight per CU and 2.4 TB/s; 16 rows quadruple the workgroups (12 per CU) with two rounds each.
constexpr int GN_MAXG = 32;
constexpr int GN_AROWS = 16;
template <typename T, typename TO>
__global__ __launch_bounds__(256) void gn_apply_rows_kernel(jatts_ragged rg, const T* __restrict__ x, TO* __restrict__ y, int C, int groups,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                            const float* __restrict__ addvec, const float* __restrict__ ws, int n_chunks) {
  __shared__ float s_mean[GN_MAXG], s_rstd[GN_MAXG];
  const int b = blockIdx.y, ch = blockIdx.x;
  const int row0 = rg.cu_rows[b];
  const int L = rg.cu_rows[b + 1] - row0;
  const int r0 = ch * GN_AROWS;
  if (r0 >= L) return;
  if (threadIdx.x < groups) {
    const float* st = ws + ((int64_t)b * groups + threadIdx.x) * n_chunks * 3;
    float na = 0.f, ma = 0.f, m2a = 0.f;   // Chan et al.: merge (n, mean, M2) pairs
    for (int i = 0; i < n_chunks; ++i) {
      const float nb = st[3 * i], mb = st[3 * i + 1], m2b = st[3 * i + 2];
      if (nb > 0.f) {
        const float nn = na + nb, dlt = mb - ma;
        ma += dlt * (nb / nn);
        m2a += m2b + dlt * dlt * (na * nb / nn);
        na = nn;
      }
    }
    s_mean[threadIdx.x] = ma;
    s_rstd[threadIdx.x] = rsqrtf(m2a / na + eps);
  }
  __syncthreads();
  const int rows = min(GN_AROWS, L - r0);
  const int gc = C / groups, upr = C >> 3, total = rows * upr;
  const T* xb = x + (int64_t)(row0 + r0) * C;
  TO* yb = y + (int64_t)(row0 + r0) * C;
  const float* av = addvec ? addvec + (int64_t)b * C : nullptr;
  for (int u = threadIdx.x; u < total; u += 256) {
    const int r = u / upr, c = (u - r * upr) * 8;
    const int g = c / gc;
    const float mean = s_mean[g], rstd = s_rstd[g];
    float v[8], o[8];
    load8f<T>(xb + (int64_t)r * C + c, v);
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + c), g1 = *reinterpret_cast<const f32x4*>(gamma + c + 4);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + c), b1 = *reinterpret_cast<const f32x4*>(beta + c + 4);
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
    if (av) { a0 = *reinterpret_cast<const f32x4*>(av + c); a1 = *reinterpret_cast<const f32x4*>(av + c + 4); }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float t = (v[e] - mean) * rstd * (e < 4 ? g0[e] : g1[e - 4]) + (e < 4 ? b0[e] : b1[e - 4]);
      o[e] = mish_f(t) + (e < 4 ? a0[e] : a1[e - 4]);
    }
    TO* dst = yb + (int64_t)r * C + c;
    if (sizeof(TO) == 2) *reinterpret_cast<f16x8*>(dst) = f16x8{(f16)o[0], (f16)o[1], (f16)o[2], (f16)o[3], (f16)o[4], (f16)o[5], (f16)o[6], (f16)o[7]};
    else {
      *reinterpret_cast<f32x4*>(dst) = f32x4{o[0], o[1], o[2], o[3]};
      *reinterpret_cast<f32x4*>((float*)dst + 4) = f32x4{o[4], o[5], o[6], o[7]};
    }
  }
}

template <typename T>
__global__ void snakebeta_kernel(const T* x, T* y, int64_t rows, int C, const float* alpha, const float* inv_beta) {
  const int64_t total = rows * C;
  if ((C & 7) == 0) {   // 8 channels per thread, 16-byte accesses
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < total; i += (int64_t)gridDim.x * blockDim.x * 8) {
      const int c = (int)(i % C);
      float v[8];
      load8f<T>(x + i, v);
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(alpha + c), a1 = *reinterpret_cast<const f32x4*>(alpha + c + 4);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(inv_beta + c), b1 = *reinterpret_cast<const f32x4*>(inv_beta + c + 4);
      T o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        o[e] = from_f32<T>(fmaf(e < 4 ? b0[e & 3] : b1[e & 3], sin2_f(v[e] * (e < 4 ? a0[e & 3] : a1[e & 3])), v[e]));   // == conv_tiles.h snake_acc
      }
      if (sizeof(T) == 2) *reinterpret_cast<f16x8*>(y + i) = f16x8{(f16)o[0], (f16)o[1], (f16)o[2], (f16)o[3], (f16)o[4], (f16)o[5], (f16)o[6], (f16)o[7]};
      else {
        *reinterpret_cast<f32x4*>(y + i) = f32x4{(float)o[0], (float)o[1], (float)o[2], (float)o[3]};
        *reinterpret_cast<f32x4*>(y + i + 4) = f32x4{(float)o[4], (float)o[5], (float)o[6], (float)o[7]};
      }
    }
    return;
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const float v = to_f32(x[i]);
    y[i] = from_f32<T>(fmaf(inv_beta[c], sin2_f(v * alpha[c]), v));
  }
}

template <typename TO>
__global__ __launch_bounds__(256) void l2norm_kernel(const float* x, int ldx, TO* y, int ldy, int64_t rows, int dim,
                                                     float eps) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  for (int c = lane; c < dim; c += 64) { const float v = x[row * ldx + c]; s += v * v; }
  const float inv = 1.f / fmaxf(sqrtf(wave_sum(s)), eps);
  for (int c = lane; c < ldy; c += 64) y[row * ldy + c] = from_f32<TO>(c < dim ? x[row * ldx + c] * inv : 0.f);
}

__global__ void gaussian_sample_kernel(const float* stats, const float* noise, float* z, int64_t rows, int C,
                                       float noise_scale) {
  const int64_t total = rows * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / C;
    const int c = (int)(i - r * C);
    z[i] = stats[r * 2 * C + c] + noise[i] * expf(stats[r * 2 * C + C + c]) * noise_scale;
  }
}

__global__ void flip_kernel(const float* x, float* y, int64_t rows, int C) {
  const int64_t total = rows * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / C;
    const int c = (int)(i - r * C);
    y[i] = x[r * C + (C - 1 - c)];
  }
}

// -------------------------------------------------------------------- length regulator
// One 256-thread block per sequence: alpha rounding + inclusive scan (int64, exact).
__global__ __launch_bounds__(256) void lr_durations_kernel(jatts_ragged rg, const int64_t* d, float alpha,
                                                           int zero_rule, int64_t* d_eff, int64_t* cum,
                                                           int64_t* olens) {
  __shared__ int64_t part[256];
  const int b = blockIdx.x;
  const int row0 = rg.cu_rows[b];
  const int L = rg.cu_rows[b + 1] - row0;
  const int per = (L + 255) / 256;
  const int lo = threadIdx.x * per, hi = min(L, lo + per);
  int64_t s = 0;
  for (int t = lo; t < hi; ++t) {
    int64_t v = d[row0 + t];
    if (zero_rule == 1) v = 1;
    else if (alpha != 1.0f) v = (int64_t)rintf((float)v * alpha);  // torch.round(ds.float()*alpha).long()
    d_eff[row0 + t] = v;
    s += v;
  }
  part[threadIdx.x] = s;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {  // Hillis-Steele inclusive scan
    int64_t add = threadIdx.x >= o ? part[threadIdx.x - o] : 0;
    __syncthreads();
    part[threadIdx.x] += add;
    __syncthreads();
  }
  // zero_rule 2: a sequence whose durations sum to 0 gets every entry = 1 -- what the reference's B=1 inference() does with
  // it (length_regulator.py:86-94: `ds[ds.sum(dim=1).eq(0)] = 1` once the batch, i.e. that utterance, sums to 0)
  const bool fallback = zero_rule == 2 && part[255] == 0 && L > 0;
  int64_t run = fallback ? lo : (threadIdx.x ? part[threadIdx.x - 1] : 0);
  for (int t = lo; t < hi; ++t) {
    if (fallback) d_eff[row0 + t] = 1;
    run += d_eff[row0 + t];
    cum[row0 + t] = run;
  }
  if (threadIdx.x == 255) {
    olens[b] = fallback ? L : part[255];
    if (zero_rule == 2) olens[gridDim.x + b] = fallback ? 1 : 0;
  }
}

// One wave per output frame: idx = #{t : cum[t] <= f} (upper bound), then copy the row.
__global__ __launch_bounds__(256) void lr_gather_kernel(jatts_ragged rg, const int64_t* cum,
                                                        const int32_t* cu_out, const float* x, int dim,
                                                        float* out, int64_t* frame_index) {
  const int b = blockIdx.y;
  const int row0 = rg.cu_rows[b];
  const int Lin = rg.cu_rows[b + 1] - row0;
  const int o0 = cu_out[b];
  const int Lout = cu_out[b + 1] - o0;
  const int lane = threadIdx.x & 63;
  for (int i = 0; i < 4; ++i) {
    const int f = (blockIdx.x * 4 + i) * 4 + (threadIdx.x >> 6);
    if (f >= Lout) return;
    int lo = 0, hi = Lin;  // first t with cum[t] > f
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cum[row0 + mid] <= (int64_t)f) lo = mid + 1; else hi = mid;
    }
    const float* src = x + (int64_t)(row0 + lo) * dim;
    float* dst = out + (int64_t)(o0 + f) * dim;
    const bool past = lo >= Lin;   // frame beyond sum(d): pad_list's zero padding (padded batches of forward())
    for (int c = lane; c < dim; c += 64) dst[c] = past ? 0.f : src[c];
    if (frame_index && lane == 0) frame_index[o0 + f] = past ? -1 : lo;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void zero_pad_rows_kernel(jatts_ragged rg, T* x, int ld, int dim, const int32_t* valid_len) {
  const int b = blockIdx.y;
  const int row0 = rg.cu_rows[b];
  const int L = rg.cu_rows[b + 1] - row0;
  const int v = max(valid_len[b], 0);
  const int64_t n = (int64_t)(L - v) * dim;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / dim;
    x[(row0 + v + r) * (int64_t)ld + (i - r * dim)] = from_f32<T>(0.f);
  }
}

// Conditional flow matching training pair (flow_matching.py:118-121): y = (1 - (1 - sigma) t_b) z + t_b x1, u = x1 - (1 - sigma) z
__global__ __launch_bounds__(256) void cfm_mix_kernel(jatts_ragged rg, const float* x1, const float* z, const float* t, float sigma,
                                                      int dim, float* y, float* u) {
  const int b = blockIdx.y;
  const int row0 = rg.cu_rows[b], L = rg.cu_rows[b + 1] - row0;
  const float tb = t[b];
  const int64_t n = (int64_t)L * dim;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t o = (int64_t)row0 * dim + i;
    y[o] = (1.f - (1.f - sigma) * tb) * z[o] + tb * x1[o];
    u[o] = x1[o] - (1.f - sigma) * z[o];
  }
}

// sum_i (a_i - b_i)^2 in double, fixed summation order (one partial per block, then one block folds the partials)
__global__ __launch_bounds__(256) void sq_err_partial_kernel(const float* a, const float* b, int64_t n, double* part) {
  __shared__ double red[256];
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const double dlt = (double)a[i] - (double)b[i];
    s += dlt * dlt;
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
__global__ void fold_partials_kernel(const double* part, int n, float scale, float* out) {
  double s = 0.0;
  for (int i = 0; i < n; ++i) s += part[i];
  *out = (float)(s * (double)scale);
}

// ------------------------------------------------------------------ gaussian upsampling
// One wave per output frame; softmax over the text axis in two passes (T_text is small).
__global__ __launch_bounds__(256) void gaussian_upsample_kernel(jatts_ragged rg, const int64_t* d,
                                                                const int32_t* cu_out, const float* hs,
                                                                int dim, float delta, float* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* cs = reinterpret_cast<float*>(smem);  // centres c_t, [Lin]
  const int b = blockIdx.y;
  const int row0 = rg.cu_rows[b];
  const int Lin = rg.cu_rows[b + 1] - row0;
  const int o0 = cu_out[b];
  const int Lout = cu_out[b + 1] - o0;
  if ((int)blockIdx.x * 4 >= Lout) return;
  if (threadIdx.x == 0) {  // c = cumsum(d) - d/2 in float (length_regulator.py:143)
    int64_t run = 0;
    for (int t = 0; t < Lin; ++t) {
      const int64_t dv = d[row0 + t];
      run += dv;
      cs[t] = (float)run - (float)dv / 2.f;
    }
  }
  __syncthreads();
  const int f = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (f >= Lout) return;
  const int lane = threadIdx.x & 63;
  float mx = -INFINITY;
  for (int t = lane; t < Lin; t += 64) {
    const float df = (float)f - cs[t];
    mx = fmaxf(mx, -delta * df * df);
  }
  mx = wave_max(mx);
  float den = 0.f;
  for (int t = lane; t < Lin; t += 64) {
    const float df = (float)f - cs[t];
    den += expf(-delta * df * df - mx);
  }
  den = wave_sum(den);
  float* dst = out + (int64_t)(o0 + f) * dim;
  for (int c = lane; c < dim; c += 64) {
    float acc = 0.f;
    for (int t = 0; t < Lin; ++t) {
      const float df = (float)f - cs[t];
      acc += expf(-delta * df * df - mx) * hs[(int64_t)(row0 + t) * dim + c];
    }
    dst[c] = acc / den;
  }
}

// ------------------------------------------------------------------ HiFi-GAN output conv
constexpr int OUT_TT = 256;

template <typename T>
__global__ __launch_bounds__(256) void hifigan_output_kernel(jatts_ragged rg, const T* x0, const T* x1,
                                                             const T* x2, int n_in, float in_scale,
                                                             float slope, int C, int K, const float* w,
                                                             float bias, float* y) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int P = C + 1;
  float* xs = reinterpret_cast<float*>(smem);  // [(OUT_TT + K - 1)][C + 1]
  float* ws = xs + (OUT_TT + K - 1) * P;       // [K][C]
  const int b = blockIdx.y;
  const int64_t row0 = (int64_t)rg.cu_rows[b] * rg.len_mul;
  const int L = (rg.cu_rows[b + 1] - rg.cu_rows[b]) * rg.len_mul;
  const int t0 = blockIdx.x * OUT_TT;
  if (t0 >= L) return;
  const int pad = (K - 1) / 2;
  for (int u = threadIdx.x; u < K * C; u += 256) ws[u] = w[u];
  const int rows = OUT_TT + K - 1;
  for (int u = threadIdx.x; u < rows * C; u += 256) {
    const int r = u / C, c = u - r * C;
    const int pos = t0 - pad + r;
    float v = 0.f;
    if (pos >= 0 && pos < L) {
      const int64_t o = (row0 + pos) * C + c;
      v = to_f32(x0[o]);
      if (n_in > 1) v += to_f32(x1[o]);
      if (n_in > 2) v += to_f32(x2[o]);
      v = lrelu(v * in_scale, slope);
    }
    xs[r * P + c] = v;
  }
  __syncthreads();
  const int t = threadIdx.x;
  if (t0 + t >= L) return;
  float acc = bias;
  for (int k = 0; k < K; ++k) {
    const float* xr = xs + (t + k) * P;
    const float* wr = ws + k * C;
    for (int c = 0; c < C; ++c) acc += wr[c] * xr[c];
  }
  y[row0 + t0 + t] = tanhf(acc);
}

// Same, for C % 8 == 0 (every HiFi-GAN config in the reference: channels / 2^n_upsamples): 16-byte batched staging,
// rows padded to C + 4 floats so that each tap is read as conflict-free float4s.
template <typename T>
__global__ __launch_bounds__(256) void hifigan_output_vec_kernel(jatts_ragged rg, const T* x0, const T* x1,
                                                                 const T* x2, int n_in, float in_scale,
                                                                 float slope, int C, int K, const float* w,
                                                                 float bias, float* y) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int P = C + 4;
  float* xs = reinterpret_cast<float*>(smem);  // [(OUT_TT + K - 1)][C + 4]
  float* ws = xs + (OUT_TT + K - 1) * P;       // [K][C]
  const int b = blockIdx.y;
  const int64_t row0 = (int64_t)rg.cu_rows[b] * rg.len_mul;
  const int L = (rg.cu_rows[b + 1] - rg.cu_rows[b]) * rg.len_mul;
  const int t0 = blockIdx.x * OUT_TT;
  if (t0 >= L) return;
  const int pad = (K - 1) / 2;
  for (int u = threadIdx.x; u < K * C; u += 256) ws[u] = w[u];
  const int rows = OUT_TT + K - 1, UPR = C / 8, total = rows * UPR;
  constexpr int UB = 4;
  for (int u0 = threadIdx.x; u0 < total; u0 += UB * 256) {
    float v[UB][8];
    bool ok[UB];
#pragma unroll
    for (int i = 0; i < UB; ++i) {
      const int u = u0 + i * 256, r = u / UPR, cu = u - r * UPR;
      const int pos = t0 - pad + r;
      ok[i] = u < total && pos >= 0 && pos < L;
      if (ok[i]) {
        const int64_t o = (row0 + pos) * C + cu * 8;
        load8f<T>(x0 + o, v[i]);
        if (n_in > 1) {
          float t[8];
          load8f<T>(x1 + o, t);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[i][e] += t[e];
        }
        if (n_in > 2) {
          float t[8];
          load8f<T>(x2 + o, t);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[i][e] += t[e];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < UB; ++i) {
      const int u = u0 + i * 256;
      if (u >= total) continue;
      const int r = u / UPR, cu = u - r * UPR;
      f32x4 lo, hi;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        lo[e] = ok[i] ? lrelu(v[i][e] * in_scale, slope) : 0.f;
        hi[e] = ok[i] ? lrelu(v[i][e + 4] * in_scale, slope) : 0.f;
      }
      *reinterpret_cast<f32x4*>(xs + r * P + cu * 8) = lo;
      *reinterpret_cast<f32x4*>(xs + r * P + cu * 8 + 4) = hi;
    }
  }
  __syncthreads();
  const int t = threadIdx.x;
  if (t0 + t >= L) return;
  float acc = bias;
  for (int k = 0; k < K; ++k) {
    const float* xr = xs + (t + k) * P;
    const float* wr = ws + k * C;
    for (int c = 0; c < C; c += 4) {
      const f32x4 xv = *reinterpret_cast<const f32x4*>(xr + c), wv = *reinterpret_cast<const f32x4*>(wr + c);
      acc += wv[0] * xv[0];   // same summation order as the scalar kernel
      acc += wv[1] * xv[1];
      acc += wv[2] * xv[2];
      acc += wv[3] * xv[3];
    }
  }
  y[row0 + t0 + t] = tanhf(acc);
}

// float [-1, 1] -> 16-bit PCM exactly as libsndfile's PCM_16 writer does for float input (sf.write(..., "PCM_16"),
// tts_decode.py:250-255): lrintf(x * 32767.f) in float arithmetic; values are clamped to [-1, 1] first (tanh output).
__global__ void pcm16_kernel(const float* x, int16_t* y, int64_t n) {
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * blockDim.x * 4) {
    if (i + 3 < n && (reinterpret_cast<uintptr_t>(x + i) & 15) == 0 && (reinterpret_cast<uintptr_t>(y + i) & 7) == 0) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(x + i);
      short o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (short)rintf(fminf(fmaxf(v[e], -1.f), 1.f) * 32767.f);
      *reinterpret_cast<uint2*>(y + i) = uint2{(unsigned)(unsigned short)o[0] | ((unsigned)(unsigned short)o[1] << 16),
                                               (unsigned)(unsigned short)o[2] | ((unsigned)(unsigned short)o[3] << 16)};
    } else {
      for (int64_t k2 = i; k2 < n && k2 < i + 4; ++k2) y[k2] = (int16_t)rintf(fminf(fmaxf(x[k2], -1.f), 1.f) * 32767.f);
    }
  }
}

}  // namespace

#define S_ ((hipStream_t)stream)

extern "C" int jatts_embed_scale(const int64_t* ids, int64_t rows, const float* table, int32_t dim,
                                 float scale, float* out, int64_t vocab, int64_t* n_bad, void* stream) {
  if (!ids || !table || !out) return jatts_set_error_msg(JATTS_ERR_ARG, "embed_scale: null pointer");
  if (rows <= 0) return JATTS_OK;
  hipLaunchKernelGGL(embed_scale_kernel, dim3((unsigned)rows), dim3(128), 0, S_, ids, rows, table, dim, scale, out, vocab, n_bad);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

extern "C" int jatts_layernorm(const void* x, int32_t in_dtype, int32_t ldx, void* y, int32_t out_dtype,
                               int32_t ldy, int64_t rows, int32_t dim, const float* gamma,
                               const float* beta, float eps, void* stream) {
  if (!x || !y || !gamma || !beta) return jatts_set_error_msg(JATTS_ERR_ARG, "layernorm: null pointer");
  if (dim > 64 * LN_MAXV) return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "layernorm: dim > 2048");
  if (rows <= 0) return JATTS_OK;
  dim3 grid((unsigned)((rows + 3) / 4)), blk(256);
#define LN_GO(TI, TO) \
  hipLaunchKernelGGL((layernorm_kernel<TI, TO>), grid, blk, 0, S_, (const TI*)x, ldx, (TO*)y, ldy, rows, dim, gamma, beta, eps)
  if (in_dtype == JATTS_F32 && out_dtype == JATTS_F32) LN_GO(float, float);
  else if (in_dtype == JATTS_F32 && out_dtype == JATTS_F16) LN_GO(float, f16);
  else if (in_dtype == JATTS_F16 && out_dtype == JATTS_F16) LN_GO(f16, f16);
  else if (in_dtype == JATTS_F16 && out_dtype == JATTS_F32) LN_GO(f16, float);
  else return jatts_set_error_msg(JATTS_ERR_ARG, "layernorm: unknown dtype");
#undef LN_GO
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

static int affine_launch(const float* x, int32_t ldx, void* y, int32_t out_dtype, int32_t ldy, int64_t rows, int32_t dim,
                         const float* scale, const float* shift, int W, void* stream) {
  if (!x || !y) return jatts_set_error_msg(JATTS_ERR_ARG, "affine_cast: null pointer");
  if (rows <= 0 || W <= 0) return JATTS_OK;
  const int64_t total = rows * (int64_t)W;
  dim3 grid((unsigned)min((int64_t)4096, (total + 255) / 256));
  if (out_dtype == JATTS_F16)
    hipLaunchKernelGGL(affine_cast_kernel<f16>, grid, dim3(256), 0, S_, x, ldx, (f16*)y, ldy, rows, dim, scale, shift, W);
  else if (out_dtype == JATTS_F32)
    hipLaunchKernelGGL(affine_cast_kernel<float>, grid, dim3(256), 0, S_, x, ldx, (float*)y, ldy, rows, dim, scale, shift, W);
  else return jatts_set_error_msg(JATTS_ERR_ARG, "affine_cast: unknown dtype");
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

extern "C" int jatts_affine_cast(const float* x, int32_t ldx, void* y, int32_t out_dtype, int32_t ldy,
                                 int64_t rows, int32_t dim, const float* scale, const float* shift,
                                 void* stream) {
  return affine_launch(x, ldx, y, out_dtype, ldy, rows, dim, scale, shift, ldy, stream);
}

extern "C" int jatts_affine_slice(const float* x, int32_t ldx, void* y, int32_t out_dtype, int32_t ldy,
                                  int64_t rows, int32_t dim, const float* scale, const float* shift,
                                  void* stream) {
  return affine_launch(x, ldx, y, out_dtype, ldy, rows, dim, scale, shift, dim, stream);
}

extern "C" int jatts_glu_dwconv_bn_swish(const jatts_ragged* rg, int32_t dtype, const void* x, void* y,
                                         int32_t channels, int32_t k_w, const float* w_dw,
                                         const float* bn_scale, const float* bn_shift, void* stream) {
  if (!rg || !x || !y || !w_dw || !bn_scale || !bn_shift) return jatts_set_error_msg(JATTS_ERR_ARG, "glu_dwconv: null pointer");
  if (!(k_w & 1)) return jatts_set_error_msg(JATTS_ERR_ARG, "glu_dwconv: k_w must be odd");
  if (rg->max_len <= 0) return JATTS_OK;
  dim3 grid((unsigned)((rg->max_len + DW_TT - 1) / DW_TT), (unsigned)rg->n_seq, (unsigned)((channels + DW_CB - 1) / DW_CB));
  const size_t lds = (size_t)(DW_TT + 2 * k_w - 1) * (DW_CB + 1) * sizeof(float);   // input tile + transposed taps
  if (lds > 64 * 1024) return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "glu_dwconv: kernel too wide");
#define DW_GO(TT, KM) hipLaunchKernelGGL((glu_dw_kernel<TT, KM>), grid, dim3(256), lds, S_, *rg, (const TT*)x, (TT*)y, channels, k_w, w_dw, bn_scale, bn_shift)
  if (dtype == JATTS_F16) {
    if (k_w <= 8) DW_GO(f16, 8); else if (k_w <= 32) DW_GO(f16, 32); else DW_GO(f16, 0);
  } else if (dtype == JATTS_F32) {
    if (k_w <= 8) DW_GO(float, 8); else if (k_w <= 32) DW_GO(float, 32); else DW_GO(float, 0);
  }
#undef DW_GO
  else return jatts_set_error_msg(JATTS_ERR_ARG, "glu_dwconv: unknown dtype");
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

extern "C" int jatts_predictor_head(int32_t dtype, const void* x, int32_t ldx, int64_t rows, int32_t dim,
                                    const float* w, float b, float* v_out, int64_t* dur_out, float offset,
                                    void* stream) {
  if (!x || !w) return jatts_set_error_msg(JATTS_ERR_ARG, "predictor_head: null pointer");
  if (rows <= 0) return JATTS_OK;
  dim3 grid((unsigned)((rows + 3) / 4));
  if (dtype == JATTS_F16)
    hipLaunchKernelGGL(predictor_head_kernel<f16>, grid, dim3(256), 0, S_, (const f16*)x, ldx, rows, dim, w, b, v_out, dur_out, offset);
  else if (dtype == JATTS_F32)
    hipLaunchKernelGGL(predictor_head_kernel<float>, grid, dim3(256), 0, S_, (const float*)x, ldx, rows, dim, w, b, v_out, dur_out, offset);
  else return jatts_set_error_msg(JATTS_ERR_ARG, "predictor_head: unknown dtype");
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

extern "C" int jatts_variance_embed_add(const jatts_ragged* rg, float* hs, int32_t dim, const float* p,
                                        const float* wp, const float* bp, int32_t kp, const float* e,
                                        const float* we, const float* be, int32_t ke, void* stream) {
  if (!rg || !hs || !p || !wp || !bp || !e || !we || !be) return jatts_set_error_msg(JATTS_ERR_ARG, "variance_embed_add: null pointer");
  if (rg->max_len <= 0) return JATTS_OK;
  hipLaunchKernelGGL(variance_embed_kernel, dim3((unsigned)rg->max_len, (unsigned)rg->n_seq), dim3(128), 0, S_, *rg, hs, dim, p, wp, bp, kp, e, we, be, ke);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

extern "C" int jatts_gated_tanh_sigmoid(const jatts_ragged* rg, int32_t dtype, const void* x, const float* gseq,
                                        void* y, int32_t channels, void* stream) {
  if (!rg || !x || !y) return jatts_set_error_msg(JATTS_ERR_ARG, "gated_tanh_sigmoid: null pointer");
  if (rg->max_len <= 0) return JATTS_OK;
  dim3 grid((unsigned)rg->max_len, (unsigned)rg->n_seq);
  if (dtype == JATTS_F16)
    hipLaunchKernelGGL(gated_kernel<f16>, grid, dim3(128), 0, S_, *rg, (const f16*)x, gseq, (f16*)y, channels);
  else if (dtype == JATTS_F32)
    hipLaunchKernelGGL(gated_kernel<float>, grid, dim3(128), 0, S_, *rg, (const float*)x, gseq, (float*)y, channels);
  else return jatts_set_error_msg(JATTS_ERR_ARG, "gated_tanh_sigmoid: unknown dtype");
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

extern "C" int jatts_groupnorm_mish(const jatts_ragged* rg, const void* x, int32_t in_dtype, void* y, int32_t out_dtype,
                                    int32_t channels, int32_t groups, const float* gamma, const float* beta, float eps,
                                    const float* addvec, float* workspace, void* stream) {
  if (!rg || !x || !y || !gamma || !beta) return jatts_set_error_msg(JATTS_ERR_ARG, "groupnorm_mish: null pointer");
  if (groups < 1 || channels % groups) return jatts_set_error_msg(JATTS_ERR_ARG, "groupnorm_mish: channels % groups != 0");
  if (rg->max_len <= 0) return JATTS_OK;
  const int gc = channels / groups;
  if (workspace && (gc & 7) == 0 && GN_TCH * gc <= 256 * GN_MAXU * 8) {   // time-split two-launch form
    const int n_chunks = (rg->max_len + GN_TCH - 1) / GN_TCH;
    dim3 grid3((unsigned)groups, (unsigned)rg->n_seq, (unsigned)n_chunks), blk3(256);
    static const int rows_env = [] { const char* e = getenv("JATTS_GN_ROWS"); return e ? atoi(e) : 1; }();
    const bool rows_form = rows_env && groups <= GN_MAXG && (channels & 7) == 0;
#define GN2_GO(TI, TO)                                                                                                   \
  do {                                                                                                                   \
    hipLaunchKernelGGL((gn_partial_kernel<TI>), grid3, blk3, 0, S_, *rg, (const TI*)x, channels, groups, workspace, n_chunks); \
    if (rows_form)                                                                                                          \
      hipLaunchKernelGGL((gn_apply_rows_kernel<TI, TO>), dim3((unsigned)((rg->max_len + GN_AROWS - 1) / GN_AROWS), (unsigned)rg->n_seq), blk3, 0, S_, *rg, (const TI*)x, (TO*)y, channels, groups, gamma, beta, eps, addvec, workspace, n_chunks); \
    else                                                                                                                    \
      hipLaunchKernelGGL((gn_apply_kernel<TI, TO>), grid3, blk3, 0, S_, *rg, (const TI*)x, (TO*)y, channels, groups, gamma, beta, eps, addvec, workspace, n_chunks); \
  } while (0)
    if (in_dtype == JATTS_F32 && out_dtype == JATTS_F32) GN2_GO(float, float);
    else if (in_dtype == JATTS_F32 && out_dtype == JATTS_F16) GN2_GO(float, f16);
    else if (in_dtype == JATTS_F16 && out_dtype == JATTS_F16) GN2_GO(f16, f16);
    else if (in_dtype == JATTS_F16 && out_dtype == JATTS_F32) GN2_GO(f16, float);
    else return jatts_set_error_msg(JATTS_ERR_ARG, "groupnorm_mish: unknown dtype");
#undef GN2_GO
    JATTS_CHECK_LAUNCH();
    return JATTS_OK;
  }
  dim3 grid((unsigned)groups, (unsigned)rg->n_seq), blk(256);
#define GN_GO(TI, TO) \
  hipLaunchKernelGGL((groupnorm_mish_kernel<TI, TO>), grid, blk, 0, S_, *rg, (const TI*)x, (TO*)y, channels, groups, gamma, beta, eps, addvec)
  if (in_dtype == JATTS_F32 && out_dtype == JATTS_F32) GN_GO(float, float);
  else if (in_dtype == JATTS_F32 && out_dtype == JATTS_F16) GN_GO(float, f16);
  else if (in_dtype == JATTS_F16 && out_dtype == JATTS_F16) GN_GO(f16, f16);
  else if (in_dtype == JATTS_F16 && out_dtype == JATTS_F32) GN_GO(f16, float);
  else return jatts_set_error_msg(JATTS_ERR_ARG, "groupnorm_mish: unknown dtype");
#undef GN_GO
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

extern "C" int jatts_snakebeta(int32_t dtype, const void* x, void* y, int64_t rows, int32_t channels, const float* alpha,
                               const float* inv_beta, void* stream) {
  if (!x || !y || !alpha || !inv_beta) return jatts_set_error_msg(JATTS_ERR_ARG, "snakebeta: null pointer");
  if (rows <= 0) return JATTS_OK;
  const int64_t total = rows * channels;
  dim3 grid((unsigned)min((int64_t)4096, (total + 255) / 256));
  if (dtype == JATTS_F16)
    hipLaunchKernelGGL(snakebeta_kernel<f16>, grid, dim3(256), 0, S_, (const f16*)x, (f16*)y, rows, channels, alpha, inv_beta);
  else if (dtype == JATTS_F32)
    hipLaunchKernelGGL(snakebeta_kernel<float>, grid, dim3(256), 0, S_, (const float*)x, (float*)y, rows, channels, alpha, inv_beta);
  else return jatts_set_error_msg(JATTS_ERR_ARG, "snakebeta: unknown dtype");
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

extern "C" int jatts_l2_normalize(const float* x, int32_t ldx, void* y, int32_t out_dtype, int32_t ldy, int64_t rows,
                                  int32_t dim, float eps, void* stream) {
  if (!x || !y) return jatts_set_error_msg(JATTS_ERR_ARG, "l2_normalize: null pointer");
  if (rows <= 0) return JATTS_OK;
  dim3 grid((unsigned)((rows + 3) / 4));
  if (out_dtype == JATTS_F16)
    hipLaunchKernelGGL(l2norm_kernel<f16>, grid, dim3(256), 0, S_, x, ldx, (f16*)y, ldy, rows, dim, eps);
  else if (out_dtype == JATTS_F32)
    hipLaunchKernelGGL(l2norm_kernel<float>, grid, dim3(256), 0, S_, x, ldx, (float*)y, ldy, rows, dim, eps);
  else return jatts_set_error_msg(JATTS_ERR_ARG, "l2_normalize: unknown dtype");
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

extern "C" int jatts_gaussian_sample(const float* stats, const float* noise, float* z, int64_t rows, int32_t channels,
                                     float noise_scale, void* stream) {
  if (!stats || !noise || !z) return jatts_set_error_msg(JATTS_ERR_ARG, "gaussian_sample: null pointer");
  if (rows <= 0) return JATTS_OK;
  const int64_t total = rows * channels;
  hipLaunchKernelGGL(gaussian_sample_kernel, dim3((unsigned)min((int64_t)4096, (total + 255) / 256)), dim3(256), 0, S_,
                     stats, noise, z, rows, channels, noise_scale);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

extern "C" int jatts_flip_channels(const float* x, float* y, int64_t rows, int32_t channels, void* stream) {
  if (!x || !y || x == y) return jatts_set_error_msg(JATTS_ERR_ARG, "flip_channels: bad pointers");
  if (rows <= 0) return JATTS_OK;
  const int64_t total = rows * channels;
  hipLaunchKernelGGL(flip_kernel, dim3((unsigned)min((int64_t)4096, (total + 255) / 256)), dim3(256), 0, S_, x, y, rows, channels);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

extern "C" int jatts_add_seq_vector(const jatts_ragged* rg, float* hs, int32_t dim, const float* vec,
                                    void* stream) {
  if (!rg || !hs || !vec) return jatts_set_error_msg(JATTS_ERR_ARG, "add_seq_vector: null pointer");
  if (rg->max_len <= 0) return JATTS_OK;
  hipLaunchKernelGGL(add_seq_vector_kernel, dim3((unsigned)rg->max_len, (unsigned)rg->n_seq), dim3(128), 0, S_, *rg, hs, dim, vec);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

extern "C" int jatts_lr_durations(const jatts_ragged* rg, const int64_t* d, float alpha, int32_t zero_rule,
                                  int64_t* d_eff, int64_t* cum, int64_t* olens, void* stream) {
  if (!rg || !d || !d_eff || !cum || !olens) return jatts_set_error_msg(JATTS_ERR_ARG, "lr_durations: null pointer");
  if (zero_rule < 0 || zero_rule > 2) return jatts_set_error_msg(JATTS_ERR_ARG, "lr_durations: zero_rule must be 0, 1 or 2");
  if (rg->n_seq <= 0) return JATTS_OK;
  hipLaunchKernelGGL(lr_durations_kernel, dim3((unsigned)rg->n_seq), dim3(256), 0, S_, *rg, d, alpha, zero_rule, d_eff, cum, olens);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

extern "C" int jatts_lr_gather(const jatts_ragged* rg_in, const int64_t* cum, const int32_t* cu_out,
                               int32_t max_out_len, const float* x, int32_t dim, float* out,
                               int64_t* frame_index, void* stream) {
  if (!rg_in || !cum || !cu_out || !x || !out) return jatts_set_error_msg(JATTS_ERR_ARG, "lr_gather: null pointer");
  if (max_out_len <= 0) return JATTS_OK;
  hipLaunchKernelGGL(lr_gather_kernel, dim3((unsigned)((max_out_len + 15) / 16), (unsigned)rg_in->n_seq), dim3(256), 0, S_, *rg_in, cum, cu_out, x, dim, out, frame_index);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

extern "C" int jatts_zero_pad_rows(const jatts_ragged* rg, void* x, int32_t dtype, int32_t ld, int32_t dim, const int32_t* valid_len,
                                   void* stream) {
  if (!rg || !x || !valid_len) return jatts_set_error_msg(JATTS_ERR_ARG, "zero_pad_rows: null pointer");
  if (rg->n_seq <= 0 || rg->max_len <= 0 || dim <= 0) return JATTS_OK;
  const int64_t per = (int64_t)rg->max_len * rg->len_mul * dim;
  dim3 grid((unsigned)((per + 1023) / 1024 < 64 ? (per + 1023) / 1024 : 64), (unsigned)rg->n_seq);
  if (dtype == JATTS_F32) hipLaunchKernelGGL(zero_pad_rows_kernel<float>, grid, dim3(256), 0, S_, *rg, (float*)x, ld, dim, valid_len);
  else if (dtype == JATTS_F16) hipLaunchKernelGGL(zero_pad_rows_kernel<f16>, grid, dim3(256), 0, S_, *rg, (f16*)x, ld, dim, valid_len);
  else return jatts_set_error_msg(JATTS_ERR_ARG, "zero_pad_rows: unknown dtype");
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

extern "C" int jatts_cfm_mix(const jatts_ragged* rg, const float* x1, const float* z, const float* t, float sigma_min, int32_t dim,
                             float* y, float* u, void* stream) {
  if (!rg || !x1 || !z || !t || !y || !u) return jatts_set_error_msg(JATTS_ERR_ARG, "cfm_mix: null pointer");
  if (rg->n_seq <= 0 || rg->max_len <= 0) return JATTS_OK;
  const int64_t per = ((int64_t)rg->max_len * dim + 255) / 256;
  hipLaunchKernelGGL(cfm_mix_kernel, dim3((unsigned)(per < 256 ? per : 256), (unsigned)rg->n_seq), dim3(256), 0, S_, *rg, x1, z, t, sigma_min, dim, y, u);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

extern "C" int jatts_sq_err_sum(const float* a, const float* b, int64_t n, float scale, float* out, double* workspace, void* stream) {
  if (!a || !b || !out || !workspace) return jatts_set_error_msg(JATTS_ERR_ARG, "sq_err_sum: null pointer");
  const int blocks = (int)((n + 4095) / 4096 < 256 ? (n + 4095) / 4096 : 256);
  hipLaunchKernelGGL(sq_err_partial_kernel, dim3((unsigned)(blocks < 1 ? 1 : blocks)), dim3(256), 0, S_, a, b, n, workspace);
  hipLaunchKernelGGL(fold_partials_kernel, dim3(1), dim3(1), 0, S_, workspace, blocks < 1 ? 1 : blocks, scale, out);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

extern "C" int jatts_gaussian_upsample(const jatts_ragged* rg_in, const int64_t* d, const int32_t* cu_out,
                                       int32_t max_out_len, const float* hs, int32_t dim, float delta,
                                       float* out, void* stream) {
  if (!rg_in || !d || !cu_out || !hs || !out) return jatts_set_error_msg(JATTS_ERR_ARG, "gaussian_upsample: null pointer");
  if (max_out_len <= 0) return JATTS_OK;
  const size_t lds = (size_t)rg_in->max_len * sizeof(float);
  if (lds > 64 * 1024) return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "gaussian_upsample: text too long");
  hipLaunchKernelGGL(gaussian_upsample_kernel, dim3((unsigned)((max_out_len + 3) / 4), (unsigned)rg_in->n_seq), dim3(256), lds, S_, *rg_in, d, cu_out, hs, dim, delta, out);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

extern "C" int jatts_hifigan_output(const jatts_ragged* rg, int32_t dtype, const void* const* x, int32_t n_in,
                                    float in_scale, float slope, int32_t c_in, int32_t k_w, const float* w,
                                    float bias, float* y, void* stream) {
  if (!rg || !x || !x[0] || !w || !y || n_in < 1 || n_in > 3) return jatts_set_error_msg(JATTS_ERR_ARG, "hifigan_output: bad arguments");
  if (rg->max_len <= 0) return JATTS_OK;
  const int64_t maxL = (int64_t)rg->max_len * rg->len_mul;
  dim3 grid((unsigned)((maxL + OUT_TT - 1) / OUT_TT), (unsigned)rg->n_seq);
  const bool vec = (c_in & 7) == 0;
  const size_t lds = ((size_t)(OUT_TT + k_w - 1) * (c_in + (vec ? 4 : 1)) + (size_t)k_w * c_in) * sizeof(float);
  if (lds > 64 * 1024) return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "hifigan_output: c_in*k_w too large");
  const void* x1 = n_in > 1 ? x[1] : nullptr;
  const void* x2 = n_in > 2 ? x[2] : nullptr;
  if (vec && dtype == JATTS_F16) {
    hipLaunchKernelGGL(hifigan_output_vec_kernel<f16>, grid, dim3(256), lds, S_, *rg, (const f16*)x[0], (const f16*)x1, (const f16*)x2, n_in, in_scale, slope, c_in, k_w, w, bias, y);
    JATTS_CHECK_LAUNCH();
    return JATTS_OK;
  }
  if (vec && dtype == JATTS_F32) {
    hipLaunchKernelGGL(hifigan_output_vec_kernel<float>, grid, dim3(256), lds, S_, *rg, (const float*)x[0], (const float*)x1, (const float*)x2, n_in, in_scale, slope, c_in, k_w, w, bias, y);
    JATTS_CHECK_LAUNCH();
    return JATTS_OK;
  }
  if (dtype == JATTS_F16)
    hipLaunchKernelGGL(hifigan_output_kernel<f16>, grid, dim3(256), lds, S_, *rg, (const f16*)x[0], (const f16*)x1, (const f16*)x2, n_in, in_scale, slope, c_in, k_w, w, bias, y);
  else if (dtype == JATTS_F32)
    hipLaunchKernelGGL(hifigan_output_kernel<float>, grid, dim3(256), lds, S_, *rg, (const float*)x[0], (const float*)x1, (const float*)x2, n_in, in_scale, slope, c_in, k_w, w, bias, y);
  else return jatts_set_error_msg(JATTS_ERR_ARG, "hifigan_output: unknown dtype");
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

extern "C" int jatts_pcm16(const float* x, int64_t n, int16_t* y, void* stream) {
  if (!x || !y) return jatts_set_error_msg(JATTS_ERR_ARG, "pcm16: null pointer");
  if (n <= 0) return JATTS_OK;
  const int64_t blocks = (n / 4 + 255) / 256;
  hipLaunchKernelGGL(pcm16_kernel, dim3((unsigned)(blocks < 8192 ? (blocks < 1 ? 1 : blocks) : 8192)), dim3(256), 0, S_, x, y, n);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}
