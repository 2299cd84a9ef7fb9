// Training-side kernels, second slice of SURVEY §8 f.4: the backward passes (and the train-mode forward pieces) of the
// FastSpeech2 layers around the conv op -- LayerNorm, ReLU / tanh / Swish / GLU, depthwise conv, batch-statistics BatchNorm,
// embedding, the legacy rel_shift + masked softmax of the attention, the length regulator, the masked losses, dropout, Adam.
// All f32, rows x channels row-major like the inference kernels.  HBM-bound element / row / column reductions: one coalesced
// read of each operand; the [C]-sized parameter gradients are summed deterministically (det_reduce.h: slabs + last-arriver sum, no atomics on data).
// Reference: torch autograd of jatts/modules/conformer/{encoder_layer,convolution}.py, modules/transformer/{attention,layer_norm}.py,
// modules/{duration_predictor,variance_predictor,length_regulator,pre_postnets}.py, jatts/losses/*.py, jatts/trainers/fastspeech2.py:24-100.
#include "common.h"
#include "det_reduce.h"

namespace {

constexpr int LN_MAXPL = 24;   // LayerNorm backward: channels per lane (C <= 1536)
constexpr int LN_GROUP = 32;   // workgroups per first-level group of the two-level slab sums (det_reduce_tree)

// ------------------------------------------------------------------ LayerNorm backward
// y = (x - mean) * rstd * g + b over the channels of each row (biased variance, eps inside the sqrt).
// dx = rstd * (dyg - mean(dyg) - xhat * mean(dyg * xhat)), dyg = dy * g;  dg += sum_rows dy * xhat;  db += sum_rows dy.
// One wave per row (lanes stride the channels); per-lane partial dg / db live in registers over the wave's rows and are
// folded through LDS into one slab per workgroup; the last-arriving workgroup adds the slabs in a fixed order (det_reduce_tree).
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ dy, int lddy,
                                                            const float* __restrict__ g, int64_t rows, int C, float eps,
                                                            float* __restrict__ dx, int lddx, float* __restrict__ dg, float* __restrict__ db,
                                                            float* __restrict__ slabs, unsigned* __restrict__ tickets) {
  __shared__ __attribute__((aligned(16))) float red[4][4][64];     // ([2][4][64] for the workgroup's own fold; all of it for the slab sums)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float pg[LN_MAXPL], pb[LN_MAXPL], gv[LN_MAXPL];
#pragma unroll
  for (int j = 0; j < LN_MAXPL; ++j) {
    pg[j] = pb[j] = 0.f;
    const int c = lane + 64 * j;
    gv[j] = c < C ? g[c] : 0.f;
  }
  const float invC = 1.f / (float)C;
  for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < rows; r += (int64_t)gridDim.x * 4) {
    const float* xr = x + r * ldx;
    const float* dr = dy + r * lddy;
    float xv[LN_MAXPL], dv[LN_MAXPL];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXPL; ++j) {
      const int c = lane + 64 * j;
      xv[j] = c < C ? xr[c] : 0.f;
      dv[j] = c < C ? dr[c] : 0.f;
      s += xv[j];
    }
    const float mean = wave_sum(s) * invC;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXPL; ++j) {
      const int c = lane + 64 * j;
      const float d = c < C ? xv[j] - mean : 0.f;
      xv[j] = d;
      q += d * d;
    }
    const float rstd = rsqrtf(wave_sum(q) * invC + eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXPL; ++j) {
      xv[j] *= rstd;                       // xhat
      const float dyg = dv[j] * gv[j];
      s1 += dyg;
      s2 += dyg * xv[j];
      pg[j] += dv[j] * xv[j];
      pb[j] += dv[j];
    }
    s1 = wave_sum(s1) * invC;
    s2 = wave_sum(s2) * invC;
    if (dx) {
      float* o = dx + r * lddx;
#pragma unroll
      for (int j = 0; j < LN_MAXPL; ++j) {
        const int c = lane + 64 * j;
        if (c < C) o[c] = rstd * (dv[j] * gv[j] - s1 - xv[j] * s2);
      }
    }
  }
  if (!dg) return;
  float* slab = slabs + (int64_t)blockIdx.x * 2 * C;     // this workgroup's partial [dg | db]
  for (int j = 0; j < LN_MAXPL; ++j) {
    if (64 * j >= C) break;
    red[0][wave][lane] = pg[j];
    red[1][wave][lane] = pb[j];
    __syncthreads();
    if (wave == 0) {
      const int c = lane + 64 * j;
      if (c < C) {
        slab[c] = red[0][0][lane] + red[0][1][lane] + red[0][2][lane] + red[0][3][lane];
        slab[C + c] = red[1][0][lane] + red[1][1][lane] + red[1][2][lane] + red[1][3][lane];
      }
    }
    __syncthreads();
  }
  det_reduce_tree<4>(slabs, tickets, (int)blockIdx.x, (int)gridDim.x, LN_GROUP, 2 * C, &red[0][0][0], 1024, reinterpret_cast<unsigned*>(&red[0][0][0]),
                     [&](int i, float t) { if (i < C) dg[i] += t; else db[i - C] += t; });
}

// ------------------------------------------------------------------ activations
__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + __expf(-v)); }
// mode: 1 relu, 2 tanh, 3 swish, 4 mish (x tanh(softplus x), torch's softplus threshold 20)
__device__ __forceinline__ float act_val(float x, int mode) {
  if (mode == 1) return fmaxf(x, 0.f);
  if (mode == 2) return tanhf(x);
  if (mode == 3) return x * sigmoidf_(x);
  return mish_f(x);
}
__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, int mode) {
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * 1024) {
    if (i + 3 < n) {
      f32x4 v = *reinterpret_cast<const f32x4*>(x + i), o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = act_val(v[e], mode);
      *reinterpret_cast<f32x4*>(y + i) = o;
    } else {
      for (int64_t k = i; k < n; ++k) y[k] = act_val(x[k], mode);
    }
  }
}
__device__ __forceinline__ float act_grad(float x, int mode) {
  if (mode == 1) return x > 0.f ? 1.f : 0.f;
  if (mode == 2) { const float t = tanhf(x); return 1.f - t * t; }
  const float s = sigmoidf_(x);
  if (mode == 3) return s * (1.f + x * (1.f - s));
  if (x > 20.f) return 1.f;
  const float w = expf(x), n = w * (w + 2.f), t = n / (n + 2.f);     // mish: d/dx [x tanh(sp)] = tanh(sp) + x (1 - tanh^2(sp)) sigmoid(x)
  return t + x * (1.f - t * t) * s;
}
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx,
                                                      int64_t n, int mode) {
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * 1024) {
    if (i + 3 < n) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(x + i), d = *reinterpret_cast<const f32x4*>(dy + i);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = d[e] * act_grad(v[e], mode);
      *reinterpret_cast<f32x4*>(dx + i) = o;
    } else {
      for (int64_t k = i; k < n; ++k) dx[k] = dy[k] * act_grad(x[k], mode);
    }
  }
}
// GLU over the channel halves of a [rows][2C] matrix: y = a * sigmoid(b)
__global__ __launch_bounds__(256) void glu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t rows, int C) {
  const int64_t n = rows * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / C;
    const int c = (int)(i - r * C);
    y[i] = x[r * 2 * C + c] * sigmoidf_(x[r * 2 * C + C + c]);
  }
}
__global__ __launch_bounds__(256) void glu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx,
                                                      int64_t rows, int C) {
  const int64_t n = rows * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / C;
    const int c = (int)(i - r * C);
    const float a = x[r * 2 * C + c], s = sigmoidf_(x[r * 2 * C + C + c]), d = dy[i];
    dx[r * 2 * C + c] = d * s;
    dx[r * 2 * C + C + c] = d * a * s * (1.f - s);
  }
}

// ------------------------------------------------------------------ depthwise conv (groups = channels), zero padding per sequence
// y[t][c] = b[c] + sum_k w[c][k'] x[t + k - pad][c], k' = k (forward) or K-1-k (flip: the data gradient)
__global__ __launch_bounds__(256) void dwconv_kernel(jatts_ragged rg, const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ b, float* __restrict__ y, int C, int K, int pad, int flip) {
  const int s = blockIdx.y;
  const int row0 = rg.cu_rows[s], L = rg.cu_rows[s + 1] - row0;
  const int64_t n = (int64_t)L * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int t = (int)(i / C), c = (int)(i - (int64_t)t * C);
    float acc = b ? b[c] : 0.f;
    for (int k = 0; k < K; ++k) {
      const int p = t + k - pad;
      if (p >= 0 && p < L) acc += w[c * K + (flip ? K - 1 - k : k)] * x[(int64_t)(row0 + p) * C + c];
    }
    y[(int64_t)(row0 + t) * C + c] = acc;
  }
}
// Tiled variants for the conformer kernel sizes (7 encoder, 31 decoder): a workgroup stages a [64 + K - 1 steps][64 channels] window
// of x in LDS once; thread (channel, quarter) slides a register window over its 16 consecutive steps, so each x value is read from
// LDS once per thread instead of K times from L1/L2 (the naive kernel above is 17x off the HBM bound at K = 31).
constexpr int DWT = 64;   // time steps per workgroup
template <int K>
__global__ __launch_bounds__(256) void dwconv_tiled_kernel(jatts_ragged rg, const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ b, float* __restrict__ y, int C, int pad, int flip) {
  __shared__ float xs[DWT + K - 1][64];
  const int s = blockIdx.y, c0 = blockIdx.z * 64, t0 = blockIdx.x * DWT;
  const int row0 = rg.cu_rows[s], L = rg.cu_rows[s + 1] - row0;
  if (t0 >= L) return;
  const int cl = threadIdx.x & 63, tq = threadIdx.x >> 6, c = c0 + cl;
  for (int i = threadIdx.x; i < (DWT + K - 1) * 64; i += 256) {
    const int r = i >> 6, j = i & 63, p = t0 - pad + r;
    xs[r][j] = (p >= 0 && p < L && c0 + j < C) ? x[(int64_t)(row0 + p) * C + c0 + j] : 0.f;
  }
  float wv[K];
#pragma unroll
  for (int k = 0; k < K; ++k) wv[k] = c < C ? w[c * K + (flip ? K - 1 - k : k)] : 0.f;
  const float bias = (b && c < C) ? b[c] : 0.f;
  __syncthreads();
  float win[16 + K - 1];
#pragma unroll
  for (int i = 0; i < 16 + K - 1; ++i) win[i] = xs[tq * 16 + i][cl];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    float acc = bias;
#pragma unroll
    for (int k = 0; k < K; ++k) acc += wv[k] * win[i + k];
    const int t = t0 + tq * 16 + i;
    if (t < L && c < C) y[(int64_t)(row0 + t) * C + c] = acc;
  }
}
template <int K>
__global__ __launch_bounds__(256) void dwconv_wgrad_tiled_kernel(jatts_ragged rg, const float* __restrict__ x, const float* __restrict__ dy,
                                                                 float* __restrict__ dw, int C, int pad, int tiles_per_block,
                                                                 float* __restrict__ slabs, unsigned* __restrict__ tickets) {
  __shared__ float xs[DWT + K - 1][64];
  __shared__ __attribute__((aligned(16))) float red[16][64];
  const int s = blockIdx.y, c0 = blockIdx.z * 64;
  const int row0 = rg.cu_rows[s], L = rg.cu_rows[s + 1] - row0;
  const int cl = threadIdx.x & 63, tq = threadIdx.x >> 6, c = c0 + cl;
  float acc[K];
#pragma unroll
  for (int k = 0; k < K; ++k) acc[k] = 0.f;
  for (int tile = 0; tile < tiles_per_block; ++tile) {
    const int t0 = (blockIdx.x * tiles_per_block + tile) * DWT;
    if (t0 >= L) break;
    __syncthreads();
    for (int i = threadIdx.x; i < (DWT + K - 1) * 64; i += 256) {
      const int r = i >> 6, j = i & 63, p = t0 - pad + r;
      xs[r][j] = (p >= 0 && p < L && c0 + j < C) ? x[(int64_t)(row0 + p) * C + c0 + j] : 0.f;
    }
    __syncthreads();
    float win[16 + K - 1];
#pragma unroll
    for (int i = 0; i < 16 + K - 1; ++i) win[i] = xs[tq * 16 + i][cl];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int t = t0 + tq * 16 + i;
      const float d = (t < L && c < C) ? dy[(int64_t)(row0 + t) * C + c] : 0.f;
#pragma unroll
      for (int k = 0; k < K; ++k) acc[k] += d * win[i + k];
    }
  }
  // group = channel tile z; parts = the (time block, sequence) workgroups of that tile; slab = [k][64 channels]
  const int n_parts = gridDim.x * gridDim.y, part = blockIdx.y * gridDim.x + blockIdx.x;
  float* gslab = slabs + (int64_t)blockIdx.z * n_parts * (64 * K);
#pragma unroll
  for (int k = 0; k < K; ++k) {
    __syncthreads();
    red[tq][cl] = acc[k];
    __syncthreads();
    if (tq == 0) gslab[(int64_t)part * (64 * K) + k * 64 + cl] = red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl];
  }
  if (det_arrive(tickets + blockIdx.z, n_parts, reinterpret_cast<unsigned*>(&red[0][0])))
    det_sum_slabs<4>(gslab, n_parts, 64 * K, &red[0][0], 1024, [&](int i, float t) {
      const int k = i >> 6, cc = c0 + (i & 63);
      if (cc < C) dw[cc * K + k] += t;
    });
}

// dw[c][k] += sum_t dy[t][c] x[t + k - pad][c]; a workgroup = 64 channels x a slice of (sequence, time); wave w takes rows t = w mod 4
constexpr int DW_KMAX = 32;
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(jatts_ragged rg, const float* __restrict__ x, const float* __restrict__ dy,
                                                           float* __restrict__ dw, int C, int K, int pad,
                                                           float* __restrict__ slabs, unsigned* __restrict__ tickets) {
  __shared__ __attribute__((aligned(16))) float red[16][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
  const int s = blockIdx.y;
  const int row0 = rg.cu_rows[s], L = rg.cu_rows[s + 1] - row0;
  float acc[DW_KMAX];
#pragma unroll
  for (int k = 0; k < DW_KMAX; ++k) acc[k] = 0.f;
  if (c < C)
    for (int t = blockIdx.z * 4 + part; t < L; t += gridDim.z * 4) {
      const float d = dy[(int64_t)(row0 + t) * C + c];
#pragma unroll
      for (int k = 0; k < DW_KMAX; ++k) {
        const int p = t + k - pad;
        if (k < K && p >= 0 && p < L) acc[k] += d * x[(int64_t)(row0 + p) * C + c];
      }
    }
  // group = channel tile x; parts = its (sequence, time split) workgroups; slab = [k][64 channels]
  const int n_parts = gridDim.y * gridDim.z, wg = blockIdx.z * gridDim.y + blockIdx.y;
  float* gslab = slabs + (int64_t)blockIdx.x * n_parts * (64 * K);
  for (int k = 0; k < K; ++k) {
    float v = 0.f;
#pragma unroll
    for (int q = 0; q < DW_KMAX; ++q) v = q == k ? acc[q] : v;
    red[part][threadIdx.x & 63] = v;
    __syncthreads();
    if (part == 0) gslab[(int64_t)wg * (64 * K) + k * 64 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    __syncthreads();
  }
  if (det_arrive(tickets + blockIdx.x, n_parts, reinterpret_cast<unsigned*>(&red[0][0])))
    det_sum_slabs<4>(gslab, n_parts, 64 * K, &red[0][0], 1024, [&](int i, float t) {
      const int k = i >> 6, cc = blockIdx.x * 64 + (i & 63);
      if (cc < C) dw[cc * K + k] += t;
    });
}

// ------------------------------------------------------------------ column statistics (BatchNorm with batch statistics)
// out0[c] += sum_r (x - shift), out1[c] += sum_r (x - shift)^2          (mode 0; shift may be null)
// out0[c] += sum_r dy,          out1[c] += sum_r dy * (x - mean) * rstd  (mode 1: x = x, y2 = dy, shift = mean, mul = rstd)
__global__ __launch_bounds__(256) void col_stats_kernel(const float* __restrict__ x, const float* __restrict__ y2, int ld, int64_t rows, int dim,
                                                        const float* __restrict__ shift, const float* __restrict__ mul, int mode,
                                                        float* __restrict__ out0, float* __restrict__ out1,
                                                        float* __restrict__ slabs, unsigned* __restrict__ tickets) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int part = threadIdx.x >> 6;
  __shared__ __attribute__((aligned(16))) float red[4][4][64];
  float s0 = 0.f, s1 = 0.f;
  if (c < dim) {
    const float sh = shift ? shift[c] : 0.f, m = mul ? mul[c] : 1.f;
    for (int64_t r = (int64_t)blockIdx.y * 4 + part; r < rows; r += (int64_t)gridDim.y * 4) {
      const float v = x[r * ld + c] - sh;
      if (mode == 0) { s0 += v; s1 += v * v; }
      else { const float d = y2[r * ld + c]; s0 += d; s1 += d * v * m; }
    }
  }
  red[0][part][threadIdx.x & 63] = s0;
  red[1][part][threadIdx.x & 63] = s1;
  __syncthreads();
  float* gslab = slabs + (int64_t)blockIdx.x * gridDim.y * 128;      // group = channel tile; slab = [out0 x 64 | out1 x 64]
  if (part == 0) {
    gslab[blockIdx.y * 128 + threadIdx.x] = red[0][0][threadIdx.x] + red[0][1][threadIdx.x] + red[0][2][threadIdx.x] + red[0][3][threadIdx.x];
    gslab[blockIdx.y * 128 + 64 + threadIdx.x] = red[1][0][threadIdx.x] + red[1][1][threadIdx.x] + red[1][2][threadIdx.x] + red[1][3][threadIdx.x];
  }
  if (det_arrive(tickets + blockIdx.x, gridDim.y, reinterpret_cast<unsigned*>(&red[0][0][0])))
    det_sum_slabs<4>(gslab, (int)gridDim.y, 128, &red[0][0][0], 1024, [&](int i, float t) {
      const int cc = blockIdx.x * 64 + (i & 63);
      if (cc < dim) (i < 64 ? out0 : out1)[cc] += t;
    });
}
// BatchNorm backward apply: dx = g * rstd * (dy - s_dy / N - xhat * s_dyxhat / N)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy, int64_t rows, int C,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ g, const float* __restrict__ s_dy,
                                                           const float* __restrict__ s_dyx, float inv_n, float* __restrict__ dx) {
  const int64_t n = rows * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const float xh = (x[i] - mean[c]) * rstd[c];
    dx[i] = g[c] * rstd[c] * (dy[i] - s_dy[c] * inv_n - xh * s_dyx[c] * inv_n);
  }
}

// ------------------------------------------------------------------ per-sequence column sums (backward of "add a vector per sequence")
// out[s][c] += sum over the rows of sequence s of x[row][c]; grid (channel tiles, sequences, time splits): the per-sequence statistics
// kernel of the speaker-embedding path (one workgroup per sequence) left 7/8 of the chip idle on a 32-utterance batch.
__global__ __launch_bounds__(256) void seq_sum_kernel(jatts_ragged rg, const float* __restrict__ x, int C, float* __restrict__ out,
                                                      float* __restrict__ slabs, unsigned* __restrict__ tickets) {
  __shared__ __attribute__((aligned(16))) float red[16][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6, s = blockIdx.y;
  const int row0 = rg.cu_rows[s], L = rg.cu_rows[s + 1] - row0;
  float a = 0.f;
  if (c < C)
    for (int t = blockIdx.z * 4 + part; t < L; t += gridDim.z * 4) a += x[(int64_t)(row0 + t) * C + c];
  red[part][threadIdx.x & 63] = a;
  __syncthreads();
  const int grp = blockIdx.y * gridDim.x + blockIdx.x;                // group = (sequence, channel tile); parts = the time splits
  float* gslab = slabs + (int64_t)grp * gridDim.z * 64;
  if (part == 0) gslab[blockIdx.z * 64 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
  if (det_arrive(tickets + grp, gridDim.z, reinterpret_cast<unsigned*>(&red[0][0])))
    det_sum_slabs<4>(gslab, (int)gridDim.z, 64, &red[0][0], 1024, [&](int i, float t) {
      const int cc = blockIdx.x * 64 + i;
      if (cc < C) out[(int64_t)s * C + cc] += t;
    });
}

// ------------------------------------------------------------------ row-indexed accumulation (embedding backward)
// dst[idx[r]][c] += scale * src[r][c], rows with idx == skip (padding_idx) or outside [0, n_dst) are dropped.
// Deterministic: one workgroup per (destination row j, 256-channel tile) walks the source rows IN ORDER -- the index list goes through
// LDS 1 024 entries at a time, a thread owns one channel and adds the matching rows as they come (an atomic scatter added them in
// arrival order).  n_dst x rows index reads in total: meant for embedding tables of a TTS vocabulary (tens of symbols).
__global__ __launch_bounds__(256) void index_add_rows_kernel(const float* __restrict__ src, int ld, const int64_t* __restrict__ idx, int64_t rows,
                                                             int C, float scale, int64_t skip, int64_t n_dst, float* __restrict__ dst) {
  __shared__ int hit[256];
  __shared__ int cnt[4];
  const int64_t j = blockIdx.x;
  const int c = blockIdx.y * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (j == skip) return;                     // (uniform per workgroup)
  float a = 0.f;
  for (int64_t r0 = 0; r0 < rows; r0 += 256) {
    const int64_t r = r0 + threadIdx.x;
    const bool m = r < rows && idx[r] == j;
    const unsigned long long bal = __ballot(m);
    __syncthreads();                         // the previous chunk's hit list is consumed
    if (lane == 0) cnt[wave] = __popcll(bal);
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += cnt[w];
    if (m) hit[base + __popcll(bal & ((1ull << lane) - 1ull))] = threadIdx.x;      // ordered compaction: row order is kept
    const int nh = cnt[0] + cnt[1] + cnt[2] + cnt[3];
    __syncthreads();
    if (c < C)
      for (int h = 0; h < nh; ++h) a += scale * src[(r0 + hit[h]) * ld + c];
  }
  if (c < C) dst[j * C + c] += a;
}

// ------------------------------------------------------------------ length-regulator backward (segment sums, deterministic)
// d_hs[token i of sequence b][c] = sum over frames t in [cum[i-1], cum[i]) of dy[cu_out[b] + t][c]   (t < output length of b)
__global__ __launch_bounds__(128) void lr_segment_sum_kernel(jatts_ragged rg, const int64_t* __restrict__ cum, const int32_t* __restrict__ cu_out,
                                                             const float* __restrict__ dy, int C, float* __restrict__ dhs) {
  const int s = blockIdx.y, i = blockIdx.x;
  const int row0 = rg.cu_rows[s], L = rg.cu_rows[s + 1] - row0;
  if (i >= L) return;
  const int64_t lo = i ? cum[row0 + i - 1] : 0, hi = cum[row0 + i];
  const int64_t o0 = cu_out[s], Lo = cu_out[s + 1] - cu_out[s];
  for (int c = threadIdx.x; c < C; c += 128) {
    float acc = 0.f;
    for (int64_t t = lo; t < hi && t < Lo; ++t) acc += dy[(o0 + t) * C + c];
    dhs[(int64_t)(row0 + i) * C + c] = acc;
  }
}

// ------------------------------------------------------------------ attention: legacy rel_shift + key mask + softmax
// Per (b, h, query i): s[j] = (ac[i][j] + shift(bd)[i][j]) * scale for j < len[b], softmax over those keys, 0 elsewhere
// (attention.py:63-93 masked_fill(min) -> softmax -> masked_fill(0)).  shift = LegacyRelPositionMultiHeadedAttention.rel_shift
// (attention.py:142-162) on a T x T matrix: out[i][j] = flat[(i + 1) T + j] of the zero-left-padded T x (T+1) matrix.
// mode 2 = RelPositionMultiHeadedAttention.rel_shift (attention.py:236-258) on a T x (2T-1) matrix: out[i][j] = bd[i][j - i + T - 1].
__device__ __forceinline__ float shifted_bd(const float* __restrict__ bd, int T, int i, int j, int mode) {
  if (mode == 2) return bd[(int64_t)i * (2 * T - 1) + (j - i + T - 1)];
  const int f = (i + 1) * T + j;
  const int r = f / (T + 1), c = f - r * (T + 1);
  return c == 0 ? 0.f : bd[(int64_t)r * T + c - 1];
}
__global__ __launch_bounds__(256) void shift_softmax_fwd_kernel(const float* __restrict__ ac, const float* __restrict__ bd, int H, int T,
                                                                const int32_t* __restrict__ lens, float scale, float* __restrict__ p,
                                                                int64_t n_rows, int mode) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 4 + wave;   // (b * H + h) * T + i
  if (row >= n_rows) return;
  const int i = (int)(row % T);
  const int64_t bh = row / T;
  const int b = (int)(bh / H);
  const int len = lens ? min(max(lens[b], 0), T) : T;
  const float* acr = ac + row * T;
  const float* bdm = bd ? bd + bh * (int64_t)T * (mode == 2 ? 2 * T - 1 : T) : nullptr;
  float m = -INFINITY;
  for (int j = lane; j < len; j += 64) {
    const float s = (acr[j] + (bdm ? shifted_bd(bdm, T, i, j, mode) : 0.f)) * scale;
    m = fmaxf(m, s);
  }
  m = wave_max(m);
  float z = 0.f;
  for (int j = lane; j < len; j += 64) {
    const float s = (acr[j] + (bdm ? shifted_bd(bdm, T, i, j, mode) : 0.f)) * scale;
    z += __expf(s - m);
  }
  z = wave_sum(z);
  const float inv = z > 0.f ? 1.f / z : 0.f;
  float* pr = p + row * T;
  for (int j = lane; j < T; j += 64) {
    float v = 0.f;
    if (j < len) v = __expf((acr[j] + (bdm ? shifted_bd(bdm, T, i, j, mode) : 0.f)) * scale - m) * inv;
    pr[j] = v;
  }
}
// ds = p * (dp - sum_j dp p) * scale  (written over dp's row into ds)
// dbd != nullptr: the same value also goes to its place in the UN-shifted gradient of the position term (the inverse of shifted_bd), so
// that no second pass over ds is needed: every dbd entry is written exactly once -- row i of ds owns, in the legacy view trick
// ((i + 1) T + j = r (T + 1) + c + 1), a run of row i and the head of row i + 1 of dbd (row 0 also zero-fills the T - 1 entries of dbd
// row 0 that no output reads); in the new style (mode 2: bd [T][2T-1], shifted[i][j] = bd[i][j - i + T - 1]) its own row, zeros outside.
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ p, const float* __restrict__ dp, int T, float scale,
                                                          float* __restrict__ ds, int64_t n_rows, float* __restrict__ dbd, int mode) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 4 + wave;
  if (row >= n_rows) return;
  const float* pr = p + row * T;
  const float* dr = dp + row * T;
  float s = 0.f;
  for (int j = lane; j < T; j += 64) s += pr[j] * dr[j];
  s = wave_sum(s);
  float* o = ds + row * T;
  if (!dbd) {
    for (int j = lane; j < T; j += 64) o[j] = pr[j] * (dr[j] - s) * scale;
    return;
  }
  const int64_t mtx = row / T;
  const int i = (int)(row - mtx * T);
  if (mode == 2) {
    const int W = 2 * T - 1, c0 = T - 1 - i;                 // ds[i][j] -> dbd[i][c0 + j]
    float* q = dbd + (mtx * T + i) * (int64_t)W;
    for (int c = lane; c < W; c += 64) {
      const int j = c - c0;
      float v = 0.f;
      if (j >= 0 && j < T) { v = pr[j] * (dr[j] - s) * scale; o[j] = v; }
      q[c] = v;
    }
    return;
  }
  float* q = dbd + mtx * T * (int64_t)T;
  if (i == 0)
    for (int c = lane; c < T - 1; c += 64) q[c] = 0.f;      // f = c + 1 < T: read by no output
  for (int j = lane; j < T; j += 64) {
    const float v = pr[j] * (dr[j] - s) * scale;
    o[j] = v;
    const int f = (i + 1) * T + j, r = f / (T + 1), cc = f - r * (T + 1);
    if (cc != 0 && r < T) q[(int64_t)r * T + cc - 1] = v;    // (cc == 0: the zero column of the padded view)
  }
}
// WaveNet gate (residual_block.py:150-156): y = tanh(a) sigmoid(b) on x = [a | b]; dx = [dy sigmoid(b) (1 - tanh^2 a) | dy tanh(a) s (1 - s)]
__global__ __launch_bounds__(256) void gate_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx,
                                                       int64_t rows, int C) {
  const int64_t n = rows * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / C;
    const int c = (int)(i - r * C);
    const float t = tanhf(x[r * 2 * C + c]), sg = sigmoidf_(x[r * 2 * C + C + c]), d = dy[i];
    dx[r * 2 * C + c] = d * sg * (1.f - t * t);
    dx[r * 2 * C + C + c] = d * t * sg * (1.f - sg);
  }
}

// torch.nn.utils.weight_norm (dim 0) as one launch each way: w[o][:] = g[o] v[o][:] / ||v[o][:]||.  One workgroup per output channel
// (row_len = c_in * k <= a few thousand); the norm is accumulated in double like the rest of the step's reductions.
__global__ __launch_bounds__(256) void weight_norm_fwd_kernel(const float* __restrict__ v, const float* __restrict__ g, int row_len,
                                                              float* __restrict__ w, float* __restrict__ inv_norm) {
  __shared__ double part[4];
  const int o = blockIdx.x;
  const float* vr = v + (int64_t)o * row_len;
  double s = 0.0;
  for (int i = threadIdx.x; i < row_len; i += 256) s += (double)vr[i] * vr[i];
  for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  const float inv = (float)(1.0 / sqrt(part[0] + part[1] + part[2] + part[3]));
  const float sc = g[o] * inv;
  for (int i = threadIdx.x; i < row_len; i += 256) w[(int64_t)o * row_len + i] = vr[i] * sc;
  if (threadIdx.x == 0) inv_norm[o] = inv;
}
// dg[o] = <dw, v> / ||v||;  dv = g / ||v|| (dw - v <dw, v> / ||v||^2)
__global__ __launch_bounds__(256) void weight_norm_bwd_kernel(const float* __restrict__ v, const float* __restrict__ g,
                                                              const float* __restrict__ inv_norm, const float* __restrict__ dw, int row_len,
                                                              float* __restrict__ dv, float* __restrict__ dg) {
  __shared__ double part[4];
  const int o = blockIdx.x;
  const float* vr = v + (int64_t)o * row_len;
  const float* dr = dw + (int64_t)o * row_len;
  double s = 0.0;
  for (int i = threadIdx.x; i < row_len; i += 256) s += (double)vr[i] * dr[i];
  for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  const float dot = (float)(part[0] + part[1] + part[2] + part[3]);
  const float inv = inv_norm[o], sc = g[o] * inv, k = dot * inv * inv;
  for (int i = threadIdx.x; i < row_len; i += 256) dv[(int64_t)o * row_len + i] = sc * (dr[i] - vr[i] * k);
  if (threadIdx.x == 0) dg[o] = dot * inv;
}

// WaveNet residual / skip bookkeeping (residual_block.py:158-167) in one pass: o = [res | skip] [rows][2 dim];
// h_out = h + res, skip_out = skip + o's second half (skip nullable: first layer).  Backward is a plain concat: do = [dh | dskip].
__global__ __launch_bounds__(256) void split_add_kernel(const float* __restrict__ o, const float* __restrict__ h, const float* __restrict__ skip,
                                                        int64_t rows, int C, float* __restrict__ h_out, float* __restrict__ skip_out) {
  const int64_t n = rows * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / C;
    const int c = (int)(i - r * C);
    h_out[i] = h[i] + o[r * 2 * C + c];
    skip_out[i] = (skip ? skip[i] : 0.f) + o[r * 2 * C + C + c];
  }
}
__global__ __launch_bounds__(256) void concat2_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t rows, int C,
                                                      float* __restrict__ out) {
  const int64_t n = rows * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / C;
    const int c = (int)(i - r * C);
    out[r * 2 * C + c] = a ? a[i] : 0.f;
    out[r * 2 * C + C + c] = b ? b[i] : 0.f;
  }
}

// ------------------------------------------------------------------ rank-1 helpers (Linear(C -> 1) heads, Conv1d(1 -> C, k=1) embeddings)
// out[r][c] (+)= v[r] * w[c] + bias[c]
__global__ __launch_bounds__(256) void outer_rows_kernel(const float* __restrict__ v, const float* __restrict__ w, const float* __restrict__ bias,
                                                         int64_t rows, int C, int accumulate, float* __restrict__ out) {
  const int64_t n = rows * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / C;
    const int c = (int)(i - r * C);
    const float val = v[r] * w[c] + (bias ? bias[c] : 0.f);
    out[i] = accumulate ? out[i] + val : val;
  }
}
// out[c] += sum_r v[r] * x[r][c]
__global__ __launch_bounds__(256) void col_wsum_kernel(const float* __restrict__ x, int ld, const float* __restrict__ v, int64_t rows, int dim,
                                                       float* __restrict__ out, float* __restrict__ slabs, unsigned* __restrict__ tickets) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int part = threadIdx.x >> 6;
  __shared__ __attribute__((aligned(16))) float red[16][64];
  float s = 0.f;
  if (c < dim)
    for (int64_t r = (int64_t)blockIdx.y * 4 + part; r < rows; r += (int64_t)gridDim.y * 4) s += v[r] * x[r * ld + c];
  red[part][threadIdx.x & 63] = s;
  __syncthreads();
  float* gslab = slabs + (int64_t)blockIdx.x * gridDim.y * 64;
  if (part == 0) gslab[blockIdx.y * 64 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
  if (det_arrive(tickets + blockIdx.x, gridDim.y, reinterpret_cast<unsigned*>(&red[0][0])))
    det_sum_slabs<4>(gslab, (int)gridDim.y, 64, &red[0][0], 1024, [&](int i, float t) {
      const int cc = blockIdx.x * 64 + i;
      if (cc < dim) out[cc] += t;
    });
}
// y[r] = bias + sum_c x[r][c] w[c]   (one wave per row)
__global__ __launch_bounds__(256) void row_dot_kernel(const float* __restrict__ x, int ld, const float* __restrict__ w, const float* __restrict__ bias,
                                                      int64_t rows, int C, float* __restrict__ y) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t r = (int64_t)blockIdx.x * 4 + wave;
  if (r >= rows) return;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += x[r * ld + c] * w[c];
  s = wave_sum(s);
  if (lane == 0) y[r] = s + (bias ? bias[0] : 0.f);
}

// ------------------------------------------------------------------ masked loss gradient
// da[t][c] = up * scale * sign(a - b') (kind 0) or up * 2 scale (a - b') (kind 1) for t < valid_len, 0 elsewhere
__global__ __launch_bounds__(256) void masked_loss_bwd_kernel(jatts_ragged rg, const float* __restrict__ a, int lda, const float* __restrict__ b,
                                                              int ldb, int dim, const int32_t* __restrict__ valid_len, int kind, float log_offset,
                                                              float scale, const float* __restrict__ upstream, float* __restrict__ da, int ldda) {
  const int s = blockIdx.y;
  const int row0 = rg.cu_rows[s], L = rg.cu_rows[s + 1] - row0;
  const int v = valid_len ? min(max(valid_len[s], 0), L) : L;
  const float up = (upstream ? *upstream : 1.f) * scale;
  const int64_t n = (int64_t)L * dim;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t t = i / dim;
    const int c = (int)(i - t * dim);
    float o = 0.f;
    if (t < v) {
      float y = b[(row0 + t) * ldb + c];
      if (log_offset >= 0.f) y = logf(y + log_offset);
      const float dlt = a[(row0 + t) * lda + c] - y;
      o = kind == 0 ? (dlt > 0.f ? up : dlt < 0.f ? -up : 0.f) : 2.f * up * dlt;
    }
    da[(row0 + t) * ldda + c] = o;
  }
}

// ------------------------------------------------------------------ dropout (counter-based mask, recomputed in the backward)
__device__ __forceinline__ uint32_t mix32(uint64_t k) {   // splitmix64 finaliser
  k += 0x9E3779B97F4A7C15ull;
  k = (k ^ (k >> 30)) * 0xBF58476D1CE4E5B9ull;
  k = (k ^ (k >> 27)) * 0x94D049BB133111EBull;
  return (uint32_t)((k ^ (k >> 31)) >> 32);
}
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, float p, uint64_t seed,
                                                      const uint64_t* __restrict__ seed_dev) {
  if (seed_dev) seed += *seed_dev;   // captured graphs: the per-step base seed lives in device memory, `seed` is the site offset
  const float keep_scale = 1.f / (1.f - p);
  const uint32_t thr = (uint32_t)(p * 4294967296.0);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    y[i] = mix32(seed * 0x100000001B3ull + (uint64_t)i) >= thr ? x[i] * keep_scale : 0.f;
}

// dropout(act(x)) in one pass each way (the FFN's ReLU -> dropout, multi_layer_conv.py:52-63): same counter-based mask as dropout_kernel
// (element index i), so it is bit-identical to act_fwd + dropout; dy == nullptr: forward (y = keep ? act(x) / (1 - p) : 0), else
// backward (dx = keep ? dy act'(x) / (1 - p) : 0).
__global__ __launch_bounds__(256) void act_dropout_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ out,
                                                          int64_t n, int mode, float p, uint64_t seed, const uint64_t* __restrict__ seed_dev) {
  if (seed_dev) seed += *seed_dev;
  const float keep_scale = 1.f / (1.f - p);
  const uint32_t thr = (uint32_t)(p * 4294967296.0);
  const uint64_t base = seed * 0x100000001B3ull;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * 1024) {
    if (i + 3 < n) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(x + i);
      f32x4 d = {1.f, 1.f, 1.f, 1.f}, o;
      if (dy) d = *reinterpret_cast<const f32x4*>(dy + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool keep = mix32(base + (uint64_t)(i + e)) >= thr;
        o[e] = keep ? (dy ? d[e] * act_grad(v[e], mode) : act_val(v[e], mode)) * keep_scale : 0.f;
      }
      *reinterpret_cast<f32x4*>(out + i) = o;
    } else {
      for (int64_t k = i; k < n; ++k) {
        const bool keep = mix32(base + (uint64_t)k) >= thr;
        out[k] = keep ? (dy ? dy[k] * act_grad(x[k], mode) : act_val(x[k], mode)) * keep_scale : 0.f;
      }
    }
  }
}

// Head split of the fused Q|K|V projection (attention.py:81-88,190-195): qkv [rows = B T][3 A] -> q + pos_bias_u, q + pos_bias_v, k, v,
// each [B][H][T][d_k] contiguous (what the batched GEMMs want), one pass.  The backward gathers the four gradients back into d qkv
// [rows][3 A] and adds the column sums of d(q + u), d(q + v) -- the bias gradients -- to du / dv (per-workgroup slabs summed in a fixed
// order by the last arriver, det_reduce.h).  As torch ops this was 2 broadcast adds + 3 permute copies forward and 4 zero-filled slice gradients + 3 adds back.
__global__ __launch_bounds__(256) void qkv_split_kernel(const float* __restrict__ qkv, const float* __restrict__ u, const float* __restrict__ v,
                                                        int B, int T, int H, int dk, float* __restrict__ qu, float* __restrict__ qv,
                                                        float* __restrict__ k, float* __restrict__ vv) {
  const int A = H * dk;
  const int64_t n = (int64_t)B * T * A;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / A;
    const int c = (int)(i - row * A), h = c / dk, d = c - h * dk;
    const int b = (int)(row / T), t = (int)(row - (int64_t)b * T);
    const int64_t o = (((int64_t)b * H + h) * T + t) * dk + d;
    const float* src = qkv + row * 3 * A + c;
    const float q = src[0];
    qu[o] = u ? q + u[c] : q;          // (u == v == nullptr: a plain head split, qv unused)
    if (v) qv[o] = q + v[c];
    k[o] = src[A];
    vv[o] = src[2 * A];
  }
}
__global__ __launch_bounds__(256) void qkv_split_bwd_kernel(const float* __restrict__ dqu, const float* __restrict__ dqv, const float* __restrict__ dk_,
                                                            const float* __restrict__ dvv, int B, int T, int H, int dk, float* __restrict__ dqkv,
                                                            float* __restrict__ du, float* __restrict__ dv, int rows_per_block,
                                                            float* __restrict__ slabs, unsigned* __restrict__ tickets) {
  __shared__ __attribute__((aligned(16))) float red[1024];
  const int A = H * dk;
  const int64_t rows = (int64_t)B * T, r0 = (int64_t)blockIdx.x * rows_per_block;
  float* slab = slabs + (int64_t)blockIdx.x * 2 * A;        // this workgroup's partial [du | dv]
  for (int c = threadIdx.x; c < A; c += 256) {
    const int h = c / dk, d = c - h * dk;
    float su = 0.f, sv = 0.f;
    for (int64_t row = r0; row < r0 + rows_per_block && row < rows; ++row) {
      const int b = (int)(row / T), t = (int)(row - (int64_t)b * T);
      const int64_t o = (((int64_t)b * H + h) * T + t) * dk + d;
      const float a = dqu[o], e = dqv ? dqv[o] : 0.f;
      float* dst = dqkv + row * 3 * A + c;
      dst[0] = a + e;
      dst[A] = dk_[o];
      dst[2 * A] = dvv[o];
      su += a;
      sv += e;
    }
    if (du || dv) {
      slab[c] = su;
      slab[A + c] = sv;
    }
  }
  if (!du && !dv) return;
  det_reduce_tree<4>(slabs, tickets, (int)blockIdx.x, (int)gridDim.x, LN_GROUP, 2 * A, red, 1024, reinterpret_cast<unsigned*>(red), [&](int i, float t) {
    if (i < A) { if (du) du[i] += t; }
    else if (dv) dv[i - A] += t;
  });
}

// y = resid + alpha * dropout(x): the residual connections of the conformer layers (encoder_layer.py:100-170: x + ff_scale * dropout(ffn),
// x + dropout(attn), x + dropout(conv)) in ONE launch instead of dropout + scale + add; resid may be NULL (the backward of the
// dropped branch: alpha * mask(dy) / (1 - p)), p may be 0 (a plain scaled add).
__global__ __launch_bounds__(256) void dropout_add_kernel(const float* __restrict__ x, const float* __restrict__ resid, float* __restrict__ y,
                                                          int64_t n, float p, float alpha, uint64_t seed, const uint64_t* __restrict__ seed_dev) {
  if (seed_dev) seed += *seed_dev;
  const float keep_scale = alpha / (1.f - p);
  const uint32_t thr = (uint32_t)(p * 4294967296.0);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const bool keep = p <= 0.f || mix32(seed * 0x100000001B3ull + (uint64_t)i) >= thr;
    const float v = keep ? x[i] * keep_scale : 0.f;
    y[i] = resid ? resid[i] + v : v;
  }
}

// ------------------------------------------------------------------ optimiser
// sum of squares into a double (gradient-norm clipping: torch.nn.utils.clip_grad_norm_, trainers/fastspeech2.py:90-94)
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, int64_t n, double* __restrict__ out,
                                                    double* __restrict__ slabs, unsigned* __restrict__ tickets) {
  __shared__ double red[256];
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) acc += (double)x[i] * (double)x[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) slabs[blockIdx.x] = red[0];
  if (det_arrive(tickets, gridDim.x, reinterpret_cast<unsigned*>(&red[1])))
    det_sum_slabs<4>(slabs, (int)gridDim.x, 1, red, 256, [&](int, double t) { *out += t; });
}
// torch.optim.Adam (no amsgrad): g' = g * gscale (+ wd * p); m = b1 m + (1-b1) g'; v = b2 v + (1-b2) g'^2;
// p -= lr / bc1 * m / (sqrt(v) / sqrt(bc2) + eps); the scalars (bias corrections, step size, 1 - beta) are computed in double on the
// host like torch's Python floats and rounded to f32 once.  gscale = min(1, max_norm / (sqrt(*sumsq) + 1e-6)) when sumsq is given.
// Gradient gather: up to 64 separately allocated gradient tensors -> their slots in the flat gradient buffer, one launch.  The tensor list
// travels BY VALUE in the kernel arguments (no pointer-table upload, nothing to keep alive; a captured graph bakes it in).  One workgroup =
// one 4 096-element chunk of one tensor; chunk0[] = prefix sum of the tensors' chunk counts.
constexpr int GATHER_MAX = 64, GATHER_CHUNK = 4096;
struct GatherBatch {
  const float* src[GATHER_MAX];
  int64_t dst[GATHER_MAX];
  int64_t numel[GATHER_MAX];
  int32_t chunk0[GATHER_MAX + 1];
  int32_t n;
};
__global__ __launch_bounds__(256) void gather_grads_kernel(GatherBatch b, float* __restrict__ flat, int accumulate) {
  const int blk = blockIdx.x;
  int lo = 0, hi = b.n;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (b.chunk0[mid] <= blk) lo = mid; else hi = mid;
  }
  const int64_t s = (int64_t)(blk - b.chunk0[lo]) * GATHER_CHUNK;
  const int64_t n = b.numel[lo] - s < GATHER_CHUNK ? b.numel[lo] - s : GATHER_CHUNK;
  const float* __restrict__ src = b.src[lo] + s;
  float* __restrict__ dst = flat + b.dst[lo] + s;
  if (accumulate) for (int i = threadIdx.x; i < n; i += 256) dst[i] += src[i];
  else for (int i = threadIdx.x; i < n; i += 256) dst[i] = src[i];
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                   int64_t n, float step_size, float w1, float b2, float w2, float eps, float wd, float bc2_sqrt,
                                                   const double* __restrict__ sumsq, float max_norm, const float* __restrict__ hyper) {
  if (hyper) {   // captured graphs: the per-step scalars [step_size, 1 - beta1, beta2, 1 - beta2, eps, weight_decay, sqrt(bc2)] live in device memory
    step_size = hyper[0]; w1 = hyper[1]; b2 = hyper[2]; w2 = hyper[3]; eps = hyper[4]; wd = hyper[5]; bc2_sqrt = hyper[6];
  }
  float gs = 1.f;
  if (sumsq && max_norm > 0.f) {
    const float c = max_norm / ((float)sqrt(*sumsq) + 1e-6f);
    gs = c < 1.f ? c : 1.f;
  }
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float gi = g[i] * gs;
    if (wd != 0.f) gi += wd * p[i];
    const float m0 = m[i];
    const float mi = m0 + w1 * (gi - m0);                  // exp_avg.lerp_(grad, 1 - beta1)
    const float vi = b2 * v[i] + w2 * gi * gi;             // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    m[i] = mi;
    v[i] = vi;
    p[i] -= step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));   // param.addcdiv_(exp_avg, denom, value=-step_size)
  }
}

// ------------------------------------------------------------------ GroupNorm over (channels of a group) x (rows of a sequence)
// torch.nn.GroupNorm(G, C) on (B, C, T) (matchatts/decoder.py:66-78 Block1D): one workgroup per (sequence, group); thread t owns
// channel t % cg of the group and the rows t / cg, t / cg + 256 / cg, ...  Two passes for the statistics (mean, then centred
// squares), a third applies; the group's [L][cg] slab is a few hundred KB and stays in L2 between them.
__device__ __forceinline__ float block_sum256(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(256) void groupnorm_fwd_kernel(jatts_ragged rg, const float* __restrict__ x, int C, int G, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps, float* __restrict__ y,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out) {
  __shared__ float red[4];
  const int s = blockIdx.y, g = blockIdx.x, cg = C / G;
  const int row0 = rg.cu_rows[s], L = rg.cu_rows[s + 1] - row0;
  const int c = g * cg + threadIdx.x % cg, r0 = threadIdx.x / cg, rs = 256 / cg;
  const float inv_n = 1.f / ((float)L * (float)cg);
  float a = 0.f;
  for (int t = r0; t < L; t += rs) a += x[(int64_t)(row0 + t) * C + c];
  const float mean = block_sum256(a, red) * inv_n;
  a = 0.f;
  for (int t = r0; t < L; t += rs) { const float d = x[(int64_t)(row0 + t) * C + c] - mean; a += d * d; }
  const float rstd = rsqrtf(block_sum256(a, red) * inv_n + eps);
  const float gm = gamma[c] * rstd, bt = beta[c] - mean * gamma[c] * rstd;
  for (int t = r0; t < L; t += rs) y[(int64_t)(row0 + t) * C + c] = x[(int64_t)(row0 + t) * C + c] * gm + bt;
  if (threadIdx.x == 0) { mean_out[s * G + g] = mean; rstd_out[s * G + g] = rstd; }
}
// dx = rstd (dy g - mean_grp(dy g) - xhat mean_grp(dy g xhat)); dgamma[c] += sum_t dy xhat; dbeta[c] += sum_t dy
__global__ __launch_bounds__(256) void groupnorm_bwd_kernel(jatts_ragged rg, const float* __restrict__ x, const float* __restrict__ dy, int C, int G,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean_in,
                                                            const float* __restrict__ rstd_in, float* __restrict__ dx, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, float* __restrict__ slabs, unsigned* __restrict__ tickets) {
  __shared__ float red[4];
  __shared__ __attribute__((aligned(16))) float cred[4][256];
  const int s = blockIdx.y, g = blockIdx.x, cg = C / G;
  const int row0 = rg.cu_rows[s], L = rg.cu_rows[s + 1] - row0;
  const int c = g * cg + threadIdx.x % cg, r0 = threadIdx.x / cg, rs = 256 / cg;
  const float mean = mean_in[s * G + g], rstd = rstd_in[s * G + g], gm = gamma[c];
  const float inv_n = 1.f / ((float)L * (float)cg);
  float s1 = 0.f, s2 = 0.f, pg = 0.f, pb = 0.f;
  for (int t = r0; t < L; t += rs) {
    const int64_t i = (int64_t)(row0 + t) * C + c;
    const float xh = (x[i] - mean) * rstd, d = dy[i];
    s1 += d * gm;
    s2 += d * gm * xh;
    pg += d * xh;
    pb += d;
  }
  s1 = block_sum256(s1, red) * inv_n;
  s2 = block_sum256(s2, red) * inv_n;
  if (dx)
    for (int t = r0; t < L; t += rs) {
      const int64_t i = (int64_t)(row0 + t) * C + c;
      const float xh = (x[i] - mean) * rstd;
      dx[i] = rstd * (dy[i] * gm - s1 - xh * s2);
    }
  if (!dgamma) return;
  cred[0][threadIdx.x] = pg;
  cred[1][threadIdx.x] = pb;
  __syncthreads();
  float* gslab = slabs + (int64_t)g * gridDim.y * 2 * cg;        // group = GroupNorm group; parts = the sequences; slab = [dgamma x cg | dbeta x cg]
  if (threadIdx.x < cg) {
    float a = 0.f, b = 0.f;
    for (int j = threadIdx.x; j < 256; j += cg) { a += cred[0][j]; b += cred[1][j]; }
    gslab[(int64_t)s * 2 * cg + threadIdx.x] = a;
    gslab[(int64_t)s * 2 * cg + cg + threadIdx.x] = b;
  }
  if (det_arrive(tickets + g, gridDim.y, reinterpret_cast<unsigned*>(&red[0])))
    det_sum_slabs<4>(gslab, (int)gridDim.y, 2 * cg, &cred[0][0], 1024, [&](int i, float t) {
      if (i < cg) dgamma[g * cg + i] += t; else dbeta[g * cg + i - cg] += t;
    });
}

// ------------------------------------------------------------------ SnakeBeta (matchatts/transformer.py:84-102, log-scale alpha / beta)
// y = x + sin^2(a x) / (b + 1e-9), a = exp(alpha[c]), b = exp(beta[c]).
__global__ __launch_bounds__(256) void snakebeta_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t rows, int C,
                                                            const float* __restrict__ alpha, const float* __restrict__ beta) {
  const int64_t n = rows * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const float sn = sinf(x[i] * expf(alpha[c]));
    y[i] = x[i] + sn * sn / (expf(beta[c]) + 1e-9f);
  }
}
// dx = dy (1 + a sin(2 a x) / (b + e)); dalpha[c] += sum dy x a sin(2 a x) / (b + e); dbeta[c] += sum dy (-b sin^2(a x) / (b + e)^2)
__global__ __launch_bounds__(256) void snakebeta_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, int64_t rows, int C,
                                                            const float* __restrict__ alpha, const float* __restrict__ beta,
                                                            float* __restrict__ dx, float* __restrict__ dalpha, float* __restrict__ dbeta,
                                                            float* __restrict__ slabs, unsigned* __restrict__ tickets) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int part = threadIdx.x >> 6;
  __shared__ __attribute__((aligned(16))) float red[4][4][64];
  float sa = 0.f, sb = 0.f;
  if (c < C) {
    const float a = expf(alpha[c]), b = expf(beta[c]), ib = 1.f / (b + 1e-9f);
    for (int64_t r = (int64_t)blockIdx.y * 4 + part; r < rows; r += (int64_t)gridDim.y * 4) {
      const float xv = x[r * C + c], d = dy[r * C + c];
      const float sn = sinf(xv * a), s2 = sinf(2.f * xv * a);
      dx[r * C + c] = d * (1.f + a * s2 * ib);
      sa += d * xv * a * s2 * ib;
      sb -= d * b * sn * sn * ib * ib;
    }
  }
  red[0][part][threadIdx.x & 63] = sa;
  red[1][part][threadIdx.x & 63] = sb;
  __syncthreads();
  float* gslab = slabs + (int64_t)blockIdx.x * gridDim.y * 128;
  if (part == 0) {
    gslab[blockIdx.y * 128 + threadIdx.x] = red[0][0][threadIdx.x] + red[0][1][threadIdx.x] + red[0][2][threadIdx.x] + red[0][3][threadIdx.x];
    gslab[blockIdx.y * 128 + 64 + threadIdx.x] = red[1][0][threadIdx.x] + red[1][1][threadIdx.x] + red[1][2][threadIdx.x] + red[1][3][threadIdx.x];
  }
  if (det_arrive(tickets + blockIdx.x, gridDim.y, reinterpret_cast<unsigned*>(&red[0][0][0])))
    det_sum_slabs<4>(gslab, (int)gridDim.y, 128, &red[0][0][0], 1024, [&](int i, float t) {
      const int cc = blockIdx.x * 64 + (i & 63);
      if (cc < C) (i < 64 ? dalpha : dbeta)[cc] += t;
    });
}

// ------------------------------------------------------------------ forward-sum (CTC) loss of the alignment framework
// ForwardSumLoss (losses/forward_sum_loss.py:41-78): per utterance F.ctc_loss on lp[t][1 + j] = log_p_attn[t][j] + prior[t][j],
// lp[t][0] = log(blank_prob), targets 1..N (every token once, in order), reduction "mean" (nll / N), zero_infinity.
// Lattice states s = 0..2N: blank for even s, token (s-1)/2 for odd s; all tokens differ, so the skip s-2 -> s is always allowed
// for odd s.  One workgroup per utterance: threads own states, time is sequential (a barrier per step); alpha / beta live in a
// global workspace [T][2N+1] each.  The gradient is torch's (LossCTC.cpp): exp(lp) - exp(logsumexp_s(alpha + beta) + nll - lp),
// which presumes log-softmax inputs -- reproduced as is, because that is what the reference back-propagates.
__device__ __forceinline__ float lse2(float a, float b) {
  if (a == -INFINITY) return b;
  if (b == -INFINITY) return a;
  const float m = fmaxf(a, b);
  return m + logf(expf(a - m) + expf(b - m));
}
__global__ __launch_bounds__(256) void ctc_forward_sum_kernel(const float* __restrict__ lp, int Tmax, int ld, const int32_t* __restrict__ ilens,
                                                              const int32_t* __restrict__ olens, float log_blank, float* __restrict__ alpha,
                                                              float* __restrict__ beta, int smax, float* __restrict__ nll_out,
                                                              float* __restrict__ grad, float grad_scale) {
  const int b = blockIdx.x;
  const int N = ilens[b], T = olens[b], S = 2 * N + 1;
  const float* l = lp + (int64_t)b * Tmax * ld;
  float* al = alpha + (int64_t)b * Tmax * smax;
  float* be = beta + (int64_t)b * Tmax * smax;
  float* g = grad ? grad + (int64_t)b * Tmax * ld : nullptr;
  auto emit = [&](int t, int s) { return (s & 1) ? l[(int64_t)t * ld + (s >> 1)] : log_blank; };
  for (int s = threadIdx.x; s < S; s += 256) al[s] = s == 0 ? log_blank : s == 1 ? l[0] : -INFINITY;
  __syncthreads();
  for (int t = 1; t < T; ++t) {
    for (int s = threadIdx.x; s < S; s += 256) {
      const float* p = al + (int64_t)(t - 1) * smax;
      float a = p[s];
      if (s >= 1) a = lse2(a, p[s - 1]);
      if ((s & 1) && s >= 3) a = lse2(a, p[s - 2]);
      al[(int64_t)t * smax + s] = a == -INFINITY ? -INFINITY : a + emit(t, s);
    }
    __syncthreads();
  }
  const float* last = al + (int64_t)(T - 1) * smax;
  const float ll = lse2(last[S - 1], S >= 2 ? last[S - 2] : -INFINITY);
  const float nll = -ll;
  const bool inf = !(nll < INFINITY);                     // zero_infinity: an impossible alignment contributes 0 loss, 0 gradient
  if (threadIdx.x == 0) nll_out[b] = inf ? 0.f : nll / (float)N;
  if (!g) return;
  for (int s = threadIdx.x; s < S; s += 256) be[(int64_t)(T - 1) * smax + s] = s == S - 1 ? log_blank : s == S - 2 ? l[(int64_t)(T - 1) * ld + (s >> 1)] : -INFINITY;
  __syncthreads();
  for (int t = T - 2; t >= 0; --t) {
    for (int s = threadIdx.x; s < S; s += 256) {
      const float* p = be + (int64_t)(t + 1) * smax;
      float a = p[s];
      if (s + 1 < S) a = lse2(a, p[s + 1]);
      if ((s & 1) && s + 2 < S) a = lse2(a, p[s + 2]);
      be[(int64_t)t * smax + s] = a == -INFINITY ? -INFINITY : a + emit(t, s);
    }
    __syncthreads();
  }
  // gradient w.r.t. the token columns (the blank column is a constant): one odd state per token
  const float sc = grad_scale / (float)N;
  for (int i = threadIdx.x; i < Tmax * ld; i += 256) {
    const int t = i / ld, j = i - t * ld;
    float v = 0.f;
    if (!inf && t < T && j < N) {
      const float lpv = l[i];
      const float ab = al[(int64_t)t * smax + 2 * j + 1] + be[(int64_t)t * smax + 2 * j + 1];
      v = (expf(lpv) - expf(ab + nll - lpv)) * sc;
    }
    g[i] = v;
  }
}

inline unsigned blocks_for(int64_t n, int per_block, unsigned cap = 8192) {
  const int64_t b = (n + per_block - 1) / per_block;
  return (unsigned)(b < 1 ? 1 : b > cap ? cap : b);
}

}  // namespace

#define S_ ((hipStream_t)stream)
#define NULLCHK(cond, msg) \
  if (cond) return jatts_set_error_msg(JATTS_ERR_ARG, msg)
// scratch of the deterministic reductions (det_reduce.h): `groups` tickets, `floats` f32 of slabs
#define WS_NEED(groups, floats)                                        \
  do {                                                                 \
    const int rc_ = jatts_ws_need((int64_t)(groups), (int64_t)(floats)); \
    if (rc_ != JATTS_OK) return rc_;                                   \
  } while (0)
#define WS_ jatts_g_ws.slabs, jatts_g_ws.tickets

extern "C" int jatts_layernorm_bwd(const float* x, int32_t ldx, const float* dy, int32_t lddy, const float* gamma, int64_t rows, int32_t dim,
                                   float eps, float* dx, int32_t lddx, float* dgamma, float* dbeta, void* stream) {
  NULLCHK(!x || !dy || !gamma, "layernorm_bwd: null pointer");
  NULLCHK((dgamma == nullptr) != (dbeta == nullptr), "layernorm_bwd: dgamma and dbeta go together");
  if (dim < 1 || dim > 64 * LN_MAXPL) return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "layernorm_bwd: 1 <= dim <= 1536");
  if (rows <= 0) return JATTS_OK;
  // <= 512 workgroups (8 waves per CU over the chip): the last arriver adds up one [dgamma | dbeta] slab per workgroup
  const unsigned nb = blocks_for(rows, 16, 512);
  if (dgamma) WS_NEED((nb + LN_GROUP - 1) / LN_GROUP + 1, (int64_t)(nb + (nb + LN_GROUP - 1) / LN_GROUP) * 2 * dim);
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(nb), dim3(256), 0, S_, x, ldx, dy, lddy, gamma, rows, dim, eps, dx, lddx, dgamma, dbeta, WS_);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

extern "C" int jatts_act_fwd(int32_t mode, const float* x, float* y, int64_t n, void* stream) {
  NULLCHK(!x || !y, "act_fwd: null pointer");
  NULLCHK(mode < 1 || mode > 4, "act_fwd: mode 1 relu, 2 tanh, 3 swish, 4 mish");
  if (n <= 0) return JATTS_OK;
  hipLaunchKernelGGL(act_fwd_kernel, dim3(blocks_for(n, 1024)), dim3(256), 0, S_, x, y, n, mode);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}
extern "C" int jatts_act_bwd(int32_t mode, const float* x, const float* dy, float* dx, int64_t n, void* stream) {
  NULLCHK(!x || !dy || !dx, "act_bwd: null pointer");
  NULLCHK(mode < 1 || mode > 4, "act_bwd: mode 1 relu, 2 tanh, 3 swish, 4 mish");
  if (n <= 0) return JATTS_OK;
  hipLaunchKernelGGL(act_bwd_kernel, dim3(blocks_for(n, 1024)), dim3(256), 0, S_, x, dy, dx, n, mode);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}
extern "C" int jatts_glu_fwd(const float* x, float* y, int64_t rows, int32_t dim, void* stream) {
  NULLCHK(!x || !y, "glu_fwd: null pointer");
  if (rows <= 0 || dim <= 0) return JATTS_OK;
  hipLaunchKernelGGL(glu_fwd_kernel, dim3(blocks_for(rows * dim, 256)), dim3(256), 0, S_, x, y, rows, dim);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}
extern "C" int jatts_glu_bwd(const float* x, const float* dy, float* dx, int64_t rows, int32_t dim, void* stream) {
  NULLCHK(!x || !dy || !dx, "glu_bwd: null pointer");
  if (rows <= 0 || dim <= 0) return JATTS_OK;
  hipLaunchKernelGGL(glu_bwd_kernel, dim3(blocks_for(rows * dim, 256)), dim3(256), 0, S_, x, dy, dx, rows, dim);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

extern "C" int jatts_dwconv(const jatts_ragged* rg, const float* x, const float* w, const float* bias, float* y, int32_t dim, int32_t k_w,
                            int32_t pad, int32_t flip, void* stream) {
  NULLCHK(!rg || !x || !w || !y, "dwconv: null pointer");
  NULLCHK(dim < 1 || k_w < 1 || k_w > DW_KMAX, "dwconv: 1 <= k_w <= 32");
  if (rg->n_seq <= 0 || rg->max_len <= 0) return JATTS_OK;
  const dim3 tgrid((unsigned)((rg->max_len + DWT - 1) / DWT), (unsigned)rg->n_seq, (unsigned)((dim + 63) / 64));
  if (k_w == 7) hipLaunchKernelGGL(dwconv_tiled_kernel<7>, tgrid, dim3(256), 0, S_, *rg, x, w, bias, y, dim, pad, flip);
  else if (k_w == 31) hipLaunchKernelGGL(dwconv_tiled_kernel<31>, tgrid, dim3(256), 0, S_, *rg, x, w, bias, y, dim, pad, flip);
  else
    hipLaunchKernelGGL(dwconv_kernel, dim3(blocks_for((int64_t)rg->max_len * dim, 256, 1024), (unsigned)rg->n_seq), dim3(256), 0, S_, *rg, x, w, bias,
                       y, dim, k_w, pad, flip);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}
extern "C" int jatts_dwconv_wgrad(const jatts_ragged* rg, const float* x, const float* dy, float* dw, int32_t dim, int32_t k_w, int32_t pad,
                                  void* stream) {
  NULLCHK(!rg || !x || !dy || !dw, "dwconv_wgrad: null pointer");
  NULLCHK(dim < 1 || k_w < 1 || k_w > DW_KMAX, "dwconv_wgrad: 1 <= k_w <= 32");
  if (rg->n_seq <= 0 || rg->max_len <= 0) return JATTS_OK;
  if (k_w == 7 || k_w == 31) {
    const int tiles = (rg->max_len + DWT - 1) / DWT;
    const int tpb = tiles > 8 ? 4 : 1;     // a few tiles per workgroup: 4x fewer slabs to sum on long sequences
    const dim3 tgrid((unsigned)((tiles + tpb - 1) / tpb), (unsigned)rg->n_seq, (unsigned)((dim + 63) / 64));
    WS_NEED(tgrid.z, (int64_t)tgrid.x * tgrid.y * tgrid.z * 64 * k_w);
    if (k_w == 7) hipLaunchKernelGGL(dwconv_wgrad_tiled_kernel<7>, tgrid, dim3(256), 0, S_, *rg, x, dy, dw, dim, pad, tpb, WS_);
    else hipLaunchKernelGGL(dwconv_wgrad_tiled_kernel<31>, tgrid, dim3(256), 0, S_, *rg, x, dy, dw, dim, pad, tpb, WS_);
    JATTS_CHECK_LAUNCH();
    return JATTS_OK;
  }
  unsigned gz = (unsigned)((rg->max_len + 63) / 64);
  if (gz > 64) gz = 64;
  WS_NEED((dim + 63) / 64, (int64_t)((dim + 63) / 64) * rg->n_seq * gz * 64 * k_w);
  hipLaunchKernelGGL(dwconv_wgrad_kernel, dim3((unsigned)((dim + 63) / 64), (unsigned)rg->n_seq, gz), dim3(256), 0, S_, *rg, x, dy, dw, dim, k_w, pad, WS_);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

extern "C" int jatts_col_stats(const float* x, const float* y2, int32_t ld, int64_t rows, int32_t dim, const float* shift, const float* mul,
                               int32_t mode, float* out0, float* out1, void* stream) {
  NULLCHK(!x || !out0 || !out1 || (mode == 1 && !y2), "col_stats: null pointer");
  NULLCHK(mode != 0 && mode != 1, "col_stats: mode 0 (moments) or 1 (dy sums)");
  if (rows <= 0 || dim <= 0) return JATTS_OK;
  const int64_t gy0 = (rows + 255) / 256;
  const unsigned gx = (unsigned)((dim + 63) / 64), gy = (unsigned)(gy0 < 256 ? gy0 : 256);
  WS_NEED(gx, (int64_t)gx * gy * 128);
  hipLaunchKernelGGL(col_stats_kernel, dim3(gx, gy), dim3(256), 0, S_, x, y2, ld, rows, dim, shift, mul, mode, out0, out1, WS_);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}
extern "C" int jatts_bn_bwd_apply(const float* x, const float* dy, int64_t rows, int32_t dim, const float* mean, const float* rstd,
                                  const float* gamma, const float* s_dy, const float* s_dyx, float* dx, void* stream) {
  NULLCHK(!x || !dy || !mean || !rstd || !gamma || !s_dy || !s_dyx || !dx, "bn_bwd_apply: null pointer");
  if (rows <= 0 || dim <= 0) return JATTS_OK;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(blocks_for(rows * dim, 256)), dim3(256), 0, S_, x, dy, rows, dim, mean, rstd, gamma, s_dy, s_dyx,
                     1.f / (float)rows, dx);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

extern "C" int jatts_index_add_rows(const float* src, int32_t ld, const int64_t* idx, int64_t rows, int32_t dim, float scale, int64_t skip,
                                    int64_t n_dst, float* dst, void* stream) {
  NULLCHK(!src || !idx || !dst, "index_add_rows: null pointer");
  if (rows <= 0 || dim <= 0) return JATTS_OK;
  NULLCHK(n_dst < 1 || n_dst > 0x7fffffff, "index_add_rows: 1 <= n_dst < 2^31");
  hipLaunchKernelGGL(index_add_rows_kernel, dim3((unsigned)n_dst, (unsigned)((dim + 255) / 256)), dim3(256), 0, S_, src, ld, idx, rows, dim, scale, skip,
                     n_dst, dst);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

extern "C" int jatts_lr_segment_sum(const jatts_ragged* rg, const int64_t* cum, const int32_t* cu_out, const float* dy, int32_t dim, float* dhs,
                                    void* stream) {
  NULLCHK(!rg || !cum || !cu_out || !dy || !dhs, "lr_segment_sum: null pointer");
  if (rg->n_seq <= 0 || rg->max_len <= 0 || dim <= 0) return JATTS_OK;
  hipLaunchKernelGGL(lr_segment_sum_kernel, dim3((unsigned)rg->max_len, (unsigned)rg->n_seq), dim3(128), 0, S_, *rg, cum, cu_out, dy, dim, dhs);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

extern "C" int jatts_gate_bwd(const float* x, const float* dy, float* dx, int64_t rows, int32_t dim, void* stream) {
  NULLCHK(!x || !dy || !dx, "gate_bwd: null pointer");
  if (rows <= 0 || dim <= 0) return JATTS_OK;
  hipLaunchKernelGGL(gate_bwd_kernel, dim3(blocks_for(rows * dim, 256)), dim3(256), 0, S_, x, dy, dx, rows, dim);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

extern "C" int jatts_weight_norm_fwd(const float* v, const float* g, int32_t n_out, int32_t row_len, float* w, float* inv_norm, void* stream) {
  NULLCHK(!v || !g || !w || !inv_norm, "weight_norm_fwd: null pointer");
  NULLCHK(n_out < 1 || row_len < 1, "weight_norm_fwd: bad geometry");
  hipLaunchKernelGGL(weight_norm_fwd_kernel, dim3((unsigned)n_out), dim3(256), 0, S_, v, g, row_len, w, inv_norm);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}
extern "C" int jatts_weight_norm_bwd(const float* v, const float* g, const float* inv_norm, const float* dw, int32_t n_out, int32_t row_len,
                                     float* dv, float* dg, void* stream) {
  NULLCHK(!v || !g || !inv_norm || !dw || !dv || !dg, "weight_norm_bwd: null pointer");
  NULLCHK(n_out < 1 || row_len < 1, "weight_norm_bwd: bad geometry");
  hipLaunchKernelGGL(weight_norm_bwd_kernel, dim3((unsigned)n_out), dim3(256), 0, S_, v, g, inv_norm, dw, row_len, dv, dg);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}
extern "C" int jatts_split_add(const float* o, const float* h, const float* skip, int64_t rows, int32_t dim, float* h_out, float* skip_out,
                               void* stream) {
  NULLCHK(!o || !h || !h_out || !skip_out, "split_add: null pointer");
  if (rows <= 0 || dim <= 0) return JATTS_OK;
  hipLaunchKernelGGL(split_add_kernel, dim3(blocks_for(rows * dim, 256)), dim3(256), 0, S_, o, h, skip, rows, dim, h_out, skip_out);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}
extern "C" int jatts_concat2(const float* a, const float* b, int64_t rows, int32_t dim, float* out, void* stream) {
  NULLCHK(!out, "concat2: null pointer");
  if (rows <= 0 || dim <= 0) return JATTS_OK;
  hipLaunchKernelGGL(concat2_kernel, dim3(blocks_for(rows * dim, 256)), dim3(256), 0, S_, a, b, rows, dim, out);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

extern "C" int jatts_shift_softmax_fwd(const float* ac, const float* bd, int32_t n_batch, int32_t n_heads, int32_t t_len, const int32_t* lens,
                                       float scale, int32_t shift_mode, float* p, void* stream) {
  NULLCHK(!ac || !p, "shift_softmax_fwd: null pointer");
  NULLCHK(n_batch < 1 || n_heads < 1 || t_len < 1, "shift_softmax_fwd: bad geometry");
  const int64_t n_rows = (int64_t)n_batch * n_heads * t_len;
  NULLCHK(shift_mode != 1 && shift_mode != 2, "shift_softmax_fwd: shift_mode 1 (legacy) or 2 (new)");
  hipLaunchKernelGGL(shift_softmax_fwd_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, S_, ac, bd, n_heads, t_len, lens, scale, p, n_rows,
                     shift_mode);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}
extern "C" int jatts_shift_softmax_bwd(const float* p, const float* dp, int32_t n_batch, int32_t n_heads, int32_t t_len, float scale,
                                       int32_t shift_mode, float* ds, float* dbd, void* stream) {
  NULLCHK(!p || !dp || !ds, "shift_softmax_bwd: null pointer");
  NULLCHK(n_batch < 1 || n_heads < 1 || t_len < 1, "shift_softmax_bwd: bad geometry");
  const int64_t n_rows = (int64_t)n_batch * n_heads * t_len;
  NULLCHK(dbd && shift_mode != 1 && shift_mode != 2, "shift_softmax_bwd: shift_mode 1 (legacy) or 2 (new)");
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, S_, p, dp, t_len, scale, ds, n_rows, dbd, shift_mode);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

extern "C" int jatts_outer_rows(const float* v, const float* w, const float* bias, int64_t rows, int32_t dim, int32_t accumulate, float* out,
                                void* stream) {
  NULLCHK(!v || !w || !out, "outer_rows: null pointer");
  if (rows <= 0 || dim <= 0) return JATTS_OK;
  hipLaunchKernelGGL(outer_rows_kernel, dim3(blocks_for(rows * dim, 256)), dim3(256), 0, S_, v, w, bias, rows, dim, accumulate, out);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}
extern "C" int jatts_col_wsum(const float* x, int32_t ld, const float* v, int64_t rows, int32_t dim, float* out, void* stream) {
  NULLCHK(!x || !v || !out, "col_wsum: null pointer");
  if (rows <= 0 || dim <= 0) return JATTS_OK;
  const int64_t gy0 = (rows + 255) / 256;
  const unsigned gx = (unsigned)((dim + 63) / 64), gy = (unsigned)(gy0 < 256 ? gy0 : 256);
  WS_NEED(gx, (int64_t)gx * gy * 64);
  hipLaunchKernelGGL(col_wsum_kernel, dim3(gx, gy), dim3(256), 0, S_, x, ld, v, rows, dim, out, WS_);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}
extern "C" int jatts_row_dot(const float* x, int32_t ld, const float* w, const float* bias, int64_t rows, int32_t dim, float* y, void* stream) {
  NULLCHK(!x || !w || !y, "row_dot: null pointer");
  if (rows <= 0 || dim <= 0) return JATTS_OK;
  hipLaunchKernelGGL(row_dot_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, S_, x, ld, w, bias, rows, dim, y);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

extern "C" int jatts_masked_loss_bwd(const jatts_ragged* rg, const float* a, int32_t lda, const float* b, int32_t ldb, int32_t dim,
                                     const int32_t* valid_len, int32_t kind, float log_offset, float scale, const float* upstream, float* da,
                                     int32_t ldda, void* stream) {
  NULLCHK(!rg || !a || !b || !da, "masked_loss_bwd: null pointer");
  NULLCHK(kind != 0 && kind != 1, "masked_loss_bwd: kind must be 0 (L1) or 1 (L2)");
  if (rg->n_seq <= 0 || rg->max_len <= 0) return JATTS_OK;
  hipLaunchKernelGGL(masked_loss_bwd_kernel, dim3(blocks_for((int64_t)rg->max_len * dim, 256, 256), (unsigned)rg->n_seq), dim3(256), 0, S_, *rg, a, lda,
                     b, ldb, dim, valid_len, kind, log_offset, scale, upstream, da, ldda);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

extern "C" int jatts_dropout(const float* x, float* y, int64_t n, float p, uint64_t seed, const uint64_t* seed_dev, void* stream) {
  NULLCHK(!x || !y, "dropout: null pointer");
  NULLCHK(!(p >= 0.f && p < 1.f), "dropout: 0 <= p < 1");
  if (n <= 0) return JATTS_OK;
  hipLaunchKernelGGL(dropout_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, S_, x, y, n, p, seed, seed_dev);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

extern "C" int jatts_qkv_split(const float* qkv, const float* u, const float* v, int32_t n_batch, int32_t t_len, int32_t n_heads, int32_t d_k,
                               float* qu, float* qv, float* k, float* vv, void* stream) {
  NULLCHK(!qkv || !qu || !k || !vv || (v && !qv) || (!u != !v), "qkv_split: null pointer");
  NULLCHK(n_batch < 1 || t_len < 1 || n_heads < 1 || d_k < 1, "qkv_split: bad geometry");
  hipLaunchKernelGGL(qkv_split_kernel, dim3(blocks_for((int64_t)n_batch * t_len * n_heads * d_k, 1024)), dim3(256), 0, S_, qkv, u, v, n_batch, t_len,
                     n_heads, d_k, qu, qv, k, vv);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}
extern "C" int jatts_qkv_split_bwd(const float* dqu, const float* dqv, const float* dk, const float* dvv, int32_t n_batch, int32_t t_len,
                                   int32_t n_heads, int32_t d_k, float* dqkv, float* du, float* dv, void* stream) {
  NULLCHK(!dqu || !dk || !dvv || !dqkv || (!dqv != !dv) || (!du != !dv), "qkv_split_bwd: null pointer");
  NULLCHK(n_batch < 1 || t_len < 1 || n_heads < 1 || d_k < 1, "qkv_split_bwd: bad geometry");
  const int64_t rows = (int64_t)n_batch * t_len;
  // 32 rows per workgroup, more when that would exceed 512 workgroups (= slabs the last arriver adds up for du / dv)
  int rpb = 32;
  while ((rows + rpb - 1) / rpb > 512) rpb += 32;
  const unsigned nb = (unsigned)((rows + rpb - 1) / rpb);
  if (du || dv) WS_NEED((nb + LN_GROUP - 1) / LN_GROUP + 1, (int64_t)(nb + (nb + LN_GROUP - 1) / LN_GROUP) * 2 * n_heads * d_k);
  hipLaunchKernelGGL(qkv_split_bwd_kernel, dim3(nb), dim3(256), 0, S_, dqu, dqv, dk, dvv, n_batch, t_len, n_heads, d_k, dqkv, du, dv, rpb, WS_);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}
extern "C" int jatts_act_dropout(int32_t mode, const float* x, const float* dy, float* out, int64_t n, float p, uint64_t seed,
                                 const uint64_t* seed_dev, void* stream) {
  NULLCHK(!x || !out, "act_dropout: null pointer");
  NULLCHK(mode < 1 || mode > 4, "act_dropout: mode 1 ReLU, 2 tanh, 3 Swish, 4 Mish");
  NULLCHK(!(p >= 0.f && p < 1.f), "act_dropout: 0 <= p < 1");
  if (n <= 0) return JATTS_OK;
  hipLaunchKernelGGL(act_dropout_kernel, dim3(blocks_for(n, 1024)), dim3(256), 0, S_, x, dy, out, n, mode, p, seed, seed_dev);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}
extern "C" int jatts_dropout_add(const float* x, const float* resid, float* y, int64_t n, float p, float alpha, uint64_t seed,
                                 const uint64_t* seed_dev, void* stream) {
  NULLCHK(!x || !y, "dropout_add: null pointer");
  NULLCHK(!(p >= 0.f && p < 1.f), "dropout_add: 0 <= p < 1");
  if (n <= 0) return JATTS_OK;
  hipLaunchKernelGGL(dropout_add_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, S_, x, resid, y, n, p, alpha, seed, seed_dev);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

extern "C" int jatts_sumsq(const float* x, int64_t n, double* out, void* stream) {
  NULLCHK(!x || !out, "sumsq: null pointer");
  if (n <= 0) return JATTS_OK;
  const unsigned nb = blocks_for(n, 4096, 1024);
  WS_NEED(1, (int64_t)nb * 2);
  hipLaunchKernelGGL(sumsq_kernel, dim3(nb), dim3(256), 0, S_, x, n, out, (double*)jatts_g_ws.slabs, jatts_g_ws.tickets);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}
extern "C" int jatts_gather_grads(const float* const* src, const int64_t* numel, const int64_t* dst_off, int32_t n, float* flat,
                                  int32_t accumulate, void* stream) {
  NULLCHK(n < 0 || (n > 0 && (!src || !numel || !dst_off || !flat)), "gather_grads: null pointer");
  for (int i0 = 0; i0 < n; i0 += GATHER_MAX) {
    GatherBatch b;
    b.n = n - i0 < GATHER_MAX ? n - i0 : GATHER_MAX;
    int32_t chunks = 0;
    for (int i = 0; i < b.n; ++i) {
      NULLCHK(!src[i0 + i] || numel[i0 + i] < 0 || dst_off[i0 + i] < 0, "gather_grads: bad entry");
      b.src[i] = src[i0 + i];
      b.dst[i] = dst_off[i0 + i];
      b.numel[i] = numel[i0 + i];
      b.chunk0[i] = chunks;
      chunks += (int32_t)((numel[i0 + i] + GATHER_CHUNK - 1) / GATHER_CHUNK);
    }
    for (int i = b.n; i < GATHER_MAX; ++i) { b.src[i] = nullptr; b.dst[i] = 0; b.numel[i] = 0; b.chunk0[i] = chunks; }
    b.chunk0[GATHER_MAX] = chunks;
    if (chunks == 0) continue;
    hipLaunchKernelGGL(gather_grads_kernel, dim3((unsigned)chunks), dim3(256), 0, S_, b, flat, accumulate);
  }
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}
extern "C" int jatts_adam_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2, double eps,
                               double weight_decay, int64_t step, const double* grad_sumsq, float max_norm, const float* hyper_dev, void* stream) {
  NULLCHK(!p || !g || !m || !v, "adam_step: null pointer");
  NULLCHK(step < 1, "adam_step: step counts from 1");
  if (n <= 0) return JATTS_OK;
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  hipLaunchKernelGGL(adam_kernel, dim3(blocks_for(n, 1024, 4096)), dim3(256), 0, S_, p, g, m, v, n, (float)(lr / bc1), (float)(1.0 - beta1),
                     (float)beta2, (float)(1.0 - beta2), (float)eps, (float)weight_decay, (float)sqrt(bc2), grad_sumsq, max_norm, hyper_dev);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

extern "C" int jatts_groupnorm_fwd(const jatts_ragged* rg, const float* x, int32_t dim, int32_t groups, const float* gamma, const float* beta,
                                   float eps, float* y, float* mean, float* rstd, void* stream) {
  NULLCHK(!rg || !x || !gamma || !beta || !y || !mean || !rstd, "groupnorm_fwd: null pointer");
  NULLCHK(groups < 1 || dim % groups != 0 || dim / groups > 256 || 256 % (dim / groups) != 0,
          "groupnorm_fwd: channels per group must divide 256");
  if (rg->n_seq <= 0 || rg->max_len <= 0) return JATTS_OK;
  hipLaunchKernelGGL(groupnorm_fwd_kernel, dim3((unsigned)groups, (unsigned)rg->n_seq), dim3(256), 0, S_, *rg, x, dim, groups, gamma, beta, eps, y, mean,
                     rstd);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}
extern "C" int jatts_groupnorm_bwd(const jatts_ragged* rg, const float* x, const float* dy, int32_t dim, int32_t groups, const float* gamma,
                                   const float* mean, const float* rstd, float* dx, float* dgamma, float* dbeta, void* stream) {
  NULLCHK(!rg || !x || !dy || !gamma || !mean || !rstd, "groupnorm_bwd: null pointer");
  NULLCHK((dgamma == nullptr) != (dbeta == nullptr), "groupnorm_bwd: dgamma and dbeta go together");
  NULLCHK(groups < 1 || dim % groups != 0 || dim / groups > 256 || 256 % (dim / groups) != 0,
          "groupnorm_bwd: channels per group must divide 256");
  if (rg->n_seq <= 0 || rg->max_len <= 0) return JATTS_OK;
  if (dgamma) WS_NEED(groups, (int64_t)rg->n_seq * 2 * dim);
  hipLaunchKernelGGL(groupnorm_bwd_kernel, dim3((unsigned)groups, (unsigned)rg->n_seq), dim3(256), 0, S_, *rg, x, dy, dim, groups, gamma, mean, rstd, dx,
                     dgamma, dbeta, WS_);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}
extern "C" int jatts_snakebeta_fwd(const float* x, float* y, int64_t rows, int32_t dim, const float* alpha, const float* beta, void* stream) {
  NULLCHK(!x || !y || !alpha || !beta, "snakebeta_fwd: null pointer");
  if (rows <= 0 || dim <= 0) return JATTS_OK;
  hipLaunchKernelGGL(snakebeta_fwd_kernel, dim3(blocks_for(rows * dim, 256)), dim3(256), 0, S_, x, y, rows, dim, alpha, beta);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}
extern "C" int jatts_snakebeta_bwd(const float* x, const float* dy, int64_t rows, int32_t dim, const float* alpha, const float* beta, float* dx,
                                   float* dalpha, float* dbeta, void* stream) {
  NULLCHK(!x || !dy || !alpha || !beta || !dx || !dalpha || !dbeta, "snakebeta_bwd: null pointer");
  if (rows <= 0 || dim <= 0) return JATTS_OK;
  const int64_t gy0 = (rows + 255) / 256;
  const unsigned gx = (unsigned)((dim + 63) / 64), gy = (unsigned)(gy0 < 256 ? gy0 : 256);
  WS_NEED(gx, (int64_t)gx * gy * 128);
  hipLaunchKernelGGL(snakebeta_bwd_kernel, dim3(gx, gy), dim3(256), 0, S_, x, dy, rows, dim, alpha, beta, dx, dalpha, dbeta, WS_);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

extern "C" int jatts_ctc_forward_sum(const float* log_p, int32_t n_batch, int32_t t_max, int32_t ld, const int32_t* ilens, const int32_t* olens,
                                     int32_t max_ilen, float log_blank, float* workspace, float* nll, float* grad, float grad_scale, void* stream) {
  NULLCHK(!log_p || !ilens || !olens || !workspace || !nll, "ctc_forward_sum: null pointer");
  NULLCHK(n_batch < 1 || t_max < 1 || ld < max_ilen || max_ilen < 1, "ctc_forward_sum: bad geometry");
  const int smax = 2 * max_ilen + 1;
  float* alpha = workspace;
  float* beta = workspace + (size_t)n_batch * t_max * smax;
  hipLaunchKernelGGL(ctc_forward_sum_kernel, dim3((unsigned)n_batch), dim3(256), 0, S_, log_p, t_max, ld, ilens, olens, log_blank, alpha, beta, smax, nll,
                     grad, grad_scale);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

extern "C" int jatts_seq_sum(const jatts_ragged* rg, const float* x, int32_t dim, float* out, void* stream) {
  NULLCHK(!rg || !x || !out, "seq_sum: null pointer");
  if (rg->n_seq <= 0 || rg->max_len <= 0 || dim <= 0) return JATTS_OK;
  unsigned gz = (unsigned)((rg->max_len + 63) / 64);
  if (gz > 16) gz = 16;
  WS_NEED((int64_t)((dim + 63) / 64) * rg->n_seq, (int64_t)((dim + 63) / 64) * rg->n_seq * gz * 64);
  hipLaunchKernelGGL(seq_sum_kernel, dim3((unsigned)((dim + 63) / 64), (unsigned)rg->n_seq, gz), dim3(256), 0, S_, *rg, x, dim, out, WS_);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}
