// Conv1d / Linear / ConvTranspose1d (polyphase) as MFMA implicit GEMM, and the fused
// HiFi-GAN ResBlock dilation unit.  gfx950 only.
//
// GEMM orientation: A operand = weights (M = output channel n), B operand =
// activations (N = time).  With 32x32 fragments a lane then owns, for ONE time step,
// groups of 4 consecutive output channels -> packed 8 B (f16) / 16 B (f32) stores into the
// time-major activation layout, and per-channel bias / residual reads vectorise.
//
// Weights arrive pre-packed in fragment order ([tap][c/16][n/32][lane][8], see
// jatts_conv_weight_index) so one A fragment is ONE fully coalesced 1 KiB (f16) wave load
// straight from L2 -- no LDS staging, no bank conflicts.  Activation tiles (+halo) are
// staged once per channel chunk in LDS with a (row bytes + 16) pitch: the 16-lane groups of
// ds_read_b128 then hit 16 distinct 16-B slots (pitch/16 is odd) -> conflict free.
#include <stdlib.h>

#include "common.h"

namespace {

template <typename T> struct Vec8IO;
template <> struct Vec8IO<f16> {
  static __device__ __forceinline__ f16x8 ldg(const f16* p) { return *reinterpret_cast<const f16x8*>(p); }
  static __device__ __forceinline__ f16x8 lds(const char* p) { return *reinterpret_cast<const f16x8*>(p); }
  static __device__ __forceinline__ void sts(char* p, const f16x8& v) { *reinterpret_cast<f16x8*>(p) = v; }
};
template <> struct Vec8IO<float> {
  static __device__ __forceinline__ f32x8 ldg(const float* p) {
    f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    return f32x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  }
  static __device__ __forceinline__ f32x8 lds(const char* p) {
    f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 16);
    return f32x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  }
  static __device__ __forceinline__ void sts(char* p, const f32x8& v) {
    *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p + 16) = f32x4{v[4], v[5], v[6], v[7]};
  }
};

// Stage `rows` x `nch` (multiple of 8) activations into LDS rows of `pitch` bytes.
// Source row for LDS row r is local position pos0 + r of a sequence of length L starting at
// global row seq_row0; positions outside [0, L) give zeros.  Up to 3 inputs are summed,
// scaled and passed through the optional leaky-ReLU before conversion to T.
// Loads are issued in batches of UB per thread BEFORE any of them is consumed: a one-load-
// per-iteration loop serialises a full L2/HBM round trip per 16 bytes (measured: the staging
// phases were as long as the MFMA phases).
template <typename T, int UB = 8>
__device__ __forceinline__ void stage_rows(char* lds, int pitch, int rows, int nch, int pos0, int L,
                                           int64_t seq_row0, const T* const* x, int n_in, int ldx,
                                           int c0, float in_scale, int pre_act, float slope) {
  typedef typename Elem<T>::vec8 V8;
  const int upr = nch >> 3;  // 8-element units per row
  const int total = rows * upr;
  const bool plain = n_in == 1 && in_scale == 1.f && pre_act == JATTS_PRE_NONE;
  for (int base = threadIdx.x; base < total; base += blockDim.x * UB) {
    V8 v[UB];
    int64_t off[UB];
    int dst[UB];
    bool ok[UB];
#pragma unroll
    for (int j = 0; j < UB; ++j) {
      const int u = base + j * blockDim.x;
      const int r = u / upr, cu = u - r * upr;
      const int pos = pos0 + r;
      ok[j] = u < total && pos >= 0 && pos < L;
      dst[j] = u < total ? r * pitch + cu * 8 * (int)sizeof(T) : -1;
      off[j] = (seq_row0 + pos) * (int64_t)ldx + c0 + cu * 8;
      if (ok[j]) v[j] = Vec8IO<T>::ldg(x[0] + off[j]);
      else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[j][e] = from_f32<T>(0.f);
      }
    }
    if (!plain) {
      V8 w1[UB], w2[UB];
      if (n_in > 1) {
#pragma unroll
        for (int j = 0; j < UB; ++j)
          if (ok[j]) w1[j] = Vec8IO<T>::ldg(x[1] + off[j]);
      }
      if (n_in > 2) {
#pragma unroll
        for (int j = 0; j < UB; ++j)
          if (ok[j]) w2[j] = Vec8IO<T>::ldg(x[2] + off[j]);
      }
#pragma unroll
      for (int j = 0; j < UB; ++j) {
        if (!ok[j]) continue;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float t = to_f32(v[j][e]);
          if (n_in > 1) t += to_f32(w1[j][e]);
          if (n_in > 2) t += to_f32(w2[j][e]);
          t *= in_scale;
          if (pre_act == JATTS_PRE_LRELU) t = lrelu(t, slope);
          v[j][e] = from_f32<T>(t);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < UB; ++j)
      if (dst[j] >= 0) Vec8IO<T>::sts(lds + dst[j], v[j]);
  }
}

// acc[f][t] += sum over (tap, 16-channel steps) of W-fragment x activation-fragment.
//  w        : packed weights; fragment (tap, kc, nf) at ((tap*KC16 + kc)*NFR + nf)*512 elements
//  kc_base  : first global 16-channel step covered by the LDS tile, kc_cnt steps staged
//  act      : LDS tile, row `col + tap*dil` holds the sample feeding output column `col`
// Software pipeline (hipcc otherwise waits vmcnt(0)/lgkmcnt right at each MFMA, and with one
// wave per SIMD nothing else hides the L2 / LDS latency): weight fragments travel through a
// register ring D iterations ahead of their use (global -> VGPR, fully coalesced 1 KiB wave
// loads), activation fragments are read from LDS one iteration ahead.  The loop is unrolled
// by D so every ring slot is a compile-time register index, and the body is STRAIGHT-LINE
// (no guards): k_w * kc_cnt must be a multiple of D, and the producers clamp at the last
// fragment instead of branching, so the tail prefetches are harmless in-bounds re-reads.
template <typename T, int NF, int NT, int D>
__device__ __forceinline__ void conv_stage(f32x16 (&acc)[NF][NT], const T* __restrict__ w, int KC16,
                                           int NFR, int nf0, int kc_base, int kc_cnt, int k_w, int dil,
                                           const char* act, int pitch, int col0, int lane) {
  typedef typename Elem<T>::vec8 V8;
  static_assert(D % 2 == 0, "ring depth must be even");
  const int g = lane >> 5;
  const int n_it = k_w * kc_cnt;
  const char* bbase = act + (size_t)(col0 + (lane & 31)) * pitch + (size_t)(8 * g) * sizeof(T);
  const T* wbase = w + (size_t)lane * 8;
  int nfo[NF];  // fragment offsets (clamped: duplicates are never stored)
#pragma unroll
  for (int f = 0; f < NF; ++f) nfo[f] = (nf0 + f < NFR ? nf0 + f : NFR - 1) * 512;
  const int last_tap = k_w - 1;

  int wp_tap = 0, wp_kk = 0;  // weight producer position
  auto fetch_w = [&](V8(&dst)[NF]) {
    const T* p = wbase + ((size_t)(wp_tap * KC16 + kc_base + wp_kk) * NFR) * 512;
#pragma unroll
    for (int f = 0; f < NF; ++f) dst[f] = Vec8IO<T>::ldg(p + nfo[f]);
    const int nk = wp_kk + 1;
    const bool wrap = nk == kc_cnt;
    wp_kk = wrap ? 0 : nk;
    wp_tap = min(wp_tap + (wrap ? 1 : 0), last_tap);
  };
  int bp_tap = 0, bp_kk = 0;  // activation producer position
  auto fetch_b = [&](V8(&dst)[NT]) {
    const char* p = bbase + (size_t)(bp_tap * dil) * pitch + (size_t)(bp_kk * 16) * sizeof(T);
#pragma unroll
    for (int t = 0; t < NT; ++t) dst[t] = Vec8IO<T>::lds(p + (size_t)(t * 32) * pitch);
    const int nk = bp_kk + 1;
    const bool wrap = nk == kc_cnt;
    bp_kk = wrap ? 0 : nk;
    bp_tap = min(bp_tap + (wrap ? 1 : 0), last_tap);
  };

  V8 ring[D][NF];
  V8 bb[2][NT];
#pragma unroll
  for (int j = 0; j < D; ++j) fetch_w(ring[j]);
  fetch_b(bb[0]);
  // sched_barrier(0): LLVM's scheduler otherwise sinks every prefetch back next to its use
  __builtin_amdgcn_sched_barrier(0);
  for (int it0 = 0; it0 < n_it; it0 += D) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      fetch_b(bb[(j + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int t = 0; t < NT; ++t) mma32(ring[j][f], bb[j & 1][t], acc[f][t]);
      fetch_w(ring[j]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

template <int NF, int NT>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[NF][NT]) {
#pragma unroll
  for (int f = 0; f < NF; ++f)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[f][t][r] = 0.f;
}

// ------------------------------------------------------------------ generic conv kernel
constexpr int KCH = 64;  // channels staged per LDS chunk

template <typename T, int NF, int NT, int WN, int WT>
__global__ __launch_bounds__(WN* WT * 64) void conv1d_kernel(jatts_conv_desc d) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BT = WT * NT * 32;
  const int b = blockIdx.y;
  const int row_b = d.rg.cu_rows[b];
  const int L = (d.rg.cu_rows[b + 1] - row_b) * d.rg.len_mul;
  const int t0 = blockIdx.x * BT;
  if (t0 >= L) return;
  const int64_t seq_row0 = (int64_t)row_b * d.rg.len_mul;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wn = wave / WT, wt = wave % WT;
  const int pitch = KCH * (int)sizeof(T) + 16;
  const int rows = BT + (d.k_w - 1) * d.dil;
  const int KC16 = d.c_in >> 4;
  const int n_pad = (d.n_out + 31) & ~31;
  const int NFR = n_pad >> 5;
  const int nf0 = (blockIdx.z * WN + wn) * NF;
  const int col0 = wt * NT * 32;

  const T* xin[3] = {(const T*)d.x[0], (const T*)d.x[1], (const T*)d.x[2]};
  f32x16 acc[NF][NT];
  zero_acc<NF, NT>(acc);

  for (int c0 = 0; c0 < d.c_in; c0 += KCH) {
    const int nch = min(KCH, d.c_in - c0);
    stage_rows<T>(smem, pitch, rows, nch, t0 - d.pad, L, seq_row0, xin, d.n_in, d.ldx, c0, d.in_scale,
                  d.pre_act, d.pre_slope);
    __syncthreads();
    conv_stage<T, NF, NT, 2>(acc, (const T*)d.w, KC16, NFR, nf0, c0 >> 4, nch >> 4, d.k_w, d.dil, smem,
                             pitch, col0, lane);
    __syncthreads();
  }

  // epilogue: lane owns column (lane&31) and channel quads n0 + {0..3}
  const int g = lane >> 5;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int pos = t0 + col0 + t * 32 + (lane & 31);
    if (pos >= L) continue;
    const int64_t row = seq_row0 + pos;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n0 = (nf0 + f) * 32 + 8 * q + 4 * g;
        if (n0 >= d.n_out) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int n = n0 + e;
          float s = acc[f][t][4 * q + e];
          if (n < d.n_out) {
            if (d.bias) s += d.bias[n];
            s = apply_act(s, d.act) * d.alpha;
            if (d.resid) s += d.resid[row * d.ldr + n];
          }
          v[e] = s;
        }
        if (d.y_transposed) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (n0 + e >= d.n_out) break;
            const int64_t o = (int64_t)(n0 + e) * d.ldy + row;
            if (d.y_is_f32) ((float*)d.y)[o] = v[e]; else ((T*)d.y)[o] = from_f32<T>(v[e]);
          }
        } else if (n0 + 3 < d.n_out && (d.ldy & 3) == 0) {
          const int64_t o = row * d.ldy + n0;
          if (d.y_is_f32 || sizeof(T) == 4) {
            *reinterpret_cast<f32x4*>((float*)d.y + o) = f32x4{v[0], v[1], v[2], v[3]};
          } else {
            *reinterpret_cast<f16x4*>((f16*)d.y + o) = f16x4{(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
          }
        } else {
          for (int e = 0; e < 4 && n0 + e < d.n_out; ++e) {
            const int64_t o = row * d.ldy + n0 + e;
            if (d.y_is_f32) ((float*)d.y)[o] = v[e]; else ((T*)d.y)[o] = from_f32<T>(v[e]);
          }
        }
      }
    }
  }
}

template <typename T, int NF, int NT, int WN, int WT>
int launch_conv(const jatts_conv_desc& d, hipStream_t s) {
  constexpr int BT = WT * NT * 32, BN = WN * NF * 32;
  const int64_t maxL = (int64_t)d.rg.max_len * d.rg.len_mul;
  dim3 grid((unsigned)((maxL + BT - 1) / BT), (unsigned)d.rg.n_seq, (unsigned)((d.n_out + BN - 1) / BN));
  const size_t lds = (size_t)(BT + (d.k_w - 1) * d.dil) * (KCH * sizeof(T) + 16);
  if (lds > 160 * 1024) return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "conv1d: halo too large for LDS");
  auto kern = conv1d_kernel<T, NF, NT, WN, WT>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return jatts_set_error(e, __FILE__, __LINE__);
  }
  hipLaunchKernelGGL(kern, grid, dim3(WN * WT * 64), lds, s, d);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

// ------------------------------------------------------------ fused HiFi-GAN dilation unit
template <typename T, int C, int WGCOLS, int WN, int NT>
__global__ __launch_bounds__(WN*(WGCOLS / (NT * 32)) * 64) void resunit_kernel(jatts_resunit_desc d) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int WT = WGCOLS / (NT * 32);
  constexpr int NF = C / (WN * 32);
  constexpr int KC16 = C / 16, NFR = C / 32;
  constexpr int pitch = C * (int)sizeof(T) + 16;
  constexpr int RD = sizeof(T) == 4 ? 2 : (KC16 < 8 ? KC16 : 8);  // weight ring depth; divides K*KC16
  static_assert(WT * NT * 32 == WGCOLS && NF * WN * 32 == C, "tile shape");
  const int K = d.k_w, dil = d.dil;
  const int p2 = (K - 1) / 2, p1 = p2 * dil;
  const int tt_out = WGCOLS - 2 * p2;

  const int b = blockIdx.y;
  const int row_b = d.rg.cu_rows[b];
  const int L = (d.rg.cu_rows[b + 1] - row_b) * d.rg.len_mul;
  const int t0 = blockIdx.x * tt_out;
  if (t0 >= L) return;
  const int64_t seq_row0 = (int64_t)row_b * d.rg.len_mul;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wn = wave / WT, wt = wave % WT;
  const int g = lane >> 5;
  const int col0 = wt * NT * 32;
  const int nf0 = wn * NF;

  const int rx = WGCOLS + 2 * p1;   // x tile rows: row r <-> position t0 - p2 - p1 + r
  // h tile: WGCOLS + K - 1 rows, row j <-> position t0 - p2 + j.  It OVERLAYS the x tile (dead once
  // stage 1 has finished everywhere): half the LDS -> 2-3 workgroups per CU, so one workgroup's
  // staging / epilogues overlap another's MFMA phase.
  char* xs = smem;
  char* hs = smem;

  const T* xin[3] = {(const T*)d.x, nullptr, nullptr};
  stage_rows<T>(xs, pitch, rx, C, t0 - p2 - p1, L, seq_row0, xin, 1, C, 0, 1.f, JATTS_PRE_LRELU, d.slope);
  __syncthreads();

  f32x16 acc[NF][NT];
  zero_acc<NF, NT>(acc);
  conv_stage<T, NF, NT, RD>(acc, (const T*)d.w1, KC16, NFR, nf0, 0, KC16, K, dil, xs, pitch, col0, lane);

  // epilogue 1: h = lrelu(acc + b1), forced to 0 outside the sequence (conv2's zero padding)
  __syncthreads();  // every wave is done reading x: the tile may now be overwritten by h
  // rows of h past the computed columns are only read by discarded output columns
  for (int u = threadIdx.x; u < (K - 1) * (C / 8); u += blockDim.x) {
    const int r = WGCOLS + u / (C / 8), cu = u % (C / 8);
    typename Elem<T>::vec8 z;
#pragma unroll
    for (int e = 0; e < 8; ++e) z[e] = from_f32<T>(0.f);
    Vec8IO<T>::sts(hs + (size_t)r * pitch + (size_t)cu * 8 * sizeof(T), z);
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = col0 + t * 32 + (lane & 31);
    const int pos = t0 - p2 + col;
    const bool inside = pos >= 0 && pos < L;
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n0 = (nf0 + f) * 32 + 8 * q + 4 * g;
        const f32x4 bb = *reinterpret_cast<const f32x4*>(d.b1 + n0);
        T o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = inside ? lrelu(acc[f][t][4 * q + e] + bb[e], d.slope) : 0.f;
          o[e] = from_f32<T>(v);
        }
        char* p = hs + (size_t)col * pitch + (size_t)n0 * sizeof(T);
        if (sizeof(T) == 2) {
          *reinterpret_cast<f16x4*>(p) = f16x4{(f16)o[0], (f16)o[1], (f16)o[2], (f16)o[3]};
        } else {
          *reinterpret_cast<f32x4*>(p) = f32x4{(float)o[0], (float)o[1], (float)o[2], (float)o[3]};
        }
      }
  }
  __syncthreads();

  zero_acc<NF, NT>(acc);
  conv_stage<T, NF, NT, RD>(acc, (const T*)d.w2, KC16, NFR, nf0, 0, KC16, K, 1, hs, pitch, col0, lane);

  // epilogue 2: y = x + acc + b2 for the tt_out valid columns
  const T* xg = (const T*)d.x;
  T* yg = (T*)d.y;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = col0 + t * 32 + (lane & 31);
    const int pos = t0 + col;
    if (col >= tt_out || pos >= L) continue;
    const int64_t rowoff = (seq_row0 + pos) * (int64_t)C;
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n0 = (nf0 + f) * 32 + 8 * q + 4 * g;
        const f32x4 bb = *reinterpret_cast<const f32x4*>(d.b2 + n0);
        if (sizeof(T) == 2) {
          const f16x4 xr = *reinterpret_cast<const f16x4*>((const f16*)xg + rowoff + n0);
          f16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (f16)(acc[f][t][4 * q + e] + bb[e] + (float)xr[e]);
          *reinterpret_cast<f16x4*>((f16*)yg + rowoff + n0) = o;
        } else {
          const f32x4 xr = *reinterpret_cast<const f32x4*>((const float*)xg + rowoff + n0);
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = acc[f][t][4 * q + e] + bb[e] + xr[e];
          *reinterpret_cast<f32x4*>((float*)yg + rowoff + n0) = o;
        }
      }
  }
}

template <typename T, int C, int WGCOLS, int WN, int NT>
int launch_resunit(const jatts_resunit_desc& d, hipStream_t s) {
  constexpr int WT = WGCOLS / (NT * 32);
  const int K = d.k_w, p2 = (K - 1) / 2, p1 = p2 * d.dil;
  const int tt_out = WGCOLS - 2 * p2;
  if (tt_out < 8) return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "resunit: kernel too wide for tile");
  const size_t pitch = C * sizeof(T) + 16;
  const size_t rows_x = WGCOLS + 2 * p1, rows_h = WGCOLS + K - 1;
  const size_t lds = (rows_x > rows_h ? rows_x : rows_h) * pitch;  // h overlays x
  if (lds > 160 * 1024) return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "resunit: tile exceeds 160 KiB LDS");
  const int64_t maxL = (int64_t)d.rg.max_len * d.rg.len_mul;
  dim3 grid((unsigned)((maxL + tt_out - 1) / tt_out), (unsigned)d.rg.n_seq);
  auto kern = resunit_kernel<T, C, WGCOLS, WN, NT>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return jatts_set_error(e, __FILE__, __LINE__);
  }
  hipLaunchKernelGGL(kern, grid, dim3(WN * WT * 64), lds, s, d);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

}  // namespace

extern "C" int64_t jatts_conv_weight_index(int32_t n, int32_t tap, int32_t c, int32_t n_pad, int32_t c_in) {
  const int64_t KC16 = c_in / 16, NFR = n_pad / 32;
  const int64_t kc = c / 16, nf = n / 32;
  const int64_t lane = 32 * ((c % 16) / 8) + (n % 32);
  return ((((int64_t)tap * KC16 + kc) * NFR + nf) * 64 + lane) * 8 + (c % 8);
}

extern "C" int jatts_conv1d(const jatts_conv_desc* d, void* stream) {
  if (!d || !d->x[0] || !d->w || !d->y || !d->rg.cu_rows) return jatts_set_error_msg(JATTS_ERR_ARG, "conv1d: null pointer");
  if (d->c_in <= 0 || d->c_in % 32) return jatts_set_error_msg(JATTS_ERR_ARG, "conv1d: c_in must be a positive multiple of 32");
  if (d->ldx % 8) return jatts_set_error_msg(JATTS_ERR_ARG, "conv1d: ldx must be a multiple of 8");
  if (d->n_in < 1 || d->n_in > 3 || d->k_w < 1 || d->dil < 1 || d->n_out < 1 || d->rg.n_seq < 1 || d->rg.len_mul < 1)
    return jatts_set_error_msg(JATTS_ERR_ARG, "conv1d: bad geometry");
  if (d->rg.max_len <= 0) return JATTS_OK;
  hipStream_t s = (hipStream_t)stream;
  const bool narrow = d->n_out <= 64;
  if (d->dtype == JATTS_F16) {
    return narrow ? launch_conv<f16, 2, 2, 1, 4>(*d, s) : launch_conv<f16, 2, 4, 2, 2>(*d, s);
  } else if (d->dtype == JATTS_F32) {
    return narrow ? launch_conv<float, 2, 2, 1, 4>(*d, s) : launch_conv<float, 2, 2, 2, 2>(*d, s);
  }
  return jatts_set_error_msg(JATTS_ERR_ARG, "conv1d: unknown dtype");
}

extern "C" int jatts_hifigan_resunit(const jatts_resunit_desc* d, void* stream) {
  if (!d || !d->x || !d->y || !d->w1 || !d->w2 || !d->b1 || !d->b2 || !d->rg.cu_rows)
    return jatts_set_error_msg(JATTS_ERR_ARG, "resunit: null pointer");
  if (d->x == d->y) return jatts_set_error_msg(JATTS_ERR_ARG, "resunit: y must not alias x");
  if (d->k_w < 1 || !(d->k_w & 1) || d->dil < 1) return jatts_set_error_msg(JATTS_ERR_ARG, "resunit: odd k_w and dil>=1 required");
  if (d->rg.max_len <= 0) return JATTS_OK;
  hipStream_t s = (hipStream_t)stream;
  if (d->dtype == JATTS_F16) {
    // tile variants: <C, workgroup columns, waves along n, 32-col fragments per wave>.
    // JATTS_RESUNIT_VARIANT (tuning knob, read once) selects alternative tilings for sweeps.
    static const int variant = [] { const char* e = getenv("JATTS_RESUNIT_VARIANT"); return e ? atoi(e) : 0; }();
    switch (d->channels * 10 + variant) {
      case 320: return launch_resunit<f16, 32, 256, 1, 2>(*d, s);
      case 321: return launch_resunit<f16, 32, 256, 1, 4>(*d, s);
      case 322: return launch_resunit<f16, 32, 128, 1, 4>(*d, s);
      case 640: return launch_resunit<f16, 64, 256, 1, 2>(*d, s);
      case 641: return launch_resunit<f16, 64, 256, 1, 4>(*d, s);
      case 642: return launch_resunit<f16, 64, 128, 1, 4>(*d, s);
      case 1280: return launch_resunit<f16, 128, 128, 2, 2>(*d, s);
      case 1281: return launch_resunit<f16, 128, 128, 2, 4>(*d, s);
      case 1282: return launch_resunit<f16, 128, 256, 2, 4>(*d, s);
      case 2560: return launch_resunit<f16, 256, 64, 4, 2>(*d, s);
      case 2561: return launch_resunit<f16, 256, 128, 4, 4>(*d, s);
      case 2562: return launch_resunit<f16, 256, 128, 4, 2>(*d, s);
      case 5120: case 5121: case 5122: return launch_resunit<f16, 512, 32, 4, 1>(*d, s);
    }
  } else if (d->dtype == JATTS_F32) {
    switch (d->channels) {
      case 32: return launch_resunit<float, 32, 256, 1, 2>(*d, s);
      case 64: return launch_resunit<float, 64, 128, 1, 2>(*d, s);
      case 128: return launch_resunit<float, 128, 64, 2, 2>(*d, s);
      case 256: return launch_resunit<float, 256, 32, 4, 1>(*d, s);
    }
  }
  return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "resunit: unsupported channels/dtype (use jatts_conv1d)");
}
