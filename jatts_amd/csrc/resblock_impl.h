// Fused HiFi-GAN ResBlock: all dilation units of one block in ONE launch,
//     for d in dils:  x <- x + conv_k,1(lrelu(conv_k,d(lrelu(x))))
// The HBM-bound stages (32 / 64 channels, k = 3 / 7) then read x once and write it once per ResBlock instead of once
// per unit (2/3 of the traffic gone), and the 2 x n_units staging / store phases of the per-unit launches collapse
// into one of each.
//
// Every conv of every unit computes the SAME window of WGCOLS columns (window column c <-> position t0 - H + c, H = the
// halo the chain consumes per side = sum over units of p2 * (dil + 1)); only the centre tt_out = WGCOLS - 2H columns
// are valid at the end -- each unit's edge columns read stale data and are never used by a valid column.  Because the
// window is fixed, a lane owns the same (column, channel) elements in every accumulator of the chain, so the residual
// stream x stays in REGISTERS (packed f16) from the first unit to the last; LDS holds one tile: lrelu(x), overwritten
// by h, overwritten by lrelu(x'), ... with M = p2 * max(dil) margin rows a side for the dilated taps.
#pragma once
#include <type_traits>

#include "resunit_impl.h"

namespace {

__device__ __forceinline__ f16x4 mask4(f16x4 v, uint32_t m) {
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  u32x2 b = __builtin_bit_cast(u32x2, v);
  b &= m;
  return __builtin_bit_cast(f16x4, b);
}
__device__ __forceinline__ f16x4 lrelu4(f16x4 v, float slope) {   // max(v, f16(float(v) * slope)) for 0 <= slope <= 1, as lrelu8
  f16x4 y;
#pragma unroll
  for (int e = 0; e < 4; ++e) y[e] = (f16)((float)v[e] * slope);
  return __builtin_elementwise_max(v, y);
}

template <typename T, int C, int WGCOLS, int WN, int NT, int KCGMAX = 8, int OCC = 2>
__global__ __launch_bounds__(WN*(WGCOLS / (NT * 32)) * 64, OCC) void resblock_kernel(jatts_resblock_desc d, unsigned bias_off) {
  // f16: the HBM-bound shapes (x read / written once per ResBlock).  f32 (round 3): every shape is MFMA-bound, but at C = 32 / 64 with k = 3
  // a conv is only 6 / 12 K-steps and the per-unit launches spend as long in their staging / store phases (instruction-slot bound,
  // profiles/r03_notes.md) as in their MFMAs: one launch per ResBlock drops two of three staging + store passes.
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int WT = WGCOLS / (NT * 32);
  constexpr int NF = C / (WN * 32);
  constexpr int KC16 = C / 16, NFR = C / 32;
  constexpr int pitch = C * (int)sizeof(T) + 16;
  constexpr int KCG = sizeof(T) == 4 ? 2 : (KC16 < KCGMAX ? KC16 : KCGMAX);   // f32: a step is 64-cycle MFMAs, two steps of look-ahead suffice
  constexpr int NTHR = WN * WT * 64;
  constexpr int UPR = C / 8;
  static_assert(WT * NT * 32 == WGCOLS && NF * WN * 32 == C, "tile shape");
  typedef typename Elem<T>::vec8 V8;

  const int K = d.k_w, p2 = (K - 1) / 2, NU = d.n_units;
  int H = 0, M = 0;
  for (int u = 0; u < NU; ++u) {
    H += p2 * (d.dil[u] + 1);
    M = max(M, p2 * d.dil[u]);
  }
  const int tt_out = WGCOLS - 2 * H;
  int b = blockIdx.y, bx = blockIdx.x;
  if (ragged_is_1d(d.rg) && !ragged_locate(d.rg, tt_out, blockIdx.x, b, bx)) return;   // 1-D grid over the real tiles of a ragged batch
  const int row_b = d.rg.cu_rows[b];
  const int L = (d.rg.cu_rows[b + 1] - row_b) * d.rg.len_mul;
  const int t0 = bx * tt_out;
  if (t0 >= L) return;
  const int64_t seq_row0 = (int64_t)row_b * d.rg.len_mul;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wn = wave / WT, wt = wave % WT;
  const int g = lane >> 5;
  const int col0 = wt * NT * 32;
  const int nf0 = wn * NF;
  const int rows = WGCOLS + 2 * M;   // tile row r <-> position t0 - H - M + r; window column c lives in row M + c

  float* bs = reinterpret_cast<float*>(smem + bias_off);   // [unit][b1 | b2][C]
  for (int u = threadIdx.x; u < 2 * C * NU; u += NTHR) {
    const int un = u / (2 * C), r = u - un * 2 * C;
    bs[u] = r < C ? d.b1[un][r] : d.b2[un][r - C];
  }

  WStream<T, NF, KCG> ws;   // weight stream of the 2 * NU convs; its first group is fetched under the x staging
  ws.prefetch((const T*)d.w1[0], NFR, nf0, lane);
  // ---- stage the raw x tile (zeros outside the sequence); the residual fragments are read from it before it is activated
  const T* xg = (const T*)d.x;
  {
    constexpr int UBX = ((WGCOLS + 64) * UPR + NTHR - 1) / NTHR;
    stage_unit<T, (UBX < 8 ? 8 : (UBX < 24 ? UBX : 24)), NTHR>(smem, pitch, rows, UPR, t0 - H - M, L, seq_row0, xg, C, false, d.slope);
  }
  __syncthreads();
  typedef typename std::conditional<sizeof(T) == 2, f16x4, f32x4>::type R4;
  R4 resid[NF][NT][4];   // x of this lane's (column, 4-channel quad) elements: C-fragment layout (packed f16 / f32)
  uint32_t keep[NT];   // all-ones / zero: ANDed onto packed f16 pairs (exact zeroing, also of an overflowed edge column)
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = col0 + t * 32 + (lane & 31);
    const int pos = t0 - H + col;
    keep[t] = (pos >= 0 && pos < L) ? 0xffffffffu : 0u;   // every conv zero-pads its own input: x and h are 0 outside the sequence
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        resid[f][t][q] = *reinterpret_cast<const R4*>(smem + (size_t)(M + col) * pitch + (size_t)((nf0 + f) * 32 + 8 * q + 4 * g) * sizeof(T));
  }
  __syncthreads();
  for (int u = threadIdx.x; u < rows * UPR; u += NTHR) {   // tile <- lrelu(tile), in place
    const int r = u / UPR, cu = u - r * UPR;
    char* p = smem + (size_t)r * pitch + (size_t)cu * 8 * sizeof(T);
    V8 v = Vec8IO<T>::lds(p);
    lrelu8(v, d.slope);
    Vec8IO<T>::sts(p, v);
  }
  __syncthreads();

  f32x16 acc[NF][NT];
  auto bias_acc = [&](const float* bv) {
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 bb = *reinterpret_cast<const f32x4*>(bv + (nf0 + f) * 32 + 8 * q + 4 * g);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[f][t][4 * q + e] = bb[e];
      }
  };

  for (int u = 0; u < NU; ++u) {
    const int dil = d.dil[u], p1 = p2 * dil;
    // conv_k,dil over lrelu(x): window column c reads tile rows (M - p1) + c + tap * dil
    bias_acc(bs + (size_t)u * 2 * C);
    conv_full_ws<T, NF, NT, KC16, KCG>(acc, ws, (const T*)d.w1[u], (const T*)d.w2[u], K, dil, smem + (size_t)(M - p1) * pitch, pitch, col0, lane);
    lds_barrier();   // every wave is done reading lrelu(x): the window rows may be overwritten by h
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int col = col0 + t * 32 + (lane & 31);
#pragma unroll
      for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          char* dst = smem + (size_t)(M + col) * pitch + (size_t)((nf0 + f) * 32 + 8 * q + 4 * g) * sizeof(T);
          if constexpr (sizeof(T) == 2) {
            // h = lrelu(acc) rounded to f16: round first, then max(h, f16(h * slope)) on packed pairs (the staging form)
            *reinterpret_cast<f16x4*>(dst) = lrelu4(mask4(f16x4{(f16)acc[f][t][4 * q], (f16)acc[f][t][4 * q + 1], (f16)acc[f][t][4 * q + 2],
                                                               (f16)acc[f][t][4 * q + 3]}, keep[t]), d.slope);
          } else {   // f32: h = lrelu(acc), zero outside the sequence
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = keep[t] ? lrelu(acc[f][t][4 * q + e], d.slope) : 0.f;
            *reinterpret_cast<f32x4*>(dst) = o;
          }
        }
    }
    lds_barrier();
    // conv_k,1 over h
    bias_acc(bs + (size_t)u * 2 * C + C);
    const bool last = u == NU - 1;
    conv_full_ws<T, NF, NT, KC16, KCG>(acc, ws, (const T*)d.w2[u], last ? nullptr : (const T*)d.w1[u + 1], K, 1,
                                       smem + (size_t)(M - p2) * pitch, pitch, col0, lane);
    lds_barrier();   // every wave is done reading h
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int col = col0 + t * 32 + (lane & 31);
#pragma unroll
      for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          char* dst = smem + (size_t)(M + col) * pitch + (size_t)((nf0 + f) * 32 + 8 * q + 4 * g) * sizeof(T);
          if constexpr (sizeof(T) == 2) {
            f16x4 xn;
#pragma unroll
            for (int e = 0; e < 4; ++e) xn[e] = (f16)(acc[f][t][4 * q + e] + (float)resid[f][t][q][e]);   // x' = x + conv(...) in f32, rounded once
            xn = mask4(xn, keep[t]);
            resid[f][t][q] = xn;
            *reinterpret_cast<f16x4*>(dst) = last ? xn : lrelu4(xn, d.slope);   // next unit's operand: lrelu(x'); after the last unit: x' itself
          } else {
            f32x4 xn, o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              xn[e] = keep[t] ? acc[f][t][4 * q + e] + resid[f][t][q][e] : 0.f;
              o[e] = last ? xn[e] : lrelu(xn[e], d.slope);
            }
            resid[f][t][q] = xn;
            *reinterpret_cast<f32x4*>(dst) = o;
          }
        }
    }
    lds_barrier();
  }

  // ---- coalesced output pass: the centre tt_out columns of the final x tile (+ the fused MRF mean, as the unit kernel)
  {
    const int vrows = min(tt_out, L - t0);
    const int64_t g0 = (seq_row0 + t0) * (int64_t)C;
    const char* ys = smem + (size_t)(M + H) * pitch;
    T* yg = (T*)d.y;
    constexpr bool keep_small = C <= 64;
    if (d.add0) unit_store_pass<T, C, keep_small ? 2 : 4, true, NTHR, false>(d.add0, d.add1, d.out_scale, ys, pitch, vrows, xg, yg, g0);
    else unit_store_pass<T, C, keep_small ? 4 : 8, false, NTHR, false>(d.add0, d.add1, d.out_scale, ys, pitch, vrows, xg, yg, g0);
  }
}

template <typename T, int C, int WGCOLS, int WN, int NT, int KCGMAX = 8, int OCC = 2>
int launch_resblock(const jatts_resblock_desc& d, hipStream_t s) {
  constexpr int WT = WGCOLS / (NT * 32);
  const int p2 = (d.k_w - 1) / 2;
  int H = 0, M = 0;
  for (int u = 0; u < d.n_units; ++u) {
    H += p2 * (d.dil[u] + 1);
    M = M > p2 * d.dil[u] ? M : p2 * d.dil[u];
  }
  const int tt_out = WGCOLS - 2 * H;
  if (tt_out < 32) return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "resblock: receptive field too wide for the tile");
  const size_t pitch = C * sizeof(T) + 16;
  size_t lds = (size_t)(WGCOLS + 2 * M) * pitch;
  const unsigned bias_off = (unsigned)lds;
  lds += (size_t)d.n_units * 2 * C * sizeof(float);
  if (lds > 160 * 1024) return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "resblock: tile exceeds 160 KiB LDS");
  const int64_t maxL = (int64_t)d.rg.max_len * d.rg.len_mul;
  dim3 grid((unsigned)((maxL + tt_out - 1) / tt_out), (unsigned)d.rg.n_seq);
  if (const int64_t n1 = ragged_tiles_1d(d.rg, tt_out)) grid = dim3((unsigned)n1);
  auto kern = resblock_kernel<T, C, WGCOLS, WN, NT, KCGMAX, OCC>;
  // the dynamic-LDS limit ONCE per kernel (function-local static of this template instantiation), not per launch: the per-launch call was seen to stall the
  // host for ~20 ms now and then (profiles/r06_notes.md section 8)
  static const hipError_t lds_attr = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (lds_attr != hipSuccess) return jatts_set_error(lds_attr, __FILE__, __LINE__);
  hipLaunchKernelGGL(kern, grid, dim3(WN * WT * 64), lds, s, d, bias_off);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

}  // namespace
