// Fused HiFi-GAN ResBlock dilation unit  y = x + conv_1(lrelu(conv_d(lrelu(x))))  (MFMA implicit GEMM, x and h tiles in LDS).
// Instantiated per dtype / channel group in resunit_*.hip.
#pragma once
#include "conv_tiles.h"

extern unsigned long long* jatts_g_trace;  // profiling hook (conv_api.hip: jatts_debug_trace)
extern unsigned jatts_g_trace_cap;

namespace {

// Unit-kernel output pass: y = (acc + b2 tile in LDS) + x [+ MRF partners] with row-contiguous 16-byte accesses;
// all global reads of a batch are issued before any is consumed (one round trip per batch, not per unit).
template <typename T, int C, int UB, bool ADD, int NTHR, bool RESID = true>
__device__ __forceinline__ void unit_store_pass(const void* add0, const void* add1, float out_scale, const char* ys, int pitch,
                                                int vrows, const T* xg, T* yg, int64_t g0) {
  typedef typename Elem<T>::vec8 V8;
  constexpr int UPR = C / 8;
  const int total = vrows * UPR;
  const bool has_add1 = ADD && add1 != nullptr;
  for (int u0 = threadIdx.x; u0 < total; u0 += UB * NTHR) {
    V8 xr[UB], a0[ADD ? UB : 1], a1[ADD ? UB : 1];
#pragma unroll
    for (int i = 0; i < UB; ++i) {
      const int u = u0 + i * NTHR;
      if (u < total) {
        if (RESID && JATTS_ABLATE != 3) xr[i] = Vec8IO<T>::ldg(xg + g0 + (int64_t)u * 8);
        if (ADD) {
          a0[i] = Vec8IO<T>::ldg((const T*)add0 + g0 + (int64_t)u * 8);
          if (has_add1) a1[i] = Vec8IO<T>::ldg((const T*)add1 + g0 + (int64_t)u * 8);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < UB; ++i) {
      const int u = u0 + i * NTHR;
      if (u >= total) continue;
      const int r = u / UPR, cu = u - r * UPR;
      V8 v = Vec8IO<T>::lds(ys + (size_t)r * pitch + (size_t)cu * 8 * sizeof(T));
      if (RESID && JATTS_ABLATE != 3) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = from_f32<T>(to_f32(v[e]) + to_f32(xr[i][e]));  // residual
      }
      if (ADD) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          v[e] = from_f32<T>((to_f32(v[e]) + to_f32(a0[i][e]) + (has_add1 ? to_f32(a1[i][e]) : 0.f)) * out_scale);
      }
      T* dst = yg + g0 + (int64_t)u * 8;
      if ((JATTS_ABLATE != 4 && JATTS_ABLATE != 12) || to_f32(v[0]) == 12345.678f) {
        if (sizeof(T) == 2) *reinterpret_cast<f16x8*>(dst) = *reinterpret_cast<const f16x8*>(&v);
        else {
          *reinterpret_cast<f32x4*>(dst) = f32x4{to_f32(v[0]), to_f32(v[1]), to_f32(v[2]), to_f32(v[3])};
          *reinterpret_cast<f32x4*>(dst + 4) = f32x4{to_f32(v[4]), to_f32(v[5]), to_f32(v[6]), to_f32(v[7])};
        }
      }
    }
  }
}

__device__ __forceinline__ void lrelu8(f16x8& v, float slope) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f16x2 x2 = {v[2 * i], v[2 * i + 1]};
    const f16x2 y2 = {(f16)((float)x2[0] * slope), (f16)((float)x2[1] * slope)};   // rounded to f16 once, as the select form
    const f16x2 m = __builtin_elementwise_max(x2, y2);                             // v_pk_max_f16
    v[2 * i] = m[0];
    v[2 * i + 1] = m[1];
  }
}
__device__ __forceinline__ void lrelu8(f32x8& v, float slope) {
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], v[e] * slope);
}

// Unit-kernel staging: the WHOLE x tile (one input, LeakyReLU) in one batch of UB 16-byte loads per thread, all in
// flight before the first is consumed.  The accumulators are not live yet, so the registers are free; the generic
// 8-per-batch form paid 3 serial HBM round trips per tile (stage x = 29 % of a k=3 workgroup's lifetime, tools/trace_unit.py).
template <typename T, int UB, int NTHR>
__device__ __forceinline__ void stage_unit(char* lds, int pitch, int rows, int upr, int pos0, int L, int64_t seq_row0,
                                           const T* x, int ldx, bool pre_lrelu, float slope) {
  typedef typename Elem<T>::vec8 V8;
  const int total = rows * upr;
  // NTHR (= blockDim.x) and upr are compile-time: unit j of a thread is (row0 + j * NTHR / upr, same column), so the
  // per-unit index arithmetic folds to one add -- every VALU op of this phase is paid ~3x under a co-resident MFMA wave
  for (int base = threadIdx.x; base < total; base += NTHR * UB) {
    V8 v[UB];
#pragma unroll
    for (int j = 0; j < UB; ++j) {
      const int u = base + j * NTHR;
      const int r = u / upr, cu = u - r * upr;
      const int pos = pos0 + r;
      if (u < total && pos >= 0 && pos < L) v[j] = Vec8IO<T>::ldg(x + (seq_row0 + pos) * (int64_t)ldx + cu * 8);
      else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[j][e] = from_f32<T>(0.f);
      }
    }
#pragma unroll
    for (int j = 0; j < UB; ++j) {
      const int u = base + j * NTHR;
      if (u >= total) continue;
      const int r = u / upr, cu = u - r * upr;
      if (pre_lrelu) {
        // LeakyReLU(x) = max(x, T(float(x) * slope)) for 0 <= slope <= 1: the product is rounded to T once (same value
        // as the select form), the max runs in T -- for f16 that is one v_fma_mix per element + packed max instead of
        // cvt, cmp, cndmask, mul, cvt
        lrelu8(v[j], slope);
      }
      Vec8IO<T>::sts(lds + (size_t)r * pitch + (size_t)cu * 8 * sizeof(T), v[j]);
    }
  }
}

// ------------------------------------------------------------ fused HiFi-GAN dilation unit
// OCC = minimum waves per SIMD the register allocation must leave room for (0: 2 when the accumulators fit 128 registers)
// RREG: the residual x of this lane's output elements is read from the RAW x tile into registers before the tile is activated in place,
// so x is fetched from HBM once (PMC: the f32 C=128 unit moved 5.98 GB per launch against 3.22 GB algorithmic, 1.61 GB of it the
// residual re-read of the store pass).  f32 tiles only: 16 more registers per accumulator fragment.
template <typename T, int C, int WGCOLS, int WN, int NT, int KCGMAX = 8, int OCC = 0, bool RREG = false>
__global__ __launch_bounds__(WN*(WGCOLS / (NT * 32)) * 64, OCC ? OCC : ((C <= 256 && (C / (WN * 32)) * NT * 16 <= 128) ? 2 : 1)) void resunit_kernel(jatts_resunit_desc d, unsigned long long* trace, unsigned trace_cap, unsigned bias_off) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int WT = WGCOLS / (NT * 32);
  constexpr int NF = C / (WN * 32);
  constexpr int KC16 = C / 16, NFR = C / 32;
  constexpr int pitch = C * (int)sizeof(T) + 16;
  constexpr int KCG = sizeof(T) == 4 ? 2 : (KC16 < KCGMAX ? KC16 : KCGMAX);  // ring depth = group size
  static_assert(WT * NT * 32 == WGCOLS && NF * WN * 32 == C, "tile shape");
  // Phase trace (profiling hook, jatts_debug_trace): thread 0 of the first trace_cap workgroups stamps s_memtime
  // at every phase boundary: [hw id, start, staged, conv1, h written, conv2, y assembled, stored, realtime x2].
  const unsigned wg_lin = blockIdx.x + blockIdx.y * gridDim.x;
  const bool tracing = trace != nullptr && wg_lin < trace_cap && threadIdx.x == 0;
#define JATTS_STAMP(i) do { if (tracing) trace[(size_t)wg_lin * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
  if (tracing) {
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    trace[(size_t)wg_lin * 16] = ((unsigned long long)xcc << 32) | hwid;
    trace[(size_t)wg_lin * 16 + 8] = __builtin_amdgcn_s_memrealtime();
  }
  JATTS_STAMP(1);
  const int K = d.k_w, dil = d.dil;
  const int p2 = (K - 1) / 2, p1 = p2 * dil;
  const int tt_out = WGCOLS - 2 * p2;

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

  const int rx = WGCOLS + 2 * p1;   // x tile rows: row r <-> position t0 - p2 - p1 + r
  // h tile: WGCOLS + K - 1 rows, row j <-> position t0 - p2 + j.  It OVERLAYS the x tile (dead once
  // stage 1 has finished everywhere): half the LDS -> 2-3 workgroups per CU, so one workgroup's
  // staging / epilogues overlap another's MFMA phase.
  char* xs = smem;
  char* hs = smem;
  // b1 | b2 live in LDS behind the tile: read per fragment in the epilogues as ds_read_b128 (~100 clk) instead of
  // 8 dependent global loads each (the epilogue-1 body measured 11 k of a 59 k-clk workgroup lifetime at C=128, k=3,
  // almost all of it load latency: tools/trace_unit.py)
  float* bs = reinterpret_cast<float*>(smem + bias_off);
  for (int u = threadIdx.x; u < 2 * C; u += blockDim.x) bs[u] = u < C ? d.b1[u] : d.b2[u - C];

  // f16: one weight stream for both convs -- conv1's first group is fetched under the x staging, conv2's under epilogue 1.
  // f32 (64-cycle MFMAs, two-deep ring): the fill is already hidden and the longer register lifetime measured 2 % slower.
  constexpr bool STREAM = sizeof(T) == 2;
  WStream<T, NF, KCG> ws;
  if constexpr (STREAM) ws.prefetch((const T*)d.w1, NFR, nf0, lane);
  const T* xin[3] = {(const T*)d.x, nullptr, nullptr};
  if (JATTS_ABLATE != 2 && JATTS_ABLATE != 7 && JATTS_ABLATE != 12)
  {
    constexpr int NTHR = WN * WT * 64;
    constexpr int UBX = ((WGCOLS + 64) * (C / 8) + NTHR - 1) / NTHR;   // covers halos up to 32 rows a side in one batch
    stage_unit<T, (UBX < 8 ? 8 : (UBX < 24 ? UBX : 24)), NTHR>(xs, pitch, rx, C / 8, t0 - p2 - p1, L, seq_row0, xin[0], C, JATTS_ABLATE != 1 && !RREG, d.slope);
  }
  __syncthreads();
  // x at (output column, channel quad) of this lane, C-fragment layout, in the tile's own type (f16 tiles: 2 registers per quad)
  typedef T resid_t __attribute__((ext_vector_type(4)));
  resid_t resid[RREG ? NF : 1][RREG ? NT : 1][4];
  if constexpr (RREG) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int col = col0 + t * 32 + (lane & 31);   // output column col <-> x tile row col + p2 + p1
#pragma unroll
      for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          resid[f][t][q] = *reinterpret_cast<const resid_t*>(xs + (size_t)(col + p2 + p1) * pitch + (size_t)((nf0 + f) * 32 + 8 * q + 4 * g) * sizeof(T));
    }
    __syncthreads();
    for (int u = threadIdx.x; u < rx * (C / 8); u += blockDim.x) {   // tile <- lrelu(tile), in place
      char* p = xs + (size_t)(u / (C / 8)) * pitch + (size_t)(u % (C / 8)) * 8 * sizeof(T);
      typename Elem<T>::vec8 v = Vec8IO<T>::lds(p);
      lrelu8(v, d.slope);
      Vec8IO<T>::sts(p, v);
    }
    __syncthreads();
  }
  JATTS_STAMP(2);

  // the accumulators start at the bias (C layout: register 4q+e of fragment f <-> channel 32(nf0+f) + 8q + 4g + e), which
  // takes 128 adds and the bias reads out of each epilogue's dependent chain
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
  bias_acc(bs);
  if (JATTS_ABLATE < 6 || JATTS_ABLATE > 9) {
    if constexpr (STREAM) conv_full_ws<T, NF, NT, KC16, KCG>(acc, ws, (const T*)d.w1, (const T*)d.w2, K, dil, xs, pitch, col0, lane);
    else conv_full<T, NF, NT, KC16, KCG>(acc, (const T*)d.w1, NFR, nf0, K, dil, xs, pitch, col0, lane);
  }

  JATTS_STAMP(3);
  // epilogue 1: h = lrelu(acc + b1), forced to 0 outside the sequence (conv2's zero padding)
  if constexpr (STREAM) lds_barrier();  // every wave is done reading x: the tile may now be overwritten by h (conv2's first weights stay in flight)
  else __syncthreads();
  JATTS_STAMP(10);
  // rows of h past the computed columns are only read by discarded output columns
  for (int u = threadIdx.x; u < (K - 1) * (C / 8); u += blockDim.x) {
    const int r = WGCOLS + u / (C / 8), cu = u % (C / 8);
    typename Elem<T>::vec8 z;
#pragma unroll
    for (int e = 0; e < 8; ++e) z[e] = from_f32<T>(0.f);
    Vec8IO<T>::sts(hs + (size_t)r * pitch + (size_t)cu * 8 * sizeof(T), z);
  }
  JATTS_STAMP(11);
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = col0 + t * 32 + (lane & 31);
    const int pos = t0 - p2 + col;
    const float keep = (pos >= 0 && pos < L) ? 1.f : 0.f;   // h is 0 outside the sequence (conv2's zero padding)
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n0 = (nf0 + f) * 32 + 8 * q + 4 * g;
        T o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a = acc[f][t][4 * q + e] * keep;
          o[e] = from_f32<T>(fmaxf(a, a * d.slope));           // LeakyReLU for 0 < slope < 1: max(a, slope * a)
        }
        char* p = hs + (size_t)col * pitch + (size_t)n0 * sizeof(T);
        if (sizeof(T) == 2) {
          if ((JATTS_ABLATE != 5 && JATTS_ABLATE != 9) || to_f32(o[0]) == 12345.678f)
            *reinterpret_cast<f16x4*>(p) = f16x4{(f16)o[0], (f16)o[1], (f16)o[2], (f16)o[3]};
        } else {
          *reinterpret_cast<f32x4*>(p) = f32x4{(float)o[0], (float)o[1], (float)o[2], (float)o[3]};
        }
      }
  }
  JATTS_STAMP(12);
  if constexpr (STREAM) lds_barrier();
  else __syncthreads();
  JATTS_STAMP(4);

  bias_acc(bs + C);
  if (JATTS_ABLATE < 6 || JATTS_ABLATE > 9) {
    if constexpr (STREAM) conv_full_ws<T, NF, NT, KC16, KCG>(acc, ws, (const T*)d.w2, nullptr, K, 1, hs, pitch, col0, lane);
    else conv_full<T, NF, NT, KC16, KCG>(acc, (const T*)d.w2, NFR, nf0, K, 1, hs, pitch, col0, lane);
  }

  JATTS_STAMP(5);
  // epilogue 2: y = x + acc + b2 for the tt_out valid columns.  acc + b2 is assembled in LDS (the h region
  // is dead once every wave has left stage 2) and the residual is added in the row-contiguous 16-byte
  // store pass below: in MFMA fragment order both the x re-read and the y store scatter every 128-byte
  // line over 8 separate 8-byte accesses (1.4 ms of a 2.7 ms launch, profiles/r01_notes.md).
  const T* xg = (const T*)d.x;
  T* yg = (T*)d.y;
  if (JATTS_ABLATE == 8) return;
  __syncthreads();
  char* ys = smem;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = col0 + t * 32 + (lane & 31);
    if (col >= tt_out || t0 + col >= L) continue;
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n0 = (nf0 + f) * 32 + 8 * q + 4 * g;
        char* p = ys + (size_t)col * pitch + (size_t)n0 * sizeof(T);
        if (sizeof(T) == 2) {
          f16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = acc[f][t][4 * q + e];
            if constexpr (RREG) v += (float)resid[f][t][q][e];
            o[e] = (f16)v;
          }
          *reinterpret_cast<f16x4*>(p) = o;
        } else {
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            o[e] = acc[f][t][4 * q + e];
            if constexpr (RREG) o[e] += (float)resid[f][t][q][e];
          }
          *reinterpret_cast<f32x4*>(p) = o;
        }
      }
  }
  __syncthreads();
  JATTS_STAMP(6);
  {
    const int vrows = min(tt_out, L - t0);
    const int64_t g0 = (seq_row0 + t0) * (int64_t)C;  // the valid rows are contiguous in y: unit u <-> 8 elements at g0 + 8u
    constexpr bool keep_small = C <= 64;   // small-channel kernels live on occupancy (6 workgroups/CU): keep the batch short
    if (d.add0) unit_store_pass<T, C, keep_small ? 2 : 4, true, WN * WT * 64, !RREG>(d.add0, d.add1, d.out_scale, ys, pitch, vrows, xg, yg, g0);   // + fused MRF mean
    else unit_store_pass<T, C, keep_small ? 4 : 8, false, WN * WT * 64, !RREG>(d.add0, d.add1, d.out_scale, ys, pitch, vrows, xg, yg, g0);
  }
  JATTS_STAMP(7);
  if (tracing) trace[(size_t)wg_lin * 16 + 9] = __builtin_amdgcn_s_memrealtime();
#undef JATTS_STAMP
}

template <typename T, int C, int WGCOLS, int WN, int NT, int KCGMAX = 8, int OCC = 0, bool RREG = false>
int launch_resunit(const jatts_resunit_desc& d, hipStream_t s) {
  constexpr int WT = WGCOLS / (NT * 32);
  const int K = d.k_w, p2 = (K - 1) / 2, p1 = p2 * d.dil;
  const int tt_out = WGCOLS - 2 * p2;
  if (tt_out < 8) return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "resunit: kernel too wide for tile");
  const size_t pitch = C * sizeof(T) + 16;
  const size_t rows_x = WGCOLS + 2 * p1, rows_h = WGCOLS + K - 1;
  size_t lds = (rows_x > rows_h ? rows_x : rows_h) * pitch;  // h overlays x
  const unsigned bias_off = (unsigned)lds;
  lds += 2 * C * sizeof(float);                               // b1 | b2
  if (lds > 160 * 1024) return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "resunit: tile exceeds 160 KiB LDS");
  static const int pad_lds = [] { const char* e = getenv("JATTS_RESUNIT_PADLDS"); return e ? atoi(e) : 0; }();
  if (pad_lds && lds < (size_t)pad_lds) lds = pad_lds;  // experiment knob: force fewer workgroups per CU
  const int64_t maxL = (int64_t)d.rg.max_len * d.rg.len_mul;
  dim3 grid((unsigned)((maxL + tt_out - 1) / tt_out), (unsigned)d.rg.n_seq);
  if (const int64_t n1 = ragged_tiles_1d(d.rg, tt_out)) grid = dim3((unsigned)n1);
  auto kern = resunit_kernel<T, C, WGCOLS, WN, NT, KCGMAX, OCC, RREG>;
  // the dynamic-LDS limit ONCE per kernel (function-local static of this template instantiation), not per launch: the per-launch call was seen to stall the
  // host for ~20 ms now and then (profiles/r06_notes.md section 8)
  static const hipError_t lds_attr = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (lds_attr != hipSuccess) return jatts_set_error(lds_attr, __FILE__, __LINE__);
  hipLaunchKernelGGL(kern, grid, dim3(WN * WT * 64), lds, s, d, jatts_g_trace, jatts_g_trace_cap, bias_off);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

}  // namespace
