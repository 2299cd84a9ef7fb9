// Fused HiFi-GAN dilation unit  y = x + conv_1(lrelu(conv_d(lrelu(x))))  with f32 activations in HBM and
// ERROR-CORRECTED SPLIT-PRECISION MFMA operands (JATTS_F32S, round 4).
//
// Why: at exact f32 (v_mfma_f32_32x32x2_f32, 157 TFLOP/s) every unit shape is matrix-pipe bound and the pipe is ~95 % busy
// (profiles/r03_notes.md): the only lever left is fewer pipe cycles per FLOP.  An f32 value v, scaled by a power of two
// into f16's range, is carried as hi = f16(v), lo = f16(v - hi) (22 significand bits); a product of two such values is
// hi.hi + hi.lo + lo.hi (+ lo.lo, below f32 resolution): three v_mfma_f32_32x32x16_f16 with f32 accumulate = 3/16 of the
// pipe cycles of the f32 chain.  [published scheme: "recovering single-precision accuracy from half-precision matrix
// units"; here with ONE accumulator -- the gfx950 f16 MFMA accumulates in f32 without a truncation bias, tools/split_probe.hip]
//
// Scales (all powers of two, so scaling and un-scaling are exact):
//  * weights: per OUTPUT channel, chosen at load time so that max |w[n]| lands in [2^14, 2^15) (hip.pack_conv_weight_split);
//    the inverse scales ride next to the biases (ws1 / ws2 of the descriptor);
//  * activations: per WORKGROUP TILE -- the block maximum of lrelu(x) over the staged tile, and of h over the tile conv1
//    produced (accumulators are in registers at that point), each mapped to [2^14, 2^15).  A tile belongs to one utterance
//    and its geometry does not depend on the batch, so an utterance's result is bit-identical alone or inside any batch.
// With the maximum at 2^15 the lo halves are normal f16 numbers down to |v| ~ 2^-3, i.e. over 18 binary orders of
// magnitude below the tile maximum; below that the absolute error floor is 2^-25 (subnormal lo), 2^-40 of the maximum.
#pragma once
#include "resunit_impl.h"

namespace {

// Exponent s of the tile scale 2^s: amax = m 2^e with m in [0.5, 1) -> s = 15 - e, the tile maximum lands in [2^14, 2^15).
// amax == 0 -> 0.  Clamped so that 2^s, 2^-s and 2^-s times an inverse weight scale stay normal f32 numbers.
__device__ __forceinline__ int split_exp(float amax) {
  const int bexp = (int)((__float_as_uint(amax) >> 23) & 0xff);          // amax in [2^(bexp-127), 2^(bexp-126))
  const int s = 15 - (bexp - 126);
  return amax > 0.f ? (s > 60 ? 60 : (s < -60 ? -60 : s)) : 0;
}
__device__ __forceinline__ float exp2i(int s) { return __uint_as_float((unsigned)(127 + s) << 23); }   // |s| <= 126

__device__ __forceinline__ float block_amax(float m, float* slots, int wave, int lane, int n_waves) {
  m = wave_max(m);
  if (lane == 0) slots[wave] = m;
  lds_barrier();
  float a = slots[0];
  for (int w = 1; w < n_waves; ++w) a = fmaxf(a, slots[w]);
  return a;
}

// 4 scaled values -> hi / lo halves (RNE both times; v - hi is exact in f32)
__device__ __forceinline__ void split4(const float (&v)[4], f16x4& hi, f16x4& lo) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    hi[e] = (f16)v[e];
    lo[e] = (f16)(v[e] - (float)hi[e]);
  }
}

template <int C, int WGCOLS, int WN, int NT, int KCG, int OCC>
__global__ __launch_bounds__(WN*(WGCOLS / (NT * 32)) * 64, OCC) void resunit_split_kernel(jatts_resunit_desc d, unsigned long long* trace,
                                                                                          unsigned trace_cap, unsigned bias_off) {
  typedef f16s T;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int WT = WGCOLS / (NT * 32);
  constexpr int NF = C / (WN * 32);
  constexpr int KC16 = C / 16, NFR = C / 32;
  constexpr int pitch = C * 4 + 16;
  constexpr int NTHR = WN * WT * 64, NW = WN * WT;
  static_assert(WT * NT * 32 == WGCOLS && NF * WN * 32 == C, "tile shape");
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
  char* xs = smem;                  // split lrelu(x) tile; h overlays it; finally the f32 y tile
  char* hs = smem;
  // behind the tile: b1 | b2 | 1/wscale1 | 1/wscale2 (4C floats), then one amax slot per wave
  float* bs = reinterpret_cast<float*>(smem + bias_off);
  float* slots = bs + 4 * C;
  for (int u = threadIdx.x; u < 4 * C; u += NTHR) bs[u] = u < C ? d.b1[u] : (u < 2 * C ? d.b2[u - C] : (u < 3 * C ? d.ws1[u - 2 * C] : d.ws2[u - 3 * C]));

  WStream<T, NF, KCG> ws;
  ws.prefetch((const T*)d.w1, NFR, nf0, lane);

  // ---- stage: lrelu(x) tile -> registers, block maximum, scale, split, LDS
  int ex;
  {
    constexpr int UPR = C / 8;
    constexpr int UBX = ((WGCOLS + 64) * UPR + NTHR - 1) / NTHR;   // halos up to 32 rows a side in one batch
    constexpr int UB = UBX < 4 ? 4 : UBX;
    const float* x = (const float*)d.x;
    const int total = rx * UPR, pos0 = t0 - p2 - p1;
    float amax = 0.f;
    const int base = threadIdx.x;          // ONE batch covers the tile (launch_resunit_split refuses halos beyond 32 rows a side)
    f32x8 v[UB];
#pragma unroll
    for (int j = 0; j < UB; ++j) {
      const int u = base + j * NTHR;
      const int r = u / UPR, cu = u - r * UPR;
      const int pos = pos0 + r;
      if (u < total && pos >= 0 && pos < L) v[j] = Vec8IO<float>::ldg(x + (seq_row0 + pos) * (int64_t)C + cu * 8);
      else v[j] = f32x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
#pragma unroll
    for (int j = 0; j < UB; ++j) {
      lrelu8(v[j], d.slope);
#pragma unroll
      for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(v[j][e]));
    }
    ex = split_exp(block_amax(amax, slots, wave, lane, NW));
    const float sx = exp2i(ex);
#pragma unroll
    for (int j = 0; j < UB; ++j) {
      const int u = base + j * NTHR;
      if (u >= total) continue;
      const int r = u / UPR, cu = u - r * UPR;
      f16sx8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float sv = v[j][e] * sx;
        o.hi[e] = (f16)sv;
        o.lo[e] = (f16)(sv - (float)o.hi[e]);
      }
      Vec8IO<T>::sts(xs + (size_t)r * pitch + (size_t)cu * 32, o);
    }
  }
  __syncthreads();
  JATTS_STAMP(2);

  f32x16 acc[NF][NT];
  zero_acc<NF, NT>(acc);
  conv_full_ws<T, NF, NT, KC16, KCG>(acc, ws, (const T*)d.w1, (const T*)d.w2, K, dil, xs, pitch, col0, lane);
  JATTS_STAMP(3);

  // ---- epilogue 1: h = lrelu(acc / (wscale1 sx) + b1), 0 outside the sequence; block maximum; scale, split, LDS (over the dead x tile)
  int eh;
  {
    const float inv_sx = exp2i(-ex);
    float amax = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int col = col0 + t * 32 + (lane & 31);
      const int pos = t0 - p2 + col;
      const float keep = (pos >= 0 && pos < L) ? 1.f : 0.f;
#pragma unroll
      for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n0 = (nf0 + f) * 32 + 8 * q + 4 * g;
          const f32x4 bb = *reinterpret_cast<const f32x4*>(bs + n0);
          const f32x4 is = *reinterpret_cast<const f32x4*>(bs + 2 * C + n0);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float a = fmaf(acc[f][t][4 * q + e], is[e] * inv_sx, bb[e]) * keep;
            a = fmaxf(a, a * d.slope);
            acc[f][t][4 * q + e] = a;
            amax = fmaxf(amax, fabsf(a));
          }
        }
    }
    // the exchange's barrier is also the hand-off "every wave is done reading x" (conv2's first weights stay in flight)
    eh = split_exp(block_amax(amax, slots, wave, lane, NW));
    const float sh = exp2i(eh);
    JATTS_STAMP(10);
    for (int u = threadIdx.x; u < (K - 1) * (C / 8); u += NTHR) {   // rows past the computed columns: read by discarded columns only
      const int r = WGCOLS + u / (C / 8), cu = u % (C / 8);
      f16sx8 z;
#pragma unroll
      for (int e = 0; e < 8; ++e) z.hi[e] = z.lo[e] = (f16)0.f;
      Vec8IO<T>::sts(hs + (size_t)r * pitch + (size_t)cu * 32, z);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int col = col0 + t * 32 + (lane & 31);
#pragma unroll
      for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int cu = (nf0 + f) * 4 + q;
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[f][t][4 * q + e] * sh;
          f16x4 hi, lo;
          split4(v, hi, lo);
          char* p = hs + (size_t)col * pitch + (size_t)cu * 32 + 8 * g;
          *reinterpret_cast<f16x4*>(p) = hi;
          *reinterpret_cast<f16x4*>(p + 16) = lo;
        }
    }
  }
  JATTS_STAMP(12);
  lds_barrier();
  JATTS_STAMP(4);

  zero_acc<NF, NT>(acc);
  conv_full_ws<T, NF, NT, KC16, KCG>(acc, ws, (const T*)d.w2, nullptr, K, 1, hs, pitch, col0, lane);
  JATTS_STAMP(5);

  // ---- epilogue 2: acc / (wscale2 sh) + b2 assembled as an f32 tile in LDS; the residual (and the MRF mean) are added in the
  // row-contiguous 16-byte store pass shared with the f32 kernel
  __syncthreads();
  char* ys = smem;
  {
    const float inv_sh = exp2i(-eh);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int col = col0 + t * 32 + (lane & 31);
      if (col >= tt_out || t0 + col >= L) continue;
#pragma unroll
      for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n0 = (nf0 + f) * 32 + 8 * q + 4 * g;
          const f32x4 bb = *reinterpret_cast<const f32x4*>(bs + C + n0);
          const f32x4 is = *reinterpret_cast<const f32x4*>(bs + 3 * C + n0);
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = fmaf(acc[f][t][4 * q + e], is[e] * inv_sh, bb[e]);
          *reinterpret_cast<f32x4*>(ys + (size_t)col * pitch + (size_t)n0 * 4) = o;
        }
    }
  }
  __syncthreads();
  JATTS_STAMP(6);
  {
    const int vrows = min(tt_out, L - t0);
    const int64_t g0 = (seq_row0 + t0) * (int64_t)C;
    constexpr bool keep_small = C <= 64;
    const float* xg = (const float*)d.x;
    float* yg = (float*)d.y;
    if (d.add0) unit_store_pass<float, C, keep_small ? 2 : 4, true, NTHR>(d.add0, d.add1, d.out_scale, ys, pitch, vrows, xg, yg, g0);
    else unit_store_pass<float, C, keep_small ? 4 : 8, false, NTHR>(d.add0, d.add1, d.out_scale, ys, pitch, vrows, xg, yg, g0);
  }
  JATTS_STAMP(7);
  if (tracing) trace[(size_t)wg_lin * 16 + 9] = __builtin_amdgcn_s_memrealtime();
#undef JATTS_STAMP
}

template <int C, int WGCOLS, int WN, int NT, int KCG = 2, int OCC = 2>
int launch_resunit_split(const jatts_resunit_desc& d, hipStream_t s) {
  constexpr int WT = WGCOLS / (NT * 32);
  const int K = d.k_w, p2 = (K - 1) / 2, p1 = p2 * d.dil;
  const int tt_out = WGCOLS - 2 * p2;
  if (tt_out < 8) return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "resunit: kernel too wide for tile");
  if (2 * p1 > 64) return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "resunit (split): halo beyond 32 rows a side");   // the staging is one batch
  const size_t pitch = C * 4 + 16;
  const size_t rows_x = WGCOLS + 2 * p1, rows_h = WGCOLS + K - 1;
  size_t lds = (rows_x > rows_h ? rows_x : rows_h) * pitch;
  const unsigned bias_off = (unsigned)lds;
  lds += 4 * C * sizeof(float) + 64;                           // b1 | b2 | 1/ws1 | 1/ws2 | amax slots
  if (lds > 160 * 1024) return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "resunit: tile exceeds 160 KiB LDS");
  const int64_t maxL = (int64_t)d.rg.max_len * d.rg.len_mul;
  dim3 grid((unsigned)((maxL + tt_out - 1) / tt_out), (unsigned)d.rg.n_seq);
  if (const int64_t n1 = ragged_tiles_1d(d.rg, tt_out)) grid = dim3((unsigned)n1);
  auto kern = resunit_split_kernel<C, WGCOLS, WN, NT, KCG, OCC>;
  // the dynamic-LDS limit ONCE per kernel (function-local static of this template instantiation), not per launch: the per-launch call was seen to stall the
  // host for ~20 ms now and then (profiles/r06_notes.md section 8)
  static const hipError_t lds_attr = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (lds_attr != hipSuccess) return jatts_set_error(lds_attr, __FILE__, __LINE__);
  hipLaunchKernelGGL(kern, grid, dim3(WN * WT * 64), lds, s, d, jatts_g_trace, jatts_g_trace_cap, bias_off);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

}  // namespace
