// Fused HiFi-GAN dilation unit  y = x + conv_1(lrelu(conv_d(lrelu(x))))  with f32 activations in HBM and
// f32-EQUIVALENT EMULATED MFMA operands (JATTS_F32E / JATTS_F32E6, round 5): every operand value is carried exactly as three
// bfloat16 terms and a product keeps the seven (six) largest of its nine partial products (common.h: bf3p<NP>, bf3_split, mma32) --
// seven (six) v_mfma_f32_32x32x16_bf16 with f32 accumulate = 7/16 (6/16) of the pipe cycles of the exact-f32 chain.
//
// Unlike the split-f16 path (resunit_split_impl.h) there is nothing to scale: bf16 has f32's exponent range, so there are no
// block maxima, no scale barriers and no element whose relative precision depends on its neighbours.  The dropped terms are
// <= 2^-24 (seven products) / 2^-23 (six) of every product for EVERY finite input with |v| >= 2^-110, accumulation is f32.
// The LDS tile holds 6 bytes per element (three planes of 16 B per 8 channels), the result tile and the residual / MRF store
// pass are the f32 kernel's.
#pragma once
#include "resunit_impl.h"

namespace {

// 4 values -> the three bf16 planes of a channel quad
__device__ __forceinline__ void bf3_split4(const float (&v)[4], bf16x4& p0, bf16x4& p1, bf16x4& p2) {
#pragma unroll
  for (int e = 0; e < 4; e += 2) {
    bf16x2 a, b, c;
    bf3_split2(v[e], v[e + 1], a, b, c);
    p0[e] = a[0]; p0[e + 1] = a[1]; p1[e] = b[0]; p1[e + 1] = b[1]; p2[e] = c[0]; p2[e + 1] = c[1];
  }
}

// conv_full_ws over ONE HALF of the input channels (KSPLIT kernels: the x tile holds C / 2 channels at a time).  KC16 = the FULL steps per tap of the
// packed weights; the tile holds KC16 / 2 of them per row.  On entry the ring holds this half's first group; on exit the first group at `w_after`
// (the other half's, or the next conv's).
template <typename T, int NF, int NT, int KC16, int KCG>
__device__ __forceinline__ void conv_khalf_ws(typename Acc32<T>::type (&acc)[NF][NT], WStream<T, NF, KCG>& ws, const T* __restrict__ w, const T* __restrict__ w_after,
                                              int half, int k_w, int dil, const char* act, int pitch_half, int col0, int lane) {
  typedef typename Elem<T>::vec8 V8;
  constexpr int GPT = KC16 / KCG, GPH = GPT / 2;
  static_assert(GPH * 2 * KCG == KC16, "group size must divide half the steps per tap");
  const T* wl = w + (size_t)lane * 8;
  const size_t gstride = (size_t)KCG * ws.wf.stride;
  const char* bbase = act + (size_t)(col0 + (lane & 31)) * pitch_half + (size_t)(8 * (lane >> 5)) * sizeof(T);
  V8 bb[2][NT];
  fetch_b<T, NT>(bb[0], bbase, pitch_half);
  __builtin_amdgcn_sched_barrier(0);
  for (int tap = 0; tap < k_w; ++tap) {
#pragma unroll
    for (int h = 0; h < GPH; ++h) {
      const bool wrap = h + 1 == GPH;
      const int nt = wrap ? tap + 1 : tap, nh = wrap ? 0 : h + 1;
      const T* nb = nt < k_w ? wl + (size_t)(nt * GPT + half * GPH + nh) * gstride : w_after + (size_t)lane * 8;
      const char* bcur = bbase + (size_t)(tap * dil) * pitch_half + (size_t)(h * KCG) * 16 * sizeof(T);
      const char* bnext = bbase + (size_t)(min(nt, k_w - 1) * dil) * pitch_half + (size_t)(nh * KCG) * 16 * sizeof(T);
      conv_group<T, NF, NT, KCG>(acc, ws.ring, bb, ws.wf, nb, bcur, bnext, pitch_half);
    }
  }
}

// KSPLIT: the x tile of conv1 is staged one channel half at a time (the second half's global loads are in flight under the first half's MFMAs).  For
// shapes whose full x tile does not fit LDS beside nothing else -- C = 256, k = 11, dilation 5: 114 rows x 1 552 B = 177 KB -- where the alternative was a
// 32-column window (9.2 ms against 6.8 ms for the other dilations).
// RREG (round 6, C <= 64): the residual comes from REGISTERS.  The rows of the x tile that are this workgroup's output rows are staged with the store
// pass's own unit -> thread map and kept (4-8 units of 8 floats per thread), so the store pass adds them without reading x a second time: the re-read was
// the larger half of these shapes' wasted traffic (1.5-1.9 x the algorithmic bytes: halo + re-read), and on a power-limited part fabric bytes are clock.
template <typename T, int C, int WGCOLS, int WN, int NT, int KCG, int OCC, bool KSPLIT = false, bool RREG = false>      // T = bf3 (seven partial products) or bf3f (six)
__global__ __launch_bounds__(WN*(WGCOLS / (NT * 32)) * 64, OCC) void resunit_emul_kernel(jatts_resunit_desc d, unsigned long long* trace,
                                                                                         unsigned trace_cap, unsigned bias_off) {
  typedef typename Elem<T>::vec8 V8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int WT = WGCOLS / (NT * 32);
  constexpr int NF = C / (WN * 32);
  constexpr int KC16 = C / 16, NFR = C / 32;
  constexpr int pitch = C * 6 + 16;
  constexpr int pitch_x = KSPLIT ? (C / 2) * 6 + 16 : pitch;      // x tile row: all channels, or one half
  constexpr int NTHR = WN * WT * 64;
  static_assert(WT * NT * 32 == WGCOLS && NF * WN * 32 == C, "tile shape");
  static_assert(sizeof(T) == 6, "bf3 is three packed bf16");
  static_assert(!(KSPLIT && RREG), "the residual registers go with the one-piece x tile");
  constexpr int MAXI = RREG ? (WGCOLS * (C / 8) + NTHR - 1) / NTHR : 1;      // interior units per thread
  f32x8 xk[MAXI];
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
  char* xs = smem;                  // bf3 lrelu(x) tile; h overlays it; finally the f32 y tile
  char* hs = smem;
  float* bs = reinterpret_cast<float*>(smem + bias_off);   // b1 | b2
  for (int u = threadIdx.x; u < 2 * C; u += NTHR) bs[u] = u < C ? d.b1[u] : d.b2[u - C];

  WStream<T, NF, KCG> ws;
  ws.prefetch((const T*)d.w1, NFR, nf0, lane);

  // ---- stage: lrelu(x) tile -> registers -> three bf16 planes in LDS (all loads of a batch in flight before the first is used)
  typename Acc32<T>::type acc[NF][NT];      // (seven products: a leading-product and a small-terms accumulator per fragment, common.h)
  auto bias_acc = [&](const float* bv) {
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 bb = *reinterpret_cast<const f32x4*>(bv + (nf0 + f) * 32 + 8 * q + 4 * g);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc_set(acc[f][t], 4 * q + e, bb[e]);
      }
  };
  auto finish_acc = [&]() {
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc_finish(acc[f][t]);
  };
  auto to_planes = [&](f32x8 v) {
    lrelu8(v, d.slope);
    V8 o;
    bf3_split8(v, o);
    return o;
  };
  const float* x = (const float*)d.x;
  const int pos0 = t0 - p2 - p1;
  if constexpr (!KSPLIT) {
    constexpr int UPR = C / 8;
    constexpr int UB = 8;
    const int total = rx * UPR;
    if constexpr (RREG) {
      // interior rows (the output rows: x-tile rows p1 + p2 .. + tt_out) first, unit v = thread + j NTHR <-> output row v / UPR: the map of the store pass
      const int r_in = p1 + p2, n_in = tt_out * UPR;
#pragma unroll
      for (int j = 0; j < MAXI; ++j) {
        const int v = threadIdx.x + j * NTHR;
        const int ro = v / UPR, cu = v - ro * UPR;
        const int pos = t0 + ro;
        if (v < n_in && pos < L) xk[j] = Vec8IO<float>::ldg(x + (seq_row0 + pos) * (int64_t)C + cu * 8);
        else xk[j] = f32x8{0, 0, 0, 0, 0, 0, 0, 0};
      }
      // the halo rows above and below, batched like the one-piece staging
      const int n_halo = (rx - tt_out) * UPR;
      for (int base = threadIdx.x; base < n_halo; base += NTHR * UB) {
        f32x8 v[UB];
#pragma unroll
        for (int j = 0; j < UB; ++j) {
          const int u = base + j * NTHR;
          int r = u / UPR;
          const int cu = u - r * UPR;
          if (r >= r_in) r += tt_out;
          const int pos = pos0 + r;
          if (u < n_halo && pos >= 0 && pos < L) v[j] = Vec8IO<float>::ldg(x + (seq_row0 + pos) * (int64_t)C + cu * 8);
          else v[j] = f32x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
#pragma unroll
        for (int j = 0; j < UB; ++j) {
          const int u = base + j * NTHR;
          if (u >= n_halo) continue;
          int r = u / UPR;
          const int cu = u - r * UPR;
          if (r >= r_in) r += tt_out;
          Vec8IO<T>::sts(xs + (size_t)r * pitch + (size_t)cu * 48, to_planes(v[j]));
        }
      }
#pragma unroll
      for (int j = 0; j < MAXI; ++j) {
        const int v = threadIdx.x + j * NTHR;
        if (v >= n_in) continue;
        const int ro = v / UPR, cu = v - ro * UPR;
        Vec8IO<T>::sts(xs + (size_t)(r_in + ro) * pitch + (size_t)cu * 48, to_planes(xk[j]));
      }
    } else {
    for (int base = threadIdx.x; base < total; base += NTHR * UB) {
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
        const int u = base + j * NTHR;
        if (u >= total) continue;
        const int r = u / UPR, cu = u - r * UPR;
        Vec8IO<T>::sts(xs + (size_t)r * pitch + (size_t)cu * 48, to_planes(v[j]));
      }
    }
    }
    __syncthreads();
    JATTS_STAMP(2);
    bias_acc(bs);
    conv_full_ws<T, NF, NT, KC16, KCG>(acc, ws, (const T*)d.w1, (const T*)d.w2, K, dil, xs, pitch, col0, lane);
  } else {
    constexpr int UPR = C / 16;                                            // 8-element units per row of one channel half
    constexpr int MAXU = ((WGCOLS + 64) * UPR + NTHR - 1) / NTHR;         // halos up to 32 rows a side (the launcher refuses more)
    const int total = rx * UPR;
    f32x8 xv[MAXU];
    auto load_half = [&](int half) {
#pragma unroll
      for (int j = 0; j < MAXU; ++j) {
        const int u = threadIdx.x + j * NTHR;
        const int r = u / UPR, cu = u - r * UPR;
        const int pos = pos0 + r;
        if (u < total && pos >= 0 && pos < L) xv[j] = Vec8IO<float>::ldg(x + (seq_row0 + pos) * (int64_t)C + half * (C / 2) + cu * 8);
        else xv[j] = f32x8{0, 0, 0, 0, 0, 0, 0, 0};
      }
    };
    auto store_half = [&]() {
#pragma unroll
      for (int j = 0; j < MAXU; ++j) {
        const int u = threadIdx.x + j * NTHR;
        if (u < total) Vec8IO<T>::sts(xs + (size_t)(u / UPR) * pitch_x + (size_t)(u % UPR) * 48, to_planes(xv[j]));
      }
    };
    load_half(0);
    store_half();
    load_half(1);                 // in flight under the first half's MFMAs
    __syncthreads();
    JATTS_STAMP(2);
    bias_acc(bs);
    const T* w1 = (const T*)d.w1;
    conv_khalf_ws<T, NF, NT, KC16, KCG>(acc, ws, w1, w1 + (size_t)(KC16 / KCG / 2) * KCG * ws.wf.stride, 0, K, dil, xs, pitch_x, col0, lane);
    lds_barrier();                // every wave is done reading the first half
    store_half();
    lds_barrier();
    conv_khalf_ws<T, NF, NT, KC16, KCG>(acc, ws, w1, (const T*)d.w2, 1, K, dil, xs, pitch_x, col0, lane);
  }
  finish_acc();
  JATTS_STAMP(3);

  // ---- epilogue 1: h = lrelu(acc), 0 outside the sequence (conv2's zero padding) -> three planes over the dead x tile
  lds_barrier();     // every wave is done reading x (conv2's first weights stay in flight)
  JATTS_STAMP(10);
  for (int u = threadIdx.x; u < (K - 1) * (C / 8); u += NTHR) {   // rows past the computed columns: read by discarded columns only
    const int r = WGCOLS + u / (C / 8), cu = u % (C / 8);
    V8 z;
#pragma unroll
    for (int e = 0; e < 8; ++e) z.b0[e] = z.b1[e] = z.b2[e] = (bf16)0.f;
    Vec8IO<T>::sts(hs + (size_t)r * pitch + (size_t)cu * 48, z);
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = col0 + t * 32 + (lane & 31);
    const int pos = t0 - p2 + col;
    const float keep = (pos >= 0 && pos < L) ? 1.f : 0.f;
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int cu = (nf0 + f) * 4 + q;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a = acc_val(acc[f][t])[4 * q + e] * keep;
          v[e] = fmaxf(a, a * d.slope);
        }
        bf16x4 q0, q1, q2;
        bf3_split4(v, q0, q1, q2);
        char* p = hs + (size_t)col * pitch + (size_t)cu * 48 + 8 * g;
        *reinterpret_cast<bf16x4*>(p) = q0;
        *reinterpret_cast<bf16x4*>(p + 16) = q1;
        *reinterpret_cast<bf16x4*>(p + 32) = q2;
      }
  }
  JATTS_STAMP(12);
  lds_barrier();
  JATTS_STAMP(4);

  bias_acc(bs + C);
  conv_full_ws<T, NF, NT, KC16, KCG>(acc, ws, (const T*)d.w2, nullptr, K, 1, hs, pitch, col0, lane);
  finish_acc();
  JATTS_STAMP(5);

  // ---- epilogue 2: acc (+ b2, already in) assembled as an f32 tile in LDS; the residual (and the MRF mean) are added in the
  // row-contiguous 16-byte store pass shared with the f32 kernel
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
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = acc_val(acc[f][t])[4 * q + e];
        *reinterpret_cast<f32x4*>(ys + (size_t)col * pitch + (size_t)n0 * 4) = o;
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
    if constexpr (RREG) {
      // y = (acc + b2 tile in LDS) + x FROM REGISTERS [+ MRF partners] (unit_store_pass with the staged interior units)
      constexpr int UPR = C / 8;
      const int n_out = vrows * UPR;
      const bool has_add1 = d.add0 != nullptr && d.add1 != nullptr;
      f32x8 a0[MAXI], a1[MAXI];
      if (d.add0) {
#pragma unroll
        for (int j = 0; j < MAXI; ++j) {
          const int v = threadIdx.x + j * NTHR;
          if (v < n_out) {
            a0[j] = Vec8IO<float>::ldg((const float*)d.add0 + g0 + (int64_t)v * 8);
            if (has_add1) a1[j] = Vec8IO<float>::ldg((const float*)d.add1 + g0 + (int64_t)v * 8);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < MAXI; ++j) {
        const int v = threadIdx.x + j * NTHR;
        if (v >= n_out) continue;
        const int ro = v / UPR, cu = v - ro * UPR;
        f32x8 o = Vec8IO<float>::lds(ys + (size_t)ro * pitch + (size_t)cu * 32);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = o[e] + xk[j][e];          // residual
        if (d.add0) {
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (o[e] + a0[j][e] + (has_add1 ? a1[j][e] : 0.f)) * d.out_scale;
        }
        float* dst = yg + g0 + (int64_t)v * 8;
        *reinterpret_cast<f32x4*>(dst) = f32x4{o[0], o[1], o[2], o[3]};
        *reinterpret_cast<f32x4*>(dst + 4) = f32x4{o[4], o[5], o[6], o[7]};
      }
    } else {
    if (d.add0) unit_store_pass<float, C, keep_small ? 2 : 4, true, NTHR>(d.add0, d.add1, d.out_scale, ys, pitch, vrows, xg, yg, g0);
    else unit_store_pass<float, C, keep_small ? 4 : 8, false, NTHR>(d.add0, d.add1, d.out_scale, ys, pitch, vrows, xg, yg, g0);
    }
  }
  JATTS_STAMP(7);
  if (tracing) trace[(size_t)wg_lin * 16 + 9] = __builtin_amdgcn_s_memrealtime();
#undef JATTS_STAMP
}

template <typename T, int C, int WGCOLS, int WN, int NT, int KCG = 2, int OCC = 2, bool KSPLIT = false, bool RREG = false>
int launch_resunit_emul(const jatts_resunit_desc& d, hipStream_t s) {
  constexpr int WT = WGCOLS / (NT * 32);
  const int K = d.k_w, p2 = (K - 1) / 2, p1 = p2 * d.dil;
  const int tt_out = WGCOLS - 2 * p2;
  if (tt_out < 8) return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "resunit: kernel too wide for tile");
  const size_t pitch = C * 6 + 16, pitch_x = KSPLIT ? (C / 2) * 6 + 16 : pitch;
  const size_t rows_x = WGCOLS + 2 * p1, rows_h = WGCOLS + K - 1;
  if (KSPLIT && 2 * p1 > 64) return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "resunit (emulated, channel halves): halo beyond 32 rows a side");
  size_t lds = rows_x * pitch_x > rows_h * pitch ? rows_x * pitch_x : rows_h * pitch;
  const unsigned bias_off = (unsigned)lds;
  lds += 2 * C * sizeof(float);                                // b1 | b2
  if (lds > 160 * 1024) return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "resunit: tile exceeds 160 KiB LDS");
  const int64_t maxL = (int64_t)d.rg.max_len * d.rg.len_mul;
  dim3 grid((unsigned)((maxL + tt_out - 1) / tt_out), (unsigned)d.rg.n_seq);
  if (const int64_t n1 = ragged_tiles_1d(d.rg, tt_out)) grid = dim3((unsigned)n1);
  auto kern = resunit_emul_kernel<T, C, WGCOLS, WN, NT, KCG, OCC, KSPLIT, RREG>;
  // the dynamic-LDS limit ONCE per kernel (function-local static of this template instantiation), not per launch: the per-launch call was seen to stall the
  // host for ~20 ms now and then (profiles/r06_notes.md section 8)
  static const hipError_t lds_attr = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (lds_attr != hipSuccess) return jatts_set_error(lds_attr, __FILE__, __LINE__);
  hipLaunchKernelGGL(kern, grid, dim3(WN * WT * 64), lds, s, d, jatts_g_trace, jatts_g_trace_cap, bias_off);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

}  // namespace
