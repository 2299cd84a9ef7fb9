// Fused HiFi-GAN ResBlock (all dilation units of a block in ONE launch) in the f32-equivalent emulated arithmetic (JATTS_F32E /
// JATTS_F32E6, round 5): resblock_impl.h's fixed-window scheme -- every conv of every unit computes the same WGCOLS columns, so a lane
// owns the same elements in all accumulators and the residual stream stays in f32 REGISTERS (nothing is rounded between units) -- on
// resunit_emul_impl.h's operands: the LDS tile holds three bf16 planes (6 bytes per element), every conv is seven (six) bf16 MFMAs per
// product into one f32 accumulator.  No scales, so no block maxima: the tile hand-offs between convs are plain LDS barriers.
//
// For the HBM-bound small-channel blocks (C = 32, k = 3 / 7; C = 64, k = 3): x is read once and y written once per ResBlock instead of
// once per unit.
#pragma once
#include "resblock_impl.h"
#include "resunit_emul_impl.h"

namespace {

template <typename T, int C, int WGCOLS, int WN, int NT, int KCG, int OCC>
__global__ __launch_bounds__(WN*(WGCOLS / (NT * 32)) * 64, OCC) void resblock_emul_kernel(jatts_resblock_desc d, unsigned bias_off) {
  typedef typename Elem<T>::vec8 V8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int WT = WGCOLS / (NT * 32);
  constexpr int NF = C / (WN * 32);
  constexpr int KC16 = C / 16, NFR = C / 32;
  constexpr int pitch = C * 6 + 16;
  constexpr int NTHR = WN * WT * 64;
  constexpr int UPR = C / 8;
  static_assert(WT * NT * 32 == WGCOLS && NF * WN * 32 == C, "tile shape");

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

  WStream<T, NF, KCG> ws;
  ws.prefetch((const T*)d.w1[0], NFR, nf0, lane);
  // ---- the RAW f32 x tile (zeros outside the sequence) -> registers (one batch of 16-byte loads); the window rows also go to LDS as f32 so
  // that the residual fragments can be read in accumulator layout; then every unit becomes three bf16 planes of lrelu(x)
  const float* xg = (const float*)d.x;
  constexpr int MAXU = ((WGCOLS + 64) * UPR + NTHR - 1) / NTHR;      // margins up to 32 rows a side (the launcher refuses more)
  f32x8 xv[MAXU];
  {
    const int total = rows * UPR, pos0 = t0 - H - M;
#pragma unroll
    for (int j = 0; j < MAXU; ++j) {
      const int u = threadIdx.x + j * NTHR;
      const int r = u / UPR, cu = u - r * UPR;
      const int pos = pos0 + r;
      if (u < total && pos >= 0 && pos < L) xv[j] = Vec8IO<float>::ldg(xg + (seq_row0 + pos) * (int64_t)C + cu * 8);
      else xv[j] = f32x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
#pragma unroll
    for (int j = 0; j < MAXU; ++j) {
      const int u = threadIdx.x + j * NTHR;
      if (u < total) Vec8IO<float>::sts(smem + (size_t)(u / UPR) * pitch + (size_t)(u % UPR) * 32, xv[j]);
    }
  }
  __syncthreads();
  f32x4 resid[NF][NT][4];   // x of this lane's (column, 4-channel quad) elements, C-fragment layout, f32
  bool keep[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = col0 + t * 32 + (lane & 31);
    const int pos = t0 - H + col;
    keep[t] = pos >= 0 && pos < L;   // every conv zero-pads its own input: x and h are 0 outside the sequence
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        resid[f][t][q] = *reinterpret_cast<const f32x4*>(smem + (size_t)(M + col) * pitch + (size_t)((nf0 + f) * 32 + 8 * q + 4 * g) * 4);
  }
  __syncthreads();      // every residual read is done: the f32 rows may be overwritten (a unit's 48 bytes overlap its neighbours' f32 data)
  {
    const int total = rows * UPR;
#pragma unroll
    for (int j = 0; j < MAXU; ++j) {
      const int u = threadIdx.x + j * NTHR;
      if (u >= total) continue;
      lrelu8(xv[j], d.slope);
      V8 o;
      bf3_split8(xv[j], o);
      Vec8IO<T>::sts(smem + (size_t)(u / UPR) * pitch + (size_t)(u % UPR) * 48, o);
    }
  }
  __syncthreads();

  typename Acc32<T>::type acc[NF][NT];
  // this lane's accumulator-layout values (final f32) -> three bf16 planes in the window rows
  auto put_planes = [&]() {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int col = col0 + t * 32 + (lane & 31);
#pragma unroll
      for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc_val(acc[f][t])[4 * q + e];
          bf16x4 q0, q1, q2;
          bf3_split4(v, q0, q1, q2);
          char* p = smem + (size_t)(M + col) * pitch + (size_t)((nf0 + f) * 4 + q) * 48 + 8 * g;
          *reinterpret_cast<bf16x4*>(p) = q0;
          *reinterpret_cast<bf16x4*>(p + 16) = q1;
          *reinterpret_cast<bf16x4*>(p + 32) = q2;
        }
    }
  };
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

  // Columns whose inputs lay outside what the chain has computed so far hold meaningless values (resblock_impl.h: "each unit's edge columns
  // read stale data and are never used by a valid column"); they are written as ZERO, as in the split kernel.  hv = invalid columns per side.
  int hv = 0;
  auto col_ok = [&](int t) {
    const int col = col0 + t * 32 + (lane & 31);
    return keep[t] && col >= hv && col < WGCOLS - hv;
  };
  for (int u = 0; u < NU; ++u) {
    const int dil = d.dil[u], p1 = p2 * dil;
    const float* bu = bs + (size_t)u * 2 * C;
    if (u > 0) hv += p1;          // (the first conv reads genuine margin rows: every window column is valid)
    // conv_k,dil over lrelu(x): window column c reads tile rows (M - p1) + c + tap * dil
    bias_acc(bu);
    conv_full_ws<T, NF, NT, KC16, KCG>(acc, ws, (const T*)d.w1[u], (const T*)d.w2[u], K, dil, smem + (size_t)(M - p1) * pitch, pitch, col0, lane);
    finish_acc();
    // h = lrelu(acc), 0 outside the sequence
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const bool ok = col_ok(t);
#pragma unroll
      for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float a = ok ? acc_val(acc[f][t])[r] : 0.f;
          acc_val(acc[f][t])[r] = fmaxf(a, a * d.slope);
        }
    }
    lds_barrier();                // every wave is done reading lrelu(x)
    put_planes();
    hv += p2;
    lds_barrier();
    // conv_k,1 over h
    const bool last = u == NU - 1;
    bias_acc(bu + C);
    conv_full_ws<T, NF, NT, KC16, KCG>(acc, ws, (const T*)d.w2[u], last ? nullptr : (const T*)d.w1[u + 1], K, 1,
                                       smem + (size_t)(M - p2) * pitch, pitch, col0, lane);
    finish_acc();
    // x' = x + acc (kept in f32 registers); next operand: lrelu(x'); after the last unit: x' as f32
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const bool ok = col_ok(t);
#pragma unroll
      for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 xn;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            xn[e] = ok ? acc_val(acc[f][t])[4 * q + e] + resid[f][t][q][e] : 0.f;
            acc_val(acc[f][t])[4 * q + e] = last ? xn[e] : fmaxf(xn[e], xn[e] * d.slope);
          }
          resid[f][t][q] = xn;
        }
    }
    lds_barrier();                // every wave is done reading h
    if (!last) put_planes();
    else {                        // the window becomes the f32 result tile
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int col = col0 + t * 32 + (lane & 31);
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *reinterpret_cast<f32x4*>(smem + (size_t)(M + col) * pitch + (size_t)((nf0 + f) * 32 + 8 * q + 4 * g) * 4) =
                f32x4{acc_val(acc[f][t])[4 * q], acc_val(acc[f][t])[4 * q + 1], acc_val(acc[f][t])[4 * q + 2], acc_val(acc[f][t])[4 * q + 3]};
      }
    }
    lds_barrier();
  }

  // ---- coalesced output pass: the centre tt_out columns of the final f32 x tile (+ the fused MRF mean, as the unit kernel)
  {
    const int vrows = min(tt_out, L - t0);
    const int64_t g0 = (seq_row0 + t0) * (int64_t)C;
    const char* ys = smem + (size_t)(M + H) * pitch;
    float* yg = (float*)d.y;
    constexpr bool keep_small = C <= 64;
    if (d.add0) unit_store_pass<float, C, keep_small ? 2 : 4, true, NTHR, false>(d.add0, d.add1, d.out_scale, ys, pitch, vrows, xg, yg, g0);
    else unit_store_pass<float, C, keep_small ? 4 : 8, false, NTHR, false>(d.add0, d.add1, d.out_scale, ys, pitch, vrows, xg, yg, g0);
  }
}

template <typename T, int C, int WGCOLS, int WN, int NT, int KCG = 2, int OCC = 2>
int launch_resblock_emul(const jatts_resblock_desc& d, hipStream_t s) {
  constexpr int WT = WGCOLS / (NT * 32);
  const int p2 = (d.k_w - 1) / 2;
  int H = 0, M = 0;
  for (int u = 0; u < d.n_units; ++u) {
    H += p2 * (d.dil[u] + 1);
    M = M > p2 * d.dil[u] ? M : p2 * d.dil[u];
  }
  const int tt_out = WGCOLS - 2 * H;
  if (tt_out < 32) return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "resblock (emulated): receptive field too wide for the tile");
  if (2 * M > 64) return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "resblock (emulated): margin beyond 32 rows a side");
  const size_t pitch = C * 6 + 16;
  size_t lds = (size_t)(WGCOLS + 2 * M) * pitch;
  const unsigned bias_off = (unsigned)lds;
  lds += (size_t)d.n_units * 2 * C * sizeof(float);
  if (lds > 160 * 1024) return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "resblock (emulated): tile exceeds 160 KiB LDS");
  const int64_t maxL = (int64_t)d.rg.max_len * d.rg.len_mul;
  dim3 grid((unsigned)((maxL + tt_out - 1) / tt_out), (unsigned)d.rg.n_seq);
  if (const int64_t n1 = ragged_tiles_1d(d.rg, tt_out)) grid = dim3((unsigned)n1);
  auto kern = resblock_emul_kernel<T, C, WGCOLS, WN, NT, KCG, OCC>;
  // the dynamic-LDS limit ONCE per kernel (function-local static of this template instantiation), not per launch: the per-launch call was seen to stall the
  // host for ~20 ms now and then (profiles/r06_notes.md section 8)
  static const hipError_t lds_attr = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (lds_attr != hipSuccess) return jatts_set_error(lds_attr, __FILE__, __LINE__);
  hipLaunchKernelGGL(kern, grid, dim3(WN * WT * 64), lds, s, d, bias_off);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

}  // namespace
