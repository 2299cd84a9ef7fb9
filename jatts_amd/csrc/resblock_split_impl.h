// Fused HiFi-GAN ResBlock (all dilation units of a block in ONE launch) in the split-precision arithmetic (JATTS_F32S, round 4):
// resblock_impl.h's fixed-window scheme -- every conv of every unit computes the same WGCOLS columns, so a lane owns the same elements in
// all accumulators and the residual stream stays in REGISTERS (f32 here: nothing is rounded between units) -- on resunit_split_impl.h's
// operands: the LDS tile holds hi / lo f16 planes (4 bytes per element, the f32 tile's geometry), every operand tile is scaled by the power
// of two that puts its block maximum in [2^14, 2^15), every conv is three f16 MFMAs per product into one f32 accumulator.
//
// For the HBM-bound small-channel blocks (C = 32, k = 3 / 7; C = 64, k = 3): x is read once and y written once per ResBlock instead of once
// per unit -- the per-unit split launches of C = 32, k = 3 already sit at 3.7 TB/s of x-in + y-out.
// The block-maximum exchanges ride on barriers the chain needs anyway (the tile hand-offs between convs).
#pragma once
#include "resblock_impl.h"
#include "resunit_split_impl.h"

namespace {

template <int C, int WGCOLS, int WN, int NT, int KCG, int OCC>
__global__ __launch_bounds__(WN*(WGCOLS / (NT * 32)) * 64, OCC) void resblock_split_kernel(jatts_resblock_desc d, unsigned bias_off) {
  typedef f16s T;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int WT = WGCOLS / (NT * 32);
  constexpr int NF = C / (WN * 32);
  constexpr int KC16 = C / 16, NFR = C / 32;
  constexpr int pitch = C * 4 + 16;
  constexpr int NTHR = WN * WT * 64, NW = WN * WT;
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

  float* bs = reinterpret_cast<float*>(smem + bias_off);   // [unit][b1 | b2 | 1/ws1 | 1/ws2][C], then one amax slot per wave
  float* slots = bs + 4 * C * NU;
  for (int u = threadIdx.x; u < 4 * C * NU; u += NTHR) {
    const int un = u / (4 * C), r = u - un * 4 * C;
    bs[u] = r < C ? d.b1[un][r] : (r < 2 * C ? d.b2[un][r - C] : (r < 3 * C ? d.ws1[un][r - 2 * C] : d.ws2[un][r - 3 * C]));
  }

  WStream<T, NF, KCG> ws;
  ws.prefetch((const T*)d.w1[0], NFR, nf0, lane);
  // ---- stage the RAW f32 x tile (zeros outside the sequence); the residual fragments are read from it before it becomes the first operand
  const float* xg = (const float*)d.x;
  {
    constexpr int UBX = ((WGCOLS + 64) * UPR + NTHR - 1) / NTHR;
    stage_unit<float, (UBX < 8 ? 8 : (UBX < 24 ? UBX : 24)), NTHR>(smem, pitch, rows, UPR, t0 - H - M, L, seq_row0, xg, C, false, d.slope);
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
  __syncthreads();
  // tile <- split(lrelu(tile) * 2^ex), in place: a thread owns whole 8-element units (32 bytes of f32 in, 16 + 16 bytes of hi | lo out)
  int ex;
  {
    float amax = 0.f;
    for (int u = threadIdx.x; u < rows * UPR; u += NTHR) {
      f32x8 v = Vec8IO<float>::lds(smem + (size_t)(u / UPR) * pitch + (size_t)(u % UPR) * 32);
      lrelu8(v, d.slope);
#pragma unroll
      for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(v[e]));
    }
    ex = split_exp(block_amax(amax, slots, wave, lane, NW));
    const float sx = exp2i(ex);
    for (int u = threadIdx.x; u < rows * UPR; u += NTHR) {
      char* p = smem + (size_t)(u / UPR) * pitch + (size_t)(u % UPR) * 32;
      f32x8 v = Vec8IO<float>::lds(p);
      lrelu8(v, d.slope);
      f16sx8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float sv = v[e] * sx;
        o.hi[e] = (f16)sv;
        o.lo[e] = (f16)(sv - (float)o.hi[e]);
      }
      Vec8IO<T>::sts(p, o);
    }
  }
  __syncthreads();

  f32x16 acc[NF][NT];
  // write this lane's accumulator-layout values v (already final f32) as a split operand at scale 2^e into the window rows
  auto put_split = [&](int e_scale) {
    const float sc = exp2i(e_scale);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int col = col0 + t * 32 + (lane & 31);
#pragma unroll
      for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[f][t][4 * q + e] * sc;
          f16x4 hi, lo;
          split4(v, hi, lo);
          char* p = smem + (size_t)(M + col) * pitch + (size_t)((nf0 + f) * 4 + q) * 32 + 8 * g;
          *reinterpret_cast<f16x4*>(p) = hi;
          *reinterpret_cast<f16x4*>(p + 16) = lo;
        }
    }
  };

  // Columns whose inputs lay outside what the chain has computed so far hold meaningless values (resblock_impl.h: "each unit's edge columns
  // read stale data and are never used by a valid column").  Here they are forced to ZERO: they must not enter the block maxima that set
  // the operand scales (stale margin rows still carry the first operand's scale).  hv = invalid columns per side after the current conv.
  int hv = 0;
  auto col_ok = [&](int t) {
    const int col = col0 + t * 32 + (lane & 31);
    return keep[t] && col >= hv && col < WGCOLS - hv;
  };
  for (int u = 0; u < NU; ++u) {
    const int dil = d.dil[u], p1 = p2 * dil;
    const float* bu = bs + (size_t)u * 4 * C;
    if (u > 0) hv += p1;          // (the first conv reads genuine margin rows: every window column is valid)
    // conv_k,dil over lrelu(x): window column c reads tile rows (M - p1) + c + tap * dil
    zero_acc<NF, NT>(acc);
    conv_full_ws<T, NF, NT, KC16, KCG>(acc, ws, (const T*)d.w1[u], (const T*)d.w2[u], K, dil, smem + (size_t)(M - p1) * pitch, pitch, col0, lane);
    // h = lrelu(acc / (ws1 2^ex) + b1), 0 outside the sequence; its block maximum (the exchange's barrier = "every wave is done reading lrelu(x)")
    int eh;
    {
      const float inv = exp2i(-ex);
      float amax = 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const bool ok = col_ok(t);
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int n0 = (nf0 + f) * 32 + 8 * q + 4 * g;
            const f32x4 bb = *reinterpret_cast<const f32x4*>(bu + n0);
            const f32x4 is = *reinterpret_cast<const f32x4*>(bu + 2 * C + n0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float a = ok ? fmaf(acc[f][t][4 * q + e], is[e] * inv, bb[e]) : 0.f;
              a = fmaxf(a, a * d.slope);
              acc[f][t][4 * q + e] = a;
              amax = fmaxf(amax, fabsf(a));
            }
          }
      }
      eh = split_exp(block_amax(amax, slots, wave, lane, NW));
      put_split(eh);
    }
    hv += p2;
    lds_barrier();
    // conv_k,1 over h
    zero_acc<NF, NT>(acc);
    const bool last = u == NU - 1;
    conv_full_ws<T, NF, NT, KC16, KCG>(acc, ws, (const T*)d.w2[u], last ? nullptr : (const T*)d.w1[u + 1], K, 1,
                                       smem + (size_t)(M - p2) * pitch, pitch, col0, lane);
    // x' = x + acc / (ws2 2^eh) + b2 (kept in f32 registers); next operand: lrelu(x') split at its block maximum; after the last unit: x' as f32
    {
      const float inv = exp2i(-eh);
      float amax = 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const bool ok = col_ok(t);
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int n0 = (nf0 + f) * 32 + 8 * q + 4 * g;
            const f32x4 bb = *reinterpret_cast<const f32x4*>(bu + C + n0);
            const f32x4 is = *reinterpret_cast<const f32x4*>(bu + 3 * C + n0);
            f32x4 xn;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              xn[e] = ok ? fmaf(acc[f][t][4 * q + e], is[e] * inv, bb[e]) + resid[f][t][q][e] : 0.f;
              const float a = last ? xn[e] : fmaxf(xn[e], xn[e] * d.slope);
              acc[f][t][4 * q + e] = a;
              amax = fmaxf(amax, fabsf(a));
            }
            resid[f][t][q] = xn;
          }
      }
      if (!last) {
        ex = split_exp(block_amax(amax, slots, wave, lane, NW));     // (its barrier: every wave is done reading h)
        put_split(ex);
      } else {
        lds_barrier();                                                // every wave is done reading h: the window becomes the f32 result tile
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int col = col0 + t * 32 + (lane & 31);
#pragma unroll
          for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int q = 0; q < 4; ++q)
              *reinterpret_cast<f32x4*>(smem + (size_t)(M + col) * pitch + (size_t)((nf0 + f) * 32 + 8 * q + 4 * g) * 4) =
                  f32x4{acc[f][t][4 * q], acc[f][t][4 * q + 1], acc[f][t][4 * q + 2], acc[f][t][4 * q + 3]};
        }
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

template <int C, int WGCOLS, int WN, int NT, int KCG = 2, int OCC = 2>
int launch_resblock_split(const jatts_resblock_desc& d, hipStream_t s) {
  constexpr int WT = WGCOLS / (NT * 32);
  const int p2 = (d.k_w - 1) / 2;
  int H = 0, M = 0;
  for (int u = 0; u < d.n_units; ++u) {
    H += p2 * (d.dil[u] + 1);
    M = M > p2 * d.dil[u] ? M : p2 * d.dil[u];
  }
  const int tt_out = WGCOLS - 2 * H;
  if (tt_out < 32) return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "resblock (split): receptive field too wide for the tile");
  if (2 * M > 64) return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "resblock (split): margin beyond 32 rows a side");
  const size_t pitch = C * 4 + 16;
  size_t lds = (size_t)(WGCOLS + 2 * M) * pitch;
  const unsigned bias_off = (unsigned)lds;
  lds += (size_t)d.n_units * 4 * C * sizeof(float) + 64;
  if (lds > 160 * 1024) return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "resblock (split): tile exceeds 160 KiB LDS");
  const int64_t maxL = (int64_t)d.rg.max_len * d.rg.len_mul;
  dim3 grid((unsigned)((maxL + tt_out - 1) / tt_out), (unsigned)d.rg.n_seq);
  if (const int64_t n1 = ragged_tiles_1d(d.rg, tt_out)) grid = dim3((unsigned)n1);
  auto kern = resblock_split_kernel<C, WGCOLS, WN, NT, KCG, OCC>;
  // the dynamic-LDS limit ONCE per kernel (function-local static of this template instantiation), not per launch: the per-launch call was seen to stall the
  // host for ~20 ms now and then (profiles/r06_notes.md section 8)
  static const hipError_t lds_attr = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (lds_attr != hipSuccess) return jatts_set_error(lds_attr, __FILE__, __LINE__);
  hipLaunchKernelGGL(kern, grid, dim3(WN * WT * 64), lds, s, d, bias_off);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

}  // namespace
