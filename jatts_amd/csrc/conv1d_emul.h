// Generic Conv1d / Linear / polyphase ConvTranspose1d with f32 activations in HBM and f32-EQUIVALENT emulated MFMA operands
// (JATTS_F32E / JATTS_F32E6, round 5): every operand value exactly as three bf16 terms, seven / six partial products per product
// (common.h: bf3p<NP>, mma32).
//
// Same implicit GEMM as conv1d_impl.h (weights = A operand in fragment order, activation chunks double-buffered in LDS).  There
// is no scale anywhere (bf16 has f32's exponent range), so -- unlike conv1d_split.h -- the result of a row does not depend on any
// tile geometry and the accumulators start at the bias like the exact-f32 kernel's.  The chunk is 32 channels: its LDS rows hold
// 6 bytes per element, and two double-buffered 160-row tiles (66 KB) keep two workgroups on a CU, one committing / storing while
// the other feeds the matrix pipe; a chunk is k_w x 2 K-steps x 7 (6) NF NT MFMAs, 56 (48) MFMAs per wave between barriers at k = 1.
#pragma once
#include "conv1d_impl.h"
#ifndef JATTS_CEMUL_DIAG
#define JATTS_CEMUL_DIAG 0   // timing probes only (wrong results): 1 = no split arithmetic in the commit, 2 = no activation loads in the chunk loop, 4 = no barriers in it;
                             // 16 x 16 x 32 kernel only: 8 = no weight refills, 16 = no B (LDS) refills, 32 = no epilogue, 64 = no commit at all
#endif

namespace {

// commit of the emulated pipeline: combine the staged f32 inputs (sum, in_scale, LeakyReLU), three bf16 planes, LDS.  PLAIN (one input, no scale, no
// activation -- every conv but the HiFi-GAN upsampling ones) skips the combine: 7.5 instead of 12.5 VALU instructions per element (round 6: the commit is
// 12 - 26 % of a workgroup's life, profiles/r06_conv16_trace.txt).
template <typename T, int MAXU, int NIN, int UPR, int NTHR, bool PLAIN>
__device__ __forceinline__ void emul_commit_as(StageRegs<float, MAXU, NIN>& sr, char* lds, int pitch, int rows, int n_in, float in_scale,
                                               int pre_act, float slope) {
  const int total = rows * UPR;
#pragma unroll
  for (int j = 0; j < MAXU; ++j) {
    const int u = threadIdx.x + j * NTHR;
    if (u >= total) continue;
    const int r = u / UPR, cu = u % UPR;
    typename Elem<T>::vec8 o;
    f32x8 tv;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = sr.v[0][j][e];
      if (!PLAIN) {
        if (NIN > 1 && n_in > 1) t += sr.v[NIN > 1 ? 1 : 0][j][e];
        if (NIN > 2 && n_in > 2) t += sr.v[NIN > 2 ? 2 : 0][j][e];
        t *= in_scale;
        if (pre_act == JATTS_PRE_LRELU) t = fmaxf(t, t * slope);     // 0 <= slope <= 1
      }
      tv[e] = t;
    }
#if JATTS_CEMUL_DIAG & 1
#pragma unroll
    for (int e = 0; e < 8; ++e) o.b0[e] = o.b1[e] = o.b2[e] = __builtin_bit_cast(bf16, (unsigned short)(__float_as_uint(tv[e]) >> 16));
#else
    bf3_split8(tv, o);
#endif
    Vec8IO<T>::sts(lds + (size_t)r * pitch + (size_t)cu * 48, o);
  }
}
template <typename T, int MAXU, int NIN, int UPR, int NTHR>
__device__ __forceinline__ void emul_commit(StageRegs<float, MAXU, NIN>& sr, char* lds, int pitch, int rows, int n_in, float in_scale,
                                            int pre_act, float slope) {
  if (n_in == 1 && in_scale == 1.f && pre_act == JATTS_PRE_NONE) emul_commit_as<T, MAXU, NIN, UPR, NTHR, true>(sr, lds, pitch, rows, n_in, in_scale, pre_act, slope);
  else emul_commit_as<T, MAXU, NIN, UPR, NTHR, false>(sr, lds, pitch, rows, n_in, in_scale, pre_act, slope);
}

// HALO: rows beyond the time tile the staging registers must cover; RD: weight ring depth in K-steps (divides KCHT / 16).
template <typename T, int NF, int NT, int WN, int WT, int NIN, int KCHT, int OCC, int HALO = 32, int RD = 2>      // T = bf3 (seven products) / bf3f (six)
__global__ __launch_bounds__(WN* WT * 64, OCC) void conv1d_emul_kernel(jatts_conv_desc d, int f32_tile, XcdOrder xo) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BT = WT * NT * 32, NTHR = WN * WT * 64;
  int bx, b, bz;
  if (!xo.decode(blockIdx.x, bx, b, bz, d.rg, BT)) return;
  const int row_b = d.rg.cu_rows[b];
  const int L = (d.rg.cu_rows[b + 1] - row_b) * d.rg.len_mul;
  const int t0 = bx * BT;
  if (t0 >= L) return;
  const int64_t seq_row0 = (int64_t)row_b * d.rg.len_mul;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wn = wave / WT, wt = wave % WT;
  constexpr int pitch = KCHT * 6 + 16;
  const int rows = BT + (d.k_w - 1) * d.dil;
  const int KC16 = d.c_in >> 4;
  const int n_pad = (d.n_out + 31) & ~31;
  const int NFR = n_pad >> 5;
  const int nf0 = (bz * WN + wn) * NF;
  const int col0 = wt * NT * 32;
  conv_second_output(d, bz * WN * NF * 32);

  const float* xin[3] = {(const float*)d.x[0], (const float*)d.x[1], (const float*)d.x[2]};
  const bool reflect = d.pad_mode == JATTS_PAD_REFLECT;
  typename Acc32<T>::type accx[NF][NT];     // (seven products: a leading-product and a small-terms accumulator per fragment, common.h)
#pragma unroll
  for (int f = 0; f < NF; ++f)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc_set(accx[f][t], r, 0.f);
  if (d.bias) {   // accumulators start at the bias (as conv1d_kernel)
    const int gq = lane >> 5;
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n0 = (nf0 + f) * 32 + 8 * q + 4 * gq;
        f32x4 bq = {0.f, 0.f, 0.f, 0.f};
        if (n0 + 3 < d.n_out) bq = *reinterpret_cast<const f32x4*>(d.bias + n0);
        else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n0 + e < d.n_out) bq[e] = d.bias[n0 + e];
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc_set(accx[f][t], 4 * q + e, bq[e]);
      }
  }

  constexpr int UPRC = KCHT / 8;
  constexpr int MAXU = ((BT + HALO) * UPRC + NTHR - 1) / NTHR;   // halo <= HALO rows (the launcher refuses more)
  static_assert((KCHT / 16) % RD == 0, "ring depth must divide the steps per chunk and tap");
  WRing<T, NF, RD> ring;
  const int n_chunks = d.c_in / KCHT;
  ring.init((const T*)d.w, KC16, NFR, nf0, d.k_w, KCHT / 16, n_chunks, lane);
  const size_t buf_bytes = (size_t)rows * pitch;
  StageRegs<float, MAXU, NIN> sr;
  stage_issue<float, MAXU, NIN, UPRC, NTHR>(sr, rows, t0 - d.pad, L, seq_row0, xin, d.n_in, d.ldx, 0, reflect);
  emul_commit<T, MAXU, NIN, UPRC, NTHR>(sr, smem, pitch, rows, d.n_in, d.in_scale, d.pre_act, d.pre_slope);
  __syncthreads();
  for (int ci = 0; ci < n_chunks; ++ci) {
    const bool more = ci + 1 < n_chunks;
    if (more && !(JATTS_CEMUL_DIAG & 2)) stage_issue<float, MAXU, NIN, UPRC, NTHR>(sr, rows, t0 - d.pad, L, seq_row0, xin, d.n_in, d.ldx, (ci + 1) * KCHT, reflect);
    conv_stage<T, NF, NT, RD>(accx, ring, KCHT / 16, d.k_w, d.dil, smem + (size_t)(ci & 1) * buf_bytes, pitch, col0, lane);
    if (more) emul_commit<T, MAXU, NIN, UPRC, NTHR>(sr, smem + (size_t)((ci + 1) & 1) * buf_bytes, pitch, rows, d.n_in, d.in_scale, d.pre_act, d.pre_slope);
    if (!(JATTS_CEMUL_DIAG & 4)) __syncthreads();
  }
  if (JATTS_CEMUL_DIAG & 4) __syncthreads();

  // close the accumulators (seven products: one correctly rounded add per element): from here on the ordinary f32 epilogues of conv1d_impl.h
  f32x16 acc[NF][NT];
#pragma unroll
  for (int f = 0; f < NF; ++f)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      acc_finish(accx[f][t]);
      acc[f][t] = acc_val(accx[f][t]);
    }
  if (d.act == JATTS_ACT_SNAKEBETA) snake_acc<NF, NT>(acc, d.act_a, d.act_b, nf0, d.n_out, lane);
  {
    constexpr int BN = WN * NF * 32;
    const int n_base = bz * BN;
    const bool rowmajor = !d.y_transposed && (d.n_out & 7) == 0 && (reinterpret_cast<uintptr_t>(d.y) & 15) == 0;
    if (rowmajor && f32_tile && (d.ldy & 3) == 0 && (!d.resid || ((d.ldr & 3) == 0 && (reinterpret_cast<uintptr_t>(d.resid) & 15) == 0))) {
      switch (d.act) {
        case JATTS_ACT_RELU: conv_epilogue_lds<float, float, JATTS_ACT_RELU, NF, NT, BN>(d, acc, smem, t0, col0, wn * NF, n_base, lane, L, seq_row0, BT); break;
        case JATTS_ACT_TANH: conv_epilogue_lds<float, float, JATTS_ACT_TANH, NF, NT, BN>(d, acc, smem, t0, col0, wn * NF, n_base, lane, L, seq_row0, BT); break;
        case JATTS_ACT_SWISH: conv_epilogue_lds<float, float, JATTS_ACT_SWISH, NF, NT, BN>(d, acc, smem, t0, col0, wn * NF, n_base, lane, L, seq_row0, BT); break;
        case JATTS_ACT_MISH: conv_epilogue_lds<float, float, JATTS_ACT_MISH, NF, NT, BN>(d, acc, smem, t0, col0, wn * NF, n_base, lane, L, seq_row0, BT); break;
        default: conv_epilogue_lds<float, float, JATTS_ACT_NONE, NF, NT, BN>(d, acc, smem, t0, col0, wn * NF, n_base, lane, L, seq_row0, BT); break;
      }
      return;
    }
  }
  switch (d.act) {
    case JATTS_ACT_RELU: conv_epilogue<float, JATTS_ACT_RELU, NF, NT>(d, acc, t0, col0, nf0, lane, L, seq_row0, b); break;
    case JATTS_ACT_TANH: conv_epilogue<float, JATTS_ACT_TANH, NF, NT>(d, acc, t0, col0, nf0, lane, L, seq_row0, b); break;
    case JATTS_ACT_SWISH: conv_epilogue<float, JATTS_ACT_SWISH, NF, NT>(d, acc, t0, col0, nf0, lane, L, seq_row0, b); break;
    case JATTS_ACT_MISH: conv_epilogue<float, JATTS_ACT_MISH, NF, NT>(d, acc, t0, col0, nf0, lane, L, seq_row0, b); break;
    default: conv_epilogue<float, JATTS_ACT_NONE, NF, NT>(d, acc, t0, col0, nf0, lane, L, seq_row0, b); break;
  }
}

template <typename T, int NF, int NT, int WN, int WT, int NIN, int KCHT, int OCC, int HALO = 32, int RD = 2>
int launch_conv_emul(const jatts_conv_desc& d, hipStream_t s) {
  constexpr int BT = WT * NT * 32, BN = WN * NF * 32;
  if ((d.k_w - 1) * d.dil > HALO) return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "conv1d (emulated): halo beyond the staging registers");
  if (d.c_in % KCHT) return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "conv1d (emulated): c_in must be a multiple of the chunk width");
  const int64_t maxL = (int64_t)d.rg.max_len * d.rg.len_mul;
  dim3 grid((unsigned)((maxL + BT - 1) / BT), (unsigned)d.rg.n_seq, (unsigned)((d.n_out + BN - 1) / BN));
  const size_t rows = (size_t)BT + (size_t)(d.k_w - 1) * d.dil;
  size_t lds = 2 * rows * (KCHT * 6 + 16);
  int f32_tile = 0;
  if (!d.y_transposed && (size_t)BT * (BN * 4 + 16) <= 159 * 1024) {     // the coalesced f32 output tile reuses the staging buffers
    f32_tile = 1;
    if (lds < (size_t)BT * (BN * 4 + 16)) lds = (size_t)BT * (BN * 4 + 16);
  }
  if (lds > 160 * 1024) return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "conv1d (emulated): tile exceeds 160 KiB LDS");
  auto kern = conv1d_emul_kernel<T, NF, NT, WN, WT, NIN, KCHT, OCC, HALO, RD>;
  // the dynamic-LDS limit ONCE per kernel (function-local static of this template instantiation), not per launch: the per-launch call was seen to stall the
  // host for ~20 ms now and then (profiles/r06_notes.md section 8)
  static const hipError_t lds_attr = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (lds_attr != hipSuccess) return jatts_set_error(lds_attr, __FILE__, __LINE__);
  XcdOrder xo;
  const int64_t total = xo.plan((int)grid.x, (int)grid.y, (int)grid.z, (int64_t)BN * d.c_in * d.k_w * 6, ragged_tiles_1d(d.rg, BT));
  if (total >= (int64_t)1 << 31) return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "conv1d: launch too large");
  hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(WN * WT * 64), lds, s, d, f32_tile, xo);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

}  // namespace
