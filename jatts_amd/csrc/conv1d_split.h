// Generic Conv1d / Linear / polyphase ConvTranspose1d with f32 activations in HBM and split-precision MFMA operands (JATTS_F32S, round 4).
//
// Same implicit GEMM as conv1d_impl.h (weights = A operand in fragment order, activation chunks of KCHT channels double-buffered in LDS),
// with every operand value carried as hi = f16(v), lo = f16(v - hi) of v scaled by a power of two, and a product spent as
// hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_f16 (common.h: mma32 on f16sx8) -- see resunit_split_impl.h for the arithmetic and
// tools/split_probe.hip for the measured error (below the exact-f32 chain's).  What is specific to the chunked conv:
//  * the activation scale is per CHUNK TILE: the workgroup takes the maximum of the (summed, scaled, activated) chunk it is about to
//    commit to LDS -- the values are in registers between stage_issue and the commit anyway -- and maps it to [2^14, 2^15).  A tile
//    belongs to one utterance and its geometry does not depend on the batch: an utterance's result is bit-identical in any batch;
//  * the accumulators live at the scale of the CURRENT chunk: when the next chunk's exponent differs they are multiplied by the exact
//    power of two in between (128 VALU per lane per chunk at most, against >= 48 k_w MFMAs);
//  * weights carry one scale per output channel (hip.pack_conv_weight_split), its inverse arrives in w_inv and is applied with the bias
//    in front of the ordinary f32 epilogues of conv1d_impl.h.
#pragma once
#include "conv1d_impl.h"
#include "resunit_split_impl.h"   // split_exp / exp2i / block_amax

namespace {

// commit of the split pipeline: combine the staged f32 inputs (sum, in_scale, LeakyReLU), block maximum, scale, split, LDS.  -> exponent
template <int MAXU, int NIN, int UPR, int NTHR>
__device__ __forceinline__ int split_commit(StageRegs<float, MAXU, NIN>& sr, char* lds, int pitch, int rows, int n_in, float in_scale,
                                            int pre_act, float slope, float* slots, int wave, int lane) {
  const int total = rows * UPR;
  const bool plain = n_in == 1 && in_scale == 1.f && pre_act == JATTS_PRE_NONE;
  float amax = 0.f;
#pragma unroll
  for (int j = 0; j < MAXU; ++j) {
    if (!plain) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float t = sr.v[0][j][e];
        if (NIN > 1 && n_in > 1) t += sr.v[NIN > 1 ? 1 : 0][j][e];
        if (NIN > 2 && n_in > 2) t += sr.v[NIN > 2 ? 2 : 0][j][e];
        t *= in_scale;
        if (pre_act == JATTS_PRE_LRELU) t = fmaxf(t, t * slope);     // 0 <= slope <= 1
        sr.v[0][j][e] = t;
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(sr.v[0][j][e]));     // (units past the tile hold zeros)
  }
#if defined(JATTS_DIAG) && defined(JATTS_DIAG_NOSPLIT)      // WRONG RESULTS: ceiling probes for tools/ only (no block maximum / no barrier; level 2: no lo half either)
  const int ex = 0;
  (void)amax;
#else
  const int ex = split_exp(block_amax(amax, slots, wave, lane, NTHR / 64));
#endif
  const float sx = exp2i(ex);
#pragma unroll
  for (int j = 0; j < MAXU; ++j) {
    const int u = threadIdx.x + j * NTHR;
    if (u >= total) continue;
    const int r = u / UPR, cu = u % UPR;
    f16sx8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float sv = sr.v[0][j][e] * sx;
      o.hi[e] = (f16)sv;
#if defined(JATTS_DIAG) && JATTS_DIAG_NOSPLIT == 2
      o.lo[e] = (f16)0.f;
#else
      o.lo[e] = (f16)(sv - (float)o.hi[e]);
#endif
    }
    Vec8IO<f16s>::sts(lds + (size_t)r * pitch + (size_t)cu * 32, o);
  }
  return ex;
}

// HALO: rows beyond the time tile the staging registers must cover (32, or 0 for the k = 1 instantiation: 128-channel chunks then fit the
// register file); RD: weight ring depth in K-steps.
template <int NF, int NT, int WN, int WT, int NIN, int KCHT, int OCC, int HALO = 32, int RD = 4>
__global__ __launch_bounds__(WN* WT * 64, OCC) void conv1d_split_kernel(jatts_conv_desc d, int f32_tile, XcdOrder xo, unsigned slot_off) {
  typedef f16s T;
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
  constexpr int pitch = KCHT * 4 + 16;
  const int rows = BT + (d.k_w - 1) * d.dil;
  const int KC16 = d.c_in >> 4;
  const int n_pad = (d.n_out + 31) & ~31;
  const int NFR = n_pad >> 5;
  const int nf0 = (bz * WN + wn) * NF;
  const int col0 = wt * NT * 32;
  conv_second_output(d, bz * WN * NF * 32);
  float* slots = reinterpret_cast<float*>(smem + slot_off);

  const float* xin[3] = {(const float*)d.x[0], (const float*)d.x[1], (const float*)d.x[2]};
  const bool reflect = d.pad_mode == JATTS_PAD_REFLECT;
  f32x16 acc[NF][NT];
  zero_acc<NF, NT>(acc);

  constexpr int UPRC = KCHT / 8;
  constexpr int MAXU = ((BT + HALO) * UPRC + NTHR - 1) / NTHR;   // halo <= HALO rows (the launcher refuses more)
  // (RD: a step is 3 NF NT 32-cycle MFMAs; four steps of look-ahead cover an L2 round trip, two do when two workgroups share the CU)
  static_assert((KCHT / 16) % RD == 0, "ring depth must divide the steps per chunk and tap");
  WRing<T, NF, RD> ring;
  const int n_chunks = d.c_in / KCHT;
  ring.init((const T*)d.w, KC16, NFR, nf0, d.k_w, KCHT / 16, n_chunks, lane);
  const size_t buf_bytes = (size_t)rows * pitch;
  StageRegs<float, MAXU, NIN> sr;
  stage_issue<float, MAXU, NIN, UPRC, NTHR>(sr, rows, t0 - d.pad, L, seq_row0, xin, d.n_in, d.ldx, 0, reflect);
  int ex = split_commit<MAXU, NIN, UPRC, NTHR>(sr, smem, pitch, rows, d.n_in, d.in_scale, d.pre_act, d.pre_slope, slots, wave, lane);
  __syncthreads();
  for (int ci = 0; ci < n_chunks; ++ci) {
    const bool more = ci + 1 < n_chunks;
    if (more) stage_issue<float, MAXU, NIN, UPRC, NTHR>(sr, rows, t0 - d.pad, L, seq_row0, xin, d.n_in, d.ldx, (ci + 1) * KCHT, reflect);
    conv_stage<T, NF, NT, RD>(acc, ring, KCHT / 16, d.k_w, d.dil, smem + (size_t)(ci & 1) * buf_bytes, pitch, col0, lane);
    if (more) {
      const int en = split_commit<MAXU, NIN, UPRC, NTHR>(sr, smem + (size_t)((ci + 1) & 1) * buf_bytes, pitch, rows, d.n_in, d.in_scale, d.pre_act,
                                                         d.pre_slope, slots, wave, lane);
      if (en != ex) {    // (uniform) the accumulators move to the next chunk's scale: an exact power of two
        const float m = exp2i(en - ex);
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[f][t][r] *= m;
        ex = en;
      }
    }
    __syncthreads();
  }

  // un-scale (exact) and add the bias: from here on the accumulators are what conv1d_impl.h's f32 epilogues expect
  {
    const float inv_sx = exp2i(-ex);
    const int gq = lane >> 5;
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n0 = (nf0 + f) * 32 + 8 * q + 4 * gq;
        f32x4 is = {0.f, 0.f, 0.f, 0.f}, bq = {0.f, 0.f, 0.f, 0.f};
        if (n0 < n_pad) is = *reinterpret_cast<const f32x4*>(d.w_inv + n0);      // (n_pad entries: whole quads)
        if (d.bias) {
          if (n0 + 3 < d.n_out) bq = *reinterpret_cast<const f32x4*>(d.bias + n0);
          else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (n0 + e < d.n_out) bq[e] = d.bias[n0 + e];
          }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[f][t][4 * q + e] = fmaf(acc[f][t][4 * q + e], is[e] * inv_sx, bq[e]);
      }
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

template <int NF, int NT, int WN, int WT, int NIN, int KCHT, int OCC, int HALO = 32, int RD = 4>
int launch_conv_split(const jatts_conv_desc& d, hipStream_t s) {
  constexpr int BT = WT * NT * 32, BN = WN * NF * 32;
  if ((d.k_w - 1) * d.dil > HALO) return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "conv1d (split): halo beyond 32 rows");
  if (d.c_in % KCHT) return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "conv1d (split): c_in must be a multiple of the chunk width");
  const int64_t maxL = (int64_t)d.rg.max_len * d.rg.len_mul;
  dim3 grid((unsigned)((maxL + BT - 1) / BT), (unsigned)d.rg.n_seq, (unsigned)((d.n_out + BN - 1) / BN));
  const size_t rows = (size_t)BT + (size_t)(d.k_w - 1) * d.dil;
  size_t lds = 2 * rows * (KCHT * 4 + 16);
  int f32_tile = 0;
  if (!d.y_transposed && (size_t)BT * (BN * 4 + 16) <= 159 * 1024) {     // the coalesced f32 output tile reuses the staging buffers
    f32_tile = 1;
    if (lds < (size_t)BT * (BN * 4 + 16)) lds = (size_t)BT * (BN * 4 + 16);
  }
  const unsigned slot_off = (unsigned)lds;
  lds += 64;                                                              // one amax slot per wave
  if (lds > 160 * 1024) return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "conv1d (split): tile exceeds 160 KiB LDS");
  auto kern = conv1d_split_kernel<NF, NT, WN, WT, NIN, KCHT, OCC, HALO, RD>;
  // the dynamic-LDS limit ONCE per kernel (function-local static of this template instantiation), not per launch: the per-launch call was seen to stall the
  // host for ~20 ms now and then (profiles/r06_notes.md section 8)
  static const hipError_t lds_attr = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (lds_attr != hipSuccess) return jatts_set_error(lds_attr, __FILE__, __LINE__);
  XcdOrder xo;
  const int64_t total = xo.plan((int)grid.x, (int)grid.y, (int)grid.z, (int64_t)BN * d.c_in * d.k_w * 4, ragged_tiles_1d(d.rg, BT));
  if (total >= (int64_t)1 << 31) return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "conv1d: launch too large");
  hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(WN * WT * 64), lds, s, d, f32_tile, xo, slot_off);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

}  // namespace
