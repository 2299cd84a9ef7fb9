// Register-streamed f32 Conv1d / Linear (the "direct" variant of jatts_conv1d, f32 operands only).
//
// Why a second f32 kernel: on v_mfma_f32_32x32x2_f32 one 16-channel step of a 64n x 64t wave tile is 32 MFMAs x 64 cycles =
// 2 048 cycles of matrix pipe for 4 KiB of weights and 4 KiB of activations per wave -- the operands are so cheap that they can
// come straight from L2 / L1 into the MFMA source registers, for the ACTIVATIONS too: no LDS tile, no chunk and NO BARRIER in the
// main loop; every wave is an independent stream with a D-step register ring for both operands (buffer loads, D x 2 048 cycles
// ahead of use), and the taps of a k > 1 conv are just more steps whose activation fragment is the same rows shifted by tap * dil
// (re-read through L1).  What the round-3 workgroup traces (tools/trace_conv.py, profiles/r03_notes.md) showed and this kernel is
// built around:
//   * while one wave streams MFMAs, every instruction of another wave on that SIMD gets ONE issue slot per MFMA (~64 cycles): a
//     prologue / epilogue of a few hundred instructions lasts 10-50 k cycles, so (a) those phases must be short in INSTRUCTIONS
//     (bias through the matrix pipe, fragment-order buffer stores: no LDS round trip, no barrier, no per-element index math),
//     (b) the main loop must carry no per-lane address arithmetic (32-bit buffer offsets fixed per lane + scalar offsets; zero
//     padding = the descriptor's range check) and (c) three workgroups per CU (<= 168 registers, no LDS) keep one wave streaming
//     while the others are outside their main loops;
//   * with the pipe > 90 % busy the part is POWER limited: the shader clock sags to 1.7-2.0 GHz (2.4 in an MFMA-only loop), which is
//     what separates the measured 0.6-0.84 of the 157 TFLOP/s spec peak from the pipe utilisation.
// Not a design for f16 (16x the MFMA rate: the L1 path could not feed it) -- the f16 kernels keep their LDS tiles.
//
// Operand layout is the one of conv_tiles.h: A = packed weights [tap][c/16][n/32][lane][8]; B fragment of time step
// (lane & 31) = 8 consecutive channels 16 kc + 8 (lane >> 5) .. of row pos (two 16-byte loads per lane).
#pragma once
#include "conv1d_impl.h"

extern unsigned long long* jatts_g_trace;  // profiling hook (conv_api.hip: jatts_debug_trace)
extern unsigned jatts_g_trace_cap;

namespace {

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

// acc = act(acc) * alpha in place: one uniform branch per activation around a plain loop (a switch inside the unrolled store code
// would inline tanh / mish once per element and per call site).
template <int NF, int NT>
__device__ __forceinline__ void apply_act_alpha(f32x16 (&acc)[NF][NT], int act, float alpha) {
#define JATTS_ACT_LOOP(A)                                      \
  _Pragma("unroll") for (int f = 0; f < NF; ++f)               \
  _Pragma("unroll") for (int t = 0; t < NT; ++t)               \
  _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[f][t][r] = act_c<A>(acc[f][t][r]) * alpha;
  switch (act) {
    case JATTS_ACT_RELU: JATTS_ACT_LOOP(JATTS_ACT_RELU) break;
    case JATTS_ACT_TANH: JATTS_ACT_LOOP(JATTS_ACT_TANH) break;
    case JATTS_ACT_SWISH: JATTS_ACT_LOOP(JATTS_ACT_SWISH) break;
    case JATTS_ACT_MISH: JATTS_ACT_LOOP(JATTS_ACT_MISH) break;
    default:
      if (alpha != 1.f) { JATTS_ACT_LOOP(JATTS_ACT_NONE) }
      break;
  }
#undef JATTS_ACT_LOOP
}

// DIAG (tools/bench_conv.py --variant 8, wrong results, kept for the record of profiles/r03_notes.md): 1 = nothing is streamed in the
// main loop (the operands of the first D steps are reused) -- the pure-MFMA ceiling of this launch geometry.
// SNAKE: the SnakeBeta epilogue (JATTS_ACT_SNAKEBETA) as its own instantiation -- behind a runtime branch its 64 inlined sin^2 bodies
// made the register allocator spill an accumulator fragment inside the main loop of EVERY launch.
template <int NF, int NT, int WN, int WT, int D, int DIAG = 0, bool PRE = false, bool SNAKE = false>
__global__ __launch_bounds__(WN* WT * 64, (D <= 2 ? 3 : 2)) void conv1d_direct_kernel(jatts_conv_desc d, unsigned long long* trace, unsigned trace_cap, XcdOrder xo) {
  const unsigned wg_lin = blockIdx.x;     // 1-D grid in XCD-aware order (conv1d_impl.h: XcdOrder)
  int bx, by, bz;
  if (!xo.decode(wg_lin, bx, by, bz, d.rg, WT * NT * 32)) return;
  // Phase trace (profiling hook, jatts_debug_trace; tools/trace_conv.py): thread 0 of the first trace_cap workgroups stamps
  // [hw id, start, main loop entered, main loop done, stored, -, -, -, realtime start, realtime end]
  const bool tracing = trace != nullptr && wg_lin < trace_cap && threadIdx.x == 0;
#define JATTS_CSTAMP(i) do { if (tracing) trace[(size_t)wg_lin * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
  if (tracing) {
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    trace[(size_t)wg_lin * 16] = ((unsigned long long)xcc << 32) | hwid;
    trace[(size_t)wg_lin * 16 + 8] = __builtin_amdgcn_s_memrealtime();
  }
  JATTS_CSTAMP(1);
  constexpr int BT = WT * NT * 32;
  const int b = by;
  const int row_b = d.rg.cu_rows[b];
  const int L = (d.rg.cu_rows[b + 1] - row_b) * d.rg.len_mul;
  const int t0 = bx * BT;
  if (t0 >= L) return;
  const int64_t seq_row0 = (int64_t)row_b * d.rg.len_mul;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wn = wave / WT, wt = wave % WT;
  const int KC16 = d.c_in >> 4;
  const int n_pad = (d.n_out + 31) & ~31;
  const int NFR = n_pad >> 5;
  const int nf0 = (bz * WN + wn) * NF;
  const int col0 = wt * NT * 32;
  const int g = lane >> 5;
  conv_second_output(d, bz * WN * NF * 32);

  // accumulators start at the bias through the matrix pipe: acc = [bias | 0] (A, k = 0 / 1) x [1 | 1] (B) + 0 -- four MFMAs with an
  // inline-zero C operand instead of 64 v_mov + 8 loads per lane
  f32x16 acc[NF][NT];
  {
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const int n = (nf0 + f) * 32 + (lane & 31);
      const float bv = (d.bias && g == 0 && n < d.n_out) ? d.bias[n] : 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[f][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, 1.f, zero, 0, 0, 0);
    }
  }

  // producer: steps run kc-major with the taps innermost (the k_w shifted reads of a 16-channel column block follow each other
  // and hit the same L1 lines); past the last step it re-reads the last one (never consumed).
  // Both operands come through BUFFER loads: a 32-bit per-lane byte offset that never changes (+ one v_add of the scalar tap shift
  // for the activations) and a scalar offset per step.  Zero padding IS the descriptor's range check: the activation descriptor
  // spans exactly this sequence's rows, a row before it (negative offset = huge unsigned) or after it returns zeros.
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)d.w, 0, (unsigned)((size_t)d.k_w * KC16 * NFR * 2048), 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)((const float*)d.x[0] + seq_row0 * (int64_t)d.ldx), 0,
                                                                    (unsigned)(((int64_t)(L - 1) * d.ldx + d.c_in) * 4), 0x00020000);
  unsigned vw[NF], vx[NT];
#pragma unroll
  for (int f = 0; f < NF; ++f) vw[f] = (unsigned)(lane * 32 + (nf0 + f < NFR ? nf0 + f : NFR - 1) * 2048);   // clamped: never stored
#pragma unroll
  for (int t = 0; t < NT; ++t) vx[t] = (unsigned)(((t0 + col0 + t * 32 + (lane & 31) - d.pad) * d.ldx + 8 * g) * 4);
  const unsigned tap_stride = (unsigned)(d.dil * d.ldx * 4);
  int p_kc = 0, p_tap = 0;
  const int last_kc = KC16 - 1, last_tap = d.k_w - 1;
  auto ldb8 = [](const __amdgpu_buffer_rsrc_t& r, unsigned voff, unsigned soff) {
    const f32x4 lo = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
    const f32x4 hi = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff + 16, soff, 0));
    return f32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  };
  auto issue = [&](f32x8(&a)[NF], f32x8(&bq)[NT]) {
    const unsigned sw = (unsigned)((p_tap * KC16 + p_kc) * NFR) << 11;
#pragma unroll
    for (int f = 0; f < NF; ++f) a[f] = ldb8(rw, vw[f], sw);
    const unsigned shift = (unsigned)p_tap * tap_stride, sx = (unsigned)p_kc << 6;
#pragma unroll
    for (int t = 0; t < NT; ++t) bq[t] = ldb8(rx, vx[t] + shift, sx);
    ++p_tap;
    if (p_tap > last_tap) { p_tap = 0; ++p_kc; }
    if (p_kc > last_kc) { p_kc = last_kc; p_tap = last_tap; }
  };
  f32x8 ra[D][NF], rb[D][NT];
#pragma unroll
  for (int j = 0; j < D; ++j) issue(ra[j], rb[j]);
  __builtin_amdgcn_sched_barrier(0);
  const int n_steps = KC16 * d.k_w;   // a multiple of 4 (c_in % 64 == 0), hence of D
  JATTS_CSTAMP(2);
  for (int s0 = 0; s0 < n_steps; s0 += D) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      if (PRE) {   // LeakyReLU prologue (HiFi-GAN upsampling convs): max(v, slope v), 0 <= slope <= 1, on the fragment about to be consumed
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int e = 0; e < 8; ++e) rb[j][t][e] = fmaxf(rb[j][t][e], rb[j][t][e] * d.pre_slope);
      }
#pragma unroll
      for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int t = 0; t < NT; ++t) mma32(ra[j][f], rb[j][t], acc[f][t]);
      if (DIAG == 0) issue(ra[j], rb[j]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  JATTS_CSTAMP(3);

  // Row-major f32 outputs (every projection / FFN / postnet conv): fragment-order epilogue through buffer stores -- a lane owns 4
  // consecutive channels of one row (16 bytes), its g-neighbour the next 4; rows past the sequence end are dropped by the range
  // check.  A quarter of the instructions of an LDS-coalesced pass (no LDS round trip, no barrier, no per-unit index math).
  const bool rowmajor = !d.y_transposed && (d.n_out & 7) == 0 && (reinterpret_cast<uintptr_t>(d.y) & 15) == 0 && (d.ldy & 3) == 0 &&
                        (!d.resid || ((d.ldr & 3) == 0 && (reinterpret_cast<uintptr_t>(d.resid) & 15) == 0));
  if constexpr (SNAKE) snake_acc<NF, NT>(acc, d.act_a, d.act_b, nf0, d.n_out, lane);
  if (rowmajor) {
    apply_act_alpha<NF, NT>(acc, d.act, d.alpha);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)((float*)d.y + seq_row0 * (int64_t)d.ldy), 0,
                                                                      (unsigned)(((int64_t)(L - 1) * d.ldy + d.n_out) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)(d.resid ? d.resid + seq_row0 * (int64_t)d.ldr : (const float*)d.y), 0,
                                                                      (unsigned)(((int64_t)(L - 1) * d.ldr + d.n_out) * 4), 0x00020000);
    const int nf0s = __builtin_amdgcn_readfirstlane(nf0);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int row = t0 + col0 + t * 32 + (lane & 31);
      const unsigned vy = (unsigned)((row * d.ldy + 4 * g) * 4), vr = (unsigned)((row * d.ldr + 4 * g) * 4);
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const int nb = (nf0s + f) * 32;
        if (nb >= d.n_out) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (nb + 8 * q >= d.n_out) continue;
          f32x4 v = {acc[f][t][4 * q], acc[f][t][4 * q + 1], acc[f][t][4 * q + 2], acc[f][t][4 * q + 3]};
          const unsigned so = (unsigned)(nb + 8 * q) * 4;
          if (d.resid) v += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, vr, so, 0));
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), ry, vy, so, 0);
        }
      }
    }
  } else {   // transposed (V^T) / ragged-width outputs: the generic fragment-order epilogue of the LDS-staged kernel
    switch (d.act) {
      case JATTS_ACT_RELU: conv_epilogue<float, JATTS_ACT_RELU, NF, NT>(d, acc, t0, col0, nf0, lane, L, seq_row0, b); break;
      case JATTS_ACT_TANH: conv_epilogue<float, JATTS_ACT_TANH, NF, NT>(d, acc, t0, col0, nf0, lane, L, seq_row0, b); break;
      case JATTS_ACT_SWISH: conv_epilogue<float, JATTS_ACT_SWISH, NF, NT>(d, acc, t0, col0, nf0, lane, L, seq_row0, b); break;
      case JATTS_ACT_MISH: conv_epilogue<float, JATTS_ACT_MISH, NF, NT>(d, acc, t0, col0, nf0, lane, L, seq_row0, b); break;
      default: conv_epilogue<float, JATTS_ACT_NONE, NF, NT>(d, acc, t0, col0, nf0, lane, L, seq_row0, b); break;
    }
  }
  JATTS_CSTAMP(4);
  if (tracing) trace[(size_t)wg_lin * 16 + 9] = __builtin_amdgcn_s_memrealtime();
#undef JATTS_CSTAMP
}

// Eligibility: one plain input (no sum / scale / LeakyReLU prologue), zero padding, 16-byte aligned rows, and sequence slabs /
// a weight set small enough for 32-bit buffer offsets.
inline bool conv_direct_ok(const jatts_conv_desc& d) {
  const int64_t maxL = (int64_t)d.rg.max_len * d.rg.len_mul;
  const int64_t n_pad = (d.n_out + 31) & ~31;
  const int64_t ld = d.ldx > d.ldy ? (d.ldx > d.ldr ? d.ldx : d.ldr) : (d.ldy > d.ldr ? d.ldy : d.ldr);
  return d.n_in == 1 && d.in_scale == 1.f && (d.pre_act == JATTS_PRE_NONE || (d.pre_act == JATTS_PRE_LRELU && d.pre_slope >= 0.f && d.pre_slope <= 1.f)) &&
         d.pad_mode == JATTS_PAD_ZERO && (d.ldx & 3) == 0 &&
         (reinterpret_cast<uintptr_t>(d.x[0]) & 15) == 0 && (reinterpret_cast<uintptr_t>(d.w) & 15) == 0 &&
         (maxL + 256 + (int64_t)d.k_w * d.dil) * ld * 4 < (int64_t)1 << 31 && (int64_t)d.k_w * d.c_in * n_pad * 4 < (int64_t)1 << 31;
}

template <int NF, int NT, int WN, int WT, int D, int DIAG = 0, bool PRE = false, bool SNAKE = false>
int launch_conv_direct(const jatts_conv_desc& d, hipStream_t s) {
  constexpr int BT = WT * NT * 32, BN = WN * NF * 32;
  const int64_t maxL = (int64_t)d.rg.max_len * d.rg.len_mul;
  XcdOrder xo;
  const int64_t total = xo.plan((int)((maxL + BT - 1) / BT), d.rg.n_seq, (d.n_out + BN - 1) / BN, (int64_t)BN * d.c_in * d.k_w * 4, ragged_tiles_1d(d.rg, BT));
  if (total >= (int64_t)1 << 31) return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "conv1d: launch too large");
  hipLaunchKernelGGL((conv1d_direct_kernel<NF, NT, WN, WT, D, DIAG, PRE, SNAKE>), dim3((unsigned)total), dim3(WN * WT * 64), 0, s, d, jatts_g_trace,
                     jatts_g_trace_cap, xo);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

}  // namespace
