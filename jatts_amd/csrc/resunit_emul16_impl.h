// Fused HiFi-GAN dilation unit  y = x + conv_1(lrelu(conv_d(lrelu(x))))  on f32-EQUIVALENT EMULATED operands (JATTS_F32E / JATTS_F32E6: three exact bf16
// terms per value, seven / six partial products per product -- resunit_emul_impl.h, common.h) issued as v_mfma_f32_16x16x32_bf16 (round 6).
//
// Why a second form.  The 16-bit matrix pipe of this part is POWER-limited (MI355X_MICROARCH.md, "DVFS give-back"), and the power follows the instruction:
// jatts_mfma_probe (csrc/probe.hip, tools/mfma_forms.py; profiles/r06_mfma_forms.txt) sustains, on the same N(0, 1) operand bits fed from LDS at the same
// operand bytes per flop,
//     v_mfma_f32_32x32x16_bf16   2 x 2 fragments per wave   1 546 - 1 587 TFLOP/s at 1.68 - 1.71 GHz
//     v_mfma_f32_16x16x32_bf16   4 x 4 fragments per wave   1 798       TFLOP/s at 1.91        GHz      (+14 %; registers only: 2 112 against 1 850)
// -- the 16 x 16 x 32 form sums 32 products before it touches the accumulator file, half the accumulator traffic per flop.  The units at C >= 128 sit at
// 0.87 of the 32 x 32 x 16 ceiling; this kernel is the same unit -- same tiles, same LDS image, same epilogues -- on the other instruction.
//
// What changes against resunit_emul_impl.h:
//  * a wave's tile is NF x NT fragments of 16 (channels) x 16 (columns); a K-step is 32 channels.  Lane l supplies row / column (l & 15) and the 8
//    contraction elements 8 (l >> 4) .. of the step: in the LDS image ([row][8-channel unit][b0 | b1 | b2]) that is unit 4 kk + (l >> 4) of its row;
//  * weights in the matching fragment order  [tap][c / 32][n / 16][lane = 16 ((c % 32) / 8) + n % 16][c % 8][b0 x8 | b1 x8 | b2 x8]
//    (jatts_resunit_desc.w_layout = 1; jatts_amd.hip.pack_unit_weight_bf16x3_k32);
//  * C / D fragment: column = l & 15, channels 4 (l >> 4) + {0..3} of the fragment: a lane still owns 4 consecutive channels of one column;
//  * a K-step is ONE straight-line body with single-buffered operands (step16): weight fragment by weight fragment, each refilled for the next step right
//    after its NT NP MFMAs (NF - 1 fragment groups ahead of its use), column pairs interleaved so an accumulator is touched every second MFMA.
// The seven-product form keeps the two accumulators per fragment (leading product | the six small ones, joined by one correctly rounded add).
#pragma once
#include <type_traits>

#include "resunit_emul_impl.h"

#ifndef JATTS_U16_ROWS
#define JATTS_U16_ROWS 1   // 1: the fragment groups but the last walk all NT column fragments in turn (row16); 0: column pairs everywhere (A/B builds)
#endif
#ifndef JATTS_U16_DIAG
#define JATTS_U16_DIAG 0   // timing probes only (wrong results; separate DIAG builds): 1 = no B (LDS) refills in the K-loop, 2 = no A (L2) refills, 3 = neither
#endif

namespace {

struct acc2x4 { f32x4 big, small; };
template <typename T> struct Acc16 { typedef f32x4 type; };
template <> struct Acc16<bf3p<7>> { typedef acc2x4 type; };
__device__ __forceinline__ void acc16_set(f32x4& a, int i, float v) { a[i] = v; }
__device__ __forceinline__ void acc16_set(acc2x4& a, int i, float v) { a.small[i] = v; a.big[i] = 0.f; }
__device__ __forceinline__ void acc16_finish(f32x4&) {}
__device__ __forceinline__ void acc16_finish(acc2x4& a) { a.small = a.big + a.small; }
__device__ __forceinline__ f32x4& acc16_val(f32x4& a) { return a; }
__device__ __forceinline__ f32x4& acc16_val(acc2x4& a) { return a.small; }

// partial product P of a fragment pair (a = weights, b = activations), smallest first -- the order of common.h's mma32
template <int P> __device__ __forceinline__ void mma16p(const bf3px8<7>& a, const bf3px8<7>& b, acc2x4& c) {
  if constexpr (P == 0) c.small = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.b1, b.b2, c.small, 0, 0, 0);
  if constexpr (P == 1) c.small = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.b2, b.b0, c.small, 0, 0, 0);
  if constexpr (P == 2) c.small = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.b0, b.b2, c.small, 0, 0, 0);
  if constexpr (P == 3) c.small = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.b1, b.b1, c.small, 0, 0, 0);
  if constexpr (P == 4) c.small = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.b1, b.b0, c.small, 0, 0, 0);
  if constexpr (P == 5) c.small = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.b0, b.b1, c.small, 0, 0, 0);
  if constexpr (P == 6) c.big = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.b0, b.b0, c.big, 0, 0, 0);
}
template <int P> __device__ __forceinline__ void mma16p(const bf3px8<6>& a, const bf3px8<6>& b, f32x4& c) {
  if constexpr (P == 0) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.b2, b.b0, c, 0, 0, 0);
  if constexpr (P == 1) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.b0, b.b2, c, 0, 0, 0);
  if constexpr (P == 2) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.b1, b.b1, c, 0, 0, 0);
  if constexpr (P == 3) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.b1, b.b0, c, 0, 0, 0);
  if constexpr (P == 4) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.b0, b.b1, c, 0, 0, 0);
  if constexpr (P == 5) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.b0, b.b0, c, 0, 0, 0);
}
template <typename T> struct NProd;
template <int NP> struct NProd<bf3p<NP>> { static constexpr int value = NP; };

// Weight stream of the 16 x 16 x 32 form: A fragments of K-step s (tap-major over `steps` = k_w KCS steps, KCS steps per tap starting at weight step kc0)
// straight from L2 into registers (one buffer, refilled fragment by fragment: step16).
template <typename T, int NF>
struct WStream16 {
  typedef typename Elem<T>::vec8 V8;
  V8 ra[NF];
  const T* wl;          // + lane * 8 elements, + this wave's first fragment
  int kc_stride;        // elements between K-steps of one tap (NFR16 * 512)
  int tap_stride;       // elements between taps (KC32 * NFR16 * 512)
  __device__ __forceinline__ void bind(const T* w, int KC32, int NFR16, int nf0, int lane) {
    wl = w + (size_t)nf0 * 512 + (size_t)lane * 8;
    kc_stride = NFR16 * 512;
    tap_stride = KC32 * kc_stride;
  }
  __device__ __forceinline__ void fetch(int tap, int kc) {
    const T* p = wl + (size_t)tap * tap_stride + (size_t)kc * kc_stride;
#pragma unroll
    for (int f = 0; f < NF; ++f) ra[f] = Vec8IO<T>::ldg(p + f * 512);
  }
};

// NP MFMAs each of fragment pairs (f, t0) and (f, t0 + 1), interleaved: an accumulator is touched every second MFMA
template <typename T, int NF, int NT>
__device__ __forceinline__ void pair16(typename Acc16<T>::type (&acc)[NF][NT], const typename Elem<T>::vec8& a, typename Elem<T>::vec8 (&rb)[NT], int f_, int t0_) {
  // (f_, t0_ are compile-time after unrolling; NT is even)
  constexpr int NP = NProd<T>::value;
  auto& c0 = acc[f_][t0_];
  auto& c1 = acc[f_][t0_ + 1];
  mma16p<0>(a, rb[t0_], c0); mma16p<0>(a, rb[t0_ + 1], c1);
  mma16p<1>(a, rb[t0_], c0); mma16p<1>(a, rb[t0_ + 1], c1);
  mma16p<2>(a, rb[t0_], c0); mma16p<2>(a, rb[t0_ + 1], c1);
  mma16p<3>(a, rb[t0_], c0); mma16p<3>(a, rb[t0_ + 1], c1);
  mma16p<4>(a, rb[t0_], c0); mma16p<4>(a, rb[t0_ + 1], c1);
  mma16p<5>(a, rb[t0_], c0); mma16p<5>(a, rb[t0_ + 1], c1);
  if constexpr (NP == 7) { mma16p<6>(a, rb[t0_], c0); mma16p<6>(a, rb[t0_ + 1], c1); }
}

// NP MFMAs each of the NT fragment pairs (f, 0 .. NT - 1), interleaved: an accumulator is touched every NT-th MFMA
template <typename T, int NF, int NT>
__device__ __forceinline__ void row16(typename Acc16<T>::type (&acc)[NF][NT], const typename Elem<T>::vec8& a, typename Elem<T>::vec8 (&rb)[NT], int f_) {
  constexpr int NP = NProd<T>::value;
#pragma unroll
  for (int t = 0; t < NT; ++t) mma16p<0>(a, rb[t], acc[f_][t]);
#pragma unroll
  for (int t = 0; t < NT; ++t) mma16p<1>(a, rb[t], acc[f_][t]);
#pragma unroll
  for (int t = 0; t < NT; ++t) mma16p<2>(a, rb[t], acc[f_][t]);
#pragma unroll
  for (int t = 0; t < NT; ++t) mma16p<3>(a, rb[t], acc[f_][t]);
#pragma unroll
  for (int t = 0; t < NT; ++t) mma16p<4>(a, rb[t], acc[f_][t]);
#pragma unroll
  for (int t = 0; t < NT; ++t) mma16p<5>(a, rb[t], acc[f_][t]);
  if constexpr (NP == 7) {
#pragma unroll
    for (int t = 0; t < NT; ++t) mma16p<6>(a, rb[t], acc[f_][t]);
  }
}

// One K-step of the 16 x 16 x 32 form: NF NT NP MFMAs, ONE body for every step (no buffer parity: with two alternating register buffers the compiler
// rotated them through the accumulator file, ~3 v_accvgpr moves per MFMA).  The A fragments (weights, L2 -> registers) are single-buffered and walked
// fragment-major: fragment f's NT NP MFMAs, then its register is refilled with the NEXT step's fragment f -- NF - 1 fragment groups (>= 1 300 pipe cycles at
// NF = 4) ahead of its use.  The B fragments (LDS) serve every group of the step; the LAST group runs column pair by column pair and refills each pair as it
// finishes, and the FIRST group of the next step starts with the pairs refilled first.
template <typename T, int NF, int NT>
__device__ __forceinline__ void step16(typename Acc16<T>::type (&acc)[NF][NT], typename Elem<T>::vec8 (&ra)[NF], typename Elem<T>::vec8 (&rb)[NT], const T* pa,
                                       const char* pb, int pitch) {
  static_assert(NT % 2 == 0, "column fragments come in pairs");
  constexpr bool ROWS16 = JATTS_U16_ROWS != 0;
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    if (f < NF - 1 && ROWS16) {       // every column fragment in turn: an accumulator every NT-th MFMA
      row16<T, NF, NT>(acc, ra[f], rb, f);
      __builtin_amdgcn_sched_barrier(0);
    } else {
#pragma unroll
      for (int tp = 0; tp < NT; tp += 2) {
        pair16<T, NF, NT>(acc, ra[f], rb, f, tp);
        if (f == NF - 1 && !(JATTS_U16_DIAG & 1)) {      // the step's last use of these two B fragments: the next step's
          rb[tp] = Vec8IO<T>::lds(pb + (size_t)(tp * 16) * pitch);
          rb[tp + 1] = Vec8IO<T>::lds(pb + (size_t)((tp + 1) * 16) * pitch);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (!(JATTS_U16_DIAG & 2)) ra[f] = Vec8IO<T>::ldg(pa + f * 512);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// One conv over an LDS tile on the 16 x 16 x 32 form.  KCS K-steps per tap, the first at weight step kc0 (KSPLIT: a channel half); `act` = the tile, unit
// (8 channels) u of a row at byte u * 48.  On entry ws.ra holds step 0 (fetched by the caller, under the staging / the previous epilogue); on exit the
// first step of what follows (`tail`: the next conv's / channel half's first fragments, or nullptr = a harmless re-read).  No branch inside the loop: the
// positions of the next step are selects.
template <typename T, int NF, int NT, int KCS>
__device__ __forceinline__ void conv16(typename Acc16<T>::type (&acc)[NF][NT], WStream16<T, NF>& ws, int kc0, int k_w, int dil, const char* act, int pitch,
                                       int col0, int lane, const T* tail) {
  typedef typename Elem<T>::vec8 V8;
  const int steps = k_w * KCS;
  const char* bbase = act + (size_t)(col0 + (lane & 15)) * pitch + (size_t)(lane >> 4) * 48;
  const T* abase = ws.wl + (size_t)kc0 * ws.kc_stride;
  V8 rb[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) rb[t] = Vec8IO<T>::lds(bbase + (size_t)(t * 16) * pitch);
  __builtin_amdgcn_sched_barrier(0);
  if (!tail) tail = abase + (size_t)(k_w - 1) * ws.tap_stride + (size_t)(KCS - 1) * ws.kc_stride;
  int tap = 0, kk = 0;        // position of the step about to run
  for (int s = 0; s < steps; ++s) {
    int nkk = kk + 1;
    const bool wrap = nkk == KCS;
    nkk = wrap ? 0 : nkk;
    const int ntap = tap + (wrap ? 1 : 0);
    const bool last = s + 1 >= steps;
    const T* pa = last ? tail : abase + (size_t)ntap * ws.tap_stride + (size_t)nkk * ws.kc_stride;
    const char* pb = bbase + (size_t)((last ? tap : ntap) * dil) * pitch + (size_t)((last ? kk : nkk) * 4) * 48;
    step16<T, NF, NT>(acc, ws.ra, rb, pa, pb, pitch);
    tap = ntap;
    kk = nkk;
  }
}

template <typename T, int C, int WGCOLS, int WN, int WT, int OCC, bool KSPLIT = false, bool RREG = false>      // T = bf3 (seven partial products) or bf3f (six)
__global__ __launch_bounds__(WN* WT * 64, OCC) void resunit_emul16_kernel(jatts_resunit_desc d, unsigned long long* trace, unsigned trace_cap,
                                                                         unsigned bias_off) {
  typedef typename Elem<T>::vec8 V8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NF = C / (WN * 16), NT = WGCOLS / (WT * 16);
  constexpr int KC32 = C / 32, NFR16 = C / 16;
  constexpr int pitch = C * 6 + 16;
  constexpr int pitch_x = KSPLIT ? (C / 2) * 6 + 16 : pitch;
  constexpr int NTHR = WN * WT * 64;
  static_assert(NF * WN * 16 == C && NT * WT * 16 == WGCOLS && C % 32 == 0, "tile shape");
  static_assert(sizeof(T) == 6, "bf3 is three packed bf16");
  static_assert(!(KSPLIT && RREG), "the residual registers go with the one-piece x tile");
  static_assert(!KSPLIT || KC32 % 2 == 0, "channel halves are whole K-steps");
  constexpr int MAXI = RREG ? (WGCOLS * (C / 8) + NTHR - 1) / NTHR : 1;      // interior units per thread (resunit_emul_impl.h: RREG)
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
  const int g4 = lane >> 4;                 // which 4 of a fragment's 16 channels this lane owns in C / D
  const int col0 = wt * NT * 16;
  const int nf0 = wn * NF;

  const int rx = WGCOLS + 2 * p1;   // x tile rows: row r <-> position t0 - p2 - p1 + r
  char* xs = smem;                  // bf3 lrelu(x) tile; h overlays it; finally the f32 y tile
  char* hs = smem;
  float* bs = reinterpret_cast<float*>(smem + bias_off);   // b1 | b2
  for (int u = threadIdx.x; u < 2 * C; u += NTHR) bs[u] = u < C ? d.b1[u] : d.b2[u - C];

  WStream16<T, NF> ws;
  ws.bind((const T*)d.w1, KC32, NFR16, nf0, lane);
  ws.fetch(0, 0);                  // conv1's first K-step, under the staging

  typename Acc16<T>::type acc[NF][NT];
  auto bias_acc = [&](const float* bv) {
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const f32x4 bb = *reinterpret_cast<const f32x4*>(bv + (nf0 + f) * 16 + 4 * g4);
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc16_set(acc[f][t], e, bb[e]);
    }
  };
  auto finish_acc = [&]() {
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc16_finish(acc[f][t]);
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
      const int r_in = p1 + p2, n_in = tt_out * UPR;
#pragma unroll
      for (int j = 0; j < MAXI; ++j) {
        const int v = threadIdx.x + j * NTHR;
        const int ro = v / UPR, cu = v - ro * UPR;
        const int pos = t0 + ro;
        if (v < n_in && pos < L) xk[j] = Vec8IO<float>::ldg(x + (seq_row0 + pos) * (int64_t)C + cu * 8);
        else xk[j] = f32x8{0, 0, 0, 0, 0, 0, 0, 0};
      }
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
      // the WHOLE tile in one batch of loads (halos up to 32 rows a side; longer ones take a second pass): the accumulators are not live yet, and a second
      // batch is a second serial HBM round trip -- 3.5 k of a C = 128 workgroup's 138 k cycles
      constexpr int UBX = ((WGCOLS + 64) * UPR + NTHR - 1) / NTHR;
      for (int base = threadIdx.x; base < total; base += NTHR * UBX) {
        f32x8 v[UBX];
#pragma unroll
        for (int j = 0; j < UBX; ++j) {
          const int u = base + j * NTHR;
          const int r = u / UPR, cu = u - r * UPR;
          const int pos = pos0 + r;
          if (u < total && pos >= 0 && pos < L) v[j] = Vec8IO<float>::ldg(x + (seq_row0 + pos) * (int64_t)C + cu * 8);
          else v[j] = f32x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
#pragma unroll
        for (int j = 0; j < UBX; ++j) {
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
    conv16<T, NF, NT, KC32>(acc, ws, 0, K, dil, xs, pitch, col0, lane, (const T*)d.w2 + (size_t)nf0 * 512 + (size_t)lane * 8);
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
    // first channel half: weight steps 0 .. KC32 / 2 - 1 of every tap; the stream continues with the second half's first step
    conv16<T, NF, NT, KC32 / 2>(acc, ws, 0, K, dil, xs, pitch_x, col0, lane, ws.wl + (size_t)(KC32 / 2) * ws.kc_stride);
    lds_barrier();                // every wave is done reading the first half
    store_half();
    lds_barrier();
    conv16<T, NF, NT, KC32 / 2>(acc, ws, KC32 / 2, K, dil, xs, pitch_x, col0, lane, (const T*)d.w2 + (size_t)nf0 * 512 + (size_t)lane * 8);
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
    const int col = col0 + t * 16 + (lane & 15);
    const int pos = t0 - p2 + col;
    const float keep = (pos >= 0 && pos < L) ? 1.f : 0.f;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const int n0 = (nf0 + f) * 16 + 4 * g4;       // this lane's 4 channels
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a = acc16_val(acc[f][t])[e] * keep;
        v[e] = fmaxf(a, a * d.slope);
      }
      bf16x4 q0, q1, q2;
      bf3_split4(v, q0, q1, q2);
      char* p = hs + (size_t)col * pitch + (size_t)(n0 >> 3) * 48 + (size_t)(n0 & 4) * 2;
      *reinterpret_cast<bf16x4*>(p) = q0;
      *reinterpret_cast<bf16x4*>(p + 16) = q1;
      *reinterpret_cast<bf16x4*>(p + 32) = q2;
    }
  }
  JATTS_STAMP(12);
  lds_barrier();
  JATTS_STAMP(4);

  bias_acc(bs + C);
  ws.bind((const T*)d.w2, KC32, NFR16, nf0, lane);     // (ra already holds w2's first step)
  conv16<T, NF, NT, KC32>(acc, ws, 0, K, 1, hs, pitch, col0, lane, nullptr);
  finish_acc();
  JATTS_STAMP(5);

  // ---- epilogue 2: acc (+ b2, already in) assembled as an f32 tile in LDS; the residual (and the MRF mean) are added in the row-contiguous store pass
  __syncthreads();
  char* ys = smem;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = col0 + t * 16 + (lane & 15);
    if (col >= tt_out || t0 + col >= L) continue;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const int n0 = (nf0 + f) * 16 + 4 * g4;
      *reinterpret_cast<f32x4*>(ys + (size_t)col * pitch + (size_t)n0 * 4) = acc16_val(acc[f][t]);
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

template <typename T, int C, int WGCOLS, int WN, int WT, int OCC = 2, bool KSPLIT = false, bool RREG = false>
int launch_resunit_emul16(const jatts_resunit_desc& d, hipStream_t s) {
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
  auto kern = resunit_emul16_kernel<T, C, WGCOLS, WN, WT, OCC, KSPLIT, RREG>;
  // the dynamic-LDS limit ONCE per kernel (function-local static of this template instantiation), not per launch: the per-launch call was seen to stall the
  // host for ~20 ms now and then (profiles/r06_notes.md section 8)
  static const hipError_t lds_attr = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (lds_attr != hipSuccess) return jatts_set_error(lds_attr, __FILE__, __LINE__);
  hipLaunchKernelGGL(kern, grid, dim3(WN * WT * 64), lds, s, d, jatts_g_trace, jatts_g_trace_cap, bias_off);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

}  // namespace
