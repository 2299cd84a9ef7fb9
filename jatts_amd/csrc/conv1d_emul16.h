// Generic Conv1d / Linear / polyphase ConvTranspose1d on f32-EQUIVALENT emulated operands (conv1d_emul.h) issued as v_mfma_f32_16x16x32_bf16 (round 6:
// the power-limited matrix pipe sustains 14 % more of that form than of 32 x 32 x 16 -- csrc/probe.hip, profiles/r06_mfma_forms.txt; the fused units moved
// first, csrc/resunit_emul16_impl.h, which also defines the fragment conventions used here).
//
// Same pipeline as conv1d_emul.h -- activation chunks of 64 channels double-buffered in LDS as three bf16 planes, the next chunk's global loads in flight
// under this chunk's MFMAs, weights streamed L2 -> registers -- with
//  * a K-step of 32 channels, a wave tile of NF x NT fragments of 16 channels x 16 time steps;
//  * weights in the fragment order [tap][c / 32][n / 16][lane = 16 ((c % 32) / 8) + n % 16][c % 8] x (b0 | b1 | b2)  (jatts_conv_desc.w_layout = 1);
//  * a two-step weight ring (the look-ahead of the 32 x 32 x 16 kernel's four 16-channel steps), each slot refilled fragment by fragment right after
//    its last MFMA; ONE set of B fragments, the last fragment group of a step refilling them column pair by column pair (resunit_emul16_impl.h: step16);
//  * epilogues on the 16 x 16 C / D layout (column = lane & 15, channels 4 (lane >> 4) + {0..3} of the fragment).
#pragma once
#include <stdlib.h>
#include "conv1d_emul.h"
#include "resunit_emul16_impl.h"

#ifndef JATTS_CEMUL_ANTIPHASE
#define JATTS_CEMUL_ANTIPHASE 1   // A/B switch of the anti-phase staging of the eight-wave tiles (conv1d_emul16_kernel)
#endif
#ifndef JATTS_CEMUL_TRACE
#define JATTS_CEMUL_TRACE 0   // DIAG builds only: phase clocks of the chunk loop into jatts_debug_trace's buffer (tools/trace_conv16.py)
#endif
#if JATTS_CEMUL_TRACE
#define JATTS_CEMUL_TRACE_PARAMS , unsigned long long* trace, unsigned trace_cap
#else
#define JATTS_CEMUL_TRACE_PARAMS
#endif

namespace {

// Weight-fragment ring of the 16 x 16 x 32 form: D slots of NF fragments, consumed in the order  for chunk: for tap: for kk (32-channel steps) ; the producer
// runs D steps ahead ACROSS chunk boundaries and clamps at the last fragment.  fetch_frag(slot j, fragment f) is called right after that fragment's last MFMA
// of the step; advance() moves the producer to the next step once all NF fragments of a slot are refilled.
template <typename T, int NF, int D>
struct WRing16 {
  typedef typename Elem<T>::vec8 V8;
  V8 r[D][NF];
  const T* wbase;
  int nfo[NF];
  int KC32, NFR16, k_w, kc_per, n_chunks;
  int p_chunk, p_tap, p_kk;

  __device__ __forceinline__ void init(const T* w, int KC32_, int NFR16_, int nf0, int k_w_, int kc_per_, int n_chunks_, int lane) {
    wbase = w + (size_t)lane * 8;
    KC32 = KC32_; NFR16 = NFR16_; k_w = k_w_; kc_per = kc_per_; n_chunks = n_chunks_;
#pragma unroll
    for (int f = 0; f < NF; ++f) nfo[f] = (nf0 + f < NFR16 ? nf0 + f : NFR16 - 1) * 512;  // clamped: never stored
    p_chunk = p_tap = p_kk = 0;
#pragma unroll
    for (int j = 0; j < D; ++j) {
#pragma unroll
      for (int f = 0; f < NF; ++f) fetch_frag(j, f);
      advance();
    }
  }
  __device__ __forceinline__ const T* pos() const { return wbase + ((size_t)(p_tap * KC32 + p_chunk * kc_per + p_kk) * NFR16) * 512; }
  __device__ __forceinline__ void fetch_frag(int j, int f) { r[j][f] = Vec8IO<T>::ldg(pos() + nfo[f]); }
  __device__ __forceinline__ void advance() {
    const int nk = p_kk + 1;
    const bool wk = nk == kc_per;
    p_kk = wk ? 0 : nk;
    const int nt = p_tap + (wk ? 1 : 0);
    const bool wt = nt == k_w;
    p_tap = wt ? 0 : nt;
    const int nc = p_chunk + (wt ? 1 : 0);
    const bool end = nc == n_chunks;  // clamp at the last fragment of the last chunk
    p_chunk = end ? n_chunks - 1 : nc;
    p_tap = end ? k_w - 1 : p_tap;
    p_kk = end ? kc_per - 1 : p_kk;
  }
};

// One chunk: k_w * kc_per K-steps (a multiple of D, so every ring slot is a compile-time register index and the body is straight-line).
template <typename T, int NF, int NT, int D>
__device__ __forceinline__ void conv_stage16(typename Acc16<T>::type (&acc)[NF][NT], WRing16<T, NF, D>& ring, int kc_per, int k_w, int dil, const char* act,
                                             int pitch, int col0, int lane) {
  typedef typename Elem<T>::vec8 V8;
  static_assert(NT % 2 == 0, "column fragments come in pairs");
  const int n_it = k_w * kc_per;
  const char* bbase = act + (size_t)(col0 + (lane & 15)) * pitch + (size_t)(lane >> 4) * 48;
  const int last_tap = k_w - 1;
  int bp_tap = 0, bp_kk = 0;  // position of the NEXT step's B fragments (clamps at the end)
  auto next_b = [&]() {
    const int nk = bp_kk + 1;
    const bool wrap = nk == kc_per;
    const bool end = wrap && bp_tap == last_tap;
    bp_kk = end ? bp_kk : (wrap ? 0 : nk);
    bp_tap = end ? bp_tap : bp_tap + (wrap ? 1 : 0);
    return bbase + (size_t)(bp_tap * dil) * pitch + (size_t)(bp_kk * 4) * 48;
  };
  V8 rb[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) rb[t] = Vec8IO<T>::lds(bbase + (size_t)(t * 16) * pitch);
  __builtin_amdgcn_sched_barrier(0);
  for (int it0 = 0; it0 < n_it; it0 += D) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const char* pb = next_b();
#pragma unroll
      for (int f = 0; f < NF; ++f) {
#pragma unroll
        for (int tp = 0; tp < NT; tp += 2) {
          pair16<T, NF, NT>(acc, ring.r[j][f], rb, f, tp);
          if (f == NF - 1 && !(JATTS_CEMUL_DIAG & 16)) {
            rb[tp] = Vec8IO<T>::lds(pb + (size_t)(tp * 16) * pitch);
            rb[tp + 1] = Vec8IO<T>::lds(pb + (size_t)((tp + 1) * 16) * pitch);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (!(JATTS_CEMUL_DIAG & 8)) ring.fetch_frag(j, f);       // this slot's fragment f of the step D ahead
        __builtin_amdgcn_sched_barrier(0);
      }
      ring.advance();
    }
  }
}

// SnakeBeta on the 16 x 16 accumulators (conv_tiles.h: snake_acc)
template <int NF, int NT>
__device__ __forceinline__ void snake_acc16(f32x4 (&acc)[NF][NT], const float* a, const float* b, int nf0, int n_out, int lane) {
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    const int n0 = min((nf0 + f) * 16 + 4 * (lane >> 4), n_out - 4);       // n_out % 4 == 0 (jatts_conv1d checks); an outside quad is never stored
    const f32x4 a4 = *reinterpret_cast<const f32x4*>(a + n0), b4 = *reinterpret_cast<const f32x4*>(b + n0);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = acc[f][t][e];
        acc[f][t][e] = fmaf(b4[e], sin2_f(v * a4[e]), v);
      }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// conv_epilogue (conv1d_impl.h) on the 16 x 16 C / D layout: y = act(acc) * alpha + resid; lane owns column lane & 15 and channels n0 .. n0 + 3
template <int ACT, int NF, int NT>
__device__ __forceinline__ void conv_epilogue16(const jatts_conv_desc& d, f32x4 (&acc)[NF][NT], int t0, int col0, int nf0, int lane, int L, int64_t seq_row0,
                                                int seq) {
  const bool vec_r = d.resid && (d.ldr & 3) == 0, vec_y = (d.ldy & 3) == 0 && !d.y_transposed;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int pos = t0 + col0 + t * 16 + (lane & 15);
    if (pos >= L) continue;
    const int64_t row = seq_row0 + pos;
    const int64_t trow = d.y_seq_col0 ? (int64_t)d.y_seq_col0[seq] * d.rg.len_mul + pos : row;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const int n0 = (nf0 + f) * 16 + 4 * (lane >> 4);
      if (n0 >= d.n_out) continue;
      const bool full = n0 + 3 < d.n_out;
      f32x4 rq = {0.f, 0.f, 0.f, 0.f};
      if (full && vec_r) rq = *reinterpret_cast<const f32x4*>(d.resid + row * d.ldr + n0);
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = act_c<ACT>(acc[f][t][e]) * d.alpha + rq[e];
      if (full && vec_y && (vec_r || !d.resid)) {
        *reinterpret_cast<f32x4*>((float*)d.y + row * d.ldy + n0) = v;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int n = n0 + e;
          if (n >= d.n_out) break;
          float s = v[e];
          if (!full) s = act_c<ACT>(acc[f][t][e]) * d.alpha;
          if (d.resid && !(full && vec_r)) s += d.resid[row * d.ldr + n];
          const int64_t o = d.y_transposed ? (int64_t)n * d.ldy + trow : row * d.ldy + n;
          ((float*)d.y)[o] = s;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// conv_epilogue_lds (conv1d_impl.h) on the 16 x 16 layout: the f32 [BT][BN] tile assembled in LDS, then row-contiguous 16-byte stores with the residual added
template <int ACT, int NF, int NT, int BN>
__device__ __forceinline__ void conv_epilogue_lds16(const jatts_conv_desc& d, f32x4 (&acc)[NF][NT], char* smem, int t0, int col0, int nf_local0, int n_base,
                                                    int lane, int L, int64_t seq_row0, int BT) {
  constexpr int opitch = BN * 4 + 16;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = col0 + t * 16 + (lane & 15);
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const int nl = (nf_local0 + f) * 16 + 4 * (lane >> 4);   // channel inside the workgroup's BN slab
      if (n_base + nl >= d.n_out) continue;                     // n_out % 8 == 0: quads are all-or-nothing
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = act_c<ACT>(acc[f][t][e]) * d.alpha;
      *reinterpret_cast<f32x4*>(smem + (size_t)col * opitch + (size_t)nl * 4) = o;
    }
  }
  __syncthreads();
  const int vrows = min(BT, L - t0);
  const int upr = min(BN, d.n_out - n_base) / 8;
  const int total = vrows * upr;
  float* yg = (float*)d.y + (seq_row0 + t0) * (int64_t)d.ldy + n_base;
  const float* rg = d.resid ? d.resid + (seq_row0 + t0) * (int64_t)d.ldr + n_base : nullptr;
  for (int u = threadIdx.x; u < total; u += blockDim.x) {
    const int r = u / upr, cu = u - r * upr;
    const char* src = smem + (size_t)r * opitch + (size_t)cu * 32;
    float* dst = yg + (int64_t)r * d.ldy + cu * 8;
    f32x4 lo = *reinterpret_cast<const f32x4*>(src), hi = *reinterpret_cast<const f32x4*>(src + 16);
    if (rg) {
      const float* rp = rg + (int64_t)r * d.ldr + cu * 8;
      lo += *reinterpret_cast<const f32x4*>(rp);
      hi += *reinterpret_cast<const f32x4*>(rp + 4);
    }
    *reinterpret_cast<f32x4*>(dst) = lo;
    *reinterpret_cast<f32x4*>(dst + 4) = hi;
  }
}

// WN x WT waves, a wave's tile NF x NT fragments of 16 x 16; HALO: rows beyond the time tile the staging registers must cover; D: ring depth in 32-channel steps
template <typename T, int NF, int NT, int WN, int WT, int NIN, int KCHT, int OCC, int HALO = 32, int D = 2>      // T = bf3 (seven products) / bf3f (six)
__global__ __launch_bounds__(WN* WT * 64, OCC) void conv1d_emul16_kernel(jatts_conv_desc d, int f32_tile, XcdOrder xo JATTS_CEMUL_TRACE_PARAMS) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BT = WT * NT * 16, NTHR = WN * WT * 64, BN = WN * NF * 16;
  int bx, b, bz;
  if (!xo.decode(blockIdx.x, bx, b, bz, d.rg, BT)) return;
#if JATTS_CEMUL_TRACE     // DIAG builds only (tools/trace_conv16.py): per workgroup, the clocks wave 0 spends in each phase of the chunk loop
  const bool tracing = trace != nullptr && blockIdx.x < trace_cap && threadIdx.x == 0;
  unsigned long long ph[7] = {0, 0, 0, 0, 0, 0, 0}, tq = __builtin_amdgcn_s_memtime();
  const unsigned long long t_begin = tq, rt_begin = __builtin_amdgcn_s_memrealtime();
#define JATTS_PH(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); ph[i] += n_ - tq; tq = n_; } while (0)
#else
#define JATTS_PH(i) do {} while (0)
#endif
  const int row_b = d.rg.cu_rows[b];
  const int L = (d.rg.cu_rows[b + 1] - row_b) * d.rg.len_mul;
  const int t0 = bx * BT;
  if (t0 >= L) return;
  const int64_t seq_row0 = (int64_t)row_b * d.rg.len_mul;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wn = wave / WT, wt = wave % WT;
  constexpr int pitch = KCHT * 6 + 16;
  const int rows = BT + (d.k_w - 1) * d.dil;
  const int KC32 = d.c_in >> 5;
  const int n_pad = (d.n_out + 31) & ~31;
  const int NFR16 = n_pad >> 4;
  const int nf0 = (bz * WN + wn) * NF;
  const int col0 = wt * NT * 16;
  conv_second_output(d, bz * BN);

  const float* xin[3] = {(const float*)d.x[0], (const float*)d.x[1], (const float*)d.x[2]};
  const bool reflect = d.pad_mode == JATTS_PAD_REFLECT;
  typename Acc16<T>::type accx[NF][NT];
  {   // accumulators start at the bias (as conv1d_kernel)
    const int g4 = lane >> 4;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const int n0 = (nf0 + f) * 16 + 4 * g4;
      f32x4 bq = {0.f, 0.f, 0.f, 0.f};
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
        for (int e = 0; e < 4; ++e) acc16_set(accx[f][t], e, bq[e]);
    }
  }

  constexpr int UPRC = KCHT / 8;
  constexpr int MAXU = ((BT + HALO) * UPRC + NTHR - 1) / NTHR;   // halo <= HALO rows (the launcher refuses more)
  static_assert((KCHT / 32) % D == 0 || D % (KCHT / 32) == 0, "ring depth and steps per chunk and tap");
  WRing16<T, NF, D> ring;
  const int n_chunks = d.c_in / KCHT;
  ring.init((const T*)d.w, KC32, NFR16, nf0, d.k_w, KCHT / 32, n_chunks, lane);
  const size_t buf_bytes = (size_t)rows * pitch;
  StageRegs<float, MAXU, NIN> sr;
  // the staging plan: a unit's row does not change from chunk to chunk -- offsets and masks once (round 6: "issue loads" was 3.5 - 8 % of a workgroup's life)
  int64_t soff[MAXU];
  bool sok[MAXU];
  {
    const int total = rows * UPRC;
#pragma unroll
    for (int j = 0; j < MAXU; ++j) {
      const int u = threadIdx.x + j * NTHR;
      const int r = u / UPRC, cu = u % UPRC;
      int pos = t0 - d.pad + r;
      if (reflect) pos = reflect_pos(pos, L);
      sok[j] = u < total && pos >= 0 && pos < L;
      soff[j] = (seq_row0 + (sok[j] ? pos : 0)) * (int64_t)d.ldx + cu * 8;
#pragma unroll
      for (int i = 0; i < NIN; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) sr.v[i][j][e] = 0.f;     // a unit outside the sequence stays zero
    }
  }
  auto issue = [&](int c0) {
#pragma unroll
    for (int j = 0; j < MAXU; ++j)
#pragma unroll
      for (int i = 0; i < NIN; ++i)
        if (i < d.n_in && sok[j]) sr.v[i][j] = Vec8IO<float>::ldg(xin[i] + soff[j] + c0);
  };
  auto commit = [&](int ci) {
    emul_commit<T, MAXU, NIN, UPRC, NTHR>(sr, smem + (size_t)(ci & 1) * buf_bytes, pitch, rows, d.n_in, d.in_scale, d.pre_act, d.pre_slope);
  };
  // Eight-wave tiles (two waves per SIMD, ONE workgroup per CU): the second wave of every SIMD runs its staging in ANTI-PHASE -- it commits chunk c + 1 (loads
  // issued a whole chunk earlier) and issues chunk c + 2 BEFORE its K-steps of chunk c, the first wave after them: a wave's split arithmetic runs under its
  // neighbour's MFMAs instead of all eight waves converting at once with the matrix pipe idle.
  const bool early = JATTS_CEMUL_ANTIPHASE && WN * WT == 8 && OCC == 1 && wave >= 4;
  issue(0);
  commit(0);
  if (early && n_chunks > 1) issue(KCHT);
  __syncthreads();
  JATTS_PH(0);      // prologue: bias, ring fill, first chunk staged
  for (int ci = 0; ci < n_chunks; ++ci) {
    const bool more = ci + 1 < n_chunks;
    if (early) {
      if (more) commit(ci + 1);
      if (ci + 2 < n_chunks) issue((ci + 2) * KCHT);
    } else if (more && !(JATTS_CEMUL_DIAG & 2)) issue((ci + 1) * KCHT);
    JATTS_PH(1);    // issue of the next chunk's loads (early waves: the commit too)
    conv_stage16<T, NF, NT, D>(accx, ring, KCHT / 32, d.k_w, d.dil, smem + (size_t)(ci & 1) * buf_bytes, pitch, col0, lane);
    JATTS_PH(2);    // K-steps
#if JATTS_CEMUL_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    JATTS_PH(3);    // waiting for the loads (and the ring's look-ahead: L2 hits)
#endif
    if (!early && more && !(JATTS_CEMUL_DIAG & 64)) commit(ci + 1);
    JATTS_PH(4);    // split + LDS writes
    if (!(JATTS_CEMUL_DIAG & 4)) __syncthreads();
    JATTS_PH(5);    // barrier
  }
  if (JATTS_CEMUL_DIAG & 4) __syncthreads();

  // close the accumulators (seven products: one correctly rounded add per element): from here on f32 epilogues
  f32x4 acc[NF][NT];
#pragma unroll
  for (int f = 0; f < NF; ++f)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      acc16_finish(accx[f][t]);
      acc[f][t] = acc16_val(accx[f][t]);
    }
  if (d.act == JATTS_ACT_SNAKEBETA) snake_acc16<NF, NT>(acc, d.act_a, d.act_b, nf0, d.n_out, lane);
#if JATTS_CEMUL_TRACE
  auto trace_out = [&]() {
    JATTS_PH(6);    // epilogue
    if (!tracing) return;
    unsigned long long* o = trace + (size_t)blockIdx.x * 16;
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    o[0] = ((unsigned long long)(xcc & 0xF) << 32) | hwid;
    for (int i = 0; i < 7; ++i) o[1 + i] = ph[i];
    o[8] = rt_begin; o[9] = __builtin_amdgcn_s_memrealtime(); o[10] = t_begin; o[11] = tq;
  };
#else
  auto trace_out = []() {};
#endif
  if (JATTS_CEMUL_DIAG & 32) {     // DIAG: no epilogue (the sum keeps every accumulator live)
    float s = 0.f;
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int t = 0; t < NT; ++t) s += acc[f][t][0] + acc[f][t][1] + acc[f][t][2] + acc[f][t][3];
    if (s != 12345.678f) return;
  }
  {
    const int n_base = bz * BN;
    const bool rowmajor = !d.y_transposed && (d.n_out & 7) == 0 && (reinterpret_cast<uintptr_t>(d.y) & 15) == 0;
    if (rowmajor && f32_tile && (d.ldy & 3) == 0 && (!d.resid || ((d.ldr & 3) == 0 && (reinterpret_cast<uintptr_t>(d.resid) & 15) == 0))) {
      switch (d.act) {
        case JATTS_ACT_RELU: conv_epilogue_lds16<JATTS_ACT_RELU, NF, NT, BN>(d, acc, smem, t0, col0, wn * NF, n_base, lane, L, seq_row0, BT); break;
        case JATTS_ACT_TANH: conv_epilogue_lds16<JATTS_ACT_TANH, NF, NT, BN>(d, acc, smem, t0, col0, wn * NF, n_base, lane, L, seq_row0, BT); break;
        case JATTS_ACT_SWISH: conv_epilogue_lds16<JATTS_ACT_SWISH, NF, NT, BN>(d, acc, smem, t0, col0, wn * NF, n_base, lane, L, seq_row0, BT); break;
        case JATTS_ACT_MISH: conv_epilogue_lds16<JATTS_ACT_MISH, NF, NT, BN>(d, acc, smem, t0, col0, wn * NF, n_base, lane, L, seq_row0, BT); break;
        default: conv_epilogue_lds16<JATTS_ACT_NONE, NF, NT, BN>(d, acc, smem, t0, col0, wn * NF, n_base, lane, L, seq_row0, BT); break;
      }
      trace_out();
      return;
    }
  }
  switch (d.act) {
    case JATTS_ACT_RELU: conv_epilogue16<JATTS_ACT_RELU, NF, NT>(d, acc, t0, col0, nf0, lane, L, seq_row0, b); break;
    case JATTS_ACT_TANH: conv_epilogue16<JATTS_ACT_TANH, NF, NT>(d, acc, t0, col0, nf0, lane, L, seq_row0, b); break;
    case JATTS_ACT_SWISH: conv_epilogue16<JATTS_ACT_SWISH, NF, NT>(d, acc, t0, col0, nf0, lane, L, seq_row0, b); break;
    case JATTS_ACT_MISH: conv_epilogue16<JATTS_ACT_MISH, NF, NT>(d, acc, t0, col0, nf0, lane, L, seq_row0, b); break;
    default: conv_epilogue16<JATTS_ACT_NONE, NF, NT>(d, acc, t0, col0, nf0, lane, L, seq_row0, b); break;
  }
  trace_out();
}

template <typename T, int NF, int NT, int WN, int WT, int NIN, int KCHT, int OCC, int HALO = 32, int D = 2>
int launch_conv_emul16(const jatts_conv_desc& d, hipStream_t s) {
  constexpr int BT = WT * NT * 16, BN = WN * NF * 16;
  if ((d.k_w - 1) * d.dil > HALO) return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "conv1d (emulated): halo beyond the staging registers");
  if (d.c_in % KCHT) return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "conv1d (emulated): c_in must be a multiple of the chunk width");
  const int64_t maxL = (int64_t)d.rg.max_len * d.rg.len_mul;
  dim3 grid((unsigned)((maxL + BT - 1) / BT), (unsigned)d.rg.n_seq, (unsigned)((d.n_out + BN - 1) / BN));
  const size_t rows = (size_t)BT + (size_t)(d.k_w - 1) * d.dil;
  size_t lds = 2 * rows * (KCHT * 6 + 16);
  int f32_tile = 0;
  // Row-major outputs WITH a residual go through an f32 tile in LDS (row-contiguous residual reads and stores); without one the accumulators are stored
  // directly -- a lane's four channels are 16 bytes, a fragment row 64: +2 - 5 % on the wide no-residual shapes, -1 .. -3 % with a residual
  // (profiles/r06_conv16_direct_epilogue.txt; JATTS_CONV_EMUL16_DIRECT_EPI = 1 / 0 forces one path for A/B).  Same values either way.
  static const int epi_env = [] { const char* e = getenv("JATTS_CONV_EMUL16_DIRECT_EPI"); return e ? (atoi(e) != 0 ? 1 : 0) : -1; }();
  const bool direct_epi = epi_env >= 0 ? epi_env == 1 : d.resid == nullptr;
  if (!direct_epi && !d.y_transposed && (size_t)BT * (BN * 4 + 16) <= 159 * 1024) {     // the coalesced f32 output tile reuses the staging buffers
    f32_tile = 1;
    if (lds < (size_t)BT * (BN * 4 + 16)) lds = (size_t)BT * (BN * 4 + 16);
  }
  if (lds > 160 * 1024) return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "conv1d (emulated): tile exceeds 160 KiB LDS");
  auto kern = conv1d_emul16_kernel<T, NF, NT, WN, WT, NIN, KCHT, OCC, HALO, D>;
  // the dynamic-LDS limit ONCE per kernel (function-local static of this template instantiation), not per launch
  static const hipError_t lds_attr = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (lds_attr != hipSuccess) return jatts_set_error(lds_attr, __FILE__, __LINE__);
  XcdOrder xo;
  const int64_t total = xo.plan((int)grid.x, (int)grid.y, (int)grid.z, (int64_t)BN * d.c_in * d.k_w * 6, ragged_tiles_1d(d.rg, BT));
  if (total >= (int64_t)1 << 31) return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "conv1d: launch too large");
#if JATTS_CEMUL_TRACE
  hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(WN * WT * 64), lds, s, d, f32_tile, xo, jatts_g_trace, jatts_g_trace_cap);
#else
  hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(WN * WT * 64), lds, s, d, f32_tile, xo);
#endif
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

}  // namespace
