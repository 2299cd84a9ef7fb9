// Conv1d / Linear / ConvTranspose1d (polyphase) as MFMA implicit GEMM, and the fused
// HiFi-GAN ResBlock dilation unit.  gfx950 only.
//
// GEMM orientation: A operand = weights (M = output channel n), B operand =
// activations (N = time).  With 32x32 fragments a lane then owns, for ONE time step,
// groups of 4 consecutive output channels -> packed 8 B (f16) / 16 B (f32) stores into the
// time-major activation layout, and per-channel bias / residual reads vectorise.
//
// Weights arrive pre-packed in fragment order ([tap][c/16][n/32][lane][8], see
// jatts_conv_weight_index) so one A fragment is ONE fully coalesced 1 KiB (f16) wave load
// straight from L2 -- no LDS staging, no bank conflicts.  Activation tiles (+halo) are
// staged once per channel chunk in LDS with a (row bytes + 16) pitch: the 16-lane groups of
// ds_read_b128 then hit 16 distinct 16-B slots (pitch/16 is odd) -> conflict free.
#include <stdlib.h>

#ifndef JATTS_ABLATE
#define JATTS_ABLATE 0  // profiling-only ablations of the fused unit (tools/ablate_unit.sh); 0 = product
#endif

#include "common.h"

namespace {

template <typename T> struct Vec8IO;
template <> struct Vec8IO<f16> {
  static __device__ __forceinline__ f16x8 ldg(const f16* p) { return *reinterpret_cast<const f16x8*>(p); }
  static __device__ __forceinline__ f16x8 lds(const char* p) { return *reinterpret_cast<const f16x8*>(p); }
  static __device__ __forceinline__ void sts(char* p, const f16x8& v) { *reinterpret_cast<f16x8*>(p) = v; }
};
template <> struct Vec8IO<float> {
  static __device__ __forceinline__ f32x8 ldg(const float* p) {
    f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    return f32x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  }
  static __device__ __forceinline__ f32x8 lds(const char* p) {
    f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 16);
    return f32x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  }
  static __device__ __forceinline__ void sts(char* p, const f32x8& v) {
    *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p + 16) = f32x4{v[4], v[5], v[6], v[7]};
  }
};

// Stage `rows` x `nch` (multiple of 8) activations into LDS rows of `pitch` bytes.
// Source row for LDS row r is local position pos0 + r of a sequence of length L starting at
// global row seq_row0; positions outside [0, L) give zeros.  Up to 3 inputs are summed,
// scaled and passed through the optional leaky-ReLU before conversion to T.
// Loads are issued in batches of UB per thread BEFORE any of them is consumed: a one-load-
// per-iteration loop serialises a full L2/HBM round trip per 16 bytes (measured: the staging
// phases were as long as the MFMA phases).
template <typename T, int UB = 8>
__device__ __forceinline__ void stage_rows(char* lds, int pitch, int rows, int nch, int pos0, int L,
                                           int64_t seq_row0, const T* const* x, int n_in, int ldx,
                                           int c0, float in_scale, int pre_act, float slope) {
  typedef typename Elem<T>::vec8 V8;
  const int upr = nch >> 3;  // 8-element units per row
  const int total = rows * upr;
  const bool plain = n_in == 1 && in_scale == 1.f && pre_act == JATTS_PRE_NONE;
  for (int base = threadIdx.x; base < total; base += blockDim.x * UB) {
    V8 v[UB];
    int64_t off[UB];
    int dst[UB];
    bool ok[UB];
#pragma unroll
    for (int j = 0; j < UB; ++j) {
      const int u = base + j * blockDim.x;
      const int r = u / upr, cu = u - r * upr;
      const int pos = pos0 + r;
      ok[j] = u < total && pos >= 0 && pos < L;
      dst[j] = u < total ? r * pitch + cu * 8 * (int)sizeof(T) : -1;
      off[j] = (seq_row0 + pos) * (int64_t)ldx + c0 + cu * 8;
      if (ok[j]) v[j] = Vec8IO<T>::ldg(x[0] + off[j]);
      else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[j][e] = from_f32<T>(0.f);
      }
    }
    if (!plain) {
      V8 w1[UB], w2[UB];
      if (n_in > 1) {
#pragma unroll
        for (int j = 0; j < UB; ++j)
          if (ok[j]) w1[j] = Vec8IO<T>::ldg(x[1] + off[j]);
      }
      if (n_in > 2) {
#pragma unroll
        for (int j = 0; j < UB; ++j)
          if (ok[j]) w2[j] = Vec8IO<T>::ldg(x[2] + off[j]);
      }
#pragma unroll
      for (int j = 0; j < UB; ++j) {
        if (!ok[j]) continue;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float t = to_f32(v[j][e]);
          if (n_in > 1) t += to_f32(w1[j][e]);
          if (n_in > 2) t += to_f32(w2[j][e]);
          t *= in_scale;
          if (pre_act == JATTS_PRE_LRELU) t = lrelu(t, slope);
          v[j][e] = from_f32<T>(t);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < UB; ++j)
      if (dst[j] >= 0) Vec8IO<T>::sts(lds + dst[j], v[j]);
  }
}

// Split staging for the double-buffered conv pipeline: issue() starts the global loads of one
// channel chunk into registers (no wait), commit() combines / activates them and writes the LDS
// tile.  Between the two the workgroup computes the previous chunk, so the HBM/L2 latency of the
// activation stream hides under the MFMA phase (async-stage split).
template <typename T, int MAXU, int NIN>
struct StageRegs {
  typedef typename Elem<T>::vec8 V8;
  V8 v[NIN][MAXU];
};

template <typename T, int MAXU, int NIN, int UPR = 8, int NTHR = 256>
__device__ __forceinline__ void stage_issue(StageRegs<T, MAXU, NIN>& sr, int rows, int pos0, int L, int64_t seq_row0,
                                            const T* const* x, int n_in, int ldx, int c0) {
  // UPR = 8-element units per row of the chunk (8 for a 64-channel chunk)
  const int total = rows * UPR;
#pragma unroll
  for (int j = 0; j < MAXU; ++j) {
    const int u = threadIdx.x + j * NTHR;   // NTHR = blockDim.x at compile time: the index math folds
    const int r = u / UPR, cu = u % UPR;
    const int pos = pos0 + r;
    const bool ok = u < total && pos >= 0 && pos < L;
    const int64_t off = (seq_row0 + pos) * (int64_t)ldx + c0 + cu * 8;
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
      if (i < n_in && ok) sr.v[i][j] = Vec8IO<T>::ldg(x[i] + off);
      else {
#pragma unroll
        for (int e = 0; e < 8; ++e) sr.v[i][j][e] = from_f32<T>(0.f);
      }
    }
  }
}

template <typename T, int MAXU, int NIN, int UPR = 8, int NTHR = 256>
__device__ __forceinline__ void stage_commit(StageRegs<T, MAXU, NIN>& sr, char* lds, int pitch, int rows, int n_in,
                                             float in_scale, int pre_act, float slope) {
  const int total = rows * UPR;
  const bool plain = n_in == 1 && in_scale == 1.f && pre_act == JATTS_PRE_NONE;
#pragma unroll
  for (int j = 0; j < MAXU; ++j) {
    const int u = threadIdx.x + j * NTHR;
    if (u >= total) continue;
    const int r = u / UPR, cu = u % UPR;
    typename Elem<T>::vec8 o = sr.v[0][j];
    if (!plain) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float t = to_f32(sr.v[0][j][e]);
        if (NIN > 1 && n_in > 1) t += to_f32(sr.v[NIN > 1 ? 1 : 0][j][e]);
        if (NIN > 2 && n_in > 2) t += to_f32(sr.v[NIN > 2 ? 2 : 0][j][e]);
        t *= in_scale;
        if (pre_act == JATTS_PRE_LRELU) t = lrelu(t, slope);
        o[e] = from_f32<T>(t);
      }
    }
    Vec8IO<T>::sts(lds + (size_t)r * pitch + (size_t)cu * 8 * sizeof(T), o);
  }
}

// ---------------------------------------------------------------- pipelined MFMA inner loop
// Work is organised in GROUPS of KCG consecutive 16-channel steps of one tap.  The packed weight
// layout ([tap][c/16][n/32][lane][8]) makes the KCG fragments of a group contiguous (stride
// NFR*512 elements), so inside a group every address is base + immediate.  A register ring of
// KCG slots holds the current group's weight fragments; slot kk is refilled with fragment kk of
// the NEXT group right after its MFMAs (global -> VGPR, one coalesced 1 KiB wave load per
// fragment, KCG iterations ahead of use).  Activation fragments are read from LDS one step ahead
// (double buffer bb).  hipcc would otherwise wait vmcnt(0)/lgkmcnt right at each MFMA, and its
// scheduler sinks prefetches back next to their uses: sched_barrier(0) pins the issue points.
// Per MFMA the loop now carries ~2 non-MFMA instructions (was ~8 with per-iteration producer
// bookkeeping), which is what a single wave per SIMD can hide behind a 32-cycle MFMA.
template <typename T, int NF>
struct WFrags {
  int nfo[NF];  // element offset of this wave's n-fragments inside one 16-channel step
  int stride;   // elements between consecutive 16-channel steps (NFR * 512)
  __device__ __forceinline__ void init(int NFR, int nf0) {
#pragma unroll
    for (int f = 0; f < NF; ++f) nfo[f] = JATTS_ABLATE == 10 ? 0 : (nf0 + f < NFR ? nf0 + f : NFR - 1) * 512;  // clamped
    stride = JATTS_ABLATE == 10 ? 0 : NFR * 512;
  }
};

template <typename T, int NF, int KCG>
__device__ __forceinline__ void ring_fill(typename Elem<T>::vec8 (&ring)[KCG][NF], const WFrags<T, NF>& wf,
                                          const T* base) {
#pragma unroll
  for (int kk = 0; kk < KCG; ++kk)
#pragma unroll
    for (int f = 0; f < NF; ++f) ring[kk][f] = Vec8IO<T>::ldg(base + (size_t)kk * wf.stride + wf.nfo[f]);
}

template <typename T, int NT>
__device__ __forceinline__ void fetch_b(typename Elem<T>::vec8 (&dst)[NT], const char* p, int pitch) {
#pragma unroll
  for (int t = 0; t < NT; ++t) dst[t] = Vec8IO<T>::lds(p + (size_t)(t * 32) * pitch);
}

// One group: acc += W[group] x act.  On entry ring = this group's fragments and bb[0] = the
// activation fragments of its first step; on exit ring = next group's fragments (from
// next_base) and bb[0] = first step of the next group (from bnext).  bcur / bnext are the
// lane-adjusted LDS addresses of step 0 of this / the next group (steps are 32*sizeof(T)/2..
// 16 channels = 16*sizeof(T) bytes apart).
template <typename T, int NF, int NT, int KCG>
__device__ __forceinline__ void conv_group(f32x16 (&acc)[NF][NT], typename Elem<T>::vec8 (&ring)[KCG][NF],
                                           typename Elem<T>::vec8 (&bb)[2][NT], const WFrags<T, NF>& wf,
                                           const T* next_base, const char* bcur, const char* bnext, int pitch) {
  static_assert(KCG % 2 == 0, "group size must be even (bb parity)");
#pragma unroll
  for (int kk = 0; kk < KCG; ++kk) {
    if (kk + 1 < KCG) fetch_b<T, NT>(bb[(kk + 1) & 1], bcur + (size_t)(kk + 1) * 16 * sizeof(T), pitch);
    else fetch_b<T, NT>(bb[0], bnext, pitch);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int t = 0; t < NT; ++t) mma32(ring[kk][f], bb[kk & 1][t], acc[f][t]);
#pragma unroll
    for (int f = 0; f < NF; ++f) ring[kk][f] = Vec8IO<T>::ldg(next_base + (size_t)kk * wf.stride + wf.nfo[f]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// A whole conv over an LDS activation tile that holds ALL input channels (fused unit): groups run
// tap-major and are LINEAR in the packed weights.  KC16 = C/16 steps per tap, GPT = KC16/KCG.
template <typename T, int NF, int NT, int KC16, int KCG>
__device__ __forceinline__ void conv_full(f32x16 (&acc)[NF][NT], const T* __restrict__ w, int NFR, int nf0, int k_w,
                                          int dil, const char* act, int pitch, int col0, int lane) {
  typedef typename Elem<T>::vec8 V8;
  constexpr int GPT = KC16 / KCG;
  static_assert(GPT * KCG == KC16, "group size must divide the steps per tap");
  WFrags<T, NF> wf;
  wf.init(NFR, nf0);
  const T* wl = w + (size_t)lane * 8;
  const size_t gstride = JATTS_ABLATE == 10 ? 0 : (size_t)KCG * wf.stride;  // elements per group
  const int n_groups = k_w * GPT;
  const char* bbase = act + (size_t)(col0 + (lane & 31)) * pitch + (size_t)(8 * (lane >> 5)) * sizeof(T);
  V8 ring[KCG][NF], bb[2][NT];
  ring_fill<T, NF, KCG>(ring, wf, wl);
  fetch_b<T, NT>(bb[0], bbase, pitch);
  __builtin_amdgcn_sched_barrier(0);
  int g = 0;
  for (int tap = 0; tap < k_w; ++tap) {
    const char* btap = bbase + (size_t)(tap * dil) * pitch;
    const char* btap_next = bbase + (size_t)(min(tap + 1, k_w - 1) * dil) * pitch;
#pragma unroll
    for (int h = 0; h < GPT; ++h, ++g) {
      const T* nb = wl + (size_t)min(g + 1, n_groups - 1) * gstride;  // clamp: harmless re-read at the end
      const char* bcur = btap + (size_t)(h * KCG) * 16 * sizeof(T);
      const char* bnext = h + 1 < GPT ? btap + (size_t)((h + 1) * KCG) * 16 * sizeof(T) : btap_next;
      conv_group<T, NF, NT, KCG>(acc, ring, bb, wf, nb, bcur, bnext, pitch);
    }
  }
}

// Weight-fragment register ring for the generic conv (chunked activations).  Fragments are consumed
// in the order  for chunk: for tap: for kk  and the producer runs D iterations ahead of the consumer
// ACROSS chunk boundaries, so the loads for the next chunk are in flight while the activation tile
// is re-staged and the workgroup sits in its barriers.  Past the last fragment the producer clamps.
template <typename T, int NF, int D>
struct WRing {
  typedef typename Elem<T>::vec8 V8;
  V8 r[D][NF];
  const T* wbase;
  int nfo[NF];
  int KC16, NFR, k_w, kc_per, n_chunks;
  int p_chunk, p_tap, p_kk;

  __device__ __forceinline__ void init(const T* w, int KC16_, int NFR_, int nf0, int k_w_, int kc_per_,
                                       int n_chunks_, int lane) {
    wbase = w + (size_t)lane * 8;
    KC16 = KC16_; NFR = NFR_; k_w = k_w_; kc_per = kc_per_; n_chunks = n_chunks_;
#pragma unroll
    for (int f = 0; f < NF; ++f) nfo[f] = (nf0 + f < NFR ? nf0 + f : NFR - 1) * 512;  // clamped: never stored
    p_chunk = p_tap = p_kk = 0;
#pragma unroll
    for (int j = 0; j < D; ++j) fetch(r[j]);
  }
  __device__ __forceinline__ void fetch(V8 (&dst)[NF]) {
    const T* p = wbase + ((size_t)(p_tap * KC16 + p_chunk * kc_per + p_kk) * NFR) * 512;
#pragma unroll
    for (int f = 0; f < NF; ++f) dst[f] = Vec8IO<T>::ldg(p + nfo[f]);
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

// One chunk: consumes k_w*kc_per ring entries (a multiple of D, so every ring slot is a compile-time
// register index and the body is straight-line); activation fragments one step ahead from LDS.
template <typename T, int NF, int NT, int D>
__device__ __forceinline__ void conv_stage(f32x16 (&acc)[NF][NT], WRing<T, NF, D>& ring, int kc_per, int k_w,
                                           int dil, const char* act, int pitch, int col0, int lane) {
  typedef typename Elem<T>::vec8 V8;
  static_assert(D % 2 == 0, "ring depth must be even");
  const int g = lane >> 5;
  const int n_it = k_w * kc_per;
  const char* bbase = act + (size_t)(col0 + (lane & 31)) * pitch + (size_t)(8 * g) * sizeof(T);
  const int last_tap = k_w - 1;
  int bp_tap = 0, bp_kk = 0;  // activation producer position (clamps at the end)
  auto fetch_bb = [&](V8(&dst)[NT]) {
    const char* p = bbase + (size_t)(bp_tap * dil) * pitch + (size_t)(bp_kk * 16) * sizeof(T);
#pragma unroll
    for (int t = 0; t < NT; ++t) dst[t] = Vec8IO<T>::lds(p + (size_t)(t * 32) * pitch);
    const int nk = bp_kk + 1;
    const bool wrap = nk == kc_per;
    bp_kk = wrap ? 0 : nk;
    bp_tap = min(bp_tap + (wrap ? 1 : 0), last_tap);
  };
  V8 bb[2][NT];
  fetch_bb(bb[0]);
  __builtin_amdgcn_sched_barrier(0);
  for (int it0 = 0; it0 < n_it; it0 += D) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      fetch_bb(bb[(j + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int t = 0; t < NT; ++t) mma32(ring.r[j][f], bb[j & 1][t], acc[f][t]);
      ring.fetch(ring.r[j]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

template <int NF, int NT>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[NF][NT]) {
#pragma unroll
  for (int f = 0; f < NF; ++f)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[f][t][r] = 0.f;
}

// ------------------------------------------------------------------ generic conv kernel
constexpr int KCH = 64;  // channels staged per LDS chunk

// Conv epilogue: y = act(acc + bias) * alpha + resid.  Lane owns column (lane&31) and the channel quads
// n0 + {0..3}; quads fully inside n_out take the 16-byte bias / residual / store path, the ragged last quad
// (n_out % 4 != 0) a scalar loop.
template <typename T, int ACT, int NF, int NT>
__device__ __forceinline__ void conv_epilogue(const jatts_conv_desc& d, f32x16 (&acc)[NF][NT], int t0, int col0, int nf0,
                                              int lane, int L, int64_t seq_row0) {
  const int g = lane >> 5;
  const bool vec_r = d.resid && (d.ldr & 3) == 0, vec_y = (d.ldy & 3) == 0 && !d.y_transposed;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int pos = t0 + col0 + t * 32 + (lane & 31);
    if (pos >= L) continue;
    const int64_t row = seq_row0 + pos;
    const int64_t trow = d.y_seq_col0 ? (int64_t)d.y_seq_col0[blockIdx.y] * d.rg.len_mul + pos : row;  // transposed output column
#pragma unroll
    for (int f = 0; f < NF; ++f) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n0 = (nf0 + f) * 32 + 8 * q + 4 * g;
        if (n0 >= d.n_out) continue;
        const bool full = n0 + 3 < d.n_out;
        f32x4 rq = {0.f, 0.f, 0.f, 0.f};   // (the bias is already in the accumulators: conv1d_kernel's init)
        if (full && vec_r) rq = *reinterpret_cast<const f32x4*>(d.resid + row * d.ldr + n0);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = act_c<ACT>(acc[f][t][4 * q + e]) * d.alpha + rq[e];
        if (full && vec_y && (vec_r || !d.resid)) {
          const int64_t o = row * d.ldy + n0;
          if (d.y_is_f32 || sizeof(T) == 4) *reinterpret_cast<f32x4*>((float*)d.y + o) = v;
          else *reinterpret_cast<f16x4*>((f16*)d.y + o) = f16x4{(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int n = n0 + e;
            if (n >= d.n_out) break;
            float s = v[e];
            if (!full) s = act_c<ACT>(acc[f][t][4 * q + e]) * d.alpha;
            if (d.resid && !(full && vec_r)) s += d.resid[row * d.ldr + n];
            const int64_t o = d.y_transposed ? (int64_t)n * d.ldy + trow : row * d.ldy + n;
            if (d.y_is_f32) ((float*)d.y)[o] = s; else ((T*)d.y)[o] = from_f32<T>(s);
          }
        }
      }
      // keep the epilogue's live ranges short: without this hipcc hoists every bias / residual
      // load of the tile to the top and the kernel loses a wave of occupancy
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// Coalesced epilogue for row-major outputs (projections, FFN, the polyphase upsampling convs): act(acc + bias) * alpha
// is assembled as a [BT][BN] tile of the OUTPUT type in LDS (the activation buffers are dead), then the f32 residual
// is added and the tile written with row-contiguous 16-byte accesses.  In fragment order every 128-byte output line
// is otherwise hit by 8 separate 8-byte stores (and residual loads) from lanes 32 rows apart.
template <typename T, typename TO, int ACT, int NF, int NT, int BN>
__device__ __forceinline__ void conv_epilogue_lds(const jatts_conv_desc& d, f32x16 (&acc)[NF][NT], char* smem, int t0,
                                                  int col0, int nf_local0, int n_base, int lane, int L,
                                                  int64_t seq_row0, int BT) {
  constexpr int opitch = BN * (int)sizeof(TO) + 16;
  const int g = lane >> 5;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = col0 + t * 32 + (lane & 31);
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int nl = (nf_local0 + f) * 32 + 8 * q + 4 * g;   // channel inside the workgroup's BN slab
        if (n_base + nl >= d.n_out) continue;                   // n_out % 8 == 0: quads are all-or-nothing
        f32x4 o;   // the bias is already in the accumulators
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = act_c<ACT>(acc[f][t][4 * q + e]) * d.alpha;
        char* p = smem + (size_t)col * opitch + (size_t)nl * sizeof(TO);
        if (sizeof(TO) == 2) *reinterpret_cast<f16x4*>(p) = f16x4{(f16)o[0], (f16)o[1], (f16)o[2], (f16)o[3]};
        else *reinterpret_cast<f32x4*>(p) = o;
      }
  }
  __syncthreads();
  const int vrows = min(BT, L - t0);
  const int upr = min(BN, d.n_out - n_base) / 8;   // 8-element units per row
  const int total = vrows * upr;
  TO* yg = (TO*)d.y + (seq_row0 + t0) * (int64_t)d.ldy + n_base;
  const float* rg = d.resid ? d.resid + (seq_row0 + t0) * (int64_t)d.ldr + n_base : nullptr;
  for (int u = threadIdx.x; u < total; u += blockDim.x) {
    const int r = u / upr, cu = u - r * upr;
    const char* src = smem + (size_t)r * opitch + (size_t)cu * 8 * sizeof(TO);
    TO* dst = yg + (int64_t)r * d.ldy + cu * 8;
    if (sizeof(TO) == 2) {
      *reinterpret_cast<f16x8*>(dst) = *reinterpret_cast<const f16x8*>(src);   // (no residual on this path)
    } else {
      f32x4 lo = *reinterpret_cast<const f32x4*>(src), hi = *reinterpret_cast<const f32x4*>(src + 16);
      if (rg) {
        const float* rp = rg + (int64_t)r * d.ldr + cu * 8;
        lo += *reinterpret_cast<const f32x4*>(rp);
        hi += *reinterpret_cast<const f32x4*>(rp + 4);
      }
      *reinterpret_cast<f32x4*>(dst) = lo;
      *reinterpret_cast<f32x4*>((float*)dst + 4) = hi;
    }
  }
}

template <typename T, int NF, int NT, int WN, int WT, int NIN, bool ASYNC, int KCHT = KCH>
__global__ __launch_bounds__(WN* WT * 64, KCHT == 128 ? 2 : 1) void conv1d_kernel(jatts_conv_desc d, int f32_tile) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BT = WT * NT * 32;
  const int b = blockIdx.y;
  const int row_b = d.rg.cu_rows[b];
  const int L = (d.rg.cu_rows[b + 1] - row_b) * d.rg.len_mul;
  const int t0 = blockIdx.x * BT;
  if (t0 >= L) return;
  const int64_t seq_row0 = (int64_t)row_b * d.rg.len_mul;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wn = wave / WT, wt = wave % WT;
  const int pitch = KCHT * (int)sizeof(T) + 16;
  const int rows = BT + (d.k_w - 1) * d.dil;
  const int KC16 = d.c_in >> 4;
  const int n_pad = (d.n_out + 31) & ~31;
  const int NFR = n_pad >> 5;
  const int nf0 = (blockIdx.z * WN + wn) * NF;
  const int col0 = wt * NT * 32;

  const T* xin[3] = {(const T*)d.x[0], (const T*)d.x[1], (const T*)d.x[2]};
  f32x16 acc[NF][NT];
  zero_acc<NF, NT>(acc);
  if (d.bias) {   // accumulators start at the bias: its loads overlap the first staging round trip instead of the epilogue
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
          for (int e = 0; e < 4; ++e) acc[f][t][4 * q + e] = bq[e];
      }
  }

  // staged 8-element units per thread in the async pipeline (halo <= 32 rows; larger halos take
  // the synchronous single-buffer path)
  constexpr int UPRC = KCHT / 8;
  constexpr int MAXU = ((BT + 32) * UPRC + WN * WT * 64 - 1) / (WN * WT * 64);
  constexpr int RD = KCHT / 16;  // ring depth: k_w * (KCHT/16) is always a multiple of it
  WRing<T, NF, RD> ring;
  const int n_chunks = d.c_in / KCHT;
  ring.init((const T*)d.w, KC16, NFR, nf0, d.k_w, KCHT / 16, n_chunks, lane);
  if constexpr (ASYNC) {
    const size_t buf_bytes = (size_t)rows * pitch;
    StageRegs<T, MAXU, NIN> sr;
    stage_issue<T, MAXU, NIN, UPRC, WN * WT * 64>(sr, rows, t0 - d.pad, L, seq_row0, xin, d.n_in, d.ldx, 0);
    stage_commit<T, MAXU, NIN, UPRC, WN * WT * 64>(sr, smem, pitch, rows, d.n_in, d.in_scale, d.pre_act, d.pre_slope);
    __syncthreads();
    for (int ci = 0; ci < n_chunks; ++ci) {
      const bool more = ci + 1 < n_chunks;
      if (more) stage_issue<T, MAXU, NIN, UPRC, WN * WT * 64>(sr, rows, t0 - d.pad, L, seq_row0, xin, d.n_in, d.ldx, (ci + 1) * KCHT);
      conv_stage<T, NF, NT, RD>(acc, ring, KCHT / 16, d.k_w, d.dil, smem + (size_t)(ci & 1) * buf_bytes, pitch, col0,
                                lane);
      if (more) stage_commit<T, MAXU, NIN, UPRC, WN * WT * 64>(sr, smem + (size_t)((ci + 1) & 1) * buf_bytes, pitch, rows, d.n_in,
                                           d.in_scale, d.pre_act, d.pre_slope);
      __syncthreads();
    }
  } else {
    for (int ci = 0; ci < n_chunks; ++ci) {
      stage_rows<T>(smem, pitch, rows, KCHT, t0 - d.pad, L, seq_row0, xin, d.n_in, d.ldx, ci * KCHT, d.in_scale,
                    d.pre_act, d.pre_slope);
      __syncthreads();
      conv_stage<T, NF, NT, RD>(acc, ring, KCHT / 16, d.k_w, d.dil, smem, pitch, col0, lane);
      __syncthreads();
    }
  }

  // epilogue, specialised per activation by ONE uniform branch: a runtime switch inside the 64-element
  // unrolled body inlined tanh/mish 128 times, the unroller gave up and the accumulators went to scratch
  // (1.8x slower conv, profiles/r01_notes.md).
  if constexpr (sizeof(T) == 2) {
    // the loop's last barrier has retired every read of the activation buffers: reuse them as the output tile
    constexpr int BN = WN * NF * 32;
    const int n_base = blockIdx.z * BN;
    const bool rowmajor = !d.y_transposed && (d.n_out & 7) == 0 && (reinterpret_cast<uintptr_t>(d.y) & 15) == 0;
#define JATTS_EPI(TO)                                                                                                  \
  switch (d.act) {                                                                                                     \
    case JATTS_ACT_RELU: conv_epilogue_lds<T, TO, JATTS_ACT_RELU, NF, NT, BN>(d, acc, smem, t0, col0, wn * NF, n_base, lane, L, seq_row0, BT); break;  \
    case JATTS_ACT_TANH: conv_epilogue_lds<T, TO, JATTS_ACT_TANH, NF, NT, BN>(d, acc, smem, t0, col0, wn * NF, n_base, lane, L, seq_row0, BT); break;  \
    case JATTS_ACT_SWISH: conv_epilogue_lds<T, TO, JATTS_ACT_SWISH, NF, NT, BN>(d, acc, smem, t0, col0, wn * NF, n_base, lane, L, seq_row0, BT); break; \
    case JATTS_ACT_MISH: conv_epilogue_lds<T, TO, JATTS_ACT_MISH, NF, NT, BN>(d, acc, smem, t0, col0, wn * NF, n_base, lane, L, seq_row0, BT); break;  \
    default: conv_epilogue_lds<T, TO, JATTS_ACT_NONE, NF, NT, BN>(d, acc, smem, t0, col0, wn * NF, n_base, lane, L, seq_row0, BT); break;              \
  }
    if (rowmajor && !d.y_is_f32 && !d.resid && (d.ldy & 7) == 0) {
      JATTS_EPI(T)
      return;
    }
    if (rowmajor && d.y_is_f32 && f32_tile && (d.ldy & 3) == 0 &&
        (!d.resid || ((d.ldr & 3) == 0 && (reinterpret_cast<uintptr_t>(d.resid) & 15) == 0))) {
      JATTS_EPI(float)
      return;
    }
#undef JATTS_EPI
  }
  switch (d.act) {
    case JATTS_ACT_RELU: conv_epilogue<T, JATTS_ACT_RELU, NF, NT>(d, acc, t0, col0, nf0, lane, L, seq_row0); break;
    case JATTS_ACT_TANH: conv_epilogue<T, JATTS_ACT_TANH, NF, NT>(d, acc, t0, col0, nf0, lane, L, seq_row0); break;
    case JATTS_ACT_SWISH: conv_epilogue<T, JATTS_ACT_SWISH, NF, NT>(d, acc, t0, col0, nf0, lane, L, seq_row0); break;
    case JATTS_ACT_MISH: conv_epilogue<T, JATTS_ACT_MISH, NF, NT>(d, acc, t0, col0, nf0, lane, L, seq_row0); break;
    default: conv_epilogue<T, JATTS_ACT_NONE, NF, NT>(d, acc, t0, col0, nf0, lane, L, seq_row0); break;
  }
}

template <typename T, int NF, int NT, int WN, int WT, int NIN, bool ASYNC, int KCHT = KCH>
int launch_conv_k(const jatts_conv_desc& d, hipStream_t s) {
  constexpr int BT = WT * NT * 32, BN = WN * NF * 32;
  const int64_t maxL = (int64_t)d.rg.max_len * d.rg.len_mul;
  dim3 grid((unsigned)((maxL + BT - 1) / BT), (unsigned)d.rg.n_seq, (unsigned)((d.n_out + BN - 1) / BN));
  const size_t rows = (size_t)BT + (size_t)(d.k_w - 1) * d.dil;
  size_t lds = (ASYNC ? 2 : 1) * rows * (KCHT * sizeof(T) + 16);
  // output tile of the coalesced epilogues (f16 kernels): T-typed always, f32 (row-major f32 outputs, e.g. the
  // in-place residual-stream updates) when the conv is long enough that the bigger LDS footprint does not matter
  int f32_tile = 0;
  if (sizeof(T) == 2) {
    if (lds < (size_t)BT * (BN * sizeof(T) + 16)) lds = (size_t)BT * (BN * sizeof(T) + 16);
    if (d.y_is_f32 && !d.y_transposed) {
      f32_tile = 1;
      if (lds < (size_t)BT * (BN * 4 + 16)) lds = (size_t)BT * (BN * 4 + 16);
    }
  }
  if (lds > 160 * 1024) return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "conv1d: halo too large for LDS");
  auto kern = conv1d_kernel<T, NF, NT, WN, WT, NIN, ASYNC, KCHT>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return jatts_set_error(e, __FILE__, __LINE__);
  }
  hipLaunchKernelGGL(kern, grid, dim3(WN * WT * 64), lds, s, d, f32_tile);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

template <typename T, int NF, int NT, int WN, int WT>
int launch_conv(const jatts_conv_desc& d, hipStream_t s) {
  const bool small_halo = (d.k_w - 1) * d.dil <= 32;
  const bool multi_chunk = d.c_in > KCH;  // a single chunk has nothing to overlap with
  if constexpr (sizeof(T) == 2 && NF == 2 && NT == 2 && WN == 2) {
    // 128-channel chunks for deep-K convs: half as many stage -> barrier -> MFMA round trips per workgroup (a k=1
    // projection with K=384 is otherwise 6 latency-bound chunk iterations around 2 us of MFMA work)
    static const int kch = [] { const char* e = getenv("JATTS_CONV_KCH"); return e ? atoi(e) : 2; }();  // 0: never, 1: k=1 only, 2: all (default)
    if (small_halo && d.n_in == 1 && d.c_in % 128 == 0 && d.c_in >= 256 && (kch == 2 || (kch == 1 && d.k_w == 1)))
      return launch_conv_k<T, NF, NT, WN, WT, 1, true, 128>(d, s);
  }
  if (small_halo && multi_chunk && d.n_in == 1) return launch_conv_k<T, NF, NT, WN, WT, 1, true>(d, s);
  if constexpr (sizeof(T) == 2) {
    if (small_halo && multi_chunk) return launch_conv_k<T, NF, NT, WN, WT, 3, true>(d, s);
  }
  return launch_conv_k<T, NF, NT, WN, WT, 3, false>(d, s);
}

// Unit-kernel output pass: y = (acc + b2 tile in LDS) + x [+ MRF partners] with row-contiguous 16-byte accesses;
// all global reads of a batch are issued before any is consumed (one round trip per batch, not per unit).
template <typename T, int C, int UB, bool ADD, int NTHR>
__device__ __forceinline__ void unit_store_pass(const jatts_resunit_desc& d, const char* ys, int pitch, int vrows,
                                                const T* xg, T* yg, int64_t g0) {
  typedef typename Elem<T>::vec8 V8;
  constexpr int UPR = C / 8;
  const int total = vrows * UPR;
  const bool has_add1 = ADD && d.add1 != nullptr;
  for (int u0 = threadIdx.x; u0 < total; u0 += UB * NTHR) {
    V8 xr[UB], a0[ADD ? UB : 1], a1[ADD ? UB : 1];
#pragma unroll
    for (int i = 0; i < UB; ++i) {
      const int u = u0 + i * NTHR;
      if (u < total) {
        if (JATTS_ABLATE != 3) xr[i] = Vec8IO<T>::ldg(xg + g0 + (int64_t)u * 8);
        if (ADD) {
          a0[i] = Vec8IO<T>::ldg((const T*)d.add0 + g0 + (int64_t)u * 8);
          if (has_add1) a1[i] = Vec8IO<T>::ldg((const T*)d.add1 + g0 + (int64_t)u * 8);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < UB; ++i) {
      const int u = u0 + i * NTHR;
      if (u >= total) continue;
      const int r = u / UPR, cu = u - r * UPR;
      V8 v = Vec8IO<T>::lds(ys + (size_t)r * pitch + (size_t)cu * 8 * sizeof(T));
      if (JATTS_ABLATE != 3) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = from_f32<T>(to_f32(v[e]) + to_f32(xr[i][e]));  // residual
      }
      if (ADD) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          v[e] = from_f32<T>((to_f32(v[e]) + to_f32(a0[i][e]) + (has_add1 ? to_f32(a1[i][e]) : 0.f)) * d.out_scale);
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
static unsigned long long* g_trace = nullptr;  // profiling hook, see jatts_debug_trace
static unsigned g_trace_cap = 0;
template <typename T, int C, int WGCOLS, int WN, int NT, int KCGMAX = 8>
__global__ __launch_bounds__(WN*(WGCOLS / (NT * 32)) * 64, (C <= 256 && (C / (WN * 32)) * NT * 16 <= 128) ? 2 : 1) void resunit_kernel(jatts_resunit_desc d, unsigned long long* trace, unsigned trace_cap, unsigned bias_off) {
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

  const int b = blockIdx.y;
  const int row_b = d.rg.cu_rows[b];
  const int L = (d.rg.cu_rows[b + 1] - row_b) * d.rg.len_mul;
  const int t0 = blockIdx.x * tt_out;
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

  const T* xin[3] = {(const T*)d.x, nullptr, nullptr};
  if (JATTS_ABLATE != 2 && JATTS_ABLATE != 7 && JATTS_ABLATE != 12)
  {
    constexpr int NTHR = WN * WT * 64;
    constexpr int UBX = ((WGCOLS + 64) * (C / 8) + NTHR - 1) / NTHR;   // covers halos up to 32 rows a side in one batch
    stage_unit<T, (UBX < 8 ? 8 : (UBX < 24 ? UBX : 24)), NTHR>(xs, pitch, rx, C / 8, t0 - p2 - p1, L, seq_row0, xin[0], C, JATTS_ABLATE != 1, d.slope);
  }
  __syncthreads();
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
  if (JATTS_ABLATE < 6 || JATTS_ABLATE > 9) conv_full<T, NF, NT, KC16, KCG>(acc, (const T*)d.w1, NFR, nf0, K, dil, xs, pitch, col0, lane);

  JATTS_STAMP(3);
  // epilogue 1: h = lrelu(acc + b1), forced to 0 outside the sequence (conv2's zero padding)
  __syncthreads();  // every wave is done reading x: the tile may now be overwritten by h
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
  __syncthreads();
  JATTS_STAMP(4);

  bias_acc(bs + C);
  if (JATTS_ABLATE < 6 || JATTS_ABLATE > 9) conv_full<T, NF, NT, KC16, KCG>(acc, (const T*)d.w2, NFR, nf0, K, 1, hs, pitch, col0, lane);

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
          for (int e = 0; e < 4; ++e) o[e] = (f16)acc[f][t][4 * q + e];
          *reinterpret_cast<f16x4*>(p) = o;
        } else {
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = acc[f][t][4 * q + e];
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
    if (d.add0) unit_store_pass<T, C, keep_small ? 2 : 4, true, WN * WT * 64>(d, ys, pitch, vrows, xg, yg, g0);   // + fused MRF mean
    else unit_store_pass<T, C, keep_small ? 4 : 8, false, WN * WT * 64>(d, ys, pitch, vrows, xg, yg, g0);
  }
  JATTS_STAMP(7);
  if (tracing) trace[(size_t)wg_lin * 16 + 9] = __builtin_amdgcn_s_memrealtime();
#undef JATTS_STAMP
}

template <typename T, int C, int WGCOLS, int WN, int NT, int KCGMAX = 8>
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
  auto kern = resunit_kernel<T, C, WGCOLS, WN, NT, KCGMAX>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return jatts_set_error(e, __FILE__, __LINE__);
  }
  hipLaunchKernelGGL(kern, grid, dim3(WN * WT * 64), lds, s, d, g_trace, g_trace_cap, bias_off);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

}  // namespace

extern "C" int jatts_debug_trace(void* buf, int64_t n_workgroups) {
  g_trace = (unsigned long long*)buf;
  g_trace_cap = buf ? (unsigned)n_workgroups : 0u;
  return JATTS_OK;
}

extern "C" int64_t jatts_conv_weight_index(int32_t n, int32_t tap, int32_t c, int32_t n_pad, int32_t c_in) {
  const int64_t KC16 = c_in / 16, NFR = n_pad / 32;
  const int64_t kc = c / 16, nf = n / 32;
  const int64_t lane = 32 * ((c % 16) / 8) + (n % 32);
  return ((((int64_t)tap * KC16 + kc) * NFR + nf) * 64 + lane) * 8 + (c % 8);
}

extern "C" int jatts_conv1d(const jatts_conv_desc* d, void* stream) {
  if (!d || !d->x[0] || !d->w || !d->y || !d->rg.cu_rows) return jatts_set_error_msg(JATTS_ERR_ARG, "conv1d: null pointer");
  if (d->c_in <= 0 || d->c_in % 64) return jatts_set_error_msg(JATTS_ERR_ARG, "conv1d: c_in must be a positive multiple of 64");
  if (d->ldx % 8) return jatts_set_error_msg(JATTS_ERR_ARG, "conv1d: ldx must be a multiple of 8");
  if (d->n_in < 1 || d->n_in > 3 || d->k_w < 1 || d->dil < 1 || d->n_out < 1 || d->rg.n_seq < 1 || d->rg.len_mul < 1)
    return jatts_set_error_msg(JATTS_ERR_ARG, "conv1d: bad geometry");
  if (d->rg.max_len <= 0) return JATTS_OK;
  hipStream_t s = (hipStream_t)stream;
  const bool narrow = d->n_out <= 64;
  // JATTS_CONV_VARIANT (tuning knob): 0 = 128n x 128t tile (64 accumulators/lane, 2-3 workgroups/CU;
  // measured 1.7x faster end to end), 1 = 128n x 256t register tile (1 workgroup/CU)
  static const int variant = [] { const char* e = getenv("JATTS_CONV_VARIANT"); return e ? atoi(e) : 0; }();
  if (d->dtype == JATTS_F16) {
    if (narrow) return launch_conv<f16, 2, 2, 1, 4>(*d, s);
    return variant == 1 ? launch_conv<f16, 2, 4, 2, 2>(*d, s) : launch_conv<f16, 2, 2, 2, 2>(*d, s);
  } else if (d->dtype == JATTS_F32) {
    return narrow ? launch_conv<float, 2, 2, 1, 4>(*d, s) : launch_conv<float, 2, 2, 2, 2>(*d, s);
  }
  return jatts_set_error_msg(JATTS_ERR_ARG, "conv1d: unknown dtype");
}

extern "C" int jatts_hifigan_resunit(const jatts_resunit_desc* d, void* stream) {
  if (!d || !d->x || !d->y || !d->w1 || !d->w2 || !d->b1 || !d->b2 || !d->rg.cu_rows)
    return jatts_set_error_msg(JATTS_ERR_ARG, "resunit: null pointer");
  if (d->x == d->y) return jatts_set_error_msg(JATTS_ERR_ARG, "resunit: y must not alias x");
  if (d->k_w < 1 || !(d->k_w & 1) || d->dil < 1) return jatts_set_error_msg(JATTS_ERR_ARG, "resunit: odd k_w and dil>=1 required");
  if (!(d->slope >= 0.f && d->slope <= 1.f)) return jatts_set_error_msg(JATTS_ERR_ARG, "resunit: LeakyReLU slope must be in [0, 1]");
  if (d->rg.max_len <= 0) return JATTS_OK;
  hipStream_t s = (hipStream_t)stream;
  if (d->dtype == JATTS_F16) {
    // tile variants: <C, workgroup columns, waves along n, 32-col fragments per wave>.
    // JATTS_RESUNIT_VARIANT (tuning knob, read once) selects alternative tilings for sweeps.
    static const int variant = [] { const char* e = getenv("JATTS_RESUNIT_VARIANT"); return e ? atoi(e) : 0; }();
    // Default (variant 0) = the fastest tiling measured per shape (profiles/r01_notes.md): 4 time fragments
    // per wave where the accumulators still allow 2 waves/SIMD -- every weight fragment fetched through the
    // 64 B/clk vector L1 then feeds 4 MFMAs instead of 2, which is what bounds the 128/256-channel units.
    const bool wide_k = d->k_w > 3;
    switch (d->channels * 10 + variant) {
      case 320: return wide_k && d->k_w > 7 ? launch_resunit<f16, 32, 512, 1, 4, 2>(*d, s) : launch_resunit<f16, 32, 256, 1, 2>(*d, s);
      case 321: return launch_resunit<f16, 32, 256, 1, 2>(*d, s);
      case 323: return launch_resunit<f16, 32, 512, 1, 4, 2>(*d, s);
      case 640: return wide_k ? launch_resunit<f16, 64, 512, 1, 4, 4>(*d, s) : launch_resunit<f16, 64, 256, 1, 2>(*d, s);
      case 641: return launch_resunit<f16, 64, 256, 1, 2>(*d, s);
      case 643: return launch_resunit<f16, 64, 256, 1, 4, 4>(*d, s);
      case 644: return launch_resunit<f16, 64, 512, 1, 4, 4>(*d, s);
      case 1280:
        // 2 x (256 + 2*p1) rows x 272 B must fit in 160 KiB for 2 workgroups/CU: k=11, d=5 misses by 5 rows -> 3-fragment tile
        if (((256 + (d->k_w - 1) * d->dil) * 272 + 1024) * 2 > 160 * 1024) return launch_resunit<f16, 128, 192, 2, 3, 4>(*d, s);
        return launch_resunit<f16, 128, 256, 2, 4, 4>(*d, s);
      case 1283: return launch_resunit<f16, 128, 256, 2, 4, 4>(*d, s);
      case 1281: return launch_resunit<f16, 128, 128, 2, 2, 8>(*d, s);
      case 1284: return launch_resunit<f16, 128, 128, 2, 4, 4>(*d, s);
      case 2560: case 2563: return launch_resunit<f16, 256, 128, 4, 4, 4>(*d, s);
      case 2561: return launch_resunit<f16, 256, 64, 4, 2, 8>(*d, s);
      case 2564: return launch_resunit<f16, 256, 128, 4, 2, 4>(*d, s);
      case 1285: return launch_resunit<f16, 128, 192, 2, 3, 4>(*d, s);
      case 5120: return launch_resunit<f16, 512, 32, 4, 1>(*d, s);
    }
  } else if (d->dtype == JATTS_F32) {
    switch (d->channels) {
      case 32: return launch_resunit<float, 32, 256, 1, 2>(*d, s);
      case 64: return launch_resunit<float, 64, 128, 1, 2>(*d, s);
      case 128: return launch_resunit<float, 128, 64, 2, 2>(*d, s);
      case 256: return launch_resunit<float, 256, 32, 4, 1>(*d, s);
    }
  }
  return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "resunit: unsupported channels/dtype (use jatts_conv1d)");
}
