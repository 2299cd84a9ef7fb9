// Shared building blocks of the MFMA implicit-GEMM kernels: LDS staging, weight-fragment register rings and the
// pipelined inner loops.  Included by conv1d_impl.h (generic conv) and resunit_impl.h (fused HiFi-GAN unit).
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
#pragma once
#include <stdlib.h>

// Profiling-only ablations of the fused unit (tools/ablate_unit.sh) exist in DIAGNOSIS builds only (-DJATTS_DIAG -DJATTS_ABLATE=n, written to
// a separate library): in the shipped library the switch is the constant 0 whatever the command line says, so every ablated branch is dead code.
#if !defined(JATTS_DIAG) || !defined(JATTS_ABLATE)
#undef JATTS_ABLATE
#define JATTS_ABLATE 0
#endif

#include "common.h"

// s_setprio around the MFMA clusters (MI355X guide T5: the scheduler then prefers the wave that is feeding the matrix pipe over a co-resident
// wave in its staging / epilogue phase).  -DJATTS_SETPRIO=1 switches it on (A/B builds: make OUT=../lib/prio CXXFLAGS+=-DJATTS_SETPRIO=1).
#ifndef JATTS_SETPRIO
#define JATTS_SETPRIO 0
#endif
#if JATTS_SETPRIO
#define JATTS_PRIO(p) __builtin_amdgcn_s_setprio(p)
#else
#define JATTS_PRIO(p) ((void)0)
#endif

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

template <> struct Vec8IO<f16s> {   // 32 bytes per 8 elements: [hi x8 | lo x8]
  static __device__ __forceinline__ f16sx8 ldg(const f16s* p) {
    const f16x8* q = reinterpret_cast<const f16x8*>(p);
    return f16sx8{q[0], q[1]};
  }
  static __device__ __forceinline__ f16sx8 lds(const char* p) {
    return f16sx8{*reinterpret_cast<const f16x8*>(p), *reinterpret_cast<const f16x8*>(p + 16)};
  }
  static __device__ __forceinline__ void sts(char* p, const f16sx8& v) {
    *reinterpret_cast<f16x8*>(p) = v.hi;
    *reinterpret_cast<f16x8*>(p + 16) = v.lo;
  }
};

template <int NP> struct Vec8IO<bf3p<NP>> {    // 48 bytes per 8 elements: [b0 x8 | b1 x8 | b2 x8]
  static __device__ __forceinline__ bf3px8<NP> ldg(const bf3p<NP>* p) {
    const bf16x8* q = reinterpret_cast<const bf16x8*>(p);
    return bf3px8<NP>{q[0], q[1], q[2]};
  }
  static __device__ __forceinline__ bf3px8<NP> lds(const char* p) {
    return bf3px8<NP>{*reinterpret_cast<const bf16x8*>(p), *reinterpret_cast<const bf16x8*>(p + 16), *reinterpret_cast<const bf16x8*>(p + 32)};
  }
  static __device__ __forceinline__ void sts(char* p, const bf3px8<NP>& v) {
    *reinterpret_cast<bf16x8*>(p) = v.b0;
    *reinterpret_cast<bf16x8*>(p + 16) = v.b1;
    *reinterpret_cast<bf16x8*>(p + 32) = v.b2;
  }
};

// Stage `rows` x `nch` (multiple of 8) activations into LDS rows of `pitch` bytes.
// Source row for LDS row r is local position pos0 + r of a sequence of length L starting at
// global row seq_row0; positions outside [0, L) give zeros.  Up to 3 inputs are summed,
// scaled and passed through the optional leaky-ReLU before conversion to T.
// Loads are issued in batches of UB per thread BEFORE any of them is consumed: a one-load-
// per-iteration loop serialises a full L2/HBM round trip per 16 bytes (measured: the staging
// phases were as long as the MFMA phases).
// Reflect padding (torch padding_mode="reflect": no edge repeat): position p of a length-L sequence, |overshoot| < L.
__device__ __forceinline__ int reflect_pos(int p, int L) {
  if (p < 0) p = -p;
  if (p >= L) p = 2 * (L - 1) - p;
  return p;
}

template <typename T, int UB = 8>
__device__ __forceinline__ void stage_rows(char* lds, int pitch, int rows, int nch, int pos0, int L,
                                           int64_t seq_row0, const T* const* x, int n_in, int ldx,
                                           int c0, float in_scale, int pre_act, float slope, bool reflect = false) {
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
      int pos = pos0 + r;
      if (reflect) pos = reflect_pos(pos, L);
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
                                            const T* const* x, int n_in, int ldx, int c0, bool reflect = false) {
  // UPR = 8-element units per row of the chunk (8 for a 64-channel chunk)
  const int total = rows * UPR;
#pragma unroll
  for (int j = 0; j < MAXU; ++j) {
    const int u = threadIdx.x + j * NTHR;   // NTHR = blockDim.x at compile time: the index math folds
    const int r = u / UPR, cu = u % UPR;
    int pos = pos0 + r;
    if (reflect) pos = reflect_pos(pos, L);
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
__device__ __forceinline__ void conv_group(typename Acc32<T>::type (&acc)[NF][NT], typename Elem<T>::vec8 (&ring)[KCG][NF],
                                           typename Elem<T>::vec8 (&bb)[2][NT], const WFrags<T, NF>& wf,
                                           const T* next_base, const char* bcur, const char* bnext, int pitch) {
  static_assert(KCG % 2 == 0, "group size must be even (bb parity)");
#pragma unroll
  for (int kk = 0; kk < KCG; ++kk) {
    if (kk + 1 < KCG) fetch_b<T, NT>(bb[(kk + 1) & 1], bcur + (size_t)(kk + 1) * 16 * sizeof(T), pitch);
    else fetch_b<T, NT>(bb[0], bnext, pitch);
    __builtin_amdgcn_sched_barrier(0);
    JATTS_PRIO(1);
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int t = 0; t < NT; ++t) mma32(ring[kk][f], bb[kk & 1][t], acc[f][t]);
    JATTS_PRIO(0);
#pragma unroll
    for (int f = 0; f < NF; ++f) ring[kk][f] = Vec8IO<T>::ldg(next_base + (size_t)kk * wf.stride + wf.nfo[f]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// A whole conv over an LDS activation tile that holds ALL input channels (fused unit): groups run
// tap-major and are LINEAR in the packed weights.  KC16 = C/16 steps per tap, GPT = KC16/KCG.
template <typename T, int NF, int NT, int KC16, int KCG>
__device__ __forceinline__ void conv_full(typename Acc32<T>::type (&acc)[NF][NT], const T* __restrict__ w, int NFR, int nf0, int k_w,
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

// Weight stream across the convs of one fused kernel: the ring outlives a conv, so the first group of the NEXT conv
// is fetched during the last group of the current one and its L2 round trip hides under the epilogue between them
// (with 32-64 channels a conv is only ~0.8-3 k cycles of MFMA: an exposed ~600-cycle ring fill per conv is a large tax).
template <typename T, int NF, int KCG>
struct WStream {
  typename Elem<T>::vec8 ring[KCG][NF];
  WFrags<T, NF> wf;
  __device__ __forceinline__ void prefetch(const T* w, int NFR, int nf0, int lane) {
    wf.init(NFR, nf0);
    ring_fill<T, NF, KCG>(ring, wf, w + (size_t)lane * 8);
  }
};

// conv_full on a pre-filled stream; w_next (or nullptr) = packed weights of the conv that follows.
template <typename T, int NF, int NT, int KC16, int KCG>
__device__ __forceinline__ void conv_full_ws(typename Acc32<T>::type (&acc)[NF][NT], WStream<T, NF, KCG>& ws, const T* __restrict__ w,
                                             const T* __restrict__ w_next, int k_w, int dil, const char* act, int pitch,
                                             int col0, int lane) {
  typedef typename Elem<T>::vec8 V8;
  constexpr int GPT = KC16 / KCG;
  static_assert(GPT * KCG == KC16, "group size must divide the steps per tap");
  const T* wl = w + (size_t)lane * 8;
  const size_t gstride = (size_t)KCG * ws.wf.stride;  // elements per group
  const int n_groups = k_w * GPT;
  const T* tail = w_next ? w_next + (size_t)lane * 8 : wl + (size_t)(n_groups - 1) * gstride;   // no successor: harmless re-read
  const char* bbase = act + (size_t)(col0 + (lane & 31)) * pitch + (size_t)(8 * (lane >> 5)) * sizeof(T);
  V8 bb[2][NT];
  fetch_b<T, NT>(bb[0], bbase, pitch);
  __builtin_amdgcn_sched_barrier(0);
  int g = 0;
  for (int tap = 0; tap < k_w; ++tap) {
    const char* btap = bbase + (size_t)(tap * dil) * pitch;
    const char* btap_next = bbase + (size_t)(min(tap + 1, k_w - 1) * dil) * pitch;
#pragma unroll
    for (int h = 0; h < GPT; ++h, ++g) {
      const T* nb = g + 1 < n_groups ? wl + (size_t)(g + 1) * gstride : tail;
      const char* bcur = btap + (size_t)(h * KCG) * 16 * sizeof(T);
      const char* bnext = h + 1 < GPT ? btap + (size_t)((h + 1) * KCG) * 16 * sizeof(T) : btap_next;
      conv_group<T, NF, NT, KCG>(acc, ws.ring, bb, ws.wf, nb, bcur, bnext, pitch);
    }
  }
}

// Workgroup barrier for LDS hand-offs only: waits for this wave's LDS traffic, NOT for its global loads -- weight
// fragments prefetched for the next conv stay in flight across it (__syncthreads() drains vmcnt too).  Global stores
// issued before it are not ordered by it: use __syncthreads() where another wave reads them.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
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
__device__ __forceinline__ void conv_stage(typename Acc32<T>::type (&acc)[NF][NT], WRing<T, NF, D>& ring, int kc_per, int k_w,
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
      JATTS_PRIO(1);
#pragma unroll
      for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int t = 0; t < NT; ++t) mma32(ring.r[j][f], bb[j & 1][t], acc[f][t]);
      JATTS_PRIO(0);
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

// SnakeBeta on the accumulators (bias already in): acc <- acc + b[n] sin^2(a[n] acc), channel n of register 4q + e being
// (nf0 + f) 32 + 8q + 4 (lane >> 5) + e.  The same expression as rowwise.hip's snakebeta_kernel, so fusing Matcha's feed-forward
// activation into the conv that produces its input changes no value; the caller then runs the JATTS_ACT_NONE epilogue.
template <int NF, int NT>
__device__ __forceinline__ void snake_acc(f32x16 (&acc)[NF][NT], const float* a, const float* b, int nf0, int n_out, int lane) {
  const int gq = lane >> 5;
#pragma unroll
  for (int f = 0; f < NF; ++f)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      // n_out is a multiple of 4 (jatts_conv1d checks): a quad is inside or outside as a whole, and an outside one (never stored) reads
      // the last quad instead of branching
      const int n0 = min((nf0 + f) * 32 + 8 * q + 4 * gq, n_out - 4);
      const f32x4 a4 = *reinterpret_cast<const f32x4*>(a + n0), b4 = *reinterpret_cast<const f32x4*>(b + n0);
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = acc[f][t][4 * q + e];
          acc[f][t][4 * q + e] = fmaf(b4[e], sin2_f(v * a4[e]), v);
        }
      __builtin_amdgcn_sched_barrier(0);   // one channel quad at a time (all 4 NF quads in flight spill registers)
    }
}

}  // namespace
