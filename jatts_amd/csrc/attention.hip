// Fused legacy relative-position self-attention (flash style) for gfx950.
//
// One workgroup = 4 waves = 64 queries of one (sequence, head); each wave owns 16 queries.
// K tiles [64 keys][d_k] and V^T tiles [d_k][64 keys] are staged in LDS and shared by
// the 4 waves.  Scores are computed TRANSPOSED (S^T = K Q^T) on 16x16 MFMA fragments so
// that a lane holds 4 consecutive keys of ONE query column: the softmax row reduction is
// in-lane adds plus two cross-lane steps (xor 16, 32), and the probabilities are already
// in B-operand position for O^T += V^T P^T -- no LDS round trip for P.
#include "common.h"

namespace {

template <typename T> struct V8 { typedef typename Elem<T>::vec8 type; };

// G<T>: what an operand of arithmetic T looks like in HBM and in the registers between a global load and the LDS store.
// f16s (JATTS_F32S, round 4): f32 in memory, split into hi / lo f16 planes on the way into LDS / into the MFMA operand registers.
template <typename T> struct G { typedef T type; typedef typename Elem<T>::vec8 vec8; static constexpr bool split = false; };
template <> struct G<f16s> { typedef float type; typedef f32x8 vec8; static constexpr bool split = true; };

template <typename T>
__device__ __forceinline__ typename G<T>::vec8 load8(const typename G<T>::type* p) {
  typename G<T>::vec8 v;
  if (sizeof(typename G<T>::type) == 2) {
    v = *reinterpret_cast<const typename G<T>::vec8*>(p);
  } else {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[e + 4] = b[e]; }
  }
  return v;
}
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
// The same 8 elements through a buffer descriptor: wave-uniform base + 32-bit per-lane byte offset + scalar byte offset, zeros past the
// descriptor's range (no 64-bit address registers, no bounds compare / select per chunk in the tile loads).
template <typename T>
__device__ __forceinline__ typename G<T>::vec8 load8_buf(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
  typename G<T>::vec8 v;
  if constexpr (sizeof(typename G<T>::type) == 2) {
    v = __builtin_bit_cast(typename G<T>::vec8, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
  } else {
    const f32x4 a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
    const f32x4 b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff + 16, soff, 0));
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[e + 4] = b[e]; }
  }
  return v;
}
// one element through the descriptor (V^T rows whose first column is not 16-byte aligned)
template <typename TG>
__device__ __forceinline__ TG load1_buf(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
  if constexpr (sizeof(TG) == 2) return __builtin_bit_cast(TG, __builtin_amdgcn_raw_buffer_load_b16(rs, voff, soff, 0));
  else return __builtin_bit_cast(TG, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
}
// LDS fragment read: 8 contraction elements of arithmetic T at `p` (f16s: one planar unit, 16 B of hi then 16 B of lo)
template <typename T>
__device__ __forceinline__ typename Elem<T>::vec8 lds8(const char* p) {
  if constexpr (G<T>::split) return f16sx8{*reinterpret_cast<const f16x8*>(p), *reinterpret_cast<const f16x8*>(p + 16)};
  else return load8<T>(reinterpret_cast<const T*>(p));
}
// 8 f32 values, scaled by the power of two `sc`, as a split operand
__device__ __forceinline__ f16sx8 split8(const f32x8& v, float sc) {
  f16sx8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float sv = v[e] * sc;
    o.hi[e] = (f16)sv;
    o.lo[e] = (f16)(sv - (float)o.hi[e]);
  }
  return o;
}
__device__ __forceinline__ float amax8(const f32x8& v, float m) {
#pragma unroll
  for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(v[e]));
  return m;
}
// exponent of the power-of-two scale that puts amax in [2^14, 2^15) (resunit_split_impl.h: split_exp), and 2^s
__device__ __forceinline__ int attn_split_exp(float amax) {
  const int bexp = (int)((__float_as_uint(amax) >> 23) & 0xff);
  const int s = 15 - (bexp - 126);
  return amax > 0.f ? (s > 60 ? 60 : (s < -60 ? -60 : s)) : 0;
}
__device__ __forceinline__ float attn_exp2i(int s) { return __uint_as_float((unsigned)(127 + s) << 23); }

#ifndef JATTS_ATTN_HALFPF
#define JATTS_ATTN_HALFPF 1
#endif
#ifndef JATTS_ATTN_PIPE
#define JATTS_ATTN_PIPE 3      // exact f32, 32-key tiles: LDS operand fragments read one step ahead of their MFMAs (1 = scores, 2 = P V; relattn_kernel: PIPE_S / PIPE_V)
#endif
#ifndef JATTS_ATTN_PIPE_PF
#define JATTS_ATTN_PIPE_PF 1   // ... in the fully prefetched kernels too (d_k 128 / 192), not only the half-tile pipeline of d_k 256
#endif
#ifndef JATTS_ATTN_DMA
#define JATTS_ATTN_DMA 2       // K / V^T tiles by LDS-direct buffer loads (no staging registers, no ds_write): 2 = every exact-f32 32-key-tile kernel (d_k 128 / 192 / 256), 1 = the bias-free d_k 256 one only, 0 = through registers
#endif
#ifndef JATTS_ATTN_DIAG
#define JATTS_ATTN_DIAG 0   // timing probes only (wrong results): 1 = no softmax arithmetic, 2 = no barriers in the key loop, 4 = no tile loads / stores in it
#endif
constexpr int KB = 64;  // keys per tile

template <typename T>
__device__ __forceinline__ void store8(char* p, const typename G<T>::vec8& v, float sc = 1.f) {
  if constexpr (G<T>::split) {
    const f16sx8 o = split8(v, sc);
    *reinterpret_cast<f16x8*>(p) = o.hi;
    *reinterpret_cast<f16x8*>(p + 16) = o.lo;
  } else if (sizeof(T) == 2) {
    *reinterpret_cast<typename Elem<T>::vec8*>(p) = v;
  } else {
    *reinterpret_cast<f32x4*>(p) = f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    *reinterpret_cast<f32x4*>(p + 16) = f32x4{(float)v[4], (float)v[5], (float)v[6], (float)v[7]};
  }
}

// 4 consecutive elements from an address that is only element-aligned (the rel-pos diagonal starts anywhere):
// gfx950 under the amdhsa ABI runs in unaligned-access mode, so this is one 8/16-byte load.
template <typename T>
__device__ __forceinline__ void load4u(const T* p, float (&o)[4]) {     // (T = the type in memory)
  T v[4];
  __builtin_memcpy(v, p, 4 * sizeof(T));
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = to_f32(v[e]);
}

// One K / V^T tile in flight in registers: the global loads of tile t+1 are issued before the MFMAs of tile t
// and land in LDS after them, so no wave ever waits on a tile load (the first version staged V^T with one
// 2-byte load per element, each a serialized round trip: 1.1 ms per decoder layer, profiles/r01_notes.md).
template <typename T, int DK, int KBT, int NW = 4>
struct TileRegs {
  static constexpr int N = KBT * DK / (512 * NW);  // 8-element chunks per thread for K and for V^T (KBT x DK elements over 64 NW threads)
  typename G<T>::vec8 k[N], v[N];
  float ku;
};

// WHICH: 1 = the K tile (+ the u . k bias of its keys), 2 = the V^T tile, 3 = both
// LATE_MASK: the columns of a V^T chunk past the sequence are zeroed by tile_store, not here -- anything that touches a loaded register
// before the store lets the scheduler pull it (and an `s_waitcnt vmcnt`) up into the MFMA loop the load is meant to hide behind.
// (The split arithmetic takes block maxima of the registers it loaded: it masks at load.)  KU = false: no u . k bias (bias-free attention: REL = false).
template <typename T, int DK, int KBT, int WHICH = 3, int NW = 4, bool KU = true, bool LATE_MASK = !G<T>::split>
__device__ __forceinline__ void tile_load(TileRegs<T, DK, KBT, NW>& tr, const jatts_relattn_desc& d, const typename G<T>::type* kg,
                                          const typename G<T>::type* vtg, int row0, int h, int j0, int Tn, bool vt_vec,
                                          __amdgpu_buffer_rsrc_t rk, __amdgpu_buffer_rsrc_t rv) {
  typedef typename G<T>::vec8 Vec;
  typedef typename G<T>::type TG;
  constexpr int UPR = DK / 8;
  if constexpr ((WHICH & 1) && KU) {   // (first: anything the compiler reloads from scratch for this address must not sit behind the tile's loads --
    tr.ku = 0.f;               //  a scratch reload waits on vmcnt(0), i.e. on every global load issued before it)
    if (d.ku && threadIdx.x < KBT) {
      const int j = j0 + (int)threadIdx.x;
      if (j >= 0 && j < Tn) tr.ku = d.ku[(int64_t)(row0 + j) * d.n_heads + h];
    }
  }
#pragma unroll
  for (int i = 0; i < TileRegs<T, DK, KBT, NW>::N; ++i) {
    const int u = threadIdx.x + 64 * NW * i;
    if constexpr (WHICH & 1) {  // K rows: keys, 8 channels per chunk
      const int r = u / UPR, cu = u - r * UPR;        // rows >= Tn lie past the descriptor: zeros (key tiles start at key 0)
      tr.k[i] = load8_buf<T>(rk, (unsigned)(r * d.ldk + cu * 8) * (unsigned)sizeof(TG), (unsigned)(j0 * d.ldk) * (unsigned)sizeof(TG));
    }
    if constexpr (WHICH & 2) {  // V^T rows: channels, 8 keys per chunk
      const int r = u / (KBT / 8), jc = j0 + 8 * (u % (KBT / 8));
      const unsigned voff = (unsigned)(r * d.ldvt + 8 * (u % (KBT / 8))) * (unsigned)sizeof(TG), soff = (unsigned)j0 * (unsigned)sizeof(TG);
      Vec z;
      if (vt_vec) {   // aligned: (row0 + j0) % 8 == 0, ldvt % 8 == 0; chunks at or past round_up(Tn, 8) lie past the descriptor
        z = load8_buf<T>(rv, voff, soff);
      } else {        // element-aligned only: one load per element through the same descriptor (cut at Tn columns of the last row then)
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = load1_buf<TG>(rv, voff + (unsigned)(e * sizeof(TG)), soff);
      }
      if constexpr (!LATE_MASK) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (jc + e < 0 || jc + e >= Tn) z[e] = from_f32<TG>(0.f);   // never multiply P = 0 by stray bits
      }
      tr.v[i] = z;
    }
  }
}

template <typename T, int DK, int KBT, int WHICH = 3, int NW = 4, bool KU = true>
__device__ __forceinline__ void tile_store(const TileRegs<T, DK, KBT, NW>& tr, char* ks, char* vs, float* kus, int KP, int VP, float sk = 1.f,
                                           float sv = 1.f, int mask_from = -1) {    // mask_from >= 0 (LATE_MASK loads): zero the tile's columns >= mask_from
  constexpr int UPR = DK / 8;
  typedef typename G<T>::type TG;
#pragma unroll
  for (int i = 0; i < TileRegs<T, DK, KBT, NW>::N; ++i) {
    const int u = threadIdx.x + 64 * NW * i;
    const int r = u / UPR, cu = u - r * UPR;
    if constexpr (WHICH & 1) store8<T>(ks + (size_t)r * KP + (size_t)cu * 8 * sizeof(T), tr.k[i], sk);
    if constexpr (WHICH & 2) {
      typename G<T>::vec8 z = tr.v[i];
      if (mask_from >= 0) {
        const int c0 = 8 * (u % (KBT / 8));
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (c0 + e >= mask_from) z[e] = from_f32<TG>(0.f);
      }
      store8<T>(vs + (size_t)(u / (KBT / 8)) * VP + (size_t)(u % (KBT / 8)) * 8 * sizeof(T), z, sv);
    }
  }
  if constexpr ((WHICH & 1) && KU) {
    if (threadIdx.x < KBT) kus[threadIdx.x] = tr.ku;
  }
}

// KBT = keys per tile: 64, or 32 for f32 at d_k >= 128 -- there a 64-key tile pair is 102-136 KB of LDS, ONE workgroup (one wave per SIMD)
// per CU, and every softmax / rescale instruction stalls the matrix pipe (0.33 of the f32 MFMA peak); half tiles fit two workgroups.
// REL = false: an instantiation without the rel-pos bias machinery (d.g is NULL: Matcha's plain attention) -- its pointers and gather
// registers are what pushed the f32 d_k 256 kernel into scratch.
// NW = 8: 512-thread workgroups of 128 queries, one per CU (the same two waves per SIMD): a K / V^T tile is staged once for twice the
// MFMA work and each thread holds half as much of it, so the whole next tile pair fits in registers again (full prefetch at d_k 256).
template <typename T, int DK, int KBT, bool REL = true, int NW = 4>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 1 : (((DK <= 256 && sizeof(T) == 2) || KBT == 32) ? 2 : 1)) void relattn_kernel(jatts_relattn_desc d) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef typename Elem<T>::vec8 Vec;
  typedef typename G<T>::type TG;          // element type in HBM (f32 for the split arithmetic)
  constexpr bool SPLIT = G<T>::split;
  // K tile pitch: the score MFMAs read 16 key rows x 4 channel groups per ds_read_b128; with the pitch = 2 (mod 4)
  // 16-byte units the hardware's 16-lane groups hit 16 distinct slots (+16 left 41 % of the LDS cycles in conflict)
  constexpr int KP = DK * (int)sizeof(T) + (sizeof(T) == 2 ? 32 : 16);
  constexpr int VP = KBT * (int)sizeof(T) + 16;  // V^T tile pitch
  constexpr int NF = KBT / 16;                   // 16-key score fragments per tile
  constexpr int NKS = DK / 32;                  // contraction steps for S
  constexpr int NDF = DK / 16;                  // output d fragments
  char* ks = smem;
  char* vs = smem + KBT * KP;
  float* kus = reinterpret_cast<float*>(smem + KBT * KP + DK * VP);
  float* slots = kus + KBT;                // split arithmetic: per-wave block maxima [4 waves][K | V] (+ Q at start)

  // XCD-aware order of the 1-D grid (launch_attn_kb): the dispatcher places workgroup id on XCD id % 8, each with its own L2; in the natural
  // (query block, sequence, head) order the query blocks of one (sequence, head) land on all eight and each XCD pulls that head's K and V^T
  // through its own L2.  XCD x takes the contiguous range [x * per, (x + 1) * per) of (head, sequence, query block) triples, query block
  // fastest: one head's query blocks run side by side on one XCD and share its tiles (f32 d_k 256: T = 384 214 -> 201 us, T = 768 unchanged --
  // there the waits on the tile loads inside the MFMA loops were the limit, see LATE_MASK; profiles/r05_notes.md section 7).
  const int gx = (d.rg.max_len + 16 * NW - 1) / (16 * NW);
  const int total = gx * d.rg.n_seq * d.n_heads, per = (total + 7) >> 3;
  const int wg = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
  if (wg >= total) return;
  const int bh = wg / gx;
  const int h = bh / d.rg.n_seq, b = bh - h * d.rg.n_seq;
  const int row0 = d.rg.cu_rows[b];
  const int Tn = d.rg.cu_rows[b + 1] - row0;
  const int i0 = (wg - bh * gx) * (16 * NW);
  if (i0 >= Tn) return;
  // keys >= Tk are masked out of the softmax (a PADDED batch, the reference's training-time forward(): Tn is then the padded
  // length, which the rel-shift geometry keeps using, and Tk the utterance's own length; attention.py:80-88)
  const int Tk = d.kv_len ? min(max(d.kv_len[b], 1), Tn) : Tn;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int qc = lane & 15, g = lane >> 4;
  const int qi = i0 + wave * 16 + qc;          // this lane's query (may be >= Tn: never stored)
  const int qi_c = qi < Tn ? qi : Tn - 1;      // clamped for loads
  const TG* qg = (const TG*)d.q + (int64_t)(row0 + qi_c) * d.ldq + h * DK;
  const TG* kg = (const TG*)d.k + (int64_t)row0 * d.ldk + h * DK;
  const int vcol0 = d.vt_col0 ? d.vt_col0[b] : row0;
  const TG* vtg = (const TG*)d.vt + (int64_t)(h * DK) * d.ldvt + vcol0;
  const TG* gg = (REL && d.g) ? (const TG*)d.g : nullptr;
  const int H = d.n_heads;
  // Key tiles always start at the sequence's first key (results do not depend on the position in the packed batch).
  // V^T is staged with aligned 16-byte loads when the sequence's first column is (vt_col0, RaggedBatch.vt_layout).
  const bool vt_vec = (d.ldvt & 7) == 0 && (vcol0 & 7) == 0 && (reinterpret_cast<uintptr_t>(d.vt) & 31) == 0;
  constexpr int j_start = 0;
  // buffer descriptors of this (utterance, head)'s K rows and V^T rows (tile_load): K = Tn rows of ldk elements from kg; V^T = d_k rows
  // of ldvt elements from vtg, the last one cut at round_up(Tn, 8) columns (aligned rows: whole 16-byte chunks) or at Tn (element loads)
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)kg, 0, (unsigned)((int64_t)Tn * d.ldk * sizeof(TG)), 0x00020000);
  const __amdgpu_buffer_rsrc_t rv =
      __builtin_amdgcn_make_buffer_rsrc((void*)vtg, 0, (unsigned)(((int64_t)(DK - 1) * d.ldvt + (vt_vec ? ((Tn + 7) & ~7) : Tn)) * sizeof(TG)), 0x00020000);

  Vec qf[NKS];
  int eq = 0;                               // split arithmetic: Q lives at scale 2^eq (one scale per workgroup = 64 queries of one head)
  if constexpr (SPLIT) {
    typename G<T>::vec8 qr[NKS];
    float m = 0.f;
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
      qr[s] = load8<T>(qg + 32 * s + 8 * g);
      m = amax8(qr[s], m);
    }
    if (qi >= Tn) m = 0.f;                  // (clamped duplicate rows of the last tile: same values, but keep the maximum a function of real queries)
    m = wave_max(m);
    if (lane == 0) slots[wave] = m;
    __syncthreads();
    float mq = slots[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) mq = fmaxf(mq, slots[w]);
    eq = attn_split_exp(mq);
    __syncthreads();
    const float sq = attn_exp2i(eq);
#pragma unroll
    for (int s = 0; s < NKS; ++s) qf[s] = split8(qr[s], sq);
  } else {
#pragma unroll
    for (int s = 0; s < NKS; ++s) qf[s] = load8<T>(qg + 32 * s + 8 * g);
  }

  f32x4 ot[NDF];
#pragma unroll
  for (int f = 0; f < NDF; ++f) ot[f] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  // rel-pos bias rows of this lane's query (legacy rel_shift reads row qi for j <= qi, row qi + 1 beyond)
  const TG* g_q = gg ? gg + ((int64_t)(row0 + qi_c) * H + h) * d.ldg : nullptr;
  const TG* g_q1 = gg ? gg + ((int64_t)(row0 + (qi_c + 1 < Tn ? qi_c + 1 : qi_c)) * H + h) * d.ldg : nullptr;

  // d_k <= 192: the next tile is prefetched into registers during the current tile's MFMAs.  d_k = 256 does not
  // have the registers for that at 2 waves/SIMD: it loads and stores the tile back to back (still 16-byte batched)
  // and relies on the second resident workgroup to cover the round trip.
  // DMA_OK (JATTS_ATTN_DMA >= 2: every exact-f32 32-key-tile kernel, not only the bias-free d_k 256 one): tiles by LDS-direct loads in the half-tile
  // schedule below; nothing is staged in registers, so there is nothing to prefetch into
  constexpr bool DMA_OK = sizeof(T) == 4 && !SPLIT && KBT == 32 && NW == 4 && DK % 64 == 0 && (JATTS_ATTN_DMA >= 2 || (JATTS_ATTN_DMA == 1 && !REL && DK == 256));
  constexpr bool PREFETCH = !DMA_OK && (NW == 8 || ((DK <= 192 || sizeof(T) != 2) && !(KBT == 32 && DK > 192)));   // (f32 d_k 256 at two workgroups per CU: no registers for it either)
  // Without the registers for a whole tile pair (plain operands only): HALF a tile in flight at a time.  V^T(t) is loaded under the score
  // MFMAs of tile t and lands in LDS before P V; K(t+1) is loaded under P V(t) and lands after it -- each global round trip behind one
  // MFMA phase, the same two barriers per tile, 32 staging registers instead of 64 (f32 d_k 256 at two workgroups per CU waited on
  // every tile before: matrix pipe 54 % busy).
  constexpr bool HALFPF = JATTS_ATTN_HALFPF && !PREFETCH && !SPLIT && sizeof(T) == 4;   // (f16 d_k 256 keeps its 64-key tiles: the half pipeline spilled 80 bytes there)
  // PIPE_S / PIPE_V (exact f32, half-tile pipeline): the compiler issues each MFMA group's ds_read_b128 right in front of it (read, wait, 4 MFMA) and
  // the LDS round trip shows in every group; reading fragment i + 1 before fragment i's MFMAs, pinned with sched_barrier, takes 8 registers.
  constexpr bool PIPE_OK = HALFPF || (JATTS_ATTN_PIPE_PF && sizeof(T) == 4 && !SPLIT && KBT == 32 && NW == 4);
  constexpr bool PIPE_S = (JATTS_ATTN_PIPE & 1) && PIPE_OK, PIPE_V = (JATTS_ATTN_PIPE & 2) && PIPE_OK;
  constexpr int DIAG = HALFPF ? JATTS_ATTN_DIAG : 0;
  // DMA: the tiles go global -> LDS directly (buffer_load_dwordx4 ... lds: 64 lanes x 16 B = 1 KB of consecutive LDS per wave instruction).
  // K: one key row (1 024 B) per instruction at the usual pitch.  V^T: eight 128-byte channel rows per instruction, so its image has NO row padding; bank
  // conflicts are avoided by an XOR swizzle of the 16-byte unit index with (row >> 1) & 7, applied to the SOURCE address here and to the fragment reads.
  typedef __attribute__((address_space(3))) void* lds_ptr;
  constexpr bool DMA = DMA_OK && HALFPF;
  constexpr int VPD = KBT * 4;     // V^T pitch in the DMA image (bytes)
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  float ku_next = 0.f;            // (rel-pos kernels: the u . k bias of the next tile's keys, one per thread < KBT, goes through a register as before)
  auto dma_k = [&](int j0) {
    if constexpr (REL) {
      ku_next = 0.f;
      if (d.ku && threadIdx.x < KBT && j0 + (int)threadIdx.x < Tn) ku_next = d.ku[(int64_t)(row0 + j0 + (int)threadIdx.x) * d.n_heads + h];
    }
    if (DK == 256 || lane < DK / 4) {     // a key row of d_k floats = d_k / 4 lanes of 16 bytes
#pragma unroll
      for (int i = 0; i < KBT / NW; ++i) {
        const int row = i * NW + wave_s;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (lds_ptr)(ks + row * KP), 16, lane * 16, (j0 + row) * d.ldk * 4, 0, 0);
      }
    }
  };
  auto ku_commit = [&]() {
    if constexpr (REL) {
      if (threadIdx.x < KBT) kus[threadIdx.x] = ku_next;
    }
  };
  auto dma_v = [&](int j0) {
    if (vt_vec) {
#pragma unroll
      for (int i = 0; i < DK / 8 / NW; ++i) {
        const int r8 = i * NW + wave_s;
        const int row = r8 * 8 + (lane >> 3);
        const int lu = (lane & 7) ^ ((row >> 1) & 7);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (lds_ptr)(vs + r8 * 8 * VPD), 16, (row * d.ldvt + lu * 4) * 4, j0 * 4, 0, 0);
      }
    } else {     // element-aligned V^T rows: 4 bytes per lane, two channel rows per instruction
      for (int i = 0; i < DK / 2 / NW; ++i) {
        const int r2 = i * NW + wave_s;
        const int row = r2 * 2 + (lane >> 5), fl = lane & 31;
        const int lu = (fl >> 2) ^ ((row >> 1) & 7);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (lds_ptr)(vs + r2 * 2 * VPD), 4, (row * d.ldvt + lu * 4 + (fl & 3)) * 4, j0 * 4, 0, 0);
      }
    }
  };
  // LOAD-BEARING: a wave's `buffer_load ... lds` pieces must have LANDED before the barrier that lets OTHER waves read those LDS rows.  A workgroup barrier
  // does not wait for vector-memory traffic (LDS-DMA requests stay in flight across s_barrier), and nothing orders another wave's ds_read behind them.  The
  // compiler happens to put s_waitcnt vmcnt(0) in front of these barriers today (checked in the gfx950 ISA); this makes it explicit.
  auto dma_landed = [] { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
  TileRegs<T, DK, KBT, NW> tr;
  if (PREFETCH) tile_load<T, DK, KBT, 3, NW, REL>(tr, d, kg, vtg, row0, h, j_start, Tn, vt_vec, rk, rv);
  if constexpr (DMA) {
    dma_k(j_start);
    ku_commit();
    dma_landed();
    __syncthreads();
  } else if constexpr (HALFPF) {
    tile_load<T, DK, KBT, 1, NW, REL>(tr, d, kg, vtg, row0, h, j_start, Tn, vt_vec, rk, rv);
    tile_store<T, DK, KBT, 1, NW, REL>(tr, ks, vs, kus, KP, VP);
    __syncthreads();
  }
  int ev_prev = 0;
  for (int j0 = j_start; j0 < Tk; j0 += KBT) {
    if constexpr (DMA) dma_v(j0);
    else if constexpr (HALFPF) { if (!(DIAG & 4)) tile_load<T, DK, KBT, 2, NW, REL, true>(tr, d, kg, vtg, row0, h, j0, Tn, vt_vec, rk, rv); }
    else if (!PREFETCH) tile_load<T, DK, KBT, 3, NW, REL>(tr, d, kg, vtg, row0, h, j0, Tn, vt_vec, rk, rv);
    int ek = 0, ev = 0;
    if constexpr (HALFPF) {
      // (K(t) is in LDS since the previous iteration's last barrier)
    } else if constexpr (SPLIT) {    // block maxima of the K and V^T tiles that sit in registers -> their power-of-two scales
      float mk = 0.f, mv = 0.f;
#pragma unroll
      for (int i = 0; i < TileRegs<T, DK, KBT, NW>::N; ++i) {
        mk = amax8(tr.k[i], mk);
        mv = amax8(tr.v[i], mv);
      }
      mk = wave_max(mk);
      mv = wave_max(mv);
      if (lane == 0) { slots[2 * wave] = mk; slots[2 * wave + 1] = mv; }
      __syncthreads();
      float sk = slots[0], sv = slots[1];
#pragma unroll
      for (int w = 1; w < NW; ++w) { sk = fmaxf(sk, slots[2 * w]); sv = fmaxf(sv, slots[2 * w + 1]); }
      ek = attn_split_exp(sk);
      ev = attn_split_exp(sv);
      tile_store<T, DK, KBT, 3, NW, REL>(tr, ks, vs, kus, KP, VP, attn_exp2i(ek), attn_exp2i(ev), j0 + KBT > Tn ? Tn - j0 : -1);
    } else {
      tile_store<T, DK, KBT, 3, NW, REL>(tr, ks, vs, kus, KP, VP, 1.f, 1.f, j0 + KBT > Tn ? Tn - j0 : -1);
    }
    if constexpr (!HALFPF) __syncthreads();
    if (PREFETCH && j0 + KBT < Tk) tile_load<T, DK, KBT, 3, NW, REL>(tr, d, kg, vtg, row0, h, j0 + KBT, Tn, vt_vec, rk, rv);

    // ---- rel-pos bias gather, issued before the score MFMAs so that its latency hides behind them ----
    float bd[NF][4];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const int jq = j0 + 16 * f + 4 * g;  // this lane's 4 consecutive keys jq .. jq+3 of fragment f
#pragma unroll
      for (int r = 0; r < 4; ++r) bd[f][r] = 0.f;
      if (REL && gg && qi < Tn) {
        const bool inside = jq >= 0 && jq + 3 < Tn;
        if (d.rel_mode == 2) {  // new rel_shift: plain diagonal map, no wrap
          const TG* p = g_q + (d.rel_center - qi + jq);
          if (inside) load4u<TG>(p, bd[f]);
          else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (jq + r >= 0 && jq + r < Tn) bd[f][r] = to_f32(p[r]);
          }
        } else {                // legacy rel_shift (view-reinterpretation wrap)
          if (inside && jq + 3 <= qi) load4u<TG>(g_q + (Tn - 1 - qi + jq), bd[f]);
          else if (inside && jq > qi + 1) load4u<TG>(g_q1 + (jq - qi - 2), bd[f]);
          else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int j = jq + r;
              if (j < 0 || j >= Tn) continue;
              if (j <= qi) bd[f][r] = to_f32(g_q[Tn - 1 - qi + j]);
              else if (j > qi + 1) bd[f][r] = to_f32(g_q1[j - qi - 2]);
            }
          }
        }
      }
    }

    // ---- S^T fragments: st[f][r] = key (j0 + 16 f + 4 g + r)  x  query qc ----
    f32x4 st[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) st[f] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (PIPE_S) {           // fragment i + 1 is read while fragment i's MFMAs issue (see PIPE_S above)
      auto kfrag = [&](int i) { return lds8<T>(ks + (size_t)(16 * (i % NF) + qc) * KP + (size_t)(32 * (i / NF) + 8 * g) * sizeof(T)); };
      Vec a0 = kfrag(0);
#pragma unroll
      for (int i = 0; i < NKS * NF; ++i) {
        Vec a1 = a0;
        if (i + 1 < NKS * NF) a1 = kfrag(i + 1);
        __builtin_amdgcn_sched_barrier(0);
        mma16(a0, qf[i / NF], st[i % NF]);
        __builtin_amdgcn_sched_barrier(0);
        a0 = a1;
      }
    } else {
#pragma unroll
      for (int s = 0; s < NKS; ++s) {   // NF independent accumulator chains per contraction step
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          const Vec a = lds8<T>(ks + (size_t)(16 * f + qc) * KP + (size_t)(32 * s + 8 * g) * sizeof(T));
          mma16(a, qf[s], st[f]);
        }
      }
    }
    // ---- bias terms, scale, mask, online softmax ----
    float mx = -INFINITY;
    if constexpr (!(DIAG & 1)) {
    const float inv_qk = SPLIT ? attn_exp2i(-(eq + ek)) : 1.f;     // exact un-scale of q . k
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      f32x4 kq = {0.f, 0.f, 0.f, 0.f};
      if constexpr (REL) kq = *reinterpret_cast<const f32x4*>(kus + 16 * f + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = j0 + 16 * f + 4 * g + r;
        float sv = SPLIT ? st[f][r] * inv_qk : st[f][r];
        if constexpr (REL) sv = sv + kq[r] + bd[f][r];      // (x + 0 is not a no-op the compiler may drop: -0)
        float s = (j >= 0 && j < Tk) ? sv * d.scale : -INFINITY;
        st[f][r] = s;
        mx = fmaxf(mx, s);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __expf(m_run - m_new);  // first tile: exp(-inf) = 0
    m_run = m_new;
    float psum = 0.f;
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __expf(st[f][r] - m_new);
        st[f][r] = p;
        psum += p;
      }
    l_run = l_run * alpha + psum;
    // split arithmetic: the O accumulators live at 2^(15 + ev) (P at the fixed scale 2^15, V^T at this tile's 2^ev): moving from the previous
    // tile's scale to this one's is an exact power of two folded into the online-softmax rescale
    const float oscale = SPLIT ? alpha * attn_exp2i(ev - ev_prev) : alpha;
    ev_prev = ev;
#pragma unroll
    for (int f = 0; f < NDF; ++f)
#pragma unroll
      for (int r = 0; r < 4; ++r) ot[f][r] *= oscale;
    }

    if constexpr (DMA) {
      dma_landed();
      __syncthreads();            // V^T(t) landed (vmcnt(0) in front of the barrier) and visible; the K buffer is free
      if (j0 + KBT > Tn) {        // the sequence's last, partial tile: zero the columns past its end (never multiply P = 0 by stray bits)
        const int mf = Tn - j0;
        for (int row = (int)threadIdx.x; row < DK; row += 64 * NW)
          for (int c = mf; c < KBT; ++c)
            *reinterpret_cast<float*>(vs + row * VPD + (((c >> 2) ^ ((row >> 1) & 7)) << 4) + ((c & 3) << 2)) = 0.f;
        __syncthreads();
      }
      if (j0 + KBT < Tk) dma_k(j0 + KBT);
    } else if constexpr (HALFPF) {
      __builtin_amdgcn_sched_barrier(0);   // nothing of the store (its waits on the loads) moves up into the score MFMAs
      if (!(DIAG & 4)) tile_store<T, DK, KBT, 2, NW, REL>(tr, ks, vs, kus, KP, VP, 1.f, 1.f, j0 + KBT > Tn ? Tn - j0 : -1);
      if (!(DIAG & 2)) __syncthreads();            // V^T(t) visible; every wave is past its score MFMAs and its u . k reads: the K buffer is free
      if (!(DIAG & 4) && j0 + KBT < Tk) tile_load<T, DK, KBT, 1, NW, REL>(tr, d, kg, vtg, row0, h, j0 + KBT, Tn, vt_vec, rk, rv);
    }
    // ---- O^T += V^T P^T over two 32-key blocks.  Contraction slots of k-group g in block kb:
    //      keys {32kb + 4g + r} (from st[2kb]) then {32kb + 16 + 4g + r} (from st[2kb+1]) ----
#pragma unroll
    for (int kb = 0; kb < KBT / 32; ++kb) {
      Vec pb;
      if constexpr (SPLIT) {          // P in [0, 1] at the fixed scale 2^15
        f32x8 pv;
#pragma unroll
        for (int r = 0; r < 4; ++r) { pv[r] = st[2 * kb][r]; pv[4 + r] = st[2 * kb + 1][r]; }
        pb = split8(pv, 32768.f);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          pb[r] = from_f32<T>(st[2 * kb][r]);
          pb[4 + r] = from_f32<T>(st[2 * kb + 1][r]);
        }
      }
      if constexpr (PIPE_V) {
        auto vfrag = [&](int f) {
          Vec a;
          if constexpr (DMA) {      // swizzled image: unit u of row r sits at unit u ^ ((r >> 1) & 7); (16 f + qc) >> 1 = 8 f + (qc >> 1)
            const char* vrow = vs + (16 * f + qc) * VPD;
            const int sw = (qc >> 1) & 7;
            const f32x4 lo = *reinterpret_cast<const f32x4*>(vrow + ((g ^ sw) << 4)), hi = *reinterpret_cast<const f32x4*>(vrow + (((g + 4) ^ sw) << 4));
#pragma unroll
            for (int r = 0; r < 4; ++r) { a[r] = lo[r]; a[4 + r] = hi[r]; }
          } else {
            const T* vr = reinterpret_cast<const T*>(vs + (size_t)(16 * f + qc) * VP) + 32 * kb + 4 * g;
#pragma unroll
            for (int r = 0; r < 4; ++r) { a[r] = vr[r]; a[4 + r] = vr[16 + r]; }
          }
          return a;
        };
        Vec a0 = vfrag(0);
#pragma unroll
        for (int f = 0; f < NDF; ++f) {
          Vec a1 = a0;
          if (f + 1 < NDF) a1 = vfrag(f + 1);
          __builtin_amdgcn_sched_barrier(0);
          mma16(a0, pb, ot[f]);
          __builtin_amdgcn_sched_barrier(0);
          a0 = a1;
        }
      } else
#pragma unroll
      for (int f = 0; f < NDF; ++f) {
        Vec a;
        if constexpr (SPLIT) {        // keys 32 kb + 4 g + {0..3} and + 16: two half-units of the planar layout (8 keys = 16 B hi | 16 B lo)
          const char* vrow = vs + (size_t)(16 * f + qc) * VP;
          const int c0 = 32 * kb + 4 * g, c1 = c0 + 16;
          const char* p0 = vrow + (size_t)(c0 >> 3) * 32 + (size_t)(c0 & 7) * 2;
          const char* p1 = vrow + (size_t)(c1 >> 3) * 32 + (size_t)(c1 & 7) * 2;
          const f16x4 h0 = *reinterpret_cast<const f16x4*>(p0), l0 = *reinterpret_cast<const f16x4*>(p0 + 16);
          const f16x4 h1 = *reinterpret_cast<const f16x4*>(p1), l1 = *reinterpret_cast<const f16x4*>(p1 + 16);
#pragma unroll
          for (int r = 0; r < 4; ++r) { a.hi[r] = h0[r]; a.hi[4 + r] = h1[r]; a.lo[r] = l0[r]; a.lo[4 + r] = l1[r]; }
        } else {
          const T* vr = reinterpret_cast<const T*>(vs + (size_t)(16 * f + qc) * VP) + 32 * kb + 4 * g;
#pragma unroll
          for (int r = 0; r < 4; ++r) { a[r] = vr[r]; a[4 + r] = vr[16 + r]; }
        }
        mma16(a, pb, ot[f]);
      }
    }
    if constexpr (DMA) {
      ku_commit();          // (every wave is past this tile's softmax since the barrier in front of the P V MFMAs)
      dma_landed();         // K(t + 1) before the barrier that publishes it
    }
    if constexpr (HALFPF && !DMA) {
      __builtin_amdgcn_sched_barrier(0);   // (the same for K(t + 1) and the P V MFMAs)
      if (!(DIAG & 4) && j0 + KBT < Tk) tile_store<T, DK, KBT, 1, NW, REL>(tr, ks, vs, kus, KP, VP);
    }
    if (!(DIAG & 2) || !HALFPF) __syncthreads();
  }

  l_run += __shfl_xor(l_run, 16);
  l_run += __shfl_xor(l_run, 32);
  if (qi < Tn) {
    const float inv = (SPLIT ? attn_exp2i(-(15 + ev_prev)) : 1.f) / l_run;
    TG* og = (TG*)d.out + (int64_t)(row0 + qi) * d.ldo + h * DK;
#pragma unroll
    for (int f = 0; f < NDF; ++f)
#pragma unroll
      for (int r = 0; r < 4; ++r) og[16 * f + 4 * g + r] = from_f32<TG>(ot[f][r] * inv);
  }
}

template <typename T, int DK, int KBT = KB, bool REL = true, int NW = 4>
int launch_attn_kb(const jatts_relattn_desc& d, hipStream_t s) {
  const size_t lds = (size_t)KBT * (DK * sizeof(T) + (sizeof(T) == 2 ? 32 : 16)) + (size_t)DK * (KBT * sizeof(T) + 16) + KBT * sizeof(float) + 64;
  const int64_t total = (int64_t)((d.rg.max_len + 16 * NW - 1) / (16 * NW)) * d.rg.n_seq * d.n_heads;
  if (total >= ((int64_t)1 << 31) - 8) return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "relpos_attention: more than 2^31 query blocks (split the batch)");
  dim3 grid((unsigned)(8 * ((total + 7) / 8)));    // 1-D, decoded in XCD-aware order by the kernel
  auto kern = relattn_kernel<T, DK, KBT, REL, NW>;
  // the dynamic-LDS limit ONCE per kernel (function-local static of this template instantiation), not per launch: the per-launch call was seen to stall the
  // host for ~20 ms now and then (profiles/r06_notes.md section 8)
  static const hipError_t lds_attr = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (lds_attr != hipSuccess) return jatts_set_error(lds_attr, __FILE__, __LINE__);
  hipLaunchKernelGGL(kern, grid, dim3(64 * NW), lds, s, d);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}
template <typename T, int DK>
int launch_attn(const jatts_relattn_desc& d, hipStream_t s) {
  if constexpr (sizeof(T) == 4 && DK >= 128 && DK % 64 == 0) {
    static const int half = [] { const char* e = getenv("JATTS_ATTN_KB32"); return e ? atoi(e) : 1; }();
    // d_k 256: the split arithmetic takes 512-thread workgroups (half the hi / lo conversion work per wave, the whole next tile pair
    // prefetched: T = 768 511 -> 466 us); exact f32 is slower that way (738 -> 771 us: eight waves in lockstep on the barriers) and
    // keeps four waves with half a tile in flight.  JATTS_ATTN_NW8 = 0 / 1 forces either (tools/ A/B only).
    static const int wide = [] { const char* e = getenv("JATTS_ATTN_NW8"); return e ? atoi(e) : -1; }();
    if constexpr (DK == 256 && sizeof(typename G<T>::type) == 4) {
      if (half && (wide >= 0 ? wide != 0 : G<T>::split)) return launch_attn_kb<T, DK, 32, true, 8>(d, s);
    }
    if constexpr (DK == 256 && sizeof(typename G<T>::type) == 4 && !G<T>::split) {
      if (half && !d.g && !d.ku) return launch_attn_kb<T, DK, 32, false>(d, s);
    }
    if (half) return launch_attn_kb<T, DK, 32>(d, s);
  }
  return launch_attn_kb<T, DK, KB>(d, s);
}

template <typename T>
int dispatch_dk(const jatts_relattn_desc& d, hipStream_t s) {
  switch (d.d_k) {
    case 32: return launch_attn<T, 32>(d, s);
    case 64: return launch_attn<T, 64>(d, s);
    case 96: return launch_attn<T, 96>(d, s);
    case 128: return launch_attn<T, 128>(d, s);
    case 192: return launch_attn<T, 192>(d, s);
    case 256: return launch_attn<T, 256>(d, s);
  }
  return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "relpos_attention: d_k must be one of 32,64,96,128,192,256");
}

template <typename T>
__global__ void rowdot_kernel(const T* x, int ldx, int64_t rows, int n_heads, int d_k, const float* vec,
                              float* out) {
  const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  for (int h = 0; h < n_heads; ++h) {
    float s = 0.f;
    for (int c = lane; c < d_k; c += 64) s += to_f32(x[row * ldx + h * d_k + c]) * vec[h * d_k + c];
    s = wave_sum(s);
    if (lane == 0) out[row * n_heads + h] = s;
  }
}

}  // namespace

extern "C" int jatts_relpos_attention(const jatts_relattn_desc* d, void* stream) {
  if (!d || !d->q || !d->k || !d->vt || !d->out || !d->rg.cu_rows)
    return jatts_set_error_msg(JATTS_ERR_ARG, "relpos_attention: null pointer");
  if (d->n_heads < 1 || d->ldq % 8 || d->ldk % 8) return jatts_set_error_msg(JATTS_ERR_ARG, "relpos_attention: bad strides");
  if (d->rg.max_len <= 0) return JATTS_OK;
  // the tile loads address one (utterance, head)'s K rows and V^T rows through 32-bit buffer offsets
  if ((int64_t)d->rg.max_len * d->ldk * 4 >= ((int64_t)1 << 32) || (int64_t)d->d_k * d->ldvt * 4 >= ((int64_t)1 << 32))
    return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "relpos_attention: a sequence's K rows / a head's V^T rows exceed 4 GiB (split the batch)");
  hipStream_t s = (hipStream_t)stream;
  if (d->dtype == JATTS_F16) return dispatch_dk<f16>(*d, s);
  if (d->dtype == JATTS_F32) return dispatch_dk<float>(*d, s);
  if (d->dtype == JATTS_F32S) return dispatch_dk<f16s>(*d, s);     // f32 tensors, split f16 hi / lo MFMA operands (round 4)
  return jatts_set_error_msg(JATTS_ERR_ARG, "relpos_attention: unknown dtype");
}

extern "C" int jatts_rowdot(int32_t dtype, const void* x, int32_t ldx, int64_t rows, int32_t n_heads,
                            int32_t d_k, const float* vec, float* out, void* stream) {
  if (!x || !vec || !out) return jatts_set_error_msg(JATTS_ERR_ARG, "rowdot: null pointer");
  if (rows <= 0) return JATTS_OK;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)((rows + 3) / 4));
  if (dtype == JATTS_F16)
    hipLaunchKernelGGL(rowdot_kernel<f16>, grid, dim3(256), 0, s, (const f16*)x, ldx, rows, n_heads, d_k, vec, out);
  else
    hipLaunchKernelGGL(rowdot_kernel<float>, grid, dim3(256), 0, s, (const float*)x, ldx, rows, n_heads, d_k, vec, out);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}
