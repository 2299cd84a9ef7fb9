// Shared device helpers for the jatts MI355X (gfx950 / CDNA4) kernels.
// Wave = 64 lanes.  Activations are packed, ragged, time-major: X[row][C] with
// cu_rows[b] = first row of sequence b (no padding rows anywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/jatts_hip.h"

typedef _Float16 f16;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define JATTS_CHECK_LAUNCH()                                 \
  do {                                                       \
    hipError_t e_ = hipGetLastError();                       \
    if (e_ != hipSuccess) return jatts_set_error(e_, __FILE__, __LINE__); \
  } while (0)

int jatts_set_error(hipError_t e, const char* file, int line);
int jatts_set_error_msg(int code, const char* msg);

// ---------------------------------------------------------------- element traits
template <typename T> struct Elem;
template <> struct Elem<f16> {
  typedef f16x8 vec8;  // 8 contraction elements per lane per MFMA K-step
  static constexpr int kBytes = 2;
};
template <> struct Elem<float> {
  typedef f32x8 vec8;
  static constexpr int kBytes = 4;
};

// Split-precision operand (round 4): an f32 value v (pre-scaled by a power of two into f16's range) carried as
// hi = f16(v), lo = f16(v - hi) -- 22 significand bits.  The element TAG is 4 bytes wide like f32, so every address
// computation of the f32 tiles carries over; 8 consecutive elements are stored PLANAR (16 B of hi, then 16 B of lo)
// so that each half is directly an MFMA operand.
struct f16s { f16 hi, lo; };
struct f16sx8 { f16x8 hi, lo; };
template <> struct Elem<f16s> {
  typedef f16sx8 vec8;
  static constexpr int kBytes = 4;
};

// f32-EQUIVALENT emulated operand (round 5, JATTS_F32E / JATTS_F32E6): every f32 value v is carried EXACTLY as three bfloat16 terms
//   b0 = bf16(v), b1 = bf16(v - b0), b2 = bf16(v - b0 - b1)      (round-to-nearest-even; both differences are exact in f32)
// 3 x 8 significand bits = f32's 24 and bf16 has f32's exponent field: v == b0 + b1 + b2 for every finite f32 whose last bit lies at or
// above bf16's smallest subnormal 2^-133 (|v| >= 2^-110) and which does not round up past bf16's largest finite value (|v| < 3.3895e38 =
// (2 - 2^-8) 2^127, a hair under FLT_MAX; beyond it b0 becomes infinity): NO scales, no block maxima, no element-dependent loss of
// relative precision.  Infinities and NaNs give NaN (inf - inf in the split), where the exact-f32 path would propagate an infinity.
// Under round-to-nearest |b1| <= 2^-8 |v| and |b2| <= 2^-16 |v|, so the nine partial products of w v have weights 1 | 2^-8 x2 | 2^-16 x3 |
// 2^-24 x2 | 2^-32; every bf16 x bf16 product is exact in the f32 accumulate.  NP = the number of partial products kept:
//   NP = 7 (JATTS_F32E): all of weight >= 2^-16 plus w1 v2 -- dropped: w2 v1 + w2 v2 <= (2^-24 + 2^-32) |w v|, one f32 rounding's worth (an
//           unfused f32 multiply).  Leading and smaller products go to separate accumulators joined by one correctly rounded add (mma32
//           below): a contraction with ONE term is within 2^-24 + 2^-24 = 2^-23 = 2 x an f32 FMA's error bound for EVERY input (the
//           acceptance rule of VERDICT r4, K_eff = 1 included).  7/16 of the pipe cycles of the exact-f32 chain, 16 more registers per fragment;
//   NP = 6 (JATTS_F32E6): the six of weight >= 2^-16 into ONE accumulator -- dropped <= 2^-23 |w v|, and the MFMA's own accumulate adds up
//           to 1 ulp at K_eff = 1 (4 x 2^-24 in all; measured 2.7).  6/16 of the cycles.
// With many terms both are MORE accurate than the exact-f32 chain (one rounding per 16-term MFMA instead of one per term).
// The element TAG is 6 bytes wide; 8 consecutive elements are stored PLANAR (16 B of b0 | b1 | b2).
typedef __bf16 bf16;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NP> struct bf3p { bf16 b0, b1, b2; };
template <int NP> struct bf3px8 { bf16x8 b0, b1, b2; };
typedef bf3p<7> bf3;        // JATTS_F32E
typedef bf3p<6> bf3f;       // JATTS_F32E6
template <int NP> struct Elem<bf3p<NP>> {
  typedef bf3px8<NP> vec8;
  static constexpr int kBytes = 6;
};
// v -> (b0, b1, b2), exact (see above).  v_cvt_pk_bf16_f32 rounds to nearest even; bf16 -> f32 is a shift.
__device__ __forceinline__ void bf3_split(float v, bf16& b0, bf16& b1, bf16& b2) {
  b0 = (bf16)v;
  const float r1 = v - (float)b0;
  b1 = (bf16)r1;
  b2 = (bf16)(r1 - (float)b1);
}
// The same split two values at a time (round 6): ONE v_cvt_pk_bf16_f32 rounds both and its result is already the stored pair; its halves widen back with a
// shift / a mask and both remainders come from one v_pk_add_f32 -- 4.5 VALU instructions per element instead of the 7.5 the scalar form compiles to in the
// staging loops (bit-identical: the same roundings in the same order).
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void bf3_split2(float v0, float v1, bf16x2& p0, bf16x2& p1, bf16x2& p2) {
  const f32x2 a = {v0, v1};
  p0 = __builtin_convertvector(a, bf16x2);
  unsigned u = __builtin_bit_cast(unsigned, p0);
  const f32x2 r1 = {v0 - __uint_as_float(u << 16), v1 - __uint_as_float(u & 0xffff0000u)};
  p1 = __builtin_convertvector(r1, bf16x2);
  u = __builtin_bit_cast(unsigned, p1);
  const f32x2 r2 = {r1[0] - __uint_as_float(u << 16), r1[1] - __uint_as_float(u & 0xffff0000u)};
  p2 = __builtin_convertvector(r2, bf16x2);
}
// 8 values -> the planar 48-byte group
template <int NP, typename V>
__device__ __forceinline__ void bf3_split8(const V& v, bf3px8<NP>& o) {
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    bf16x2 p0, p1, p2;
    bf3_split2(v[e], v[e + 1], p0, p1, p2);
    o.b0[e] = p0[0]; o.b0[e + 1] = p0[1];
    o.b1[e] = p1[0]; o.b1[e + 1] = p1[1];
    o.b2[e] = p2[0]; o.b2[e + 1] = p2[1];
  }
}

// One "K=16" step of a 32x32 output fragment.  Lane l supplies row/col (l&31) and
// contraction elements 8*(l>>5) .. 8*(l>>5)+7 of both operands.
//  f16 : v_mfma_f32_32x32x16_f16 (dense f16 MFMA rate, f32 accumulate)
//  f32 : 8 x v_mfma_f32_32x32x2_f32 (exact f32 fma chain; the parity mode).  MFMA j
//        consumes element j of each lane, i.e. k = 8*(l>>5)+j: both k-groups agree
//        between A and B, and the order of the contraction sum is irrelevant.
__device__ __forceinline__ void mma32(const f16x8& a, const f16x8& b, f32x16& c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ void mma32(const f32x8& a, const f32x8& b, f32x16& c) {
#pragma unroll
  for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], c, 0, 0, 0);
}
//  f16s: hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_f16 into ONE f32 accumulator (lo.lo is below f32 resolution);
//        the caller un-scales the sum by an exact power of two
__device__ __forceinline__ void mma32(const f16sx8& a, const f16sx8& b, f32x16& c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.lo, b.hi, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.hi, b.lo, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.hi, b.hi, c, 0, 0, 0);
}
//  bf3p<6>: the six partial products of weight >= 2^-16, smallest first, into ONE f32 accumulator (a = weights, b = activations)
__device__ __forceinline__ void mma32(const bf3px8<6>& a, const bf3px8<6>& b, f32x16& c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.b2, b.b0, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.b0, b.b2, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.b1, b.b1, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.b1, b.b0, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.b0, b.b1, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.b0, b.b0, c, 0, 0, 0);
}
//  bf3p<7>: TWO accumulators.  The bf16 MFMA's f32 accumulate is correctly rounded when the accumulator is at least as large as the arriving
//  products, but TRUNCATES the low bits of an accumulator that is smaller (tools/bf16_acc_probe.hip: |c| ~ 2^-8 |a b| -> up to 1 ulp of
//  error, 17 % of results differ from the correctly rounded one; |c| >= |a b| -> never).  So the leading product w0 v0 goes to `big` (sums of
//  like-sized terms: correctly rounded), the six smaller partial products to `small` (their truncations are at 2^-31 of the product), and the
//  epilogue adds the two with ONE correctly rounded v_add_f32 (acc_finish): a one-term contraction is then within 2^-24 (dropped w2 v1 + w2 v2)
//  + 2^-24 (that add) = 2^-23 of w v for every input.
struct acc2x16 { f32x16 big, small; };
__device__ __forceinline__ void mma32(const bf3px8<7>& a, const bf3px8<7>& b, acc2x16& c) {
  c.small = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.b1, b.b2, c.small, 0, 0, 0);
  c.small = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.b2, b.b0, c.small, 0, 0, 0);
  c.small = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.b0, b.b2, c.small, 0, 0, 0);
  c.small = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.b1, b.b1, c.small, 0, 0, 0);
  c.small = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.b1, b.b0, c.small, 0, 0, 0);
  c.small = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.b0, b.b1, c.small, 0, 0, 0);
  c.big = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.b0, b.b0, c.big, 0, 0, 0);
}
// Accumulator fragment of a 32x32 output tile per operand type, and the three things a kernel does with it besides mma32: start it at a
// value (the bias), close it (acc_finish) and read the result (acc_val)
template <typename T> struct Acc32 { typedef f32x16 type; };
template <> struct Acc32<bf3p<7>> { typedef acc2x16 type; };
__device__ __forceinline__ void acc_set(f32x16& a, int i, float v) { a[i] = v; }
__device__ __forceinline__ void acc_set(acc2x16& a, int i, float v) { a.small[i] = v; a.big[i] = 0.f; }
__device__ __forceinline__ void acc_finish(f32x16&) {}
__device__ __forceinline__ void acc_finish(acc2x16& a) { a.small = a.big + a.small; }
__device__ __forceinline__ f32x16& acc_val(f32x16& a) { return a; }
__device__ __forceinline__ f32x16& acc_val(acc2x16& a) { return a.small; }
// 16x16 output fragment, "K=32" step: lane supplies row/col (l&15) and contraction
// elements 8*(l>>4) .. +7.
__device__ __forceinline__ void mma16(const f16x8& a, const f16x8& b, f32x4& c) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ void mma16(const f32x8& a, const f32x8& b, f32x4& c) {
#pragma unroll
  for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], c, 0, 0, 0);
}

__device__ __forceinline__ void mma16(const f16sx8& a, const f16sx8& b, f32x4& c) {   // split operands: lo.hi + hi.lo + hi.hi, one accumulator
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.lo, b.hi, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi, b.lo, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi, b.hi, c, 0, 0, 0);
}

// C/D fragment maps (dtype independent on gfx950):
//  32x32: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5), reg in [0,16)
//  16x16: col = lane&15, row = 4*(lane>>4) + reg,                  reg in [0,4)

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(f16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ f16 from_f32<f16>(float v) { return (f16)v; }

__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }
// Mish = v tanh(softplus v).  tanh(log(1 + w)) = (w^2 + 2w) / (w^2 + 2w + 2) with w = e^v: one exp and one divide instead of
// exp + log1p + tanh (the activation was VALU-bound in the GroupNorm apply kernel).  v > 20: torch's softplus threshold (= v).
// Branch-free: at the threshold w^2 + 2w = 2.4e17 and the ratio rounds to exactly 1, so clamping the exponent's argument IS the
// threshold rule.  v_exp_f32 / v_rcp_f32 (1 ulp each) instead of libm's expf and an IEEE divide: 54 -> 20 VALU per element in the
// GroupNorm apply pass, relative error <= 3e-7.
__device__ __forceinline__ float mish_f(float v) {
  const float w = __expf(fminf(v, 20.f)), n = w * (w + 2.f);
  return v * (n * __builtin_amdgcn_rcpf(n + 2.f));
}

// sin^2(x) for SnakeBeta: three-constant Cody-Waite reduction by pi/2 (exact products under FMA) and the Cephes single-precision
// sin / cos kernels on [-pi/4, pi/4].  No table and no scratch: libm's sinf carries a Payne-Hanek slow path that puts 320 bytes
// of private memory into every kernel that inlines it.  |error| < 2e-7 absolute for |x| up to 1e5 (libm's own sinf, squared: 1.3e-7).
__device__ __forceinline__ float sin2_f(float x) {
  const float k = rintf(x * 0.636619772f);
  float r = fmaf(-k, 1.5707963705062866f, x);
  r = fmaf(-k, -4.371138828673793e-08f, r);
  r = fmaf(-k, -1.7151245100058819e-15f, r);
  const float z = r * r;
  const float sn = fmaf(r * z, fmaf(z, fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f), r);
  const float cs = fmaf(z * z, fmaf(z, fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f), fmaf(-0.5f, z, 1.f));
  const float t = ((int)k & 1) ? cs : sn;
  return t * t;
}

// Apply the fused epilogue activation.
__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case JATTS_ACT_RELU: return v > 0.f ? v : 0.f;
    case JATTS_ACT_TANH: return tanhf(v);
    case JATTS_ACT_SWISH: return v / (1.f + __expf(-v));
    case JATTS_ACT_MISH: return mish_f(v);
    default: return v;
  }
}

// Compile-time activation (one specialised epilogue per activation, selected by a uniform branch).
template <int ACT> __device__ __forceinline__ float act_c(float v) { return apply_act(v, ACT); }

// Sequence lookup for a ragged launch: tile -> (sequence, first local row).
// cu_tiles[b] = number of tiles before sequence b (n_seq+1 entries).
__device__ __forceinline__ int find_seq(const int32_t* __restrict__ cu_tiles, int n_seq, int tile) {
  int lo = 0, hi = n_seq;  // invariant: cu_tiles[lo] <= tile < cu_tiles[hi]
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (cu_tiles[mid] <= tile) lo = mid; else hi = mid;
  }
  return lo;
}

// 1-D grids over the REAL tiles of a ragged batch (jatts_ragged.host_lens != NULL): workgroup `wg` -> (sequence b, tile inside it) with
// `tt` rows per tile, tiles numbered sequence by sequence.  Every wave of the workgroup computes it redundantly from cu_rows (a 64-lane
// prefix sum per 64 sequences: no LDS, no barrier; ~1 k cycles against workgroup lifetimes of tens of thousands).  false: a
// workgroup past the last tile.  All 64 lanes must be active (call it first thing in the kernel).
__device__ __forceinline__ bool ragged_locate(const jatts_ragged& rg, int tt, unsigned wg, int& b, int& tile) {
  const int lane = threadIdx.x & 63;
  unsigned base = 0;
  for (int b0 = 0; b0 < rg.n_seq; b0 += 64) {
    const int i = b0 + lane;
    int tiles = 0;
    if (i < rg.n_seq) tiles = (int)(((int64_t)(rg.cu_rows[i + 1] - rg.cu_rows[i]) * rg.len_mul + tt - 1) / tt);
    int incl = tiles;   // inclusive prefix sum over the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(incl, o);
      if (lane >= o) incl += v;
    }
    const unsigned total = (unsigned)__shfl(incl, 63);
    if (wg < base + total) {
      const unsigned long long m = __ballot(base + (unsigned)incl > wg);
      const int l = __ffsll((long long)m) - 1;
      b = b0 + l;
      tile = (int)(wg - base) - (__shfl(incl, l) - __shfl(tiles, l));
      return true;
    }
    base += total;
  }
  return false;
}
// tiles of `tt` rows a launch over rg covers in the 1-D form: the exact count sum_b ceil(L_b len_mul / tt) from the host lengths, or 0 = the
// rectangular grid (uniform batches and callers without host lengths).  Exact, not an upper bound: the conv kernels deal the tile range out to
// the 8 XCDs in contiguous pieces, and spare tiles at the end would leave whole XCDs idle on small launches.
__host__ __device__ __forceinline__ bool ragged_is_1d(const jatts_ragged& rg) { return rg.host_lens != nullptr && rg.total_rows > 0; }
static inline int64_t ragged_tiles_1d(const jatts_ragged& rg, int tt) {
  if (!ragged_is_1d(rg)) return 0;
  int64_t n = 0;
  for (int b = 0; b < rg.n_seq; ++b) n += ((int64_t)rg.host_lens[b] * rg.len_mul + tt - 1) / tt;
  return n;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// Maximum over the wave, in every lane.  DPP inside the rows of 16 (quad swaps, then row rotations) and four v_readlane across them: ~10
// dependent VALU issues instead of six ds_bpermute round trips (each a full LDS latency with a wait) -- the block maxima of the split
// arithmetic sit on the critical path of every operand tile.  All 64 lanes must be active (every caller is a block-level reduction in
// converged code): v_readlane of an inactive lane returns stale data.
__device__ __forceinline__ float wave_max(float v) {
#define JATTS_DPP_MAX(ctrl) v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, false)))
  JATTS_DPP_MAX(0xB1);    // quad_perm [1, 0, 3, 2]
  JATTS_DPP_MAX(0x4E);    // quad_perm [2, 3, 0, 1]
  JATTS_DPP_MAX(0x124);   // row_ror 4
  JATTS_DPP_MAX(0x128);   // row_ror 8
#undef JATTS_DPP_MAX
  const int i = __builtin_bit_cast(int, v);
  const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 0)), b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 16));
  const float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 32)), e = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 48));
  return fmaxf(fmaxf(a, b), fmaxf(c, e));
}
