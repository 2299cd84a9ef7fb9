// Batched f32 GEMM on the exact-f32 matrix pipe (v_mfma_f32_32x32x2_f32) for the TRAINING step's attention products (round 4):
//   C[b] (M x N) = alpha * op(A[b]) (M x K) * op(B[b]) (K x N)
// q k^T, q p^T, P v and their gradients (attention.py:164-206 under autograd) went to rocBLAS through torch.matmul -- the one library GEMM
// inside a SURVEY row counted as implemented (missing #5).  A two-level batch index (outer, inner) with independent strides per
// operand covers (B, H, ...) tensors with an operand shared over B (the position projection p_h: outer stride 0).
//
// One workgroup = 4 waves = a 128 x 128 (or 128 x 64) tile of C (each wave 64 x 64 = 2 x 2 fragments, or 64 x 32); K runs in chunks of 32
// through a double-buffered LDS pair As[k][m], Bs[k][n] (pitch 132: the transposing store of a K-contiguous operand hits 64 distinct
// banks); the loads of chunk i + 1 are issued before the MFMAs of chunk i.  A fragment operand is one ds_read_b32 per lane and K-pair.
#include <stdlib.h>

#include "common.h"

#ifndef JATTS_BGEMM_DIAG
#define JATTS_BGEMM_DIAG 0   // timing probes only (wrong results): 1 = no operand loads in the chunk loop, 2 = no LDS stores in it, 4 = no barriers in it
#endif

namespace {

// Tile rows BM = 64 MF (MF = 32-row fragments per wave along m: 2 -> 128 rows, 1 -> 64), LDS pitch of the A tile BP = BM + 4 floats; BN = 64 / 128 / 192 and
// the chunk depth BK are template parameters too

struct BgemmArgs {
  const float* a;
  const float* b;
  float* c;
  int64_t sa_o, sa_i, sb_o, sb_i, sc_o, sc_i;   // element strides of the (outer, inner) batch index
  int lda, ldb, ldc, ta, tb;
  int n_inner, M, N, K;
  float alpha;
  int accumulate;
  int n_batch, gx, gy;      // XCD-aware 1-D grid: workgroup id -> (matrix, tile), see bgemm_kernel
};

// One operand chunk (RW rows of the "long" dimension from r0, BK contraction steps from k0) -> registers, 16 bytes per load.  k_contig = the
// operand is stored with K contiguous (A untransposed / B transposed): a thread takes 4 consecutive k of one row per load; otherwise the
// long dimension is contiguous: 4 consecutive rows of one k.
template <int RW, int BK, int PITCH, bool KC, bool VEC>      // KC: K contiguous in memory; VEC: 16-byte loads allowed (compile time: a
struct Chunk {                                               // run-time flag made hipcc emit every combination inside one kernel, waits and all)
  static constexpr int NL = (RW * BK / 4 + 255) / 256;      // 16-byte loads per thread
  static constexpr int TOTAL = RW * BK / 4;
  f32x4 v[NL];
  // vec (the matrix starts on a 16-byte boundary and ld % 4 == 0: every attention operand): ALL loads are issued unconditionally from clamped
  // addresses and masked afterwards -- a load inside a per-element branch makes hipcc wait vmcnt(0) right behind it, i.e. one L2 round trip
  // per 16 bytes, eight times per chunk (the first version of this kernel spent 4/5 of its time there).  A 16-byte piece that straddles the
  // edge (k + 3 >= K or r + 3 >= R) stays inside its row because ld % 4 == 0.
  __device__ __forceinline__ void load(const float* base, int ld, int r0, int R, int k0, int K) {
    constexpr bool k_contig = KC;
    if constexpr (VEC) {
      int lim[NL], at[NL];
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        const int u = (int)threadIdx.x + 256 * j;
        int r, k;
        if (k_contig) { r = r0 + u / (BK / 4); k = k0 + 4 * (u % (BK / 4)); }
        else { k = k0 + u / (RW / 4); r = r0 + 4 * (u % (RW / 4)); }
        const bool in = (TOTAL % 256 == 0 || u < TOTAL) && r < R && k < K;
        const int rc = r < R ? r : R - 1, kc = k < K ? k : K - 1;
        // vector index along the contiguous dimension: clamp to the last full-or-partial piece of the row
        const int64_t off = k_contig ? (int64_t)rc * ld + (kc & ~3) : (int64_t)kc * ld + (rc & ~3);
        v[j] = *reinterpret_cast<const f32x4*>(base + off);
        at[j] = k_contig ? k : r;
        lim[j] = in ? (k_contig ? K : R) : -1;      // component e is valid iff at + e < lim
      }
#pragma unroll
      for (int j = 0; j < NL; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[j][e] = at[j] + e < lim[j] ? v[j][e] : 0.f;
      return;
    }
    // element loads (a matrix that starts off a 16-byte boundary or ld % 4 != 0: a padded batch length T with T % 4 != 0 puts P v and its
    // gradients here): the same discipline -- every load unconditional from a clamped address, masks afterwards.  (The first version loaded
    // inside per-element branches: one exec-masked load and one L2 round trip per element.)
    // Element (j, e) of a thread is element tid + 256 (4 j + e) of the chunk counted along the CONTIGUOUS dimension first, so one wave
    // instruction reads 64 consecutive floats of a row (or 64 / BK rows of BK): the 4-consecutive-elements-per-lane map of the 16-byte
    // path would touch four times the cache lines per instruction here.
    bool ok[NL][4];
#pragma unroll
    for (int j = 0; j < NL; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int idx = (int)threadIdx.x + 256 * (4 * j + e);
        const int re = r0 + (k_contig ? idx / BK : idx % RW), ke = k0 + (k_contig ? idx % BK : idx / RW);
        ok[j][e] = (RW * BK % 1024 == 0 || idx < RW * BK) && re < R && ke < K;
        const int rc = re < R ? re : R - 1, kc = ke < K ? ke : K - 1;
        v[j][e] = base[k_contig ? (int64_t)rc * ld + kc : (int64_t)kc * ld + rc];
      }
#pragma unroll
    for (int j = 0; j < NL; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) v[j][e] = ok[j][e] ? v[j][e] : 0.f;
  }
  __device__ __forceinline__ void store(float* s) const {
    constexpr bool k_contig = KC;
    if constexpr (!VEC) {      // (the element map of load(): consecutive lanes = consecutive k at pitch 132 = 64 distinct banks, or consecutive r)
#pragma unroll
      for (int j = 0; j < NL; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int idx = (int)threadIdx.x + 256 * (4 * j + e);
          if (RW * BK % 1024 != 0 && idx >= RW * BK) continue;
          const int r = k_contig ? idx / BK : idx % RW, k = k_contig ? idx % BK : idx / RW;
          s[k * PITCH + r] = v[j][e];
        }
      return;
    }
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const int u = (int)threadIdx.x + 256 * j;
      if (TOTAL % 256 != 0 && u >= TOTAL) continue;
      if (k_contig) {      // transpose on the way in: s[k][r]
        const int r = u / (BK / 4), k = 4 * (u % (BK / 4));
#pragma unroll
        for (int e = 0; e < 4; ++e) s[(k + e) * PITCH + r] = v[j][e];
      } else {
        const int k = u / (RW / 4), r = 4 * (u % (RW / 4));
        *reinterpret_cast<f32x4*>(s + k * PITCH + r) = v[j];
      }
    }
  }
};

// WNF = 32-column fragments per wave along n: 1 -> 128 x 64 tile, 2 -> 128 x 128, 3 -> 128 x 192 (n = d_k = 192 in ONE tile: the T x T operand of
// P v and of the key / value gradients is then read once instead of once per n tile).
// 1-D grid in XCD-aware order: workgroup id lands on XCD id % 8 (each with its own L2), so matrix b = 8 (m / tiles) + id % 8 with m = id / 8 --
// all tiles of a matrix run on ONE XCD, back to back, and its operands are fetched from HBM once instead of once per XCD that happens to
// hold one of its tiles (the first version, a 3-D grid, ran at 45-65 TFLOP/s against rocBLAS' 100-110).
// MF = 1: 64-row tiles for launches whose 128-row tiles would not fill the two-per-CU slots evenly (P v and the key / value gradients at T = 768: 384
// workgroups of 128 x 192 on 512 slots = half the CUs carrying two and half one; 768 of 64 x 192 = three each)
template <int WNF, int BK, bool AK, bool BKC, bool VEC, int MF = 2>      // AK / BKC: K contiguous in A (trans_a == 0) / in B (trans_b != 0)
__global__ __launch_bounds__(256, 2) void bgemm_kernel(BgemmArgs g) {
  constexpr int BN = 64 * WNF, BPN = BN + 4, BM = 64 * MF, BP = BM + 4;
  extern __shared__ __attribute__((aligned(16))) float sm_raw[];      // [buffer][A: BK x BP | B: BK x BPN]
  constexpr int BUF = BK * (BP + BPN);
  const int xcd = (int)(blockIdx.x & 7u), mloc = (int)(blockIdx.x >> 3);
  const int tiles = g.gx * g.gy;
  const int bz = 8 * (mloc / tiles) + xcd, tile = mloc % tiles;
  if (bz >= g.n_batch) return;
  const int bo = bz / g.n_inner, bi = bz - bo * g.n_inner;
  const float* A = g.a + bo * g.sa_o + bi * g.sa_i;
  const float* B = g.b + bo * g.sb_o + bi * g.sb_i;
  float* C = g.c + bo * g.sc_o + bi * g.sc_i;
  const int m0 = (tile / g.gx) * BM, n0 = (tile % g.gx) * BN;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = (wave >> 1) * 32 * MF, wn = (wave & 1) * 32 * WNF;
  const int lo = lane & 31, hi = lane >> 5;

  f32x16 acc[MF][WNF];
#pragma unroll
  for (int i = 0; i < MF; ++i)
#pragma unroll
    for (int j = 0; j < WNF; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  Chunk<BM, BK, BP, AK, VEC> ra;
  Chunk<BN, BK, BPN, BKC, VEC> rb;
  ra.load(A, g.lda, m0, g.M, 0, g.K);
  rb.load(B, g.ldb, n0, g.N, 0, g.K);
  ra.store(sm_raw);
  rb.store(sm_raw + BK * BP);
  __syncthreads();
  const int n_chunks = (g.K + BK - 1) / BK;
  for (int ci = 0; ci < n_chunks; ++ci) {
    const bool more = ci + 1 < n_chunks;
    if (more && !(JATTS_BGEMM_DIAG & 1)) {
      ra.load(A, g.lda, m0, g.M, (ci + 1) * BK, g.K);
      rb.load(B, g.ldb, n0, g.N, (ci + 1) * BK, g.K);
    }
    const float* as = sm_raw + (ci & 1) * BUF + hi * BP + wm + lo;
    const float* bs = sm_raw + (ci & 1) * BUF + BK * BP + hi * BPN + wn + lo;
    // fragments of 8 K-pairs at a time into registers, THEN their MFMAs: a ds_read right in front of its MFMA exposes the LDS latency 16 times
    // per chunk (the first version ran at 55-63 TFLOP/s)
#pragma unroll
    for (int k8 = 0; k8 < BK / 2; k8 += 8) {
      float av[8][MF], bv[8][WNF];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
#pragma unroll
        for (int i = 0; i < MF; ++i) av[q][i] = as[2 * (k8 + q) * BP + 32 * i];
#pragma unroll
        for (int j = 0; j < WNF; ++j) bv[q][j] = bs[2 * (k8 + q) * BPN + 32 * j];
      }
      __builtin_amdgcn_sched_barrier(0);      // (hipcc otherwise sinks every read next to its MFMA behind an lgkmcnt(0))
#pragma unroll
      for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int j = 0; j < WNF; ++j) {
#pragma unroll
          for (int i = 0; i < MF; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][i], bv[q][j], acc[i][j], 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (more && !(JATTS_BGEMM_DIAG & 2)) {
      ra.store(sm_raw + ((ci + 1) & 1) * BUF);
      rb.store(sm_raw + ((ci + 1) & 1) * BUF + BK * BPN * 0 + BK * BP);
    }
    if (!(JATTS_BGEMM_DIAG & 4)) __syncthreads();
  }
  // C/D map: column (lane & 31) = n, row (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) = m
#pragma unroll
  for (int i = 0; i < MF; ++i)
#pragma unroll
    for (int j = 0; j < WNF; ++j) {
      const int n = n0 + wn + 32 * j + lo;
      if (n >= g.N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (m < g.M) {
          float* o = C + (int64_t)m * g.ldc + n;
          *o = g.accumulate ? *o + g.alpha * acc[i][j][r] : g.alpha * acc[i][j][r];
        }
      }
    }
}

}  // namespace

extern "C" int jatts_bgemm(const float* a, int64_t sa_outer, int64_t sa_inner, int32_t lda, int32_t trans_a, const float* b, int64_t sb_outer,
                           int64_t sb_inner, int32_t ldb, int32_t trans_b, float* c, int64_t sc_outer, int64_t sc_inner, int32_t ldc,
                           int32_t n_outer, int32_t n_inner, int32_t m, int32_t n, int32_t k, float alpha, int32_t accumulate, void* stream) {
  if (!a || !b || !c) return jatts_set_error_msg(JATTS_ERR_ARG, "bgemm: null pointer");
  if (n_outer < 1 || n_inner < 1 || m < 1 || n < 1 || k < 1 || lda < 1 || ldb < 1 || ldc < 1)
    return jatts_set_error_msg(JATTS_ERR_ARG, "bgemm: bad geometry");
  BgemmArgs g{a, b, c, sa_outer, sa_inner, sb_outer, sb_inner, sc_outer, sc_inner, lda, ldb, ldc, trans_a, trans_b, n_inner, m, n, k, alpha, accumulate, 0, 0, 0};
  // n tile: 192 in one piece (n = d_k = 192), else 64-wide where a 128-wide tile would be more than a quarter empty
  const int n_batch = n_outer * n_inner;
  auto launch = [&](auto kern, int bn, int bk, int bm) -> int {
    const int lds = 2 * bk * (bm + 4 + bn + 4) * (int)sizeof(float);
    if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return jatts_set_error_msg(JATTS_ERR_HIP, "bgemm: could not raise the dynamic LDS limit");
    g.n_batch = n_batch;
    g.gx = (n + bn - 1) / bn;
    g.gy = (m + bm - 1) / bm;
    const int64_t total = (int64_t)8 * ((n_batch + 7) / 8) * g.gx * g.gy;
    if (total >= (int64_t)1 << 31) return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "bgemm: launch too large");
    hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(256), lds, (hipStream_t)stream, g);
    return JATTS_OK;
  };
  const int n128 = (n + 127) / 128 * 128, n64 = (n + 63) / 64 * 64;
  // 16-byte loads when EVERY matrix of both operands starts on a 16-byte boundary with a leading dimension of 4 k elements
  const bool vec = ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0 &&
                   ((sa_outer | sa_inner | sb_outer | sb_inner | (int64_t)lda | (int64_t)ldb) & 3) == 0;
  const int tile = (n > 128 && n <= 192) ? 3 : (n64 * 4 <= n128 * 3 ? 1 : 2);
  int rc = JATTS_ERR_UNSUPPORTED;
#define JATTS_BGEMM_CASE(WNF, BKC_, AKv, BKv, VECv, MFv) rc = launch(bgemm_kernel<WNF, BKC_, AKv, BKv, VECv, MFv>, 64 * WNF, BKC_, 64 * MFv)
#define JATTS_BGEMM_ORIENT(WNF, BKC_, VECv, MFv)                                         \
  do {                                                                                   \
    if (!trans_a && trans_b) JATTS_BGEMM_CASE(WNF, BKC_, true, true, VECv, MFv);         \
    else if (!trans_a && !trans_b) JATTS_BGEMM_CASE(WNF, BKC_, true, false, VECv, MFv);  \
    else if (trans_a && !trans_b) JATTS_BGEMM_CASE(WNF, BKC_, false, false, VECv, MFv);  \
    else JATTS_BGEMM_CASE(WNF, BKC_, false, true, VECv, MFv);                            \
  } while (0)
  // 64-row tiles when the 128-row tiling leaves fewer than two rounds of the 512 two-per-CU slots (see bgemm_kernel); JATTS_BGEMM_MF = 1 / 2 forces either
  static const int mf_env = [] { const char* e = getenv("JATTS_BGEMM_MF"); return e ? atoi(e) : 0; }();
  const int64_t wg128 = (int64_t)n_batch * ((m + 127) / 128) * ((n + 191) / 192);
  const bool half_m = mf_env ? mf_env == 1 : (m > 64 && wg128 < 1024);
  if (!vec && tile == 3) JATTS_BGEMM_ORIENT(3, 16, false, 1);     // (n = d_k = 192 in one 64 x 192 tile here too: the T x T operand is read once)
  else if (!vec) JATTS_BGEMM_ORIENT(2, 16, false, 2);            // element loads: any alignment (a padded batch length T % 4 != 0 takes it for P v and its gradients)
  else if (tile == 3 && half_m) JATTS_BGEMM_ORIENT(3, 16, true, 1);
  else if (tile == 3) JATTS_BGEMM_ORIENT(3, 16, true, 2);
  else if (tile == 1) JATTS_BGEMM_ORIENT(1, 32, true, 2);
  else JATTS_BGEMM_ORIENT(2, 32, true, 2);
#undef JATTS_BGEMM_ORIENT
#undef JATTS_BGEMM_CASE
  if (rc != JATTS_OK) return rc;
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}
