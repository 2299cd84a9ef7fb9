// Deterministic cross-workgroup reductions for the training kernels (round 4).
//
// The small parameter-gradient reductions (LayerNorm gamma / beta, bias and column sums, depthwise-conv weights, BatchNorm / GroupNorm /
// SnakeBeta statistics, the gradient norm) used to add per-workgroup partial sums into the result with f32 atomics: the order of the
// additions, and with it the last bits of the sum, changed from run to run.  The reference is bit-identical run to run, so is
// this: every workgroup stores its partial vector to a SLAB in a library-wide scratch buffer, takes a ticket, and the workgroup that
// draws the last ticket adds the slabs up in a FIXED order (slab index, then a fixed tree over its waves) and accumulates the total into
// the result -- one launch, no atomics on data, no zero-filled accumulator required by the reduction itself.
//
// Publication protocol (MI355X: 8 XCDs with private L2s; /opt/skills guide, "in-launch combine"): plain slab stores -> every wave
// s_waitcnt vmcnt(0) -> barrier -> lane 0: agent-scope RELEASE fence, s_waitcnt vmcnt(0), relaxed agent-scope ticket fetch_add; the last
// arriver: agent-scope ACQUIRE fence (drops its CU's L1) -> barrier -> plain slab loads.  The ticket returns to 0 before the launch ends.
// The scratch is shared by every launch: launches of ONE stream serialise, so one training stream per process may use these kernels at
// a time (jatts_amd.training does).
#pragma once
#include "common.h"

constexpr int JATTS_WS_TICKETS = 16384;         // uint32 tickets at the head of the scratch (zero between launches)
struct jatts_ws_t {
  unsigned* tickets;
  float* slabs;
  int64_t slab_floats;
};
extern jatts_ws_t jatts_g_ws;                                  // api.hip: jatts_set_workspace
int jatts_ws_need(int64_t groups, int64_t slab_floats);        // JATTS_OK, or an error naming the bytes jatts_set_workspace must provide

namespace {

// All threads of the workgroup call this after the workgroup's slab stores.  -> true, workgroup-uniformly, in the last arriver of `ticket`.
// `flag`: one 32-bit LDS word of an array the kernel already owns (free at this point).
__device__ __forceinline__ bool det_arrive(unsigned* ticket, unsigned n_parts, unsigned* flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = t == n_parts - 1;
    if (last) {
      __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // nobody else touches it before the next launch
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    *flag = last ? 1u : 0u;
  }
  __syncthreads();
  return *flag != 0u;
}

// The last arriver's sum: total[i] = sum_p slab[p * n + i] for i < n, parts in a FIXED order -- wave w of the NW waves takes the
// contiguous part range [w P / NW, (w + 1) P / NW), the NW wave sums are added in wave order through `red` (NW x 64 x VEC elements of
// LDS) -- then add(i, total).  A function of (P, NW, n) only: bit-identical from run to run.  T = float or double.
// n % 4 == 0 (floats): a lane owns FOUR consecutive columns and moves them as one 16-byte load, eight loads in flight: a single
// workgroup pulling slabs that other XCDs wrote is latency-bound, bytes in flight are its bandwidth (4-byte loads, four in flight:
// 16 GB/s -- a 512-part LayerNorm reduction took 80 us; profiles/r04_notes.md).
// red_cap: elements of T available at `red` (the 16-byte form needs NW x 256 floats).
template <int NW, typename T, typename F>
__device__ __forceinline__ void det_sum_slabs(const T* slab, int n_parts, int n, T* red, int red_cap, F&& add) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p0 = (int)((int64_t)wave * n_parts / NW), p1 = (int)((int64_t)(wave + 1) * n_parts / NW);
  if constexpr (sizeof(T) == 4) {
    if ((n & 3) == 0 && (reinterpret_cast<uintptr_t>(slab) & 15) == 0 && red_cap >= NW * 256) {
      const int nv = n >> 2;
      for (int c0 = 0; c0 < nv; c0 += 64) {
        const int i = c0 + lane;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        if (i < nv) {
          const f32x4* q = reinterpret_cast<const f32x4*>(slab) + (int64_t)p0 * nv + i;
          int p = p0;
          for (; p + 8 <= p1; p += 8, q += 8 * (int64_t)nv) {
            f32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = q[j * (int64_t)nv];
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[j];          // part order
          }
          for (; p < p1; ++p, q += nv) s += q[0];
        }
        __syncthreads();                                     // (red is free again)
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(red) + (wave * 64 + lane) * 4) = s;
        __syncthreads();
        if (wave == 0 && i < nv) {
          f32x4 t = *reinterpret_cast<const f32x4*>(reinterpret_cast<float*>(red) + lane * 4);
#pragma unroll
          for (int w = 1; w < NW; ++w) t += *reinterpret_cast<const f32x4*>(reinterpret_cast<float*>(red) + (w * 64 + lane) * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) add(4 * i + e, (T)t[e]);
        }
      }
      __syncthreads();
      return;
    }
  }
  // callers alias det_arrive's flag word into `red`: every wave must have read it before the first store below (a wave without slab
  // loads of its own gets here early; each pass of the loop ends on a barrier, so one barrier in front covers them all)
  __syncthreads();
  for (int c0 = 0; c0 < n; c0 += 64) {
    const int i = c0 + lane;
    T s = 0;
    if (i < n) {
      const T* q = slab + (int64_t)p0 * n + i;
      int p = p0;
      for (; p + 4 <= p1; p += 4, q += 4 * (int64_t)n) {      // four loads in flight, added in part order
        const T a = q[0], b = q[n], c = q[2 * (int64_t)n], d = q[3 * (int64_t)n];
        s += a; s += b; s += c; s += d;
      }
      for (; p < p1; ++p, q += n) s += q[0];
    }
    red[wave * 64 + lane] = s;
    __syncthreads();
    if (wave == 0 && i < n) {
      T t = red[lane];
#pragma unroll
      for (int w = 1; w < NW; ++w) t += red[w * 64 + lane];
      add(i, t);
    }
    __syncthreads();
  }
}

// Two-level form for reductions over HUNDREDS of workgroups (LayerNorm / the Q|K|V split: one slab per workgroup of a row-parallel
// launch): parts are grouped G at a time; the last arriver of a group adds its G slabs into a level-2 slab, the last of those group
// reducers adds the level-2 slabs and accumulates into the result.  The group reductions run concurrently on different CUs while other
// groups are still computing; only the last group's sum and the short final sum are exposed.  Order: parts within a group, then groups
// -- fixed.  tickets: n_groups + 1 words; slabs: (n_parts + n_groups) * n floats.
template <int NW, typename F>
__device__ __forceinline__ void det_reduce_tree(float* slabs, unsigned* tickets, int part, int n_parts, int G, int n, float* red,
                                                int red_cap, unsigned* flag, F&& add) {
  const int n_groups = (n_parts + G - 1) / G, g = part / G;
  const int in_g = min(G, n_parts - g * G);
  float* l2 = slabs + (int64_t)n_parts * n;
  if (!det_arrive(tickets + g, in_g, flag)) return;
  det_sum_slabs<NW>(slabs + (int64_t)g * G * n, in_g, n, red, red_cap, [&](int i, float t) { l2[(int64_t)g * n + i] = t; });
  if (!det_arrive(tickets + n_groups, n_groups, flag)) return;
  det_sum_slabs<NW>(l2, n_groups, n, red, red_cap, add);
}

}  // namespace
