// jatts_mfma_probe: the matrix pipe's PRACTICAL ceiling on this part, measured live (bench.py: roofline.practical_peak).
//
// The dense 16-bit MFMAs of MI355X are power-limited: with realistic operand bits the shader clock settles well under the 2.4 GHz
// the 2.5 PFLOP/s spec peak is quoted at (MI355X_MICROARCH.md, "DVFS give-back": 1 247-1 483 TFLOP/s for tuned bf16 kernels).  A
// kernel priced against the spec peak alone reads "half the machine idle" when its pipe is saturated at the clock the power budget
// allows.  This probe is the same MFMA stream with everything else stripped: 2 x 2 fragments of 32 x 32 per wave, four waves per
// workgroup, operands re-read every K-step
//   feed 1: from LDS (ds_read_b128; both operands -- the way any kernel that does not keep its operands in registers must feed them),
//   feed 0: from registers only (no operand traffic at all: the upper bound of the issue rate at the sustained clock),
// no global traffic, no barriers, no epilogue.  It reports nothing itself: the caller times the launch (HIP events) and divides
// jatts_mfma_probe_flops() by it; clocks[0..1] = s_memtime / s_memrealtime ticks of workgroup 0 (sustained shader clock =
// clocks[0] / (clocks[1] * 10 ns)).
//   dtype JATTS_F32E / JATTS_F32E6 : v_mfma_f32_32x32x16_bf16      JATTS_F16 / JATTS_F32S : v_mfma_f32_32x32x16_f16
//   dtype JATTS_F32               : v_mfma_f32_32x32x2_f32 (8 per K-step: the exact-f32 chain)
#include "common.h"

namespace {

constexpr int PROBE_NF = 2, PROBE_NT = 2, PROBE_WAVES = 4;
constexpr int PROBE_ROWS = 128, PROBE_PITCH = 272;      // LDS image: 128 rows x (128 16-bit elements + 16 B pad) -- conflict-free ds_read_b128

template <int KIND> struct ProbeOp;                      // 0 bf16, 1 f16, 2 f32
template <> struct ProbeOp<0> { typedef bf16x8 V; static constexpr int kBytes = 16; };
template <> struct ProbeOp<1> { typedef f16x8 V; static constexpr int kBytes = 16; };
template <> struct ProbeOp<2> { typedef f32x8 V; static constexpr int kBytes = 32; };

__device__ __forceinline__ void probe_mma(const bf16x8& a, const bf16x8& b, f32x16& c) { c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ void probe_mma(const f16x8& a, const f16x8& b, f32x16& c) { mma32(a, b, c); }
__device__ __forceinline__ void probe_mma(const f32x8& a, const f32x8& b, f32x16& c) { mma32(a, b, c); }

template <int KIND, bool LDSFED>
__global__ __launch_bounds__(PROBE_WAVES * 64, 2) void mfma_probe_kernel(const char* __restrict__ src, int64_t src_bytes, int iters,
                                                                          unsigned long long* clocks, float* sink) {
  typedef typename ProbeOp<KIND>::V V;
  constexpr int EB = ProbeOp<KIND>::kBytes;              // bytes one lane supplies per fragment
  constexpr int PITCH = KIND == 2 ? 2 * PROBE_PITCH - 16 : PROBE_PITCH;
  __shared__ __attribute__((aligned(16))) char lds[PROBE_ROWS * PITCH];
  // fill the LDS image from the caller's operand bytes (wrapping), 16 B per thread per pass
  for (int u = threadIdx.x; u < PROBE_ROWS * PITCH / 16; u += PROBE_WAVES * 64)
    *reinterpret_cast<f32x4*>(lds + (size_t)u * 16) =
        *reinterpret_cast<const f32x4*>(src + (((size_t)blockIdx.x * 4096 + (size_t)u * 16) % (size_t)(src_bytes - 16) & ~(size_t)15));
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16 acc[PROBE_NF][PROBE_NT];
#pragma unroll
  for (int f = 0; f < PROBE_NF; ++f)
#pragma unroll
    for (int t = 0; t < PROBE_NT; ++t)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[f][t][j] = 0.f;
  // fragment (row block r, K-step s): row r * 32 + (lane & 31), bytes (s * 2 + (lane >> 5)) * EB ..
  auto frag = [&](int rblk, int s) -> V {
    return *reinterpret_cast<const V*>(lds + (size_t)(rblk * 32 + (lane & 31)) * PITCH + (size_t)(((s & 3) * 2 + (lane >> 5)) * EB));
  };
  V a0[PROBE_NF], b0[PROBE_NT], a1[PROBE_NF], b1[PROBE_NT];
#pragma unroll
  for (int f = 0; f < PROBE_NF; ++f) a0[f] = frag(f, wave), a1[f] = frag(f, wave + 1);
#pragma unroll
  for (int t = 0; t < PROBE_NT; ++t) b0[t] = frag(2 + t, wave), b1[t] = frag(2 + t, wave + 1);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it += 2) {
    if constexpr (LDSFED) {
#pragma unroll
      for (int f = 0; f < PROBE_NF; ++f) a1[f] = frag(f, it + 1);
#pragma unroll
      for (int t = 0; t < PROBE_NT; ++t) b1[t] = frag(2 + t, it + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int f = 0; f < PROBE_NF; ++f)
#pragma unroll
      for (int t = 0; t < PROBE_NT; ++t) probe_mma(a0[f], b0[t], acc[f][t]);
    if constexpr (LDSFED) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int f = 0; f < PROBE_NF; ++f) a0[f] = frag(f, it + 2);
#pragma unroll
      for (int t = 0; t < PROBE_NT; ++t) b0[t] = frag(2 + t, it + 2);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int f = 0; f < PROBE_NF; ++f)
#pragma unroll
      for (int t = 0; t < PROBE_NT; ++t) probe_mma(a1[f], b1[t], acc[f][t]);
    if constexpr (LDSFED) __builtin_amdgcn_sched_barrier(0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int f = 0; f < PROBE_NF; ++f)
#pragma unroll
    for (int t = 0; t < PROBE_NT; ++t)
#pragma unroll
      for (int j = 0; j < 16; ++j) s += acc[f][t][j];
  if (s == 12345.678f) sink[0] = s;                     // keeps the accumulators alive; never true for the probe's operands
  if (blockIdx.x == 0 && threadIdx.x == 0 && clocks) { clocks[0] = t1 - t0; clocks[1] = r1 - r0; }
}

template <bool F16>
__device__ __forceinline__ f32x4 probe16_mma(const bf16x8& a, const bf16x8& b, const f32x4& c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// The same probe on v_mfma_f32_16x16x32_bf16: 4 x 4 fragments of 16 x 16 per wave (the operand bytes per flop of the 2 x 2 x 32 x 32 form), K = 32 per step.
// Half the accumulator elements written per flop of the 32 x 32 x 16 form: does the power-limited pipe sustain more with it?  (dtype code 16 + JATTS_F32E)
template <bool LDSFED, bool F16 = false>      // F16: v_mfma_f32_16x16x32_f16 on the same fragment geometry (dtype code 16 + JATTS_F16)
__global__ __launch_bounds__(PROBE_WAVES * 64, 2) void mfma_probe16_kernel(const char* __restrict__ src, int64_t src_bytes, int iters,
                                                                            unsigned long long* clocks, float* sink) {
  constexpr int PITCH = PROBE_PITCH;
  __shared__ __attribute__((aligned(16))) char lds[PROBE_ROWS * PITCH];
  for (int u = threadIdx.x; u < PROBE_ROWS * PITCH / 16; u += PROBE_WAVES * 64)
    *reinterpret_cast<f32x4*>(lds + (size_t)u * 16) =
        *reinterpret_cast<const f32x4*>(src + (((size_t)blockIdx.x * 4096 + (size_t)u * 16) % (size_t)(src_bytes - 16) & ~(size_t)15));
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 acc[4][4];
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[f][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  // fragment (row block r of 16 rows, K-step s): row r * 16 + (lane & 15), bytes ((s & 1) * 4 + (lane >> 4)) * 16 ..
  auto frag = [&](int rblk, int s) -> bf16x8 {
    return *reinterpret_cast<const bf16x8*>(lds + (size_t)(rblk * 16 + (lane & 15)) * PITCH + (size_t)(((s & 1) * 4 + (lane >> 4)) * 16));
  };
  bf16x8 a0[4], b0[4], a1[4], b1[4];
#pragma unroll
  for (int f = 0; f < 4; ++f) a0[f] = frag(f, wave), a1[f] = frag(f, wave + 1), b0[f] = frag(4 + f, wave), b1[f] = frag(4 + f, wave + 1);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it += 2) {
    if constexpr (LDSFED) {
#pragma unroll
      for (int f = 0; f < 4; ++f) a1[f] = frag(f, it + 1), b1[f] = frag(4 + f, it + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[f][t] = probe16_mma<F16>(a0[f], b0[t], acc[f][t]);
    if constexpr (LDSFED) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int f = 0; f < 4; ++f) a0[f] = frag(f, it + 2), b0[f] = frag(4 + f, it + 2);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[f][t] = probe16_mma<F16>(a1[f], b1[t], acc[f][t]);
    if constexpr (LDSFED) __builtin_amdgcn_sched_barrier(0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) s += acc[f][t][j];
  if (s == 12345.678f) sink[0] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0 && clocks) { clocks[0] = t1 - t0; clocks[1] = r1 - r0; }
}

template <int KIND>
int probe_launch(int feed, const void* src, int64_t src_bytes, int iters, int wgs, unsigned long long* clocks, float* sink, hipStream_t s) {
  if (feed) hipLaunchKernelGGL((mfma_probe_kernel<KIND, true>), dim3(wgs), dim3(PROBE_WAVES * 64), 0, s, (const char*)src, src_bytes, iters, clocks, sink);
  else hipLaunchKernelGGL((mfma_probe_kernel<KIND, false>), dim3(wgs), dim3(PROBE_WAVES * 64), 0, s, (const char*)src, src_bytes, iters, clocks, sink);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

}  // namespace

extern "C" double jatts_mfma_probe_flops(int32_t dtype, int32_t iters, int32_t workgroups) {
  // one K-step = 32 x 32 x 16 multiply-adds per fragment in every dtype (f32: eight 32x32x2 MFMAs); the 16 x 16 x 32 form: 16 fragments of 16 x 16 x 32
  if (dtype >= 16) return (double)workgroups * PROBE_WAVES * (double)(iters + (iters & 1)) * 16 * (2.0 * 16 * 16 * 32);
  return (double)workgroups * PROBE_WAVES * (double)(iters + (iters & 1)) * PROBE_NF * PROBE_NT * (2.0 * 32 * 32 * 16);
}

extern "C" int jatts_mfma_probe(int32_t dtype, int32_t feed, const void* operands, int64_t operand_bytes, int32_t iters, int32_t workgroups,
                                uint64_t* clocks, float* sink, void* stream) {
  if (!operands || operand_bytes < 65536 || ((uintptr_t)operands & 15) || iters < 2 || workgroups < 1 || !sink)
    return jatts_set_error_msg(JATTS_ERR_ARG, "mfma_probe: operands (>= 64 KiB, 16-byte aligned), iters >= 2, workgroups >= 1, sink required");
  hipStream_t s = (hipStream_t)stream;
  unsigned long long* c = (unsigned long long*)clocks;
  if (dtype == 16 + JATTS_F32E) {
    if (feed) hipLaunchKernelGGL((mfma_probe16_kernel<true>), dim3(workgroups), dim3(PROBE_WAVES * 64), 0, s, (const char*)operands, operand_bytes, iters, c, sink);
    else hipLaunchKernelGGL((mfma_probe16_kernel<false>), dim3(workgroups), dim3(PROBE_WAVES * 64), 0, s, (const char*)operands, operand_bytes, iters, c, sink);
    JATTS_CHECK_LAUNCH();
    return JATTS_OK;
  }
  if (dtype == 16 + JATTS_F16) {
    if (feed) hipLaunchKernelGGL((mfma_probe16_kernel<true, true>), dim3(workgroups), dim3(PROBE_WAVES * 64), 0, s, (const char*)operands, operand_bytes, iters, c, sink);
    else hipLaunchKernelGGL((mfma_probe16_kernel<false, true>), dim3(workgroups), dim3(PROBE_WAVES * 64), 0, s, (const char*)operands, operand_bytes, iters, c, sink);
    JATTS_CHECK_LAUNCH();
    return JATTS_OK;
  }
  switch (dtype) {
    case JATTS_F32E: case JATTS_F32E6: return probe_launch<0>(feed, operands, operand_bytes, iters, workgroups, c, sink, s);
    case JATTS_F16: case JATTS_F32S: return probe_launch<1>(feed, operands, operand_bytes, iters, workgroups, c, sink, s);
    case JATTS_F32: return probe_launch<2>(feed, operands, operand_bytes, iters, workgroups, c, sink, s);
  }
  return jatts_set_error_msg(JATTS_ERR_ARG, "mfma_probe: unknown dtype");
}
