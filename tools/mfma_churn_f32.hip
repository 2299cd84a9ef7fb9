// What does workgroup turnover cost an f32-MFMA-bound kernel?  (round 3: conv1d<float> sits at ~100 of 157 TFLOP/s even with NOTHING
// streamed in its main loop -- tools/bench_conv.py --variant 8.)  One kernel: every wave runs `iters` x 64 independent-accumulator
// v_mfma_f32_32x32x2_f32 from registers, optionally after a global-load prologue and before a store epilogue, with `lds` bytes of
// dynamic LDS allocated.  The same total work is launched as (a) 512 long workgroups (one round), (b) many short ones.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_churn_f32.hip -o tools/mfma_churn_f32 && tools/mfma_churn_f32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int EPI>
__global__ __launch_bounds__(256, 2) void churn(const float* src, float* out, int iters, unsigned long long* clk) {
  extern __shared__ char smem[];
  const int lane = threadIdx.x & 63;
  float a[2][8], b[2][8];
  const float* p = src + ((size_t)blockIdx.x * 256 + threadIdx.x) % 4096 * 8;   // 128 KB region: L2 resident
  for (int h = 0; h < 2; ++h)
    for (int j = 0; j < 8; ++j) { a[h][j] = p[h * 8 + j]; b[h][j] = p[16 + h * 8 + j]; }
  f32x16 acc[2][2];
  for (int f = 0; f < 2; ++f) for (int t = 0; t < 2; ++t) for (int r = 0; r < 16; ++r) acc[f][t][r] = 0.f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[f][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[f][j], b[t][j], acc[f][t], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (EPI == 1) {          // 64 KB per workgroup, row-contiguous 16-byte stores straight from the accumulators
    float* o = out + (size_t)blockIdx.x * 16384 + (size_t)threadIdx.x * 64;
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<f32x4*>(o + (f * 2 + t) * 16 + q * 4) = f32x4{acc[f][t][4 * q], acc[f][t][4 * q + 1], acc[f][t][4 * q + 2], acc[f][t][4 * q + 3]};
  } else {
    float s = 0.f;
    for (int f = 0; f < 2; ++f) for (int t = 0; t < 2; ++t) for (int r = 0; r < 16; ++r) s += acc[f][t][r];
    if (s == 12345.678f) out[0] = s + smem[lane];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

int main() {
  std::vector<float> h(4096 * 8 + 64);
  unsigned s = 12345;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.f - 0.5f; }
  float *d, *o; unsigned long long* c;
  hipMalloc(&d, h.size() * 4); hipMalloc(&o, (size_t)16384 * 4 * 65536); hipMalloc(&c, 16);
  hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const long total_iters = 512L * 8192;   // wave-iterations of 32 MFMAs per workgroup-wave, summed over workgroups
  for (int epi = 0; epi < 2; ++epi)
    for (int lds : {0, 66 * 1024, 100 * 1024})
      for (int blocks : {256, 512, 1024, 2048, 8192, 32768}) {
        const int iters = (int)(total_iters / blocks);
        auto kern = epi ? churn<1> : churn<0>;
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        float best = 1e9f; unsigned long long hc[2] = {0, 1};
        for (int rep = 0; rep < 4; ++rep) {
          hipEventRecord(e0);
          hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, d, o, iters, c);
          hipEventRecord(e1); hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1);
          if (ms < best) { best = ms; hipMemcpy(hc, c, 16, hipMemcpyDeviceToHost); }
        }
        const double flop = (double)blocks * 4 * iters * 32 * 4096.0;
        printf("epi %d lds %3d KB  %5d WGs x %5d iters (%6.1f us of MFMA per wave at 2.4 GHz): %.3f ms  %.1f TFLOP/s  wave-0 clock %.2f GHz\n", epi, lds / 1024,
               blocks, iters, iters * 2048 / 2400.0, best, flop / best / 1e9, (double)hc[0] / ((double)hc[1] * 10.0));
      }
  return 0;
}
