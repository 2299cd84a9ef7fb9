// Does a chain of DEPENDENT v_mfma_f32_16x16x4_f32 (same accumulator back to back -- what csrc/attention.hip's mma16(f32x8) issues: 8 in
// a row per fragment) run at the full matrix-pipe rate?  One kernel, NCH independent accumulators used round-robin in runs of RUN
// dependent instructions; cycles per MFMA from s_memtime at 1 and 2 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 -Wno-unused-value tools/mfma_chain_f32.hip -o tools/mfma_chain_f32 && tools/mfma_chain_f32
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NCH, int RUN>
__global__ __launch_bounds__(256) void chain(const float* src, float* out, int iters, unsigned long long* clk) {
  float a[8], b[8];
  for (int j = 0; j < 8; ++j) { a[j] = src[(threadIdx.x * 16 + j) & 4095]; b[j] = src[(threadIdx.x * 16 + 8 + j) & 4095]; }
  f32x4 acc[NCH];
  for (int c = 0; c < NCH; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j0 = 0; j0 < 8; j0 += RUN)
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int j = j0; j < j0 + RUN; ++j) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc[c], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  f32x4 s = acc[0];
  for (int c = 1; c < NCH; ++c) s += acc[c];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int NCH, int RUN>
void run(const float* src, float* out, unsigned long long* clk, int wgs_per_cu) {
  const int iters = 4096 / NCH;     // 8 NCH MFMAs per iteration: the same 32 768 MFMAs per wave in every variant
  hipLaunchKernelGGL((chain<NCH, RUN>), dim3(256 * wgs_per_cu), dim3(256), 0, 0, src, out, iters, clk);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((chain<NCH, RUN>), dim3(256 * wgs_per_cu), dim3(256), 0, 0, src, out, iters, clk);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
  const double n = 32768.0;
  const double flops = 256.0 * wgs_per_cu * 4 * n * 2048;
  printf("chains %2d x runs of %d, %d wave(s)/SIMD: %6.1f memtime ticks / MFMA (100 MHz clock), %7.3f ms, %6.1f TFLOP/s\n", NCH, RUN, wgs_per_cu,
         (double)c / n, ms, flops / ms * 1e-9);
}

int main() {
  float *src, *out; unsigned long long* clk;
  hipMalloc(&src, 4096 * 4); hipMalloc(&out, 256 * 2 * 256 * 4); hipMalloc(&clk, 8);
  hipMemset(src, 0, 4096 * 4);
  for (int w = 1; w <= 2; ++w) {
    run<1, 8>(src, out, clk, w);     // one accumulator, all dependent
    run<2, 8>(src, out, clk, w);     // attention's S loop: two fragments, 8 dependent each
    run<16, 8>(src, out, clk, w);    // attention's P V loop: 16 fragments, 8 dependent each
    run<2, 1>(src, out, clk, w);     // two accumulators alternating
    run<4, 1>(src, out, clk, w);
    run<16, 1>(src, out, clk, w);    // fully interleaved
  }
  return 0;
}
