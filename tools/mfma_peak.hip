// Register-only MFMA ceiling of this GPU: v_mfma_f32_32x32x16_f16 back to back on 4 (or 8) independent accumulators,
// no memory traffic.  Prints TFLOP/s and the sustained shader clock (s_memtime vs the 100 MHz s_memrealtime): dense
// MFMA is power-limited on MI355X, so the "peak" to price an MFMA-bound kernel against is what this prints.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o tools/mfma_peak && tools/mfma_peak [zero|rand]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k(const _Float16* src, float* out, int iters, unsigned long long* clk) {
  f16x8 a = *reinterpret_cast<const f16x8*>(src + (threadIdx.x & 63) * 8);
  f16x8 b = *reinterpret_cast<const f16x8*>(src + 512 + (threadIdx.x & 63) * 8);
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  if (s == 12345.678f) out[0] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

int main(int argc, char** argv) {
  const bool zero = argc > 1 && !strcmp(argv[1], "zero");
  std::vector<_Float16> h(1024);
  unsigned s = 12345;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = zero ? (_Float16)0.f : (_Float16)(((s >> 8) & 0xffff) / 65536.f - 0.5f); }
  _Float16* d; float* o; unsigned long long* c;
  hipMalloc(&d, 2048); hipMalloc(&o, 4); hipMalloc(&c, 16);
  hipMemcpy(d, h.data(), 2048, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd) {
    const int blocks = 256 * waves_per_simd, iters = 40000;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, d, o, iters, c);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long hc[2]; hipMemcpy(hc, c, 16, hipMemcpyDeviceToHost);
      double flops = (double)blocks * 4 * iters * 4 * 32768.0;
      printf("%s operands, %d wave(s)/SIMD: %.2f ms  %.0f TFLOP/s  shader clock %.3f GHz\n", zero ? "zero" : "random", waves_per_simd, ms,
             flops / ms / 1e9, (double)hc[0] / ((double)hc[1] * 10.0));
    }
  }
  return 0;
}
