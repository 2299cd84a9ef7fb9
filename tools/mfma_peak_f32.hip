// Register-only f32 MFMA ceiling: v_mfma_f32_32x32x2_f32 vs v_mfma_f32_16x16x4_f32, random / zero operands, 1 or 2 waves per SIMD.
// Same FLOP rate by the book (64 FLOP/clk/SIMD); which one the power limit lets run at a higher clock is what this measures.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_peak_f32.hip -o tools/mfma_peak_f32 && tools/mfma_peak_f32 [zero]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k32(const float* src, float* out, int iters, unsigned long long* clk) {
  float a[8], b[8];
  for (int j = 0; j < 8; ++j) { a[j] = src[(threadIdx.x & 63) * 8 + j]; b[j] = src[512 + (threadIdx.x & 63) * 8 + j]; }
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  if (s == 12345.678f) out[0] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int NACC>
__global__ __launch_bounds__(256) void k16(const float* src, float* out, int iters, unsigned long long* clk) {
  float a[8], b[8];
  for (int j = 0; j < 8; ++j) { a[j] = src[(threadIdx.x & 63) * 8 + j]; b[j] = src[512 + (threadIdx.x & 63) * 8 + j]; }
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j];
  if (s == 12345.678f) out[0] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

int main(int argc, char** argv) {
  const bool zero = argc > 1 && !strcmp(argv[1], "zero");
  std::vector<float> h(1024);
  unsigned s = 12345;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = zero ? 0.f : ((s >> 8) & 0xffff) / 65536.f - 0.5f; }
  float *d, *o; unsigned long long* c;
  hipMalloc(&d, 4096); hipMalloc(&o, 4); hipMalloc(&c, 16);
  hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int shape = 0; shape < 2; ++shape)
    for (int wps = 1; wps <= 2; ++wps) {
      const int blocks = 256 * wps, iters = shape == 0 ? 4000 : 8000;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (shape == 0) hipLaunchKernelGGL(k32<4>, dim3(blocks), dim3(256), 0, 0, d, o, iters, c);
        else hipLaunchKernelGGL(k16<8>, dim3(blocks), dim3(256), 0, 0, d, o, iters, c);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long hc[2]; hipMemcpy(hc, c, 16, hipMemcpyDeviceToHost);
        const double per = shape == 0 ? 4 * 8 * 4096.0 : 8 * 8 * 2048.0;   // FLOP per wave per iteration
        printf("%s  %s operands, %d wave(s)/SIMD: %.2f ms  %.1f TFLOP/s  shader clock %.3f GHz\n", shape == 0 ? "32x32x2 " : "16x16x4 ",
               zero ? "zero" : "random", wps, ms, (double)blocks * 4 * iters * per / ms / 1e9, (double)hc[0] / ((double)hc[1] * 10.0));
      }
    }
  return 0;
}
