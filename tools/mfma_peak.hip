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

// Same MFMA stream fed the way the fused unit feeds it: per K-step NF weight fragments from global memory (L2-resident,
// 1 KiB per wave-load) and NT activation fragments from LDS (ds_read_b128), NF*NT MFMAs.  Shows how far the operand
// traffic alone pulls the power-limited ceiling down.
template <int NF, int NT>
__global__ __launch_bounds__(256) void kfed(const _Float16* w, float* out, int iters, unsigned long long* clk) {
  __shared__ __attribute__((aligned(16))) _Float16 lds[128 * 136];
  for (int i = threadIdx.x; i < 128 * 136; i += 256) lds[i] = w[i & 8191];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16 acc[NF][NT];
  for (int f = 0; f < NF; ++f) for (int t = 0; t < NT; ++t) for (int j = 0; j < 16; ++j) acc[f][t][j] = 0.f;
  const _Float16* wl = w + lane * 8 + wave * 1024;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  f16x8 a0[NF], a1[NF], b0[NT], b1[NT];
#define LOADA(dst, it) for (int f = 0; f < NF; ++f) dst[f] = *reinterpret_cast<const f16x8*>(wl + (size_t)((it) & 255) * 4096 + f * 512)
#define LOADB(dst, it) for (int t = 0; t < NT; ++t) dst[t] = *reinterpret_cast<const f16x8*>(&lds[((lane & 31) + 32 * t) * 136 + ((((it) & 63) * 16) & 127) + 8 * (lane >> 5)])
#define MMA(aa, bb) _Pragma("unroll") for (int f = 0; f < NF; ++f) _Pragma("unroll") for (int t = 0; t < NT; ++t) acc[f][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aa[f], bb[t], acc[f][t], 0, 0, 0)
  LOADA(a0, 0); LOADB(b0, 0);
  for (int it = 0; it < iters; it += 2) {
    LOADA(a1, it + 1); LOADB(b1, it + 1);
    __builtin_amdgcn_sched_barrier(0);
    MMA(a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    LOADA(a0, it + 2); LOADB(b0, it + 2);
    __builtin_amdgcn_sched_barrier(0);
    MMA(a1, b1);
    __builtin_amdgcn_sched_barrier(0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int f = 0; f < NF; ++f) for (int t = 0; t < NT; ++t) for (int j = 0; j < 16; ++j) s += acc[f][t][j];
  if (s == 12345.678f) out[0] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int NF, int NT>
void run_fed(const _Float16* dw, float* o, unsigned long long* c, hipEvent_t e0, hipEvent_t e1) {
  const int blocks = 512, iters = 20000;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((kfed<NF, NT>), dim3(blocks), dim3(256), 0, 0, dw, o, iters, c);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long hc[2]; hipMemcpy(hc, c, 16, hipMemcpyDeviceToHost);
    double flops = (double)blocks * 4 * iters * NF * NT * 32768.0;
    printf("fed NF=%d NT=%d (2 waves/SIMD, weights from L2, activations from LDS): %.2f ms  %.0f TFLOP/s  shader clock %.3f GHz\n", NF, NT, ms,
           flops / ms / 1e9, (double)hc[0] / ((double)hc[1] * 10.0));
  }
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
  {
    std::vector<_Float16> hw(256 * 4096 + 8192);
    for (auto& v : hw) { s = s * 1664525u + 1013904223u; v = zero ? (_Float16)0.f : (_Float16)(((s >> 8) & 0xffff) / 65536.f - 0.5f); }
    _Float16* dw; hipMalloc(&dw, hw.size() * 2);
    hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    run_fed<2, 2>(dw, o, c, e0, e1);
    run_fed<2, 4>(dw, o, c, e0, e1);
    run_fed<4, 2>(dw, o, c, e0, e1);
    run_fed<4, 4>(dw, o, c, e0, e1);
  }
  return 0;
}
