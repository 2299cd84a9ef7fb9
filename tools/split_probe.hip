// Numerics probe for the error-corrected split-precision MFMA path (round 4).
//   hipcc --offload-arch=gfx950 -O2 -o tools/split_probe tools/split_probe.hip && tools/split_probe
// Questions it answers on the hardware:
//  1. does v_mfma_f32_32x32x16_f16 keep f16 SUBNORMAL inputs (or flush them)?
//  2. how does its f32 accumulation round (RNE-like zero-mean error, or a truncation bias)?
//  3. x = hi + lo (f16 each, power-of-two tile / channel scales), products hi.hi + hi.lo + lo.hi:
//     error against fp64 of (a) ONE f32 accumulator for all three products, (b) main + correction accumulators,
//     (c) the exact-f32 MFMA chain (v_mfma_f32_32x32x2_f32) -- on conv-shaped dot products (K = C k = 96 ... 2816).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// A: [32][K] row-major (f16 hi / lo planes, or f32), B: [K][32] stored as Bt[32][K] (column n contiguous in k).
// mode 0: f32 MFMA chain; 1: split, one accumulator; 2: split, main + correction accumulator; 3: hi only (plain f16)
__global__ void probe(const float* A, const float* Bt, const f16* Ahi, const f16* Alo, const f16* Bhi, const f16* Blo, int K, int mode,
                      float* C) {
  const int lane = threadIdx.x, r = lane & 31, g = lane >> 5;
  f32x16 acc = {0}, cor = {0};
  for (int k0 = 0; k0 < K; k0 += 16) {
    if (mode == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = k0 + 8 * g + j;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + k], Bt[r * K + k], acc, 0, 0, 0);
      }
    } else {
      const f16x8 ah = *reinterpret_cast<const f16x8*>(Ahi + r * K + k0 + 8 * g), al = *reinterpret_cast<const f16x8*>(Alo + r * K + k0 + 8 * g);
      const f16x8 bh = *reinterpret_cast<const f16x8*>(Bhi + r * K + k0 + 8 * g), bl = *reinterpret_cast<const f16x8*>(Blo + r * K + k0 + 8 * g);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
      if (mode == 1) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
      } else if (mode == 2) {
        cor = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, cor, 0, 0, 0);
        cor = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, cor, 0, 0, 0);
      }
    }
  }
  for (int i = 0; i < 16; ++i) {
    const int row = (i & 3) + 8 * (i >> 2) + 4 * g;   // row = A row (n), col = B column (t)
    C[row * 32 + r] = acc[i] + cor[i];
  }
}

static double urand() { return (rand() + 0.5) / ((double)RAND_MAX + 1.0); }
static double nrand() { return sqrt(-2.0 * log(urand())) * cos(2.0 * M_PI * urand()); }

struct Dev {
  float *A, *Bt, *C;
  f16 *Ahi, *Alo, *Bhi, *Blo;
};

static void split(const std::vector<float>& v, const std::vector<float>& scale_per_row, int K, std::vector<f16>& hi, std::vector<f16>& lo) {
  hi.resize(v.size());
  lo.resize(v.size());
  for (size_t i = 0; i < v.size(); ++i) {
    const float s = v[i] * scale_per_row[i / K];
    const f16 h = (f16)s;
    hi[i] = h;
    lo[i] = (f16)(s - (float)h);
  }
}

static float pow2_scale(float amax, int target_exp) {   // 2^(target_exp - e) with amax = m 2^e, m in [0.5, 1)
  if (!(amax > 0.f)) return 1.f;
  int e;
  frexpf(amax, &e);
  return ldexpf(1.f, target_exp - e);
}

int main() {
  srand(1234);
  Dev d;
  const int KMAX = 4096;
  CK(hipMalloc(&d.A, 32 * KMAX * 4)); CK(hipMalloc(&d.Bt, 32 * KMAX * 4)); CK(hipMalloc(&d.C, 32 * 32 * 4));
  CK(hipMalloc(&d.Ahi, 32 * KMAX * 2)); CK(hipMalloc(&d.Alo, 32 * KMAX * 2)); CK(hipMalloc(&d.Bhi, 32 * KMAX * 2)); CK(hipMalloc(&d.Blo, 32 * KMAX * 2));

  // ---- 1. subnormal inputs
  {
    const int K = 16;
    std::vector<f16> ah(32 * K, (f16)0.f), z(32 * K, (f16)0.f), bh(32 * K, (f16)0.f);
    const float tiny = ldexpf(1.f, -20);              // f16 subnormal (min normal 2^-14)
    for (int r = 0; r < 32; ++r) { ah[r * K] = (f16)tiny; bh[r * K] = (f16)1.f; ah[r * K + 1] = (f16)1.f; bh[r * K + 1] = (f16)tiny; }
    CK(hipMemcpy(d.Ahi, ah.data(), ah.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(d.Alo, z.data(), z.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(d.Bhi, bh.data(), bh.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(d.Blo, z.data(), z.size() * 2, hipMemcpyHostToDevice));
    probe<<<1, 64>>>(d.A, d.Bt, d.Ahi, d.Alo, d.Bhi, d.Blo, K, 3, d.C);
    float c[1024];
    CK(hipMemcpy(c, d.C, sizeof(c), hipMemcpyDeviceToHost));
    printf("subnormal f16 inputs: A=2^-20 * B=1 + A=1 * B=2^-20 -> %.9g (expected %.9g; 0 = flushed)\n", c[0], 2.0 * tiny);
  }

  // ---- 2 / 3. conv-shaped dot products
  printf("%-34s %6s | %-23s | %-23s | %-23s | %-23s\n", "case", "K", "f32 MFMA chain", "split, 1 accumulator", "split, 2 accumulators", "plain f16 (hi only)");
  printf("%-34s %6s | %-23s | %-23s | %-23s | %-23s\n", "", "", "max      rms     bias", "max      rms     bias", "max      rms     bias", "max      rms     bias");
  struct Case { const char* name; int K; double wstd; int xkind; };
  const Case cases[] = {
      {"w N(0,.01), x lrelu N(0,1)", 96, 0.01, 0},   {"w N(0,.01), x lrelu N(0,1)", 352, 0.01, 0},  {"w N(0,.01), x lrelu N(0,1)", 1408, 0.01, 0},
      {"w N(0,.01), x lrelu N(0,1)", 2816, 0.01, 0}, {"w N(0,.05), x positive |N|", 1408, 0.05, 1}, {"w N(0,.01), x wide dynamic 1e-4..10", 1408, 0.01, 2},
      {"w N(0,1), x N(0,1)*300 (large)", 1408, 1.0, 3}, {"w all +, x all + (biased sums)", 1408, 0.01, 4},
  };
  for (const Case& cs : cases) {
    const int K = cs.K, reps = 24;
    double mx[4] = {0}, sq[4] = {0}, bias[4] = {0};
    size_t cnt = 0;
    for (int rep = 0; rep < reps; ++rep) {
      std::vector<float> A(32 * K), Bt(32 * K);
      for (auto& v : A) v = (float)(nrand() * cs.wstd);
      if (cs.xkind == 4) for (auto& v : A) v = fabsf(v);
      for (auto& v : Bt) {
        double x = nrand();
        if (cs.xkind == 0) x = x > 0 ? x : 0.1 * x;
        else if (cs.xkind == 1 || cs.xkind == 4) x = fabs(x);
        else if (cs.xkind == 2) x = x * pow(10.0, -4.0 + 5.0 * urand());
        else if (cs.xkind == 3) x = x * 300.0;
        v = (float)x;
      }
      // scales: weights per output channel (row of A) to [2^14, 2^15); activations one scale for the tile
      std::vector<float> sa(32), sb(32);
      for (int r = 0; r < 32; ++r) {
        float m = 0;
        for (int k = 0; k < K; ++k) m = fmaxf(m, fabsf(A[r * K + k]));
        sa[r] = pow2_scale(m, 15);
      }
      float mb = 0;
      for (float v : Bt) mb = fmaxf(mb, fabsf(v));
      for (int r = 0; r < 32; ++r) sb[r] = pow2_scale(mb, 15);
      std::vector<f16> ah, al, bh, bl;
      split(A, sa, K, ah, al);
      split(Bt, sb, K, bh, bl);
      CK(hipMemcpy(d.A, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d.Bt, Bt.data(), Bt.size() * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(d.Ahi, ah.data(), ah.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(d.Alo, al.data(), al.size() * 2, hipMemcpyHostToDevice));
      CK(hipMemcpy(d.Bhi, bh.data(), bh.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(d.Blo, bl.data(), bl.size() * 2, hipMemcpyHostToDevice));
      std::vector<double> ref(1024), mag(1024);
      for (int n = 0; n < 32; ++n)
        for (int t = 0; t < 32; ++t) {
          double s = 0, m = 0;
          for (int k = 0; k < K; ++k) { const double p = (double)A[n * K + k] * (double)Bt[t * K + k]; s += p; m += fabs(p); }
          ref[n * 32 + t] = s; mag[n * 32 + t] = m;
        }
      for (int mode = 0; mode < 4; ++mode) {
        probe<<<1, 64>>>(d.A, d.Bt, d.Ahi, d.Alo, d.Bhi, d.Blo, K, mode, d.C);
        float c[1024];
        CK(hipMemcpy(c, d.C, sizeof(c), hipMemcpyDeviceToHost));
        for (int n = 0; n < 32; ++n)
          for (int t = 0; t < 32; ++t) {
            double v = c[n * 32 + t];
            if (mode) v = v / ((double)sa[n] * (double)sb[0]);       // epilogue: exact power-of-two unscale
            v = (double)(float)v;
            const double e = (v - ref[n * 32 + t]) / mag[n * 32 + t];  // relative to sum |a b| (the natural error scale of a dot product)
            mx[mode] = fmax(mx[mode], fabs(e)); sq[mode] += e * e; bias[mode] += e;
          }
      }
      cnt += 1024;
    }
    printf("%-34s %6d |", cs.name, K);
    for (int m = 0; m < 4; ++m) printf(" %8.2e %8.2e %+8.1e |", mx[m], sqrt(sq[m] / cnt), bias[m] / cnt);
    printf("\n");
  }
  printf("(errors relative to sum_k |a_k b_k|; 2^-24 = 5.96e-08; split scales: weights per output channel and activations per tile to [2^14, 2^15))\n");
  return 0;
}
