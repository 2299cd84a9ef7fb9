// How does v_mfma_f32_32x32x16_bf16 round its f32 accumulate?  (round 5: the emulated arithmetic's one-term error measured 2.4 x 2^-24
// where "dropped terms + one correctly rounded add" allows 2.0.)
//   hipcc --offload-arch=gfx950 -O2 -o tools/bf16_acc_probe tools/bf16_acc_probe.hip && tools/bf16_acc_probe
// One non-zero product per output (A row r has a[r][0] != 0, B column c has b[0][c] != 0, everything else 0) and an accumulator input
// c[r][c] of chosen relative magnitude: d = mfma(a, b, c) against the correctly rounded fl(a b + c) from fp64.
//  case 1: |c| ~ 2^-8 |a b|  (the emulation's last accumulate: small partial sums in c, the leading product arrives)
//  case 2: |c| ~ 2^+8 |a b|  (the reverse order: leading product already in c, a 2^-8 partial product arrives)
//  case 3: |c| ~ |a b|
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void probe(const float* a, const float* b, const float* cin, float* dout) {   // a[32], b[32] (bf16-representable), c[32][32]
  const int lane = threadIdx.x, r = lane & 31, g = lane >> 5;
  bf16x8 av = {0, 0, 0, 0, 0, 0, 0, 0}, bv = {0, 0, 0, 0, 0, 0, 0, 0};
  if (g == 0) { av[0] = (bf16)a[r]; bv[0] = (bf16)b[r]; }     // contraction element k = 0 only
  f32x16 c;
  for (int i = 0; i < 16; ++i) { const int row = (i & 3) + 8 * (i >> 2) + 4 * g; c[i] = cin[row * 32 + r]; }
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c, 0, 0, 0);
  for (int i = 0; i < 16; ++i) { const int row = (i & 3) + 8 * (i >> 2) + 4 * g; dout[row * 32 + r] = c[i]; }
}

static double urand() { return (rand() + 0.5) / ((double)RAND_MAX + 1.0); }
static float bf16_round(float v) { uint32_t u; memcpy(&u, &v, 4); u = (u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000u; memcpy(&v, &u, 4); return v; }

int main() {
  srand(7);
  float *da, *db, *dc, *dd;
  CK(hipMalloc(&da, 128)); CK(hipMalloc(&db, 128)); CK(hipMalloc(&dc, 4096)); CK(hipMalloc(&dd, 4096));
  const double rel[3] = {ldexp(1.0, -8), ldexp(1.0, 8), 1.0};
  const char* name[3] = {"|c| ~ 2^-8 |a b|", "|c| ~ 2^+8 |a b|", "|c| ~ |a b|"};
  for (int cs = 0; cs < 3; ++cs) {
    double max_ulp = 0, sum = 0, bias = 0;
    long n = 0, wrong = 0;
    for (int rep = 0; rep < 400; ++rep) {
      float a[32], b[32], c[1024], d[1024];
      for (int i = 0; i < 32; ++i) { a[i] = bf16_round((float)((1.0 + urand()) * (urand() < 0.5 ? -1 : 1))); b[i] = bf16_round((float)((1.0 + urand()) * (urand() < 0.5 ? -1 : 1))); }
      for (int r = 0; r < 32; ++r)
        for (int q = 0; q < 32; ++q) c[r * 32 + q] = (float)((double)a[r] * b[q] * rel[cs] * (0.5 + urand()) * (urand() < 0.5 ? -1 : 1));
      CK(hipMemcpy(da, a, 128, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b, 128, hipMemcpyHostToDevice)); CK(hipMemcpy(dc, c, 4096, hipMemcpyHostToDevice));
      probe<<<1, 64>>>(da, db, dc, dd);
      CK(hipMemcpy(d, dd, 4096, hipMemcpyDeviceToHost));
      for (int r = 0; r < 32; ++r)
        for (int q = 0; q < 32; ++q) {
          const double exact = (double)a[r] * (double)b[q] + (double)c[r * 32 + q];
          const float rn = (float)exact;                         // correctly rounded
          int e;
          frexp(exact, &e);
          const double ulp = ldexp(1.0, e - 24);
          const double err = ((double)d[r * 32 + q] - exact) / ulp;
          max_ulp = fmax(max_ulp, fabs(err)); sum += fabs(err); bias += err * (exact > 0 ? 1 : -1);
          wrong += d[r * 32 + q] != rn;
          ++n;
        }
    }
    printf("%-18s max |error| %.3f ulp, mean %.3f ulp, signed mean (towards larger magnitude +) %+.3f ulp, differs from the correctly rounded result in %.1f %% of %ld\n",
           name[cs], max_ulp, sum / n, bias / n, 100.0 * wrong / n, n);
  }
  return 0;
}
