// Fused legacy relative-position self-attention (flash style) for gfx950.
//
// One workgroup = 4 waves = 64 queries of one (sequence, head); each wave owns 16 queries.
// K tiles [64 keys][d_k] and V^T tiles [d_k][64 keys] are staged in LDS and shared by
// the 4 waves.  Scores are computed TRANSPOSED (S^T = K Q^T) on 16x16 MFMA fragments so
// that a lane holds 4 consecutive keys of ONE query column: the softmax row reduction is
// in-lane adds plus two cross-lane steps (xor 16, 32), and the probabilities are already
// in B-operand position for O^T += V^T P^T -- no LDS round trip for P.
#include "common.h"

namespace {

template <typename T> struct V8 { typedef typename Elem<T>::vec8 type; };

template <typename T>
__device__ __forceinline__ typename Elem<T>::vec8 load8(const T* p) {
  typename Elem<T>::vec8 v;
  if (sizeof(T) == 2) {
    v = *reinterpret_cast<const typename Elem<T>::vec8*>(p);
  } else {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[e + 4] = b[e]; }
  }
  return v;
}

constexpr int QB = 64;  // queries per workgroup
constexpr int KB = 64;  // keys per tile

template <typename T, int DK>
__global__ __launch_bounds__(256) void relattn_kernel(jatts_relattn_desc d) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef typename Elem<T>::vec8 Vec;
  constexpr int KP = DK * (int)sizeof(T) + 16;  // K tile pitch (bytes)
  constexpr int VP = KB * (int)sizeof(T) + 16;  // V^T tile pitch
  constexpr int NKS = DK / 32;                  // contraction steps for S
  constexpr int NDF = DK / 16;                  // output d fragments
  char* ks = smem;
  char* vs = smem + KB * KP;

  const int b = blockIdx.y, h = blockIdx.z;
  const int row0 = d.rg.cu_rows[b];
  const int Tn = d.rg.cu_rows[b + 1] - row0;
  const int i0 = blockIdx.x * QB;
  if (i0 >= Tn) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int qc = lane & 15, g = lane >> 4;
  const int qi = i0 + wave * 16 + qc;          // this lane's query (may be >= Tn: never stored)
  const int qi_c = qi < Tn ? qi : Tn - 1;      // clamped for loads
  const T* qg = (const T*)d.q + (int64_t)(row0 + qi_c) * d.ldq + h * DK;
  const T* kg = (const T*)d.k + (int64_t)row0 * d.ldk + h * DK;
  const T* vtg = (const T*)d.vt + (int64_t)(h * DK) * d.ldvt + row0;
  const T* gg = d.g ? (const T*)d.g : nullptr;
  const int H = d.n_heads;

  Vec qf[NKS];
#pragma unroll
  for (int s = 0; s < NKS; ++s) qf[s] = load8<T>(qg + 32 * s + 8 * g);

  f32x4 ot[NDF];
#pragma unroll
  for (int f = 0; f < NDF; ++f) ot[f] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  for (int j0 = 0; j0 < Tn; j0 += KB) {
    // ---- stage K tile (rows = keys) and V^T tile (rows = d) ----
    {
      constexpr int UPR = DK / 8;
      for (int u = threadIdx.x; u < KB * UPR; u += 256) {
        const int r = u / UPR, cu = u - r * UPR;
        Vec v;
        if (j0 + r < Tn) v = load8<T>(kg + (int64_t)(j0 + r) * d.ldk + cu * 8);
        else {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = from_f32<T>(0.f);
        }
        T* dst = reinterpret_cast<T*>(ks + (size_t)r * KP) + cu * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) dst[e] = v[e];
      }
      // V^T rows are contiguous in time in global memory; unaligned start -> scalar loads
      for (int u = threadIdx.x; u < DK * KB; u += 256) {
        const int r = u / KB, c = u - r * KB;
        T v = from_f32<T>(0.f);
        if (j0 + c < Tn) v = vtg[(int64_t)r * d.ldvt + j0 + c];
        reinterpret_cast<T*>(vs + (size_t)r * VP)[c] = v;
      }
    }
    __syncthreads();

    // ---- S^T fragments: st[f][r] = key (j0 + 16 f + 4 g + r)  x  query qc ----
    f32x4 st[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      st[f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < NKS; ++s) {
        const T* ap = reinterpret_cast<const T*>(ks + (size_t)(16 * f + qc) * KP) + 32 * s + 8 * g;
        Vec a = load8<T>(ap);
        mma16(a, qf[s], st[f]);
      }
    }
    // ---- bias terms, scale, mask, online softmax ----
    float mx = -INFINITY;
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = j0 + 16 * f + 4 * g + r;
        float s = st[f][r];
        if (j < Tn) {
          if (d.ku) s += d.ku[(int64_t)(row0 + j) * H + h];
          if (gg && qi < Tn) {
            if (d.rel_mode == 2) {  // new rel_shift: plain diagonal map, no wrap
              s += to_f32(gg[((int64_t)(row0 + qi) * H + h) * d.ldg + (d.rel_center - qi + j)]);
            } else {                // legacy rel_shift (view-reinterpretation wrap)
              if (j <= qi) s += to_f32(gg[((int64_t)(row0 + qi) * H + h) * d.ldg + (Tn - 1 - qi + j)]);
              else if (j > qi + 1) s += to_f32(gg[((int64_t)(row0 + qi + 1) * H + h) * d.ldg + (j - qi - 2)]);
            }
          }
          s *= d.scale;
        } else {
          s = -INFINITY;
        }
        st[f][r] = s;
        mx = fmaxf(mx, s);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __expf(m_run - m_new);  // first tile: exp(-inf) = 0
    m_run = m_new;
    float psum = 0.f;
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __expf(st[f][r] - m_new);
        st[f][r] = p;
        psum += p;
      }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int f = 0; f < NDF; ++f)
#pragma unroll
      for (int r = 0; r < 4; ++r) ot[f][r] *= alpha;

    // ---- O^T += V^T P^T over two 32-key blocks.  Contraction slots of k-group g in block kb:
    //      keys {32kb + 4g + r} (from st[2kb]) then {32kb + 16 + 4g + r} (from st[2kb+1]) ----
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      Vec pb;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pb[r] = from_f32<T>(st[2 * kb][r]);
        pb[4 + r] = from_f32<T>(st[2 * kb + 1][r]);
      }
#pragma unroll
      for (int f = 0; f < NDF; ++f) {
        const T* vr = reinterpret_cast<const T*>(vs + (size_t)(16 * f + qc) * VP) + 32 * kb + 4 * g;
        Vec a;
#pragma unroll
        for (int r = 0; r < 4; ++r) { a[r] = vr[r]; a[4 + r] = vr[16 + r]; }
        mma16(a, pb, ot[f]);
      }
    }
    __syncthreads();
  }

  l_run += __shfl_xor(l_run, 16);
  l_run += __shfl_xor(l_run, 32);
  if (qi < Tn) {
    const float inv = 1.f / l_run;
    T* og = (T*)d.out + (int64_t)(row0 + qi) * d.ldo + h * DK;
#pragma unroll
    for (int f = 0; f < NDF; ++f)
#pragma unroll
      for (int r = 0; r < 4; ++r) og[16 * f + 4 * g + r] = from_f32<T>(ot[f][r] * inv);
  }
}

template <typename T, int DK>
int launch_attn(const jatts_relattn_desc& d, hipStream_t s) {
  const size_t lds = (size_t)KB * (DK * sizeof(T) + 16) + (size_t)DK * (KB * sizeof(T) + 16);
  dim3 grid((unsigned)((d.rg.max_len + QB - 1) / QB), (unsigned)d.rg.n_seq, (unsigned)d.n_heads);
  auto kern = relattn_kernel<T, DK>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return jatts_set_error(e, __FILE__, __LINE__);
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, d);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

template <typename T>
int dispatch_dk(const jatts_relattn_desc& d, hipStream_t s) {
  switch (d.d_k) {
    case 32: return launch_attn<T, 32>(d, s);
    case 64: return launch_attn<T, 64>(d, s);
    case 96: return launch_attn<T, 96>(d, s);
    case 128: return launch_attn<T, 128>(d, s);
    case 192: return launch_attn<T, 192>(d, s);
    case 256: return launch_attn<T, 256>(d, s);
  }
  return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "relpos_attention: d_k must be one of 32,64,96,128,192,256");
}

template <typename T>
__global__ void rowdot_kernel(const T* x, int ldx, int64_t rows, int n_heads, int d_k, const float* vec,
                              float* out) {
  const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  for (int h = 0; h < n_heads; ++h) {
    float s = 0.f;
    for (int c = lane; c < d_k; c += 64) s += to_f32(x[row * ldx + h * d_k + c]) * vec[h * d_k + c];
    s = wave_sum(s);
    if (lane == 0) out[row * n_heads + h] = s;
  }
}

}  // namespace

extern "C" int jatts_relpos_attention(const jatts_relattn_desc* d, void* stream) {
  if (!d || !d->q || !d->k || !d->vt || !d->out || !d->rg.cu_rows)
    return jatts_set_error_msg(JATTS_ERR_ARG, "relpos_attention: null pointer");
  if (d->n_heads < 1 || d->ldq % 8 || d->ldk % 8) return jatts_set_error_msg(JATTS_ERR_ARG, "relpos_attention: bad strides");
  if (d->rg.max_len <= 0) return JATTS_OK;
  hipStream_t s = (hipStream_t)stream;
  if (d->dtype == JATTS_F16) return dispatch_dk<f16>(*d, s);
  if (d->dtype == JATTS_F32) return dispatch_dk<float>(*d, s);
  return jatts_set_error_msg(JATTS_ERR_ARG, "relpos_attention: unknown dtype");
}

extern "C" int jatts_rowdot(int32_t dtype, const void* x, int32_t ldx, int64_t rows, int32_t n_heads,
                            int32_t d_k, const float* vec, float* out, void* stream) {
  if (!x || !vec || !out) return jatts_set_error_msg(JATTS_ERR_ARG, "rowdot: null pointer");
  if (rows <= 0) return JATTS_OK;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)((rows + 3) / 4));
  if (dtype == JATTS_F16)
    hipLaunchKernelGGL(rowdot_kernel<f16>, grid, dim3(256), 0, s, (const f16*)x, ldx, rows, n_heads, d_k, vec, out);
  else
    hipLaunchKernelGGL(rowdot_kernel<float>, grid, dim3(256), 0, s, (const float*)x, ldx, rows, n_heads, d_k, vec, out);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}
