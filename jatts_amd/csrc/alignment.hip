// Alignment learning pieces: the reference's only JIT-native code is the monotonic alignment search,
// a per-utterance numba loop on the host with device<->host copies (jatts/modules/alignments.py:63-93,281-310).  Here the
// whole batch runs on the GPU: one workgroup per utterance sweeps the mel frames; the T_inp cells of a frame are
// independent given the previous frame, so a column is one parallel step + one barrier; the argmax decisions are kept as
// bit masks in LDS (T_mel x T_inp bits) and a single thread walks them backwards.  Q is float64 as in the reference.
#include "common.h"

namespace {

constexpr int MAS_SLOTS = 4;    // tokens per thread: T_inp <= 1024
constexpr int MAS_PF = 8;       // log-prob columns prefetched ahead (registers)

__device__ __forceinline__ double block_sum_f64(double v, double* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void mas_kernel(jatts_ragged rg_f, const int32_t* cu_text, const float* logp, int ld,
                                                  int64_t* path, int64_t* dur, double* score) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = blockIdx.x;
  const int row0 = rg_f.cu_rows[b];
  const int T_mel = rg_f.cu_rows[b + 1] - row0;
  const int tok0 = cu_text[b];
  const int T_inp = cu_text[b + 1] - tok0;
  if (T_mel <= 0 || T_inp <= 0) {
    if (threadIdx.x == 0 && score) score[b] = 0.0;
    for (int i = threadIdx.x; i < T_inp; i += 256) dur[tok0 + i] = 0;
    return;
  }
  const int W = (T_inp + 63) >> 6;                       // 64-bit decision words per frame
  unsigned long long* dec = reinterpret_cast<unsigned long long*>(smem);              // [T_mel][W]
  double* qbuf = reinterpret_cast<double*>(dec + (size_t)T_mel * W);                  // [2][T_inp + 1]
  int* cnt = reinterpret_cast<int*>(qbuf + 2 * (T_inp + 1));                          // [T_inp]
  int* pth = cnt + T_inp;                                                              // [T_mel]
  __shared__ double red[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* lp = logp + (int64_t)row0 * ld;
  const double NEG = -INFINITY;

  double q[MAS_SLOTS];
#pragma unroll
  for (int m = 0; m < MAS_SLOTS; ++m) {
    const int i = threadIdx.x + 256 * m;
    q[m] = (i == 0) ? (double)lp[0] : NEG;               // column 0: Q[0,0] = log_prob[0,0]; Q[i>0,0] = -inf
    if (i < T_inp) { qbuf[i] = q[m]; cnt[i] = 0; }
  }
  // register ring of the next MAS_PF columns of log-probabilities
  float pf[MAS_PF][MAS_SLOTS];
#pragma unroll
  for (int p = 0; p < MAS_PF; ++p)
#pragma unroll
    for (int m = 0; m < MAS_SLOTS; ++m) {
      const int i = threadIdx.x + 256 * m, j = 1 + p;
      pf[p][m] = (i < T_inp && j < T_mel) ? lp[(int64_t)j * ld + i] : 0.f;
    }
  __syncthreads();
  for (int j0 = 1; j0 < T_mel; j0 += MAS_PF) {
#pragma unroll
    for (int p = 0; p < MAS_PF; ++p) {
      const int j = j0 + p;
      if (j >= T_mel) break;                              // uniform
      const double* qp = qbuf + ((j - 1) & 1) * (T_inp + 1);
      double* qn = qbuf + (j & 1) * (T_inp + 1);
#pragma unroll
      for (int m = 0; m < MAS_SLOTS; ++m) {
        const int i = threadIdx.x + 256 * m;
        const double up = (i >= 1 && i < T_inp) ? qp[i - 1] : NEG;    // Q[i-1, j-1]
        const bool take_up = (i >= 1 && i < T_inp) && (up >= q[m]);   // alignments.py:85: Q[i_a, j] >= Q[i_b, j]
        const unsigned long long mask = __ballot(take_up);
        if (lane == 0 && 4 * m + wave < W) dec[(size_t)j * W + 4 * m + wave] = mask;
        double nq = NEG;
        if (i == 0) nq = q[m] + (double)pf[p][m];                      // running sum of row 0 (:70-71)
        else if (i < T_inp && i <= j) nq = fmax(up, q[m]) + (double)pf[p][m];   // (:74-76)
        q[m] = nq;
        if (i < T_inp) qn[i] = nq;
        // refill this ring slot with column j + MAS_PF
        const int jn = j + MAS_PF;
        pf[p][m] = (i < T_inp && jn < T_mel) ? lp[(int64_t)jn * ld + i] : 0.f;
      }
      __syncthreads();
    }
  }
  // backtrack (alignments.py:78-92): one thread, decisions from LDS
  if (threadIdx.x == 0) {
    int a = T_inp - 1;
    pth[T_mel - 1] = a;
    for (int j = T_mel - 2; j >= 0; --j) {
      if (a > 0) a -= (int)((dec[(size_t)(j + 1) * W + (a >> 6)] >> (a & 63)) & 1ull);
      pth[j] = a;
    }
  }
  __syncthreads();
  double s = 0.0;
  for (int j = threadIdx.x; j < T_mel; j += 256) {
    const int a = pth[j];
    path[row0 + j] = a;
    atomicAdd(&cnt[a], 1);
    s += (double)lp[(int64_t)j * ld + a];
  }
  s = block_sum_f64(s, red);
  if (threadIdx.x == 0 && score) score[b] = s;
  __syncthreads();
  for (int i = threadIdx.x; i < T_inp; i += 256) dur[tok0 + i] = cnt[i];   // np.bincount(viterbi) (:303-304)
}

// score[f][i] = -|| feats[f] - text[i] ||_2 ; log_softmax over the utterance's tokens (alignments.py:50-59).
constexpr int AL_FB = 16;    // frames per workgroup
constexpr int AL_CC = 64;    // channels per staged chunk
constexpr int AL_PAIRS = 32; // (frame, token) pairs per thread: T_text <= 512

__global__ __launch_bounds__(256) void align_logp_kernel(jatts_ragged rg_f, const int32_t* cu_text, const float* feats, const float* text,
                                                         int adim, float* logp, int ld) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = blockIdx.y;
  const int row0 = rg_f.cu_rows[b];
  const int T_f = rg_f.cu_rows[b + 1] - row0;
  const int f0 = blockIdx.x * AL_FB;
  if (f0 >= T_f) return;
  const int tok0 = cu_text[b];
  const int T_t = cu_text[b + 1] - tok0;
  const int nf = min(AL_FB, T_f - f0);
  float* ts = reinterpret_cast<float*>(smem);            // [T_t][AL_CC + 1]
  float* fs = ts + (size_t)T_t * (AL_CC + 1);            // [AL_FB][AL_CC + 1]
  float* sc = fs + AL_FB * (AL_CC + 1);                  // [AL_FB][T_t]
  const int pairs = nf * T_t;
  float acc[AL_PAIRS];
#pragma unroll
  for (int p = 0; p < AL_PAIRS; ++p) acc[p] = 0.f;
  for (int c0 = 0; c0 < adim; c0 += AL_CC) {
    const int cc = min(AL_CC, adim - c0);
    __syncthreads();
    for (int u = threadIdx.x; u < T_t * cc; u += 256) {
      const int i = u / cc, c = u - i * cc;
      ts[i * (AL_CC + 1) + c] = text[(int64_t)(tok0 + i) * adim + c0 + c];
    }
    for (int u = threadIdx.x; u < nf * cc; u += 256) {
      const int f = u / cc, c = u - f * cc;
      fs[f * (AL_CC + 1) + c] = feats[(int64_t)(row0 + f0 + f) * adim + c0 + c];
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < AL_PAIRS; ++p) {
      const int u = threadIdx.x + 256 * p;
      if (u < pairs) {
        const int f = u / T_t, i = u - f * T_t;
        const float* tr = ts + i * (AL_CC + 1);
        const float* fr = fs + f * (AL_CC + 1);
        float a = acc[p];
        for (int c = 0; c < cc; ++c) { const float dlt = fr[c] - tr[c]; a += dlt * dlt; }
        acc[p] = a;
      }
    }
  }
#pragma unroll
  for (int p = 0; p < AL_PAIRS; ++p) {
    const int u = threadIdx.x + 256 * p;
    if (u < pairs) sc[u] = -sqrtf(acc[p]);               // u = f * T_t + i
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int f = wave; f < nf; f += 4) {
    float mx = -INFINITY;
    for (int i = lane; i < T_t; i += 64) mx = fmaxf(mx, sc[f * T_t + i]);
    mx = wave_max(mx);
    float se = 0.f;
    for (int i = lane; i < T_t; i += 64) se += expf(sc[f * T_t + i] - mx);
    const float lse = mx + logf(wave_sum(se));
    float* out = logp + (int64_t)(row0 + f0 + f) * ld;
    for (int i = lane; i < ld; i += 64) out[i] = i < T_t ? sc[f * T_t + i] - lse : -INFINITY;   // masked_fill(-inf) columns
  }
}

}  // namespace

extern "C" int jatts_mas_viterbi(const jatts_ragged* rg_feats, const int32_t* cu_text, const float* log_p, int32_t ld,
                                 int32_t max_text_len, int64_t* path, int64_t* dur, double* score, void* stream) {
  if (!rg_feats || !rg_feats->cu_rows || !cu_text || !log_p || !path || !dur) return jatts_set_error_msg(JATTS_ERR_ARG, "mas_viterbi: null pointer");
  if (rg_feats->n_seq < 1 || rg_feats->max_len <= 0) return JATTS_OK;
  if (max_text_len < 1 || max_text_len > 256 * MAS_SLOTS || ld < max_text_len)
    return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "mas_viterbi: text length must be 1..1024 and <= ld");
  const size_t W = (max_text_len + 63) / 64;
  const size_t lds = (size_t)rg_feats->max_len * W * 8 + 2 * (size_t)(max_text_len + 1) * 8 + (size_t)max_text_len * 4 + (size_t)rg_feats->max_len * 4 + 16;
  if (lds > 160 * 1024) return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "mas_viterbi: T_feats x T_text decision bits exceed 160 KiB of LDS");
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)mas_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return jatts_set_error(e, __FILE__, __LINE__);
  }
  hipLaunchKernelGGL(mas_kernel, dim3((unsigned)rg_feats->n_seq), dim3(256), lds, (hipStream_t)stream, *rg_feats, cu_text, log_p, ld, path, dur, score);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

extern "C" int jatts_alignment_logp(const jatts_ragged* rg_feats, const int32_t* cu_text, int32_t max_text_len, const float* feats,
                                    const float* text, int32_t adim, float* log_p, int32_t ld, void* stream) {
  if (!rg_feats || !rg_feats->cu_rows || !cu_text || !feats || !text || !log_p) return jatts_set_error_msg(JATTS_ERR_ARG, "alignment_logp: null pointer");
  if (rg_feats->n_seq < 1 || rg_feats->max_len <= 0) return JATTS_OK;
  if (max_text_len < 1 || AL_FB * max_text_len > 256 * AL_PAIRS || ld < max_text_len)
    return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "alignment_logp: text length must be 1..512 and <= ld");
  const size_t lds = ((size_t)max_text_len * (AL_CC + 1) + (size_t)AL_FB * (AL_CC + 1) + (size_t)AL_FB * max_text_len) * sizeof(float);
  if (lds > 160 * 1024) return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "alignment_logp: text too long for LDS");
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)align_logp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return jatts_set_error(e, __FILE__, __LINE__);
  }
  dim3 grid((unsigned)((rg_feats->max_len + AL_FB - 1) / AL_FB), (unsigned)rg_feats->n_seq);
  hipLaunchKernelGGL(align_logp_kernel, grid, dim3(256), lds, (hipStream_t)stream, *rg_feats, cu_text, feats, text, adim, log_p, ld);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}
