// Row-wise kernels of the speaker-embedding front end: log-mel filterbank features and the pooling /
// squeeze-excitation pieces of ECAPA-TDNN.  The contractions (DFT, mel projection, every TDNN conv) run on jatts_conv1d.
// The reference delegates all of this to SpeechBrain (jatts/modules/feature_extract/spkemb_speechbrain.py:14-28,
// third party, not vendored): arithmetic restated from the public recipe, see oracle/ecapa_oracle.py.
#include "common.h"

namespace {

// frames[row(b, t)][n] = window[n] * x_b[t * hop + n - n_fft / 2], zero outside the signal (torch.stft, center=True,
// pad_mode="constant": SpeechBrain's STFT default [recalled]; torch.stft's own default would be "reflect")
__global__ __launch_bounds__(256) void frame_signal_kernel(jatts_ragged rg, const int32_t* cu_samples, const float* x, const float* window,
                                                           int n_fft, int hop, float* out, int ldo) {
  const int b = blockIdx.y;
  const int row0 = rg.cu_rows[b], T = rg.cu_rows[b + 1] - row0;
  const int s0 = cu_samples[b], n = cu_samples[b + 1] - s0;
  for (int t = blockIdx.x; t < T; t += gridDim.x) {
    float* o = out + (int64_t)(row0 + t) * ldo;
    for (int i = threadIdx.x; i < ldo; i += 256) {
      float v = 0.f;
      if (i < n_fft) {
        const int p = t * hop + i - n_fft / 2;
        if (p >= 0 && p < n) v = window[i] * x[s0 + p];
      }
      o[i] = v;
    }
  }
}

// out[row][k] = re[row][k]^2 + im[row][k]^2 for k < n_bins (re at column k, im at column n_bins + k), 0 for k >= n_bins
__global__ __launch_bounds__(256) void power_spectrum_kernel(const float* x, int ldx, int n_bins, int64_t rows, float* out, int ldo) {
  const int64_t total = rows * ldo;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / ldo;
    const int k = (int)(i - r * ldo);
    float v = 0.f;
    if (k < n_bins) {
      const float re = x[r * ldx + k], im = x[r * ldx + n_bins + k];
      v = re * re + im * im;
    }
    out[i] = v;
  }
}

// Per utterance: dB = 10 log10(max(p, amin)); dB = max(dB, max(dB) - top_db); out = dB - mean over time (per mel channel).
__global__ __launch_bounds__(256) void fbank_post_kernel(jatts_ragged rg, const float* p, int ldp, int n_mels, float amin, float top_db,
                                                         float* out, int ldo) {
  __shared__ float red[256];
  __shared__ float mean[256];
  const int b = blockIdx.x;
  const int row0 = rg.cu_rows[b], T = rg.cu_rows[b + 1] - row0;
  if (T <= 0) return;
  float mx = -INFINITY;
  for (int i = threadIdx.x; i < T * n_mels; i += 256) {
    const int t = i / n_mels, c = i - t * n_mels;
    mx = fmaxf(mx, 10.f * log10f(fmaxf(p[(int64_t)(row0 + t) * ldp + c], amin)));
  }
  red[threadIdx.x] = mx;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
    __syncthreads();
  }
  const float floor_db = red[0] - top_db;
  for (int c = threadIdx.x; c < n_mels; c += 256) {   // n_mels <= 256
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += fmaxf(10.f * log10f(fmaxf(p[(int64_t)(row0 + t) * ldp + c], amin)), floor_db);
    mean[c] = s / (float)T;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < T * ldo; i += 256) {
    const int t = i / ldo, c = i - t * ldo;
    out[(int64_t)(row0 + t) * ldo + c] =
        c < n_mels ? fmaxf(10.f * log10f(fmaxf(p[(int64_t)(row0 + t) * ldp + c], amin)), floor_db) - mean[c] : 0.f;
  }
}

// Per (utterance, channel): weights w_t = softmax_t(logits[t][c]) (or 1/T when logits == NULL),
// mean = sum_t w_t x_t, std = sqrt(max(sum_t w_t (x_t - mean)^2, eps)).  One wave per 64 channels, lanes = channels.
__global__ __launch_bounds__(64) void seq_mean_std_kernel(jatts_ragged rg, const float* x, int ldx, int dim, const float* logits, int ldl,
                                                          float* mean, float* stdv, int ldm, float eps) {
  const int b = blockIdx.y, c = blockIdx.x * 64 + threadIdx.x;
  const int row0 = rg.cu_rows[b], T = rg.cu_rows[b + 1] - row0;
  if (c >= dim || T <= 0) return;
  float mx = -INFINITY, z = (float)T;
  if (logits) {
    for (int t = 0; t < T; ++t) mx = fmaxf(mx, logits[(int64_t)(row0 + t) * ldl + c]);
    z = 0.f;
    for (int t = 0; t < T; ++t) z += expf(logits[(int64_t)(row0 + t) * ldl + c] - mx);
  }
  const float inv = 1.f / z;
  float m = 0.f;
  for (int t = 0; t < T; ++t) {
    const float w = logits ? expf(logits[(int64_t)(row0 + t) * ldl + c] - mx) * inv : inv;
    m += w * x[(int64_t)(row0 + t) * ldx + c];
  }
  float v = 0.f;
  for (int t = 0; t < T; ++t) {
    const float w = logits ? expf(logits[(int64_t)(row0 + t) * ldl + c] - mx) * inv : inv;
    const float dlt = x[(int64_t)(row0 + t) * ldx + c] - m;
    v += w * dlt * dlt;
  }
  mean[(int64_t)b * ldm + c] = m;
  if (stdv) stdv[(int64_t)b * ldm + c] = sqrtf(fmaxf(v, eps));
}

// y = post( scale[c] * pre(x + vec[b][c]) + shift[c] ); pre: 0 none / 1 relu, post: 0 none / 2 tanh (JATTS_ACT_* codes)
template <typename TO>
__global__ __launch_bounds__(256) void seq_affine_act_kernel(jatts_ragged rg, const float* x, int ldx, int dim, const float* vec, int pre,
                                                             const float* scale, const float* shift, int post, TO* y, int ldy) {
  const int b = blockIdx.y;
  const int row0 = rg.cu_rows[b], T = rg.cu_rows[b + 1] - row0;
  const int64_t total = (int64_t)T * ldy;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t t = i / ldy;
    const int c = (int)(i - t * ldy);
    float v = 0.f;
    if (c < dim) {
      v = x[(row0 + t) * ldx + c] + (vec ? vec[(int64_t)b * dim + c] : 0.f);
      if (pre == JATTS_ACT_RELU) v = fmaxf(v, 0.f);
      v = v * (scale ? scale[c] : 1.f) + (shift ? shift[c] : 0.f);
      if (post == JATTS_ACT_TANH) v = tanhf(v);
    }
    y[(row0 + t) * ldy + c] = from_f32<TO>(v);
  }
}

// y = x * sigmoid(s[b][c]) + resid   (squeeze-excitation gate + the block's residual connection)
__global__ __launch_bounds__(256) void se_scale_add_kernel(jatts_ragged rg, const float* x, int dim, const float* s, const float* resid, float* y, int ldy) {
  const int b = blockIdx.y;
  const int row0 = rg.cu_rows[b], T = rg.cu_rows[b + 1] - row0;
  const int64_t total = (int64_t)T * dim;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t t = i / dim;
    const int c = (int)(i - t * dim);
    const int64_t o = (int64_t)row0 * dim + i;
    const float g = 1.f / (1.f + expf(-s[(int64_t)b * dim + c]));
    y[(row0 + t) * ldy + c] = x[o] * g + (resid ? resid[o] : 0.f);
  }
}

}  // namespace

#define S_ ((hipStream_t)stream)

extern "C" int jatts_frame_signal(const jatts_ragged* rg_frames, const int32_t* cu_samples, const float* x, const float* window,
                                  int32_t n_fft, int32_t hop, float* out, int32_t ldo, void* stream) {
  if (!rg_frames || !cu_samples || !x || !window || !out) return jatts_set_error_msg(JATTS_ERR_ARG, "frame_signal: null pointer");
  if (n_fft < 2 || hop < 1 || ldo < n_fft) return jatts_set_error_msg(JATTS_ERR_ARG, "frame_signal: bad geometry");
  if (rg_frames->n_seq <= 0 || rg_frames->max_len <= 0) return JATTS_OK;
  const unsigned gx = (unsigned)(rg_frames->max_len < 1024 ? rg_frames->max_len : 1024);
  hipLaunchKernelGGL(frame_signal_kernel, dim3(gx, (unsigned)rg_frames->n_seq), dim3(256), 0, S_, *rg_frames, cu_samples, x, window, n_fft, hop, out, ldo);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

extern "C" int jatts_power_spectrum(const float* x, int32_t ldx, int32_t n_bins, int64_t rows, float* out, int32_t ldo, void* stream) {
  if (!x || !out) return jatts_set_error_msg(JATTS_ERR_ARG, "power_spectrum: null pointer");
  if (ldx < 2 * n_bins || ldo < n_bins) return jatts_set_error_msg(JATTS_ERR_ARG, "power_spectrum: bad strides");
  if (rows <= 0) return JATTS_OK;
  const int64_t blocks = (rows * ldo + 255) / 256;
  hipLaunchKernelGGL(power_spectrum_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, S_, x, ldx, n_bins, rows, out, ldo);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

extern "C" int jatts_fbank_post(const jatts_ragged* rg, const float* p, int32_t ldp, int32_t n_mels, float amin, float top_db,
                                float* out, int32_t ldo, void* stream) {
  if (!rg || !p || !out) return jatts_set_error_msg(JATTS_ERR_ARG, "fbank_post: null pointer");
  if (n_mels < 1 || n_mels > 256 || ldo < n_mels || ldp < n_mels) return jatts_set_error_msg(JATTS_ERR_ARG, "fbank_post: bad geometry");
  if (rg->n_seq <= 0) return JATTS_OK;
  hipLaunchKernelGGL(fbank_post_kernel, dim3((unsigned)rg->n_seq), dim3(256), 0, S_, *rg, p, ldp, n_mels, amin, top_db, out, ldo);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

extern "C" int jatts_seq_mean_std(const jatts_ragged* rg, const float* x, int32_t ldx, int32_t dim, const float* logits, int32_t ldl,
                                  float* mean, float* stdv, int32_t ldm, float eps, void* stream) {
  if (!rg || !x || !mean) return jatts_set_error_msg(JATTS_ERR_ARG, "seq_mean_std: null pointer");
  if (rg->n_seq <= 0 || dim <= 0) return JATTS_OK;
  hipLaunchKernelGGL(seq_mean_std_kernel, dim3((unsigned)((dim + 63) / 64), (unsigned)rg->n_seq), dim3(64), 0, S_, *rg, x, ldx, dim, logits, ldl,
                     mean, stdv, ldm, eps);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

extern "C" int jatts_seq_affine_act(const jatts_ragged* rg, const float* x, int32_t ldx, int32_t dim, const float* seq_vec, int32_t pre_act,
                                    const float* scale, const float* shift, int32_t post_act, void* y, int32_t out_dtype, int32_t ldy,
                                    void* stream) {
  if (!rg || !x || !y) return jatts_set_error_msg(JATTS_ERR_ARG, "seq_affine_act: null pointer");
  if (ldy < dim || ldx < dim) return jatts_set_error_msg(JATTS_ERR_ARG, "seq_affine_act: bad strides");
  if (rg->n_seq <= 0 || rg->max_len <= 0) return JATTS_OK;
  const int64_t per = ((int64_t)rg->max_len * ldy + 255) / 256;
  dim3 grid((unsigned)(per < 1024 ? per : 1024), (unsigned)rg->n_seq);
  if (out_dtype == JATTS_F32)
    hipLaunchKernelGGL(seq_affine_act_kernel<float>, grid, dim3(256), 0, S_, *rg, x, ldx, dim, seq_vec, pre_act, scale, shift, post_act, (float*)y, ldy);
  else if (out_dtype == JATTS_F16)
    hipLaunchKernelGGL(seq_affine_act_kernel<f16>, grid, dim3(256), 0, S_, *rg, x, ldx, dim, seq_vec, pre_act, scale, shift, post_act, (f16*)y, ldy);
  else return jatts_set_error_msg(JATTS_ERR_ARG, "seq_affine_act: unknown dtype");
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}

extern "C" int jatts_se_scale_add(const jatts_ragged* rg, const float* x, int32_t dim, const float* s, const float* resid, float* y, int32_t ldy,
                                  void* stream) {
  if (!rg || !x || !s || !y) return jatts_set_error_msg(JATTS_ERR_ARG, "se_scale_add: null pointer");
  if (rg->n_seq <= 0 || rg->max_len <= 0) return JATTS_OK;
  const int64_t per = ((int64_t)rg->max_len * dim + 255) / 256;
  hipLaunchKernelGGL(se_scale_add_kernel, dim3((unsigned)(per < 1024 ? per : 1024), (unsigned)rg->n_seq), dim3(256), 0, S_, *rg, x, dim, s, resid, y, ldy);
  JATTS_CHECK_LAUNCH();
  return JATTS_OK;
}
