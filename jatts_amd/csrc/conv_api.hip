// C-ABI entry points of the MFMA convolution family: argument checks and tile selection.  The kernels
// are instantiated in per-dtype translation units (conv1d_*.hip, resunit_*.hip) that build in parallel.
#include <stdlib.h>

#include "common.h"

unsigned long long* jatts_g_trace = nullptr;  // profiling hook, see jatts_debug_trace
unsigned jatts_g_trace_cap = 0;

int jatts_conv1d_f16(const jatts_conv_desc& d, hipStream_t s);
int jatts_conv1d_f32(const jatts_conv_desc& d, hipStream_t s);
int jatts_conv1d_split(const jatts_conv_desc& d, hipStream_t s);           // JATTS_F32S
int jatts_conv1d_emul(const jatts_conv_desc& d, hipStream_t s);            // JATTS_F32E / JATTS_F32E6
int jatts_resunit_f16_narrow(const jatts_resunit_desc& d, hipStream_t s);  // C = 32, 64
int jatts_resunit_f16_wide(const jatts_resunit_desc& d, hipStream_t s);    // C = 128, 256, 512
int jatts_resunit_f32(const jatts_resunit_desc& d, hipStream_t s);
int jatts_resunit_split(const jatts_resunit_desc& d, hipStream_t s);       // JATTS_F32S
int jatts_resunit_emul(const jatts_resunit_desc& d, hipStream_t s);        // JATTS_F32E
int jatts_resblock_f16(const jatts_resblock_desc& d, hipStream_t s);
int jatts_resblock_f32(const jatts_resblock_desc& d, hipStream_t s);
int jatts_resblock_split(const jatts_resblock_desc& d, hipStream_t s);     // JATTS_F32S
int jatts_resblock_emul(const jatts_resblock_desc& d, hipStream_t s);      // JATTS_F32E / JATTS_F32E6

extern "C" int jatts_debug_trace(void* buf, int64_t n_workgroups) {
  jatts_g_trace = (unsigned long long*)buf;
  jatts_g_trace_cap = buf ? (unsigned)n_workgroups : 0u;
  return JATTS_OK;
}

extern "C" int64_t jatts_conv_weight_index(int32_t n, int32_t tap, int32_t c, int32_t n_pad, int32_t c_in) {
  const int64_t KC16 = c_in / 16, NFR = n_pad / 32;
  const int64_t kc = c / 16, nf = n / 32;
  const int64_t lane = 32 * ((c % 16) / 8) + (n % 32);
  return ((((int64_t)tap * KC16 + kc) * NFR + nf) * 64 + lane) * 8 + (c % 8);
}

extern "C" int64_t jatts_unit_weight_index_k32(int32_t n, int32_t tap, int32_t c, int32_t channels) {
  const int64_t KC32 = channels / 32, NFR16 = channels / 16;
  const int64_t kc = c / 32, nf = n / 16;
  const int64_t lane = 16 * ((c % 32) / 8) + (n % 16);
  return ((((int64_t)tap * KC32 + kc) * NFR16 + nf) * 64 + lane) * 8 + (c % 8);
}

extern "C" int jatts_conv1d(const jatts_conv_desc* d, void* stream) {
  if (!d || !d->x[0] || !d->w || !d->y || !d->rg.cu_rows) return jatts_set_error_msg(JATTS_ERR_ARG, "conv1d: null pointer");
  if (d->c_in <= 0 || d->c_in % 64) return jatts_set_error_msg(JATTS_ERR_ARG, "conv1d: c_in must be a positive multiple of 64");
  if (d->ldx % 8) return jatts_set_error_msg(JATTS_ERR_ARG, "conv1d: ldx must be a multiple of 8");
  if (d->n_in < 1 || d->n_in > 3 || d->k_w < 1 || d->dil < 1 || d->n_out < 1 || d->rg.n_seq < 1 || d->rg.len_mul < 1)
    return jatts_set_error_msg(JATTS_ERR_ARG, "conv1d: bad geometry");
  if (d->act == JATTS_ACT_SNAKEBETA && (!d->act_a || !d->act_b || (d->n_out & 3) || ((uintptr_t)d->act_a & 15) || ((uintptr_t)d->act_b & 15)))
    return jatts_set_error_msg(JATTS_ERR_ARG, "conv1d: JATTS_ACT_SNAKEBETA needs 16-byte aligned act_a / act_b and n_out % 4 == 0");
  if (d->n_split != 0 && (d->n_split < 0 || d->n_split % 256 || d->n_split >= d->n_out || !d->y2 || d->y_transposed || d->resid || d->ldy2 <= 0))
    return jatts_set_error_msg(JATTS_ERR_ARG, "conv1d: n_split must be a multiple of 256 below n_out, with y2 / ldy2 set, y row-major and no residual");
  if (d->w_layout != 0 && d->dtype != JATTS_F32E && d->dtype != JATTS_F32E6) return jatts_set_error_msg(JATTS_ERR_ARG, "conv1d: w_layout = 1 goes with JATTS_F32E / JATTS_F32E6 only");
  if (d->rg.max_len <= 0) return JATTS_OK;
  hipStream_t s = (hipStream_t)stream;
  if (d->dtype == JATTS_F16) return jatts_conv1d_f16(*d, s);
  if (d->dtype == JATTS_F32) return jatts_conv1d_f32(*d, s);
  if (d->dtype == JATTS_F32S) {
    if (!d->w_inv) return jatts_set_error_msg(JATTS_ERR_ARG, "conv1d: JATTS_F32S needs w_inv");
    if (!d->y_is_f32) return jatts_set_error_msg(JATTS_ERR_ARG, "conv1d: JATTS_F32S writes f32 (y_is_f32 = 1)");
    return jatts_conv1d_split(*d, s);
  }
  if (d->dtype == JATTS_F32E || d->dtype == JATTS_F32E6) {
    if (!d->y_is_f32) return jatts_set_error_msg(JATTS_ERR_ARG, "conv1d: JATTS_F32E / JATTS_F32E6 write f32 (y_is_f32 = 1)");
    if (d->w_layout != 0 && d->w_layout != 1) return jatts_set_error_msg(JATTS_ERR_ARG, "conv1d: w_layout must be 0 or 1");
    return jatts_conv1d_emul(*d, s);
  }
  return jatts_set_error_msg(JATTS_ERR_ARG, "conv1d: unknown dtype");
}

extern "C" int jatts_hifigan_resunit(const jatts_resunit_desc* d, void* stream) {
  if (!d || !d->x || !d->y || !d->w1 || !d->w2 || !d->b1 || !d->b2 || !d->rg.cu_rows)
    return jatts_set_error_msg(JATTS_ERR_ARG, "resunit: null pointer");
  if (d->x == d->y) return jatts_set_error_msg(JATTS_ERR_ARG, "resunit: y must not alias x");
  if (d->k_w < 1 || !(d->k_w & 1) || d->dil < 1) return jatts_set_error_msg(JATTS_ERR_ARG, "resunit: odd k_w and dil>=1 required");
  if (!(d->slope >= 0.f && d->slope <= 1.f)) return jatts_set_error_msg(JATTS_ERR_ARG, "resunit: LeakyReLU slope must be in [0, 1]");
  if (d->rg.max_len <= 0) return JATTS_OK;
  hipStream_t s = (hipStream_t)stream;
  if (d->dtype == JATTS_F16) return d->channels <= 64 ? jatts_resunit_f16_narrow(*d, s) : jatts_resunit_f16_wide(*d, s);
  if (d->dtype == JATTS_F32) return jatts_resunit_f32(*d, s);
  if (d->dtype == JATTS_F32S) {
    if (!d->ws1 || !d->ws2) return jatts_set_error_msg(JATTS_ERR_ARG, "resunit: JATTS_F32S needs ws1 / ws2");
    return jatts_resunit_split(*d, s);
  }
  if (d->dtype == JATTS_F32E || d->dtype == JATTS_F32E6) {
    if (d->w_layout != 0 && d->w_layout != 1) return jatts_set_error_msg(JATTS_ERR_ARG, "resunit: w_layout must be 0 or 1");
    return jatts_resunit_emul(*d, s);
  }
  if (d->w_layout != 0) return jatts_set_error_msg(JATTS_ERR_ARG, "resunit: w_layout = 1 goes with JATTS_F32E / JATTS_F32E6 only");
  return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "resunit: unsupported channels/dtype (use jatts_conv1d)");
}

extern "C" int jatts_hifigan_resblock(const jatts_resblock_desc* d, void* stream) {
  if (!d || !d->x || !d->y || !d->rg.cu_rows) return jatts_set_error_msg(JATTS_ERR_ARG, "resblock: null pointer");
  if (d->n_units < 1 || d->n_units > 3) return jatts_set_error_msg(JATTS_ERR_ARG, "resblock: 1..3 units");
  for (int u = 0; u < d->n_units; ++u) {
    if (!d->w1[u] || !d->w2[u] || !d->b1[u] || !d->b2[u]) return jatts_set_error_msg(JATTS_ERR_ARG, "resblock: null weights");
    if (d->dil[u] < 1) return jatts_set_error_msg(JATTS_ERR_ARG, "resblock: dil >= 1 required");
  }
  if (d->x == d->y) return jatts_set_error_msg(JATTS_ERR_ARG, "resblock: y must not alias x");
  if (d->k_w < 1 || !(d->k_w & 1)) return jatts_set_error_msg(JATTS_ERR_ARG, "resblock: odd k_w required");
  if (!(d->slope >= 0.f && d->slope <= 1.f)) return jatts_set_error_msg(JATTS_ERR_ARG, "resblock: LeakyReLU slope must be in [0, 1]");
  if (d->dtype == JATTS_F32S) {
    for (int u = 0; u < d->n_units; ++u)
      if (!d->ws1[u] || !d->ws2[u]) return jatts_set_error_msg(JATTS_ERR_ARG, "resblock: JATTS_F32S needs ws1 / ws2");
    if (d->rg.max_len <= 0) return JATTS_OK;
    return jatts_resblock_split(*d, (hipStream_t)stream);
  }
  if (d->dtype == JATTS_F32E || d->dtype == JATTS_F32E6) {
    if (d->rg.max_len <= 0) return JATTS_OK;
    return jatts_resblock_emul(*d, (hipStream_t)stream);
  }
  if (d->dtype != JATTS_F16 && d->dtype != JATTS_F32) return jatts_set_error_msg(JATTS_ERR_ARG, "resblock: unknown dtype");
  if (d->rg.max_len <= 0) return JATTS_OK;
  return d->dtype == JATTS_F16 ? jatts_resblock_f16(*d, (hipStream_t)stream) : jatts_resblock_f32(*d, (hipStream_t)stream);
}
