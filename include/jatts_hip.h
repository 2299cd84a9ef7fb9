/* jatts_hip.h -- C ABI of libjatts_hip.so: the MI355X (gfx950) kernels behind the
 * jatts stage-4 hot path (text2mel `model.inference` + HiFi-GAN `Vocoder.decode`).
 *
 * The reference (unilight/jatts) is pure Python/PyTorch and has no FFI of its own
 * (SURVEY.md §8b); each entry point below names the reference code whose arithmetic
 * it replaces (paths under /root/reference/jatts/).  INTEGRATION.md shows the
 * ctypes binding a reference maintainer would add.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes only; every pointer is DEVICE memory owned
 *    by the caller unless marked HOST.  No allocation, no ownership transfer.
 *  - All launches are asynchronous on `stream` (a hipStream_t passed as void*).
 *  - Return 0 on success, negative on error; jatts_last_error() gives the text.
 *  - Activations are packed ragged, time-major: X[row][ld] where sequence b owns rows
 *    cu_rows[b]*len_mul .. cu_rows[b+1]*len_mul-1.  cu_rows has n_seq+1 int32 entries.
 *    `max_len` is the HOST-known max of (cu_rows[b+1]-cu_rows[b]) and only sizes grids.
 *  - dtype: JATTS_F16 = f16 operands on v_mfma_f32_32x32x16_f16 with f32 accumulate
 *    (fast mode); JATTS_F32 = f32 operands on v_mfma_f32_32x32x2_f32 (exact-f32 parity
 *    mode).  Biases, residual streams and reductions are always f32.
 */
#ifndef JATTS_HIP_H_
#define JATTS_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JATTS_ABI_VERSION 4   /* 4 (round 6): + jatts_mfma_probe / jatts_mfma_probe_flops, jatts_conv_desc + n_split / ldy2 / y2 / y2_seq_col0 / w_layout and jatts_resunit_desc + w_layout (appended), jatts_unit_weight_index_k32; 3 (round 5): jatts_ragged + total_rows AND host_lens -- the struct grew from 24 to 32 bytes,
                                * so every descriptor that embeds it (jatts_conv_desc, jatts_resunit_desc, jatts_resblock_desc, jatts_relattn_desc) shifted by 8 bytes; JATTS_F32E; 2 (round 4): jatts_conv_desc + w_inv / act_a / act_b, jatts_resunit_desc + ws1 / ws2, jatts_resblock_desc + ws1 / ws2;
                                * bumped whenever a descriptor's layout or an entry point's signature changes: a stale library is refused at load */

#define JATTS_F32 0
#define JATTS_F16 1
/* f32 activations in HBM, error-corrected split-precision MFMA operands (round 4): every operand value, scaled by a
 * power of two, travels as hi = f16(v), lo = f16(v - hi) and a product is hi.hi + hi.lo + lo.hi on
 * v_mfma_f32_32x32x16_f16 with f32 accumulate (3/16 of the matrix-pipe cycles of the exact-f32 chain, ~22 significand
 * bits per operand).  Accepted by jatts_hifigan_resunit and jatts_conv1d; weights packed by the host as [hi x8 | lo x8] per lane
 * (jatts_amd.hip.pack_conv_weight_split) with the inverse per-output-channel scales in ws1 / ws2 (unit) or w_inv (conv).
 * Activation scales are chosen per workgroup tile inside the kernels (powers of two: scaling and un-scaling are exact). */
#define JATTS_F32S 2
/* f32 activations in HBM, f32-EQUIVALENT emulated MFMA operands (round 5): every operand value v travels EXACTLY as three
 * bfloat16 terms b0 = bf16(v), b1 = bf16(v - b0), b2 = bf16(v - b0 - b1) (3 x 8 significand bits, f32's exponent range: no
 * scales, no block maxima; exact for |v| >= 2^-110) and a product w v keeps the largest of its nine partial products on
 * v_mfma_f32_32x32x16_bf16 with f32 accumulate:
 *   JATTS_F32E  : seven (all of weight >= 2^-16 plus w1 v2).  Dropped <= 2^-24 |w v|: a one-term contraction, product plus its
 *                 rounding into the f32 result, is within 2^-23 = 2 x an f32 FMA's bound for EVERY input.  7/16 of the pipe cycles
 *                 of the exact-f32 chain.
 *   JATTS_F32E6 : six (weight >= 2^-16).  Dropped <= 2^-23 |w v| (3 x 2^-24 with the rounding at one term).  6/16 of the cycles.
 * Accepted by jatts_hifigan_resunit and jatts_conv1d (x, resid, y f32); weights packed by the host as [b0 x8 | b1 x8 | b2 x8] per lane
 * (jatts_amd.hip.pack_conv_weight_bf16x3; the same operand serves both), ws1 / ws2 / w_inv unused. */
#define JATTS_F32E 3
#define JATTS_F32E6 4

#define JATTS_ACT_NONE 0
#define JATTS_ACT_RELU 1
#define JATTS_ACT_TANH 2
#define JATTS_ACT_SWISH 3
#define JATTS_ACT_MISH 4
#define JATTS_ACT_SNAKEBETA 5 /* v + act_b[n] * sin^2(act_a[n] * v): Matcha's feed-forward activation (jatts_snakebeta) in the conv's epilogue */

#define JATTS_PAD_ZERO 0
#define JATTS_PAD_REFLECT 1

#define JATTS_PRE_NONE 0
#define JATTS_PRE_LRELU 1

#define JATTS_OK 0
#define JATTS_ERR_ARG (-1)
#define JATTS_ERR_HIP (-2)
#define JATTS_ERR_UNSUPPORTED (-3)

int jatts_abi_version(void);
const char* jatts_last_error(void);
/* Device name / arch string of the current device (HOST buffer). */
int jatts_device_info(char* buf, int buflen);

/* ---------------------------------------------------------------------------------
 * Ragged geometry shared by the sequence kernels.
 * ------------------------------------------------------------------------------- */
typedef struct jatts_ragged {
  const int32_t* cu_rows; /* device, n_seq+1 */
  int32_t n_seq;
  int32_t max_len; /* host-known max base length */
  int32_t len_mul; /* rows per base row (HiFi-GAN stage rate); 1 elsewhere */
  int32_t total_rows; /* (ABI 3) HOST-known sum of the base lengths = cu_rows[n_seq] - cu_rows[0]; 0 with host_lens == NULL */
  const int32_t* host_lens; /* (ABI 3) HOST memory, n_seq base lengths (= the differences of cu_rows), or NULL.  Non-NULL: the MFMA conv / fused-unit
                             * kernels launch a 1-D grid over exactly the REAL tiles of a ragged batch (the launcher counts them from these
                             * lengths for the tile it picks) and every workgroup finds its (sequence, tile) from cu_rows; NULL: the rectangular
                             * grid n_seq x tiles of the longest sequence, whose workgroups past a shorter sequence's end exit at once (a quarter
                             * of the grid at T ~ U{64..128}).  Read at launch time only (a captured graph does not keep the pointer).  Results
                             * are identical either way.  jatts_amd.hip passes it for non-uniform batches only.  It is a HOST pointer that travels
                             * inside descriptors copied to the device as kernel arguments: DEVICE CODE MUST NEVER DEREFERENCE IT -- kernels only
                             * compare it with NULL (common.h: ragged_is_1d) to learn which grid form they were launched with. */
} jatts_ragged;

/* ---------------------------------------------------------------------------------
 * Conv1d / Linear as MFMA implicit GEMM.
 * Replaces torch.nn.Conv1d / Linear / ConvTranspose1d call sites on the path:
 *   modules/transformer/multi_layer_conv.py:52-63 (FFN w_1/w_2),
 *   modules/transformer/attention.py:39-61,93,184 (linear_q/k/v/out/pos),
 *   modules/conformer/convolution.py:66,77 (pointwise convs),
 *   modules/duration_predictor.py:78-84, variance_predictor.py:75-80 (predictor convs),
 *   models/fastspeech2.py:641 (feat_out), modules/pre_postnets.py:183-185 (postnet),
 *   parallel_wavegan HiFiGANGenerator input_conv / upsamples (vocoder.py:64).
 *
 *   y[t, n] = resid[t, n] + alpha * act( bias[n] + sum_{tap, c} W[n, tap, c] *
 *                                pre( in_scale * sum_i x_i[t + tap*dil - pad, c] ) )
 * with rows outside the sequence reading as zero.  W is given in MFMA fragment order
 * (jatts_conv_weight_index).  c_in must be a multiple of 64 (zero-pad channels otherwise).
 * ------------------------------------------------------------------------------- */
typedef struct jatts_conv_desc {
  jatts_ragged rg;
  int32_t dtype;       /* operand type of x_i and w */
  int32_t n_in;        /* 1..3 inputs summed */
  const void* x[3];    /* [rows][ldx] */
  int32_t ldx;
  float in_scale;
  int32_t pre_act;     /* JATTS_PRE_* applied while staging */
  float pre_slope;
  const void* w;       /* packed weights */
  int32_t c_in;        /* multiple of 64 */
  int32_t n_out;       /* true output channels (packed rows = round_up(n_out, 32)) */
  int32_t k_w;         /* taps */
  int32_t dil;
  int32_t pad;         /* input offset: tap 0 reads row t - pad */
  const float* bias;   /* [n_out] or NULL */
  int32_t act;         /* JATTS_ACT_* */
  float alpha;
  const float* resid;  /* f32 [rows][ldr] or NULL */
  int32_t ldr;
  void* y;             /* [rows][ldy] (or [n][ldy] when y_transposed) */
  int32_t ldy;
  int32_t y_is_f32;    /* 1: y is f32 regardless of dtype */
  int32_t y_transposed;/* 1: y[n*ldy + row] (used for V^T) */
  const int32_t* y_seq_col0; /* transposed output only, or NULL: sequence b writes column y_seq_col0[b] + t
                              * instead of its packed row (V^T with 8-aligned sequence starts, see vt_col0) */
  int32_t pad_mode;    /* JATTS_PAD_ZERO (rows outside the sequence read as zero) or JATTS_PAD_REFLECT (torch
                        * padding_mode="reflect", the SpeechBrain Conv1d default used by ECAPA-TDNN; halo < length) */
  int32_t variant;     /* 0: the library picks the kernel / tile (the product setting).  Non-zero forces one where it applies
                        * (parity tests reach every kernel; tools/bench_conv.py tunes the heuristic).  f32, n_out > 64:
                        * 1 = LDS-staged 128n x 64t, 2 = LDS-staged 128n x 128t, 3 = register-streamed ("direct": one zero-padded input, optionally with the
                        * LeakyReLU prologue; csrc/conv1d_direct.h) 128n x 128t with a 2-step operand ring, 4 = the same with a
                        * 4-step ring, 5 = register-streamed 128n x 64t.  Unknown / inapplicable values fall back to 0. */
  const float* w_inv;  /* JATTS_F32S only (NULL otherwise): round_up(n_out, 32) floats, 2^-s[n] where w holds the hi / lo f16 halves of
                        * W[n] * 2^s[n] (jatts_amd.hip.pack_conv_weight_split); x_i, resid and y are f32 */
  const float* act_a;  /* JATTS_ACT_SNAKEBETA only (NULL otherwise): [n_out] exp(alpha) and ... */
  const float* act_b;  /* ... [n_out] 1 / (exp(beta) + 1e-9), the two per-channel vectors jatts_snakebeta takes */
  /* (ABI 4) TWO outputs from one launch -- the Q | K | V projection of an attention block (modules/transformer/attention.py:39-61 forward_qkv; Matcha's
   * attn1.to_q / to_k / to_v, modules/matchatts/transformer.py:222-260) as ONE conv over the concatenated weights: output channels n < n_split go to y as
   * described above (row-major), channels n >= n_split go to y2 TRANSPOSED, y2[(n - n_split) * ldy2 + column] with the column rule of y_seq_col0 taken
   * from y2_seq_col0 (V^T in the attention kernel's layout).  n_split = 0: one output.  Requires n_split % 256 == 0 (a workgroup's channel slab lies on
   * one side), y_transposed = 0, resid = NULL; y2 has y's element type.  Same arithmetic as two launches: every output element is the same contraction. */
  int32_t n_split;
  int32_t ldy2;
  void* y2;
  const int32_t* y2_seq_col0;
  /* (ABI 4) JATTS_F32E / JATTS_F32E6 only: fragment order of w.  0: the order of jatts_conv_weight_index ([tap][c / 16][n / 32][lane][c % 8], v_mfma_f32_32x32x16_bf16
   * kernels); 1: [tap][c / 32][n / 16][lane = 16 ((c % 32) / 8) + n % 16][c % 8] x (b0 | b1 | b2) with n padded to 32 and c to 64 (v_mfma_f32_16x16x32_bf16 kernels,
   * csrc/conv1d_emul16.h: the form the power-limited matrix pipe sustains 14 % more of; jatts_amd.hip.pack_conv_weight_bf16x3_k32). */
  int32_t w_layout;
} jatts_conv_desc;

int jatts_conv1d(const jatts_conv_desc* d, void* stream);

/* Index of W[n][tap][c] inside the packed buffer (HOST helper, pure function):
 * fragments ordered [tap][c/16][n/32][lane = 32*((c%16)/8) + n%32][c%8].
 * n_pad = round_up(n_out, 32). */
int64_t jatts_conv_weight_index(int32_t n, int32_t tap, int32_t c, int32_t n_pad, int32_t c_in);

/* ---------------------------------------------------------------------------------
 * HiFi-GAN ResBlock dilation unit, fused (the dominant kernel):
 *   y = x + conv_k,1( lrelu( conv_k,d( lrelu(x) ) + b1 ) ) + b2
 * parallel_wavegan.layers.HiFiGANResidualBlock.forward [third party; call site
 * jatts/vocoder/vocoder.py:64].  x, y: [rows][channels]; w1/w2 packed as above with
 * c_in = n_out = channels.  y must not alias x.  channels in {32, 64, 128, 256, 512}
 * (f32 mode: up to 256); other widths go through two jatts_conv1d launches.
 * ------------------------------------------------------------------------------- */
typedef struct jatts_resunit_desc {
  jatts_ragged rg;
  int32_t dtype;
  int32_t channels;
  int32_t k_w;
  int32_t dil;
  float slope;
  const void* x;
  void* y;
  const void* w1;
  const float* b1;
  const void* w2;
  const float* b2;
  /* optional MRF mix fused into the (coalesced) output pass: y = out_scale * (unit(x) + add0 + add1);
   * used by the last unit of the last ResBlock so that the mean over ResBlocks
   * (HiFiGANGenerator.forward: cs / num_blocks) is written once.  NULL = plain unit. */
  const void* add0;
  const void* add1;
  float out_scale;
  /* JATTS_F32S only: channels floats each, 2^-s[n] where w1 / w2 were packed as hi/lo of w[n] * 2^s[n] (NULL otherwise) */
  const float* ws1;
  const float* ws2;
  /* (ABI 4) JATTS_F32E / JATTS_F32E6 only: fragment order of w1 / w2.  0: [tap][c / 16][n / 32][lane = 32 ((c % 16) / 8) + n % 32][c % 8] x (b0 | b1 | b2), the
   * order of jatts_conv_weight_index (v_mfma_f32_32x32x16_bf16 kernels).  1: [tap][c / 32][n / 16][lane = 16 ((c % 32) / 8) + n % 16][c % 8] x (b0 | b1 | b2)
   * (v_mfma_f32_16x16x32_bf16 kernels, csrc/resunit_emul16_impl.h: the form the power-limited matrix pipe sustains 14 % more of; jatts_unit_weight_index_k32). */
  int32_t w_layout;
} jatts_resunit_desc;

int jatts_hifigan_resunit(const jatts_resunit_desc* d, void* stream);

/* Index of W[n][tap][c] (one of the three bf16 planes' 8-element groups counted as one element) inside a w_layout = 1 buffer (HOST helper, pure function). */
int64_t jatts_unit_weight_index_k32(int32_t n, int32_t tap, int32_t c, int32_t channels);

/* ---------------------------------------------------------------------------------
 * HiFi-GAN ResBlock, all dilation units fused in one launch (f16 operands; f32 for the small-channel k = 3 blocks):
 *   for u in 0..n_units-1:  x <- x + conv_k,1( lrelu( conv_k,dil[u]( lrelu(x) ) + b1[u] ) ) + b2[u]
 * parallel_wavegan.layers.HiFiGANResidualBlock.forward [third party; call site jatts/vocoder/vocoder.py:64].
 * x is read once and y written once per ResBlock (the per-unit launches move 3x the bytes); the residual stream stays
 * in registers between units, rounded to f16 once per unit (where the per-unit launches round it on its way to HBM);
 * at f32 (channels 32 / 64, k_w = 3: round 3) nothing is rounded: the result equals the per-unit launches to f32 summation order.
 * channels in {32, 64, 128}, k_w in {3, 7} (wider receptive fields waste too much of the tile on halo), n_units <= 3;
 * returns JATTS_ERR_UNSUPPORTED otherwise -- callers then issue jatts_hifigan_resunit per unit.
 * add0 / add1 / out_scale: the MRF mean fused into the output pass, as in jatts_resunit_desc.
 * ------------------------------------------------------------------------------- */
typedef struct jatts_resblock_desc {
  jatts_ragged rg;
  int32_t dtype;     /* JATTS_F16 (channels 32 / 64 / 128), JATTS_F32 for channels 32 / 64 with a chain halo of <= 16 rows a side (k = 3), or JATTS_F32S */
  int32_t channels;
  int32_t k_w;
  int32_t n_units;
  int32_t dil[3];
  float slope;
  const void* x;
  void* y;           /* must not alias x */
  const void* w1[3];
  const float* b1[3];
  const void* w2[3];
  const float* b2[3];
  const void* add0;
  const void* add1;
  float out_scale;
  /* JATTS_F32S (round 4: channels 32 with k_w 3 / 7, channels 64 with k_w 3; x / y f32, weights as for jatts_hifigan_resunit): channels
   * floats each, the inverse per-output-channel weight scales of w1[u] / w2[u] (NULL otherwise) */
  const float* ws1[3];
  const float* ws2[3];
} jatts_resblock_desc;

int jatts_hifigan_resblock(const jatts_resblock_desc* d, void* stream);

/* ---------------------------------------------------------------------------------
 * Training side, first slice of SURVEY 8 f.4 (jatts/trainers/fastspeech2.py:24-100): the criterion sums on forward()'s
 * outputs and the parameter gradients of jatts_conv1d.  dx of a conv is jatts_conv1d itself on W'[c][n][k-1-tap] with
 * pad' = (k-1)*dil - pad.
 * ------------------------------------------------------------------------------- */
/* *out = scale * sum over sequences b, rows t < valid_len[b] (NULL: all rows), columns c < dim of |a - b'| (kind 0, L1Loss) or
 * (a - b')^2 (kind 1, MSELoss); b' = log(b + log_offset) when log_offset >= 0 (DurationPredictorLoss, duration_predictor_loss.py:55),
 * else b.  losses/l1l2_loss.py:43-63, variance_predictor_loss.py.  workspace: 4 * n_seq doubles. */
int jatts_masked_loss(const jatts_ragged* rg, const float* a, int32_t lda, const float* b, int32_t ldb, int32_t dim,
                      const int32_t* valid_len, int32_t kind, float log_offset, double scale, float* out, double* workspace,
                      void* stream);
/* dw[n][c][tap] = sum_t dy[t][n] * x[t + tap*dil - pad][c] over all sequences (torch weight layout (n_out, c_in, k_w), f32).
 * workspace != NULL (n_seq * (k_w * pad64(n_out) * pad64(c_in) + pad64(n_out)) floats): split-K partials are written there and summed
 * by a second launch, dw is OVERWRITTEN (deterministic, no atomics).  workspace == NULL: refused on the MFMA path (k_w 1 / 3 / 5); the VALU
 * path (other widths) then ACCUMULATES into dw, through the fixed-order slabs of jatts_set_workspace (round 4: no f32 atomics anywhere).
 * db (nullable, n_out floats, OVERWRITTEN): the bias gradient sum_t dy[t][n] of the same convolution (torch.nn.Conv1d's bias.grad).  On
 * the MFMA path it falls out of the dy tiles the kernel stages anyway (no second pass over dy); otherwise a fixed-order column-sum launch. */
int jatts_conv1d_wgrad(const jatts_ragged* rg, const float* x, int32_t ldx, const float* dy, int32_t ldy, int32_t c_in,
                       int32_t n_out, int32_t k_w, int32_t dil, int32_t pad, float* dw, float* db, float* workspace, void* stream);
/* Pack a torch-layout f32 weight (n_out, c_in, k_w) into jatts_conv1d's fragment order (zero padded: n to 32, c to c_mult) as
 * `dtype`, in one launch.  mode 0: the weight itself; mode 1: the data-gradient operand W'[c][n][k'] = W[n][c][k_w-1-k'] (a conv from
 * n_out to c_in channels; padded sizes follow the swapped roles).  out: k_w * pad32(n) * pad(c, c_mult) elements. */
int jatts_pack_conv_weight(const float* w, int32_t n_out, int32_t c_in, int32_t k_w, int32_t c_mult, int32_t mode, int32_t dtype,
                           void* out, void* stream);
/* The same for the JATTS_F32S operand (round 4): per packed row n (mode 0: output channel; mode 1: input channel of W) the power-of-two scale
 * that puts max |w| of the row in [2^14, 2^15); out: 2 * k_w * pad32(n) * pad(c, c_mult) f16 = [tap][c/16][n/32][lane][hi x8 | lo x8] of
 * w * 2^s[n]; inv: pad32(n) floats, 2^-s[n] (1 for all-zero and padding rows).  Two launches (row maxima, pack); deterministic. */
int jatts_pack_conv_weight_split(const float* w, int32_t n_out, int32_t c_in, int32_t k_w, int32_t c_mult, int32_t mode, void* out, float* inv,
                                 void* stream);
/* out[c] += sum over rows of x[row][c] (bias gradient; caller zeroes out). */
int jatts_col_sum(const float* x, int32_t ld, int64_t rows, int32_t dim, float* out, void* stream);

/* ---------------------------------------------------------------------------------
 * Training side, second slice of SURVEY 8 f.4: the backward passes (and train-mode forward pieces) of the FastSpeech2 layers
 * around the conv op.  All f32, [rows][dim] row-major; every "+=" output is accumulated with f32 atomics (caller zeroes).
 * The reference has no interface for these: they stand where torch autograd runs for the jatts/modules layers under
 * jatts/trainers/fastspeech2.py:86-96 (gen_loss.backward(); clip_grad_norm_; optimizer.step()).
 * ------------------------------------------------------------------------------- */
/* LayerNorm over the channels (modules/transformer/layer_norm.py:12-42): dx (nullable), dgamma += , dbeta += (both or neither). dim <= 1536. */
int jatts_layernorm_bwd(const float* x, int32_t ldx, const float* dy, int32_t lddy, const float* gamma, int64_t rows, int32_t dim,
                        float eps, float* dx, int32_t lddx, float* dgamma, float* dbeta, void* stream);
/* Element-wise activations, mode 1 ReLU, 2 tanh, 3 Swish (x sigmoid x, modules/conformer/swish.py), 4 Mish; bwd: dx = dy * act'(x). */
int jatts_act_fwd(int32_t mode, const float* x, float* y, int64_t n, void* stream);
int jatts_act_bwd(int32_t mode, const float* x, const float* dy, float* dx, int64_t n, void* stream);
/* GLU over the channel halves (convolution.py:66: F.glu(dim=1)): x [rows][2 dim] -> y [rows][dim]. */
int jatts_glu_fwd(const float* x, float* y, int64_t rows, int32_t dim, void* stream);
int jatts_glu_bwd(const float* x, const float* dy, float* dx, int64_t rows, int32_t dim, void* stream);
/* Depthwise Conv1d (groups = channels, convolution.py:44-52), zero padding inside each sequence: y[t][c] = bias[c] +
 * sum_k w[c][k] x[t + k - pad][c]; flip = 1 uses w[c][k_w-1-k] (the data gradient, with pad' = k_w - 1 - pad). k_w <= 32. */
int jatts_dwconv(const jatts_ragged* rg, const float* x, const float* w, const float* bias, float* y, int32_t dim, int32_t k_w,
                 int32_t pad, int32_t flip, void* stream);
/* dw[c][k] += sum_t dy[t][c] x[t + k - pad][c]. */
int jatts_dwconv_wgrad(const jatts_ragged* rg, const float* x, const float* dy, float* dw, int32_t dim, int32_t k_w, int32_t pad,
                       void* stream);
/* Column sums for batch-statistics BatchNorm1d (train mode of convolution.py:53 / pre_postnets.py:112-143):
 * mode 0: out0[c] += sum_r (x - shift[c]), out1[c] += sum_r (x - shift[c])^2 (shift nullable);
 * mode 1: out0[c] += sum_r y2, out1[c] += sum_r y2 * (x - shift[c]) * mul[c]   (y2 = dy, shift = mean, mul = rstd). */
int jatts_col_stats(const float* x, const float* y2, int32_t ld, int64_t rows, int32_t dim, const float* shift, const float* mul,
                    int32_t mode, float* out0, float* out1, void* stream);
/* dx = gamma rstd (dy - s_dy / rows - xhat s_dyx / rows), xhat = (x - mean) rstd. */
int jatts_bn_bwd_apply(const float* x, const float* dy, int64_t rows, int32_t dim, const float* mean, const float* rstd,
                       const float* gamma, const float* s_dy, const float* s_dyx, float* dx, void* stream);
/* dst[idx[r]][c] += scale * src[r][c]; rows with idx == skip or outside [0, n_dst) dropped (Embedding backward, padding_idx). */
int jatts_index_add_rows(const float* src, int32_t ld, const int64_t* idx, int64_t rows, int32_t dim, float scale, int64_t skip,
                         int64_t n_dst, float* dst, void* stream);
/* Length-regulator backward: dhs[token i of sequence b] = sum of dy over the frames [cum[i-1], cum[i]) of b (deterministic). */
int jatts_lr_segment_sum(const jatts_ragged* rg, const int64_t* cum, const int32_t* cu_out, const float* dy, int32_t dim, float* dhs,
                         void* stream);
/* Attention probabilities of LegacyRelPositionMultiHeadedAttention (modules/transformer/attention.py:142-206) on explicit
 * [n_batch][n_heads][T][T] score matrices: p = softmax_j((ac + rel_shift(bd)) * scale) over the keys j < lens[b], 0 elsewhere
 * (bd nullable: plain attention).  bwd: ds = p (dp - sum_j dp p) * scale (= d ac), dbd = rel_shift^-1(ds) (nullable). */
int jatts_shift_softmax_fwd(const float* ac, const float* bd, int32_t n_batch, int32_t n_heads, int32_t t_len, const int32_t* lens,
                            float scale, int32_t shift_mode, float* p, void* stream);
int jatts_shift_softmax_bwd(const float* p, const float* dp, int32_t n_batch, int32_t n_heads, int32_t t_len, float scale,
                            int32_t shift_mode, float* ds, float* dbd, void* stream);
/* shift_mode 1: the legacy rel_shift above (bd [T][T]); 2: RelPositionMultiHeadedAttention.rel_shift (attention.py:236-258, VITS):
 * bd [T][2T-1], shifted[i][j] = bd[i][j - i + T - 1]. */
/* WaveNet gate backward (vits/wavenet/residual_block.py:150-156): y = tanh(a) sigmoid(b), x = [a | b] [rows][2 dim]. */
int jatts_gate_bwd(const float* x, const float* dy, float* dx, int64_t rows, int32_t dim, void* stream);
/* torch.nn.utils.weight_norm (dim 0; the WaveNet convolutions of jatts/modules/vits/wavenet/residual_block.py:60-103 are wrapped by
 * apply_weight_norm, vits.py:403-411): w [n_out][row_len] = g[o] v[o][:] / ||v[o][:]||, inv_norm[o] = 1 / ||v[o]|| kept for the backward
 * (dv [n_out][row_len], dg [n_out]). */
int jatts_weight_norm_fwd(const float* v, const float* g, int32_t n_out, int32_t row_len, float* w, float* inv_norm, void* stream);
int jatts_weight_norm_bwd(const float* v, const float* g, const float* inv_norm, const float* dw, int32_t n_out, int32_t row_len,
                          float* dv, float* dg, void* stream);
/* WaveNet residual / skip split (residual_block.py:158-167): o [rows][2 dim] = [res | skip]; h_out = h + res, skip_out = skip (nullable) +
 * skip part.  jatts_concat2 is its backward: out [rows][2 dim] = [a | b] (either nullable = zeros). */
int jatts_split_add(const float* o, const float* h, const float* skip, int64_t rows, int32_t dim, float* h_out, float* skip_out, void* stream);
int jatts_concat2(const float* a, const float* b, int64_t rows, int32_t dim, float* out, void* stream);
/* Rank-1 pieces of Linear(dim -> 1) heads and Conv1d(1 -> dim, k=1) embeddings:
 * out[r][c] (+)= v[r] w[c] + bias[c];  out[c] += sum_r v[r] x[r][c];  y[r] = bias[0] + sum_c x[r][c] w[c]. */
int jatts_outer_rows(const float* v, const float* w, const float* bias, int64_t rows, int32_t dim, int32_t accumulate, float* out,
                     void* stream);
int jatts_col_wsum(const float* x, int32_t ld, const float* v, int64_t rows, int32_t dim, float* out, void* stream);
int jatts_row_dot(const float* x, int32_t ld, const float* w, const float* bias, int64_t rows, int32_t dim, float* y, void* stream);
/* Gradient of jatts_masked_loss w.r.t. a: da = *upstream (NULL: 1) * scale * {sign(a - b'), 2 (a - b')} on valid rows, 0 elsewhere. */
int jatts_masked_loss_bwd(const jatts_ragged* rg, const float* a, int32_t lda, const float* b, int32_t ldb, int32_t dim,
                          const int32_t* valid_len, int32_t kind, float log_offset, float scale, const float* upstream, float* da,
                          int32_t ldda, void* stream);
/* torch.nn.GroupNorm(groups, dim) over each sequence's [rows][dim / groups] slabs (matchatts/decoder.py:66-78 Block1D; the
 * statistics run over every row of the sequence as laid out, i.e. the padded length of a padded batch).  fwd also returns the
 * per-(sequence, group) mean / rstd [n_seq * groups] for the backward; bwd: dx (nullable), dgamma +=, dbeta += (both or neither).
 * dim / groups must divide 256. */
int jatts_groupnorm_fwd(const jatts_ragged* rg, const float* x, int32_t dim, int32_t groups, const float* gamma, const float* beta,
                        float eps, float* y, float* mean, float* rstd, void* stream);
int jatts_groupnorm_bwd(const jatts_ragged* rg, const float* x, const float* dy, int32_t dim, int32_t groups, const float* gamma,
                        const float* mean, const float* rstd, float* dx, float* dgamma, float* dbeta, void* stream);
/* SnakeBeta with log-scale parameters (matchatts/transformer.py:84-102): y = x + sin^2(exp(alpha) x) / (exp(beta) + 1e-9);
 * bwd: dx, dalpha +=, dbeta += (gradients w.r.t. the LOG parameters, as stored). */
int jatts_snakebeta_fwd(const float* x, float* y, int64_t rows, int32_t dim, const float* alpha, const float* beta, void* stream);
int jatts_snakebeta_bwd(const float* x, const float* dy, int64_t rows, int32_t dim, const float* alpha, const float* beta, float* dx,
                        float* dalpha, float* dbeta, void* stream);
/* ForwardSumLoss of the alignment framework (losses/forward_sum_loss.py:41-78): per utterance b, F.ctc_loss(reduction "mean",
 * zero_infinity) on log_p[b][t][j] (t < olens[b], j < ilens[b]; the caller has already added the beta-binomial prior) with a
 * constant blank column log_blank and targets 1..ilens[b].  nll[b] = -log p / ilens[b] (0 when impossible).  grad (nullable,
 * [n_batch][t_max][ld]) = grad_scale * d nll[b] / d log_p as torch's ctc backward computes it.  workspace: 2 * n_batch * t_max *
 * (2 * max_ilen + 1) floats. */
int jatts_ctc_forward_sum(const float* log_p, int32_t n_batch, int32_t t_max, int32_t ld, const int32_t* ilens, const int32_t* olens,
                          int32_t max_ilen, float log_blank, float* workspace, float* nll, float* grad, float grad_scale, void* stream);
/* out[s][c] += sum over the rows of sequence s of x[row][c] (f32 atomics; caller zeroes out [n_seq][dim]): the gradient of a
 * per-sequence vector added to every row (speaker / time embeddings, WaveNet global conditioning). */
int jatts_seq_sum(const jatts_ragged* rg, const float* x, int32_t dim, float* out, void* stream);
/* Inverted dropout with a counter-based mask: y[i] = keep(seed, i) ? x[i] / (1 - p) : 0; the backward is the same call on dy. */
/* seed_dev (may be NULL): the mask seed is *seed_dev + seed -- a captured graph replays with a new base seed per step (device memory) and
 * `seed` as the call site's offset. */
int jatts_dropout(const float* x, float* y, int64_t n, float p, uint64_t seed, const uint64_t* seed_dev, void* stream);
/* Head split of the fused Q|K|V projection for the training-time attention (modules/transformer/attention.py:81-88,190-195):
 * qkv [n_batch t_len][3 A] (A = n_heads d_k) -> qu = q + pos_bias_u, qv = q + pos_bias_v, k, vv, each [n_batch][n_heads][t_len][d_k]
 * contiguous.  _bwd: the four gradients -> dqkv [rows][3 A] (overwritten) and du / dv [A] += column sums (caller zeroes them).
 * u == v == NULL (with qv / dqv / du / dv NULL): the plain head split of an attention without position biases (matchatts/transformer.py:16-25). */
int jatts_qkv_split(const float* qkv, const float* u, const float* v, int32_t n_batch, int32_t t_len, int32_t n_heads, int32_t d_k, float* qu,
                    float* qv, float* k, float* vv, void* stream);
int jatts_qkv_split_bwd(const float* dqu, const float* dqv, const float* dk, const float* dvv, int32_t n_batch, int32_t t_len, int32_t n_heads,
                        int32_t d_k, float* dqkv, float* du, float* dv, void* stream);
/* dropout(act(x)) in one pass (the FFN's ReLU -> Dropout, modules/transformer/multi_layer_conv.py:52-63; modes as jatts_act_fwd, mask as
 * jatts_dropout: bit-identical to the two launches).  dy == NULL: out = keep ? act(x) / (1 - p) : 0; else out = keep ? dy act'(x) / (1 - p) : 0. */
int jatts_act_dropout(int32_t mode, const float* x, const float* dy, float* out, int64_t n, float p, uint64_t seed, const uint64_t* seed_dev,
                      void* stream);
/* y[i] = (resid ? resid[i] : 0) + alpha * dropout(x)[i] with jatts_dropout's mask for (seed, i): the residual connections of the
 * conformer layers (jatts/modules/conformer/encoder_layer.py:100-170) in one launch; p == 0 is a plain scaled add. */
int jatts_dropout_add(const float* x, const float* resid, float* y, int64_t n, float p, float alpha, uint64_t seed, const uint64_t* seed_dev,
                      void* stream);
/* Gradient gather for the flat-buffer optimiser: n separately allocated f32 gradient tensors (HOST arrays of device pointers, element
 * counts and destination offsets in elements) -> flat[dst_off[i] .. + numel[i]), = or += (accumulate).  ceil(n / 64) launches, the tensor
 * list passed by value (graph-capturable, no table upload).  Replaces torch's one `param.grad += g` kernel per parameter
 * (AccumulateGrad, 260-650 launches per step of the reference's loss.backward(), jatts/trainers/fastspeech2.py:86). */
int jatts_gather_grads(const float* const* src, const int64_t* numel, const int64_t* dst_off, int32_t n, float* flat, int32_t accumulate,
                       void* stream);
/* *out += sum x^2 (double); Adam step (torch.optim.Adam semantics, step counts from 1) with the gradient scaled by
 * min(1, max_norm / (sqrt(*grad_sumsq) + 1e-6)) when grad_sumsq != NULL and max_norm > 0 (clip_grad_norm_).  The hyper-parameters
 * are doubles: bias corrections and step size are computed in double (torch computes them as Python floats) and rounded once. */
int jatts_sumsq(const float* x, int64_t n, double* out, void* stream);
int jatts_adam_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2, double eps,
                    double weight_decay, int64_t step, const double* grad_sumsq, float max_norm, const float* hyper_dev, void* stream);
/* hyper_dev (may be NULL): 7 device floats [lr / bc1, 1 - beta1, beta2, 1 - beta2, eps, weight_decay, sqrt(bc2)] that REPLACE the host
 * scalars -- a captured training step replays with the scalars of the current optimiser step (jatts_amd.training: graph mode). */

/* Profiling hook (not part of the reference interface): while `buf` is non-NULL, thread 0 of the first n_workgroups
 * workgroups of every jatts_hifigan_resunit launch writes 16 uint64 to buf[16*wg ..]: {XCC_ID<<32 | HW_ID, s_memtime
 * at start, x staged, conv1 done, h written, conv2 done, y assembled, y stored, then s_memrealtime (100 MHz) at start
 * and end, 6 unused}.  buf: device memory of n_workgroups*128 bytes.  Pass NULL to switch tracing off (the default). */
int jatts_debug_trace(void* buf, int64_t n_workgroups);

/* Measurement hook (not part of the reference interface; bench.py's `roofline.practical_peak`): the matrix pipe's sustained issue rate on THIS
 * part at the clock its power budget allows.  One launch of `workgroups` x 256 threads, every wave issuing 2 x 2 fragments of 32 x 32 per K-step
 * for `iters` K-steps (rounded up to even) with nothing else in the kernel: dtype JATTS_F32E / JATTS_F32E6 -> v_mfma_f32_32x32x16_bf16, JATTS_F16 /
 * JATTS_F32S -> v_mfma_f32_32x32x16_f16, JATTS_F32 -> eight v_mfma_f32_32x32x2_f32 per K-step, 16 + JATTS_F32E -> 4 x 4 fragments of v_mfma_f32_16x16x32_bf16 (the
 * same operand bytes per flop; a comparison of the two bf16 forms under the power limit), 16 + JATTS_F16 -> that geometry on v_mfma_f32_16x16x32_f16; feed = 1: both operands re-read from LDS every K-step
 * (ds_read_b128), feed = 0: operands stay in registers.  operands: >= 64 KiB of DEVICE memory, 16-byte aligned, holding operand bits of the dtype (the
 * matrix pipe's power, hence its clock, follows them: zeros run ~19 % faster than random bits); clocks (device, 2 x uint64, may be NULL): s_memtime and
 * s_memrealtime (100 MHz) ticks of workgroup 0 across its MFMA loop; sink: 1 device float, never written.  The caller times the launch on `stream`
 * and divides jatts_mfma_probe_flops() by it. */
int jatts_mfma_probe(int32_t dtype, int32_t feed, const void* operands, int64_t operand_bytes, int32_t iters, int32_t workgroups,
                     uint64_t* clocks, float* sink, void* stream);
double jatts_mfma_probe_flops(int32_t dtype, int32_t iters, int32_t workgroups);

/* Batched f32 GEMM on the exact-f32 matrix pipe, for the attention products of the TRAINING step and their gradients (round 4; reference:
 * torch.matmul inside LegacyRelPositionMultiHeadedAttention.forward / forward_attention, modules/transformer/attention.py:63-93,164-206,
 * under autograd):  C[o][i] (m x n) = alpha * op(A[o][i]) (m x k) * op(B[o][i]) (k x n)  [+ C[o][i] if accumulate]
 * for o < n_outer, i < n_inner; matrix (o, i) of an operand starts at base + o * s?_outer + i * s?_inner elements (a stride may be 0: an
 * operand shared over that index, e.g. the position projection over the batch).  trans_a = 0: A is stored (m x k) with leading dimension
 * lda, 1: stored (k x m); trans_b = 0: B stored (k x n), 1: stored (n x k).  Rows are contiguous (unit inner stride); a matrix whose start and
 * leading dimension are 16-byte multiples is read with 16-byte loads.  Deterministic (no split-K, no atomics). */
int jatts_bgemm(const float* a, int64_t sa_outer, int64_t sa_inner, int32_t lda, int32_t trans_a, const float* b, int64_t sb_outer,
                int64_t sb_inner, int32_t ldb, int32_t trans_b, float* c, int64_t sc_outer, int64_t sc_inner, int32_t ldc, int32_t n_outer,
                int32_t n_inner, int32_t m, int32_t n, int32_t k, float alpha, int32_t accumulate, void* stream);

/* Scratch of the DETERMINISTIC reductions of the training kernels (round 4): jatts_layernorm_bwd, jatts_groupnorm_bwd,
 * jatts_snakebeta_bwd, jatts_dwconv_wgrad, jatts_col_stats, jatts_col_sum, jatts_col_wsum, jatts_seq_sum, jatts_qkv_split_bwd,
 * jatts_sumsq and the VALU path / bias sums of jatts_conv1d_wgrad reduce across workgroups through per-workgroup slabs that the
 * last-arriving workgroup adds up in a fixed order (no f32 atomics: results are bit-identical from run to run, like the
 * reference's).  buf: device memory, 16-byte aligned, ZERO-filled by the caller once, owned by the caller and kept
 * alive until replaced (a captured graph bakes its address in); the first 64 KiB hold tickets, the rest slabs.  A launch that needs
 * more fails with JATTS_ERR_ARG naming the size.  Launches of one stream serialise on it; two streams must not run these kernels
 * concurrently: the scratch is process-wide (jatts_amd.hip records the stream of the last reduction launch and makes any other stream
 * wait for it with an event before its own).  buf = NULL unregisters. */
int jatts_set_workspace(void* buf, int64_t bytes);

/* Output stage: y[t] = tanh( b + sum_{tap,c} w[tap][c] * lrelu( in_scale * sum_i x_i[t+tap-pad][c] ) )
 * (HiFiGANGenerator.output_conv: LeakyReLU(0.01) -> Conv1d(C,1,k) -> Tanh).  w: f32 [k_w][c_in]. */
int jatts_hifigan_output(const jatts_ragged* rg, int32_t dtype, const void* const* x, int32_t n_in,
                         float in_scale, float slope, int32_t c_in, int32_t k_w, const float* w,
                         float bias, float* y, void* stream);

/* ---------------------------------------------------------------------------------
 * Legacy relative-position multi-head self-attention (fused, flash style).
 * Replaces LegacyRelPositionMultiHeadedAttention.forward core
 * (modules/transformer/attention.py:164-206 incl. rel_shift :142-162 and
 * forward_attention :63-93) for one unmasked sequence each:
 *   score[i,j] = ( q_i.k_j + ku[j] + BD'[i,j] ) * scale
 *   BD'[i,j]  = g[i][T-1-i+j]    (j <= i)
 *             = 0                (j == i+1)
 *             = g[i+1][j-i-2]    (j >  i+1)      (the view-reinterpretation wrap)
 * where g[row][h][m] = (q_row,h + pos_bias_v_h) . p_h[m] is produced by jatts_conv1d and
 * ku[row][h] = pos_bias_u_h . k_row,h by jatts_rowdot.  out = softmax(score) V.
 * d_k must be a multiple of 32.  One sequence's K rows (max_len * ldk * 4 bytes) and one head's V^T rows (d_k * ldvt * 4 bytes) are
 * addressed through 32-bit buffer offsets: launches beyond 4 GiB of either are refused with JATTS_ERR_UNSUPPORTED (split the batch).
 * ------------------------------------------------------------------------------- */
typedef struct jatts_relattn_desc {
  jatts_ragged rg;
  int32_t dtype;
  int32_t n_heads;
  int32_t d_k;
  const void* q;  int32_t ldq;   /* [rows][ldq], head h at column h*d_k */
  const void* k;  int32_t ldk;
  const void* vt; int32_t ldvt;  /* V^T: [(h*d_k + d)][ldvt]; column = vt_col0[b] + t, or the packed row if vt_col0 is NULL */
  const void* g;  int32_t ldg;   /* g[row][h][m], row stride n_heads*ldg; NULL = no rel-pos */
  const float* ku;               /* [rows][n_heads] or NULL */
  float scale;
  void* out; int32_t ldo;        /* [rows][ldo], head h at column h*d_k */
  /* rel_mode 0/1: legacy map above (LegacyRelPositionMultiHeadedAttention).
   * rel_mode 2: RelPositionMultiHeadedAttention (attention.py:209-305, new rel_shift :237-261, used
   * by VITS): BD'[i,j] = g[i][rel_center - i + j] for every j (no wrap); g has 2*cap-1 columns
   * whose column m encodes relative position rel_center - m, rel_center = cap - 1. */
  int32_t rel_mode;
  int32_t rel_center;
  /* Per-sequence first column of V^T (n_seq entries) or NULL.  With every entry a multiple of 8, ldvt % 8 == 0,
   * a 32-byte aligned vt and >= 8 readable columns of slack after each sequence, the kernel stages V^T with aligned
   * 16-byte loads; key tiles always start at the sequence's first key, so a sequence's result does not depend on
   * where it sits in the packed batch. */
  const int32_t* vt_col0;
  /* Per-sequence number of valid keys (n_seq entries) or NULL = all.  Keys >= kv_len[b] are excluded from the softmax
   * (the key padding mask of the reference's batched forward(), attention.py:80-88); queries, the rel-shift geometry
   * and V^T keep the full (padded) sequence length. */
  const int32_t* kv_len;
} jatts_relattn_desc;

int jatts_relpos_attention(const jatts_relattn_desc* d, void* stream);

/* out[row][h] = sum_d x[row][h*d_k + d] * vec[h][d]   (pos_bias_u . k ; pos_bias_v . p) */
int jatts_rowdot(int32_t dtype, const void* x, int32_t ldx, int64_t rows, int32_t n_heads,
                 int32_t d_k, const float* vec, float* out, void* stream);

/* ---------------------------------------------------------------------------------
 * Row-wise kernels (HBM bound).
 * ------------------------------------------------------------------------------- */
/* Embedding lookup * scale (encoder.py:133-137 + positional_encoding.py:232): f32 out.  vocab > 0: ids outside [0, vocab)
 * give a zero row and are counted into *n_bad (device int64, caller-zeroed, may be NULL) -- torch.nn.Embedding raises
 * IndexError for them; the host raises it at its next synchronisation point instead of paying one here. */
int jatts_embed_scale(const int64_t* ids, int64_t rows, const float* table, int32_t dim,
                      float scale, float* out, int64_t vocab, int64_t* n_bad, void* stream);

/* LayerNorm over the last dim (modules/transformer/layer_norm.py:12-42, eps 1e-12):
 * x (in_dtype) -> y (out_dtype); gamma/beta f32. */
int jatts_layernorm(const void* x, int32_t in_dtype, int32_t ldx, void* y, int32_t out_dtype,
                    int32_t ldy, int64_t rows, int32_t dim, const float* gamma,
                    const float* beta, float eps, void* stream);

/* y[row][c] = x[row][c] * scale[c] + shift[c]  (f32 in, `out_dtype` out; cols >= dim and
 * < ldy are zero-filled).  Vocoder.decode normalisation (vocoder/vocoder.py:56-61),
 * f32 -> operand casts. */
int jatts_affine_cast(const float* x, int32_t ldx, void* y, int32_t out_dtype, int32_t ldy,
                      int64_t rows, int32_t dim, const float* scale, const float* shift,
                      void* stream);
/* The same map into a column slice: exactly `dim` columns per row are written (row stride ldy), nothing is zero-filled. */
int jatts_affine_slice(const float* x, int32_t ldx, void* y, int32_t out_dtype, int32_t ldy,
                      int64_t rows, int32_t dim, const float* scale, const float* shift,
                      void* stream);

/* Conformer conv-module core (modules/conformer/convolution.py:67-75):
 *   h = GLU(x[:, :C], x[:, C:]) ; y = swish( dwconv_k(h) * bn_scale + bn_shift )
 * x: [rows][2C], y: [rows][C] (dtype); w_dw f32 [C][k]; BatchNorm(eval) and the
 * depthwise bias folded into bn_scale/bn_shift by the host. */
int jatts_glu_dwconv_bn_swish(const jatts_ragged* rg, int32_t dtype, const void* x, void* y,
                              int32_t channels, int32_t k_w, const float* w_dw,
                              const float* bn_scale, const float* bn_shift, void* stream);

/* Predictor head: v[row] = x[row,:].w + b  (duration_predictor.py:85, variance_predictor.py:80).
 * If dur_out != NULL also dur_out[row] = clamp(round(exp(v) - offset), 0) as int64
 * (duration_predictor.py:87-90; round half to even). */
int jatts_predictor_head(int32_t dtype, const void* x, int32_t ldx, int64_t rows, int32_t dim,
                         const float* w, float b, float* v_out, int64_t* dur_out, float offset,
                         void* stream);

/* hs[row][c] += sum_k p[row+k-pad] wp[c][k] + bp[c] + sum_k e[row+k-pad] we[c][k] + be[c]
 * (pitch_embed / energy_embed Conv1d(1, adim, k), models/fastspeech2.py:614-616). */
int jatts_variance_embed_add(const jatts_ragged* rg, float* hs, int32_t dim, const float* p,
                             const float* wp, const float* bp, int32_t kp, const float* e,
                             const float* we, const float* be, int32_t ke, void* stream);

/* WaveNet gated activation (modules/wavenet/residual_block.py:143-160):
 *   y[row][c] = tanh(x[row][c] + g[seq][c]) * sigmoid(x[row][C + c] + g[seq][C + c]),  g optional.
 * x: [rows][2C] (dtype), gseq: f32 [n_seq][2C] or NULL, y: [rows][C] (dtype). */
int jatts_gated_tanh_sigmoid(const jatts_ragged* rg, int32_t dtype, const void* x, const float* gseq,
                             void* y, int32_t channels, void* stream);

/* GroupNorm over (channels/groups x time) of each utterance + Mish (+ per-utterance vector), the body of
 * Matcha's Block1D / ResnetBlock1D (modules/matchatts/decoder.py:66-97):
 *   y[t][c] = mish( (x[t][c] - mean_g) * rstd_g * gamma[c] + beta[c] ) + addvec[seq][c]
 * x: [rows][C] (in_dtype); y: [rows][C] (out_dtype); addvec: f32 [n_seq][C] or NULL.
 * workspace: caller-owned f32 scratch of n_seq * groups * ceil(max_len / 64) * 3 floats (chunk statistics of the
 * time-split two-launch form) or NULL (single-launch form, one workgroup per (utterance, group)). */
int jatts_groupnorm_mish(const jatts_ragged* rg, const void* x, int32_t in_dtype, void* y, int32_t out_dtype,
                         int32_t channels, int32_t groups, const float* gamma, const float* beta, float eps,
                         const float* addvec, float* workspace, void* stream);

/* SnakeBeta (modules/matchatts/transformer.py:84-102): y = x + inv_beta[c] * sin(x * alpha[c])^2 with
 * alpha = exp(log_alpha), inv_beta = 1 / (exp(log_beta) + 1e-9) precomputed by the host. */
int jatts_snakebeta(int32_t dtype, const void* x, void* y, int64_t rows, int32_t channels, const float* alpha,
                    const float* inv_beta, void* stream);

/* y[row][:] = x[row][:] / max(||x[row]||_2, eps)  (torch.nn.functional.normalize on speaker embeddings,
 * models/fastspeech2.py:751, vits.py:706); f32 in, `out_dtype` out, columns >= dim and < ldy zero-filled. */
int jatts_l2_normalize(const float* x, int32_t ldx, void* y, int32_t out_dtype, int32_t ldy, int64_t rows,
                       int32_t dim, float eps, void* stream);

/* VITS prior sampling (models/vits.py:478-480): z[row][c] = stats[row][c] + noise[row][c] *
 * exp(stats[row][C + c]) * noise_scale;  stats: f32 [rows][2C] (m_p | logs_p), noise/z: f32 [rows][C]. */
int jatts_gaussian_sample(const float* stats, const float* noise, float* z, int64_t rows, int32_t channels,
                          float noise_scale, void* stream);

/* y[row][c] = x[row][C-1-c]  (FlipFlow, modules/vits/flow.py:17-40); f32, x != y. */
int jatts_flip_channels(const float* x, float* y, int64_t rows, int32_t channels, void* stream);

/* hs[row][:] += vec[seq(row)][:]  (speaker embedding add, fastspeech2.py:591-597,751-753) */
int jatts_add_seq_vector(const jatts_ragged* rg, float* hs, int32_t dim, const float* vec,
                         void* stream);

/* ---------------------------------------------------------------------------------
 * Speaker-embedding front end: what jatts/modules/feature_extract/spkemb_speechbrain.py:14-28 gets from
 * SpeechBrain's EncoderClassifier.encode_batch (third party, not vendored: log-mel filterbank -> sentence mean
 * normalisation -> ECAPA-TDNN; called per utterance at jatts/bin/tts_decode.py:209-212).  The contractions (DFT, mel
 * projection, TDNN / Res2Net / SE / pooling convs) are jatts_conv1d launches (pad_mode = JATTS_PAD_REFLECT where
 * SpeechBrain's Conv1d pads); these are the row-wise pieces.  All f32 unless noted.
 * ------------------------------------------------------------------------------- */
/* Windowed frames of packed waveforms: frames[row(b,t)][n] = window[n] * x_b[reflect(t*hop + n - n_fft/2)] (torch.stft
 * center=True); rg_frames counts frames per utterance, cu_samples (n_seq+1) delimits the waveforms; columns n_fft..ldo-1 = 0. */
int jatts_frame_signal(const jatts_ragged* rg_frames, const int32_t* cu_samples, const float* x, const float* window,
                       int32_t n_fft, int32_t hop, float* out, int32_t ldo, void* stream);
/* out[row][k] = x[row][k]^2 + x[row][n_bins + k]^2 (k < n_bins), zero up to ldo. */
int jatts_power_spectrum(const float* x, int32_t ldx, int32_t n_bins, int64_t rows, float* out, int32_t ldo, void* stream);
/* Per utterance: dB = 10 log10(max(p, amin)), clamped from below at max(dB) - top_db, minus its mean over time per channel. */
int jatts_fbank_post(const jatts_ragged* rg, const float* p, int32_t ldp, int32_t n_mels, float amin, float top_db,
                     float* out, int32_t ldo, void* stream);
/* Per (utterance, channel) statistics over time with weights softmax_t(logits) (logits == NULL: uniform):
 * mean[b*ldm + c], std[b*ldm + c] = sqrt(max(weighted variance, eps)) (std may be NULL).  SE squeeze + attentive statistics pooling. */
int jatts_seq_mean_std(const jatts_ragged* rg, const float* x, int32_t ldx, int32_t dim, const float* logits, int32_t ldl,
                       float* mean, float* stdv, int32_t ldm, float eps, void* stream);
/* y = post_act( scale[c] * pre_act(x + seq_vec[b][c]) + shift[c] ); pre_act: JATTS_ACT_NONE / _RELU, post_act: _NONE / _TANH;
 * seq_vec / scale / shift may be NULL; y in out_dtype with columns dim..ldy-1 zeroed (TDNNBlock: conv -> ReLU -> BatchNorm). */
int jatts_seq_affine_act(const jatts_ragged* rg, const float* x, int32_t ldx, int32_t dim, const float* seq_vec, int32_t pre_act,
                         const float* scale, const float* shift, int32_t post_act, void* y, int32_t out_dtype, int32_t ldy,
                         void* stream);
/* y[row*ldy + c] = x * sigmoid(s[b][c]) + resid (x, resid: [rows][dim]; resid may be NULL): SE gate + residual connection. */
int jatts_se_scale_add(const jatts_ragged* rg, const float* x, int32_t dim, const float* s, const float* resid, float* y, int32_t ldy,
                       void* stream);

/* ---------------------------------------------------------------------------------
 * Length regulator (modules/length_regulator.py:70-97), bit-exact integer path.
 * Step 1: d_eff = (alpha == 1) ? d : (int64) rint((float)d * alpha)   [half-to-even],
 *         cum[row] = inclusive prefix sum of d_eff inside each sequence, olens[b] = total.
 *         zero_rule (the all-zero rule, :85-94): 0 = none; 1 = every d_eff of every sequence is 1; 2 = per sequence,
 *         as the reference's B=1 inference() behaves: a sequence whose d_eff sum to 0 gets every entry 1, and
 *         olens must then hold 2*n_seq entries, olens[n_seq + b] = 1 where the rule fired (the host logs the
 *         reference's warning), else 0.
 * Step 2: out[cu_out[b] + f][:] = x[cu_rows[b] + idx][:], idx = #{t : cum[t] <= f};
 *         frame_index (optional) receives idx as int64.
 * ------------------------------------------------------------------------------- */
int jatts_lr_durations(const jatts_ragged* rg, const int64_t* d, float alpha, int32_t zero_rule,
                       int64_t* d_eff, int64_t* cum, int64_t* olens, void* stream);
int jatts_lr_gather(const jatts_ragged* rg_in, const int64_t* cum, const int32_t* cu_out,
                    int32_t max_out_len, const float* x, int32_t dim, float* out,
                    int64_t* frame_index, void* stream);
/* Output sequences may be longer than sum(d_eff): frames past the end are zero (pad_list, length_regulator.py:16-43;
 * the padded batches of forward()), frame_index -1. */

/* Zero the rows t >= valid_len[b] of every sequence of a packed matrix x[rows][ld] (first `dim` columns; f32 or f16):
 * the `xs * x_masks` / `x * mask` steps of the reference's batched forward() passes (variance_predictor.py:81-83,
 * duration_predictor.py:93-96, matchatts/decoder.py:75-77,93-96). */
int jatts_zero_pad_rows(const jatts_ragged* rg, void* x, int32_t dtype, int32_t ld, int32_t dim, const int32_t* valid_len,
                        void* stream);
/* Conditional-flow-matching training pair (matchatts/flow_matching.py:115-121): per sequence b with time t[b],
 * y = (1 - (1 - sigma_min) t) z + t x1, u = x1 - (1 - sigma_min) z; all [rows][dim] f32. */
int jatts_cfm_mix(const jatts_ragged* rg, const float* x1, const float* z, const float* t, float sigma_min, int32_t dim,
                  float* y, float* u, void* stream);
/* *out = scale * sum_i (a_i - b_i)^2 (double accumulation, fixed order): F.mse_loss(..., reduction="sum") / normaliser
 * (flow_matching.py:123-125).  workspace: 256 doubles. */
int jatts_sq_err_sum(const float* a, const float* b, int64_t n, float scale, float* out, double* workspace, void* stream);

/* Gaussian upsampling (modules/length_regulator.py:111-154), one unmasked sequence each:
 * out[f] = softmax_t( -delta (f - c_t)^2 ) @ hs,  c_t = cumsum(d)_t - d_t/2 (float math). */
int jatts_gaussian_upsample(const jatts_ragged* rg_in, const int64_t* d, const int32_t* cu_out,
                            int32_t max_out_len, const float* hs, int32_t dim, float delta,
                            float* out, void* stream);

/* Stage-4 output: float waveform -> little-endian 16-bit PCM, the conversion libsndfile applies for
 * sf.write(path, y, fs, "PCM_16") (jatts/bin/tts_decode.py:250-255): y[i] = lrintf(clamp(x[i], -1, 1) * 32767.f). */
int jatts_pcm16(const float* x, int64_t n, int16_t* y, void* stream);

/* ---------------------------------------------------------------------------------
 * Alignment learning (training-side neighbours of the stage-4 path, Matcha-MAS / VITS).
 * ------------------------------------------------------------------------------- */
/* AlignmentModule.forward after its convolutions (modules/alignments.py:50-59):
 *   log_p[f][i] = log_softmax_i( -|| feats[f] - text[i] ||_2 )  over the utterance's own tokens;
 * columns >= T_text of a row are -inf (the reference's masked_fill of padded tokens).
 * feats: f32 [frame rows][adim] (f_conv3 output), text: f32 [token rows][adim] (t_conv2 output),
 * cu_text: int32 [n_seq + 1] token offsets, log_p: f32 [frame rows][ld], ld >= max_text_len <= 512. */
int jatts_alignment_logp(const jatts_ragged* rg_feats, const int32_t* cu_text, int32_t max_text_len, const float* feats,
                         const float* text, int32_t adim, float* log_p, int32_t ld, void* stream);

/* Monotonic alignment search + duration extraction for the whole batch (modules/alignments.py:63-93
 * _monotonic_alignment_search, :281-310 viterbi_decode; the reference runs a numba loop per utterance on the host):
 *   path[f]  = token index of frame f (int64, per utterance),   dur[i] = number of frames of token i (np.bincount),
 *   score[b] = sum_f log_p[f][path[f]]  (bin_loss = -mean over frames, averaged over the batch by the caller).
 * Q is float64 as in the reference; row 0 is a float64 running sum of the float32 inputs (see oracle/mas_oracle.py).
 * Limits: T_text <= 1024 and T_feats * ceil(T_text / 64) * 8 bytes of decision bits must fit LDS. score may be NULL. */
int jatts_mas_viterbi(const jatts_ragged* rg_feats, const int32_t* cu_text, const float* log_p, int32_t ld,
                      int32_t max_text_len, int64_t* path, int64_t* dur, double* score, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* JATTS_HIP_H_ */
