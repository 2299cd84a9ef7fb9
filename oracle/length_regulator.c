/* CPU oracle (C restatement): LengthRegulator index arithmetic.
 * TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Built by oracle/Makefile
 * into oracle/_build/liblr_oracle.so; only tests/, smoke() and bench.py's
 * cpu_baseline leg may load it.
 *
 * Follows /root/reference/jatts/modules/length_regulator.py:
 *   :81-83  alpha != 1  ->  ds = round(ds.float() * alpha).long()   (half-to-even,
 *           fp32 product, as torch.round on a float32 tensor)
 *   :85-94  whole-batch ds.sum() == 0 -> rows whose own sum is 0 become all-ones
 *   :96     repeat_interleave(x_b, d_b, dim=0) per utterance
 *   :97     pad_list(..., 0.0) (:16-43): zero-pad to the batch max
 * Pinned by the KATs in SURVEY.md §8 A9 (measured on the real reference) and by
 * tests/golden/lr_kat.npz.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

/* Effective durations after alpha scaling and the all-zero fix-up.
 * ds: (B, T) int64 row-major (entries past ilens[b] are ignored and written 0
 * in the output, matching a zero-padded duration tensor).
 * Returns the batch maximum of the per-utterance frame counts. */
int64_t lr_effective_durations(const int64_t* ds, const int32_t* ilens, int B, int T,
                               float alpha, int64_t* d_eff, int64_t* olens) {
  int64_t total = 0;
  for (int b = 0; b < B; ++b) {
    int64_t s = 0;
    for (int t = 0; t < T; ++t) {
      int64_t d = (t < ilens[b]) ? ds[(int64_t)b * T + t] : 0;
      if (alpha != 1.0f) {
        float v = (float)d * alpha;
        d = (int64_t)nearbyintf(v); /* default rounding mode: half-to-even */
      }
      d_eff[(int64_t)b * T + t] = d;
      s += d;
    }
    olens[b] = s;
    total += s;
  }
  if (total == 0) { /* length_regulator.py:85-94 (fills the WHOLE row, pads included) */
    for (int b = 0; b < B; ++b) {
      for (int t = 0; t < T; ++t) d_eff[(int64_t)b * T + t] = 1;
      olens[b] = T;
    }
  }
  int64_t mx = 0;
  for (int b = 0; b < B; ++b) mx = olens[b] > mx ? olens[b] : mx;
  return mx;
}

/* Frame -> token index map for one utterance: idx[f] = token whose repeat covers
 * frame f (== searchsorted(cumsum(d), f, right=True)). */
void lr_frame_index(const int64_t* d, int T, int64_t* idx) {
  int64_t f = 0;
  for (int t = 0; t < T; ++t)
    for (int64_t r = 0; r < d[t]; ++r) idx[f++] = t;
}

/* Full gather on fp32 rows: out (B, Tmax, D) zero-padded. */
void lr_gather_f32(const float* x, const int64_t* d_eff, const int64_t* olens, int B, int T,
                   int D, int64_t Tmax, float* out) {
  (void)olens;
  memset(out, 0, sizeof(float) * (size_t)B * (size_t)Tmax * (size_t)D);
  for (int b = 0; b < B; ++b) {
    int64_t f = 0;
    for (int t = 0; t < T; ++t)
      for (int64_t r = 0; r < d_eff[(int64_t)b * T + t]; ++r, ++f)
        memcpy(out + ((int64_t)b * Tmax + f) * D, x + ((int64_t)b * T + t) * D,
               sizeof(float) * (size_t)D);
  }
}
