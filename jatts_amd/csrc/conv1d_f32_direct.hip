// jatts_conv1d, f32 operands, register-streamed variants (conv1d_direct.h): no LDS tile, no barrier in the main loop.
#include "conv1d_direct.h"

// -> JATTS_OK / error, or 1 when the variant does not apply (the caller falls back to the LDS-staged kernel)
int jatts_conv1d_f32_direct(const jatts_conv_desc& d, int variant, hipStream_t s) {
  if (!conv_direct_ok(d)) return 1;
  const bool snake = d.act == JATTS_ACT_SNAKEBETA;   // Matcha's feed-forward convs (256 -> 1024 ... k1): the two wide tiles carry the epilogue
  if (snake && (d.n_out <= 64 || d.pre_act != JATTS_PRE_NONE || (variant != 0 && variant != 3 && variant != 5))) return 1;
  if (d.n_out <= 64) {   // narrow outputs (HiFi-GAN's last upsampling conv: 64 -> 2 x 32 channels at 6.3 M rows): 64n x 256t, four waves along time
    if (variant != 0 && variant != 3) return 1;
    return d.pre_act != JATTS_PRE_NONE ? launch_conv_direct<2, 2, 1, 4, 2, 0, true>(d, s) : launch_conv_direct<2, 2, 1, 4, 2>(d, s);
  }
  if (variant == 0) {
    // 128n x 128t unless the launch would leave most of the 768 workgroup slots (3 per CU) empty or its time tiles half empty
    // (sequences of <= 64 rows): then 128n x 64t -- tools/bench_conv.py, profiles/r03_notes.md: +40-80 % at 4 096 rows, +5-17 % at
    // 8 192 rows x n_out <= 768, -13 % at 8 192 rows x 384 -> 1536 k3 (768 workgroups), which therefore stays on the big tile
    const int64_t maxL = (int64_t)d.rg.max_len * d.rg.len_mul;
    const int64_t t128 = (maxL + 127) / 128, t64 = (maxL + 63) / 64;
    const int64_t wgs = t128 * d.rg.n_seq * ((d.n_out + 127) / 128);
    variant = (wgs <= 512 || t64 * 64 * 23 <= t128 * 128 * 20) ? 5 : 3;
  }
  if (d.pre_act != JATTS_PRE_NONE) {
    // LeakyReLU prologue (the HiFi-GAN upsampling convs): max(v, slope v) on the activation fragment about to be consumed -- 32 VALU per
    // 32-MFMA step, and still 5-8 % faster than the LDS-staged kernel on the large launches (256 -> 1024 k3 at 393 216 rows: 134.0 vs 126.9
    // TFLOP/s, 512 -> 2048 k1: 111.9 vs 104.1).  Both tiles: an utterance must take the same kernel family (same summation order) alone
    // and inside a batch -- results are bit-identical wherever a sequence sits (tests/test_fullsize_gpu.py).
    if (variant == 5) return launch_conv_direct<2, 1, 2, 2, 2, 0, true>(d, s);
    return launch_conv_direct<2, 2, 2, 2, 2, 0, true>(d, s);
  }
  if (snake) return variant == 5 ? launch_conv_direct<2, 1, 2, 2, 2, 0, false, true>(d, s) : launch_conv_direct<2, 2, 2, 2, 2, 0, false, true>(d, s);
  switch (variant) {
    case 3: return launch_conv_direct<2, 2, 2, 2, 2>(d, s);      // 128n x 128t, 256 threads, ring 2: three workgroups per CU
    case 4: return launch_conv_direct<2, 2, 2, 2, 4>(d, s);      // ring 4 (two workgroups per CU)
    case 5: return launch_conv_direct<2, 1, 2, 2, 2>(d, s);      // 128n x 64t (64n x 32t wave tiles): twice the workgroups for small launches
#ifdef JATTS_DIAG   // diagnosis builds only (make DIAG=1, used by tools/): never in the shipped library
    case 8: return launch_conv_direct<2, 2, 2, 2, 2, 1>(d, s);   // WRONG RESULTS by design: nothing streamed in the main loop (profiles/r03_notes.md)
#endif
    default: return 1;
  }
}
