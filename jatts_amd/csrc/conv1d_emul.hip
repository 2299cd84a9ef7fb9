// jatts_conv1d, JATTS_F32E / JATTS_F32E6: f32 activations in HBM, f32-equivalent emulated MFMA operands (three bf16 terms, seven / six
// partial products; conv1d_emul.h).
#include <stdlib.h>

#include "conv1d_emul16.h"

// The arithmetic has no tile-dependent scale, so -- unlike the split kernels -- the tile may follow the launch: every variant gives the
// same bits for a row.  (variants: JATTS_CONV_EMUL_VARIANT, measured by tools/bench_conv.py --dtype emul)
template <typename T>
static int conv1d_emul(const jatts_conv_desc& d, hipStream_t s) {
  static const int variant = [] { const char* e = getenv("JATTS_CONV_EMUL_VARIANT"); return e ? atoi(e) : 0; }();
  // seven products carry two accumulators per fragment (common.h): the 2 x 2-fragment wave tile no longer fits 256 registers, so the
  // 128 n x 128 t tile runs as eight waves of 1 x 2 fragments (64-channel chunks, one workgroup per CU, two waves per SIMD)
  constexpr bool TWO = sizeof(typename Acc32<T>::type) > sizeof(f32x16);
  if (d.n_in > 1)     // summed inputs (an unfused MRF mean in front of a HiFi-GAN upsampling conv; rare): 3x the staging registers, one workgroup per CU
    return launch_conv_emul<T, 2, 2, 2, 2, 3, 32, 1>(d, s);
  if (d.n_out <= 64) return launch_conv_emul<T, 2, 1, 1, 4, 1, 32, 2>(d, s);          // 64 n x 128 t, light on registers: the HBM-bound last upsampling conv
  if (variant == 1) return launch_conv_emul<T, 2, 2, 2, 2, 1, 64, 1, 32, 4>(d, s);    // 64-channel chunks, one workgroup per CU
  // k = 1 with two accumulators: 128 n x 64 t, four waves of 2 x 1 fragments, two workgroups per CU (+3-8 % on the large launches, +20-35 % at
  // 4 096 / 8 192 rows against the eight-wave tile; at k = 3 the eight-wave tile is 3-4 % ahead: profiles/r05_notes.md)
  // (six products: the same tile for SMALL k = 1 launches -- <= 768 workgroups of 128 x 128, i.e. one round of the two-per-CU slots: 40-72 -> 58-79 TFLOP/s at
  //  4 096 / 8 192 rows.  Nothing in this arithmetic depends on the tile, so the choice may follow the launch.)
  const int64_t maxL = (int64_t)d.rg.max_len * d.rg.len_mul;
  const int64_t wgs128 = ((maxL + 127) / 128) * d.rg.n_seq * ((d.n_out + 127) / 128);
  const bool small_k1 = d.k_w == 1 && wgs128 <= 768;
  // round 6, the B = 1 drop-in path (one utterance: 768 rows): a k = 3 feed-forward conv is 72 (384 -> 1536) or 18 (1536 -> 384) workgroups of 128 x 128 on 256
  // CUs, each walking the whole contraction -- 44 % of the utterance's GPU time.  Launches that cannot give every CU one such workgroup take the 128 n x 64 t
  // tile (twice the workgroups, two per CU).  Seven products only: their k > 1 product tile walks the contraction in the same 64-channel chunks, so a row's
  // bits do not change (the six-product k > 1 tile uses 32-channel chunks: another summation order at k > 1).
  if (variant == 0 && TWO && d.k_w > 1 && wgs128 <= 128 && (d.k_w - 1) * d.dil <= 32) return launch_conv_emul<T, 1, 2, 4, 1, 1, 64, 2, 32, 4>(d, s);
  // The same 128 n x 64 t tile as four waves of 1 x 2 fragments SIDE BY SIDE IN n (each wave 32 n x 64 t): half the weight fragments fetched per
  // MFMA (a 2 x 1 wave pulls 6 KB of weights per 14 MFMAs through the vector memory path, eight waves ~110 B / clk / CU): another +4-9 % on every
  // k = 1 shape, +20 % at 4 096 rows (profiles/r05_notes.md); 64 n x 128 t (variant 5) stages twice the activations per MFMA and loses 30 %.
  if (variant == 6 || (variant == 0 && d.k_w == 1 && (TWO || small_k1))) return launch_conv_emul<T, 1, 2, 4, 1, 1, 64, 2, 32, 4>(d, s);
  // (round 6, measured and dropped -- profiles/r06_notes.md: 256 n x 64 t / 128 n x 128 t tiles of 2 x 2-fragment waves with 64- and 128-channel chunks and
  //  HALO = 0 staging for k = 1: +-3 % except 2048 -> 512 (+14 % with 128-channel chunks), -10..-25 % on the 384-wide shapes; the next chunk's commit dealt out
  //  between the MFMAs of the current one with sched_group_barrier: spills at 256 registers, -20 % at k = 1, -6 % at k = 3; starting the CUs' second
  //  workgroup slot half a K-loop late: +-1 %)
  if (variant == 4) return launch_conv_emul<T, 2, 1, 2, 2, 1, 64, 2, 32, 4>(d, s);    // (2 x 1 fragments per wave, 2 x 2 waves: the first k = 1 tile)
  if (variant == 5) return launch_conv_emul<T, 1, 2, 2, 2, 1, 64, 2, 32, 4>(d, s);    // 64 n x 128 t, four waves of 1 x 2 fragments, two workgroups per CU
  // (measured and dropped: 256 n x 64 t and 256 n x 128 t eight-wave tiles -- more output channels per staged tile -- were 5-25 % slower on most shapes and
  //  +3-12 % only on 512 -> 2048 / 2048 -> 512 k1 in the six-product mode; profiles/r05_notes.md)
  if (variant == 2 || (TWO && variant != 3)) return launch_conv_emul<T, 1, 2, 4, 2, 1, 64, 1, 32, 4>(d, s);    // 8 waves, 64-channel chunks, one workgroup per CU
  if constexpr (TWO) return launch_conv_emul<T, 2, 2, 2, 2, 1, 32, 1>(d, s);          // (variant 3) four waves, one workgroup per CU
  else return launch_conv_emul<T, 2, 2, 2, 2, 1, 32, 2>(d, s);                        // 128 n x 128 t, two workgroups per CU
}

// The 16 x 16 x 32 form (w_layout = 1; conv1d_emul16.h).  Every tile walks the contraction in 64-channel chunks, so -- in BOTH arithmetics -- a row's bits do
// not depend on the tile and the choice may follow the launch (tests/test_emul_gpu.py: test_conv1d_emul16_tiles_agree forces each tile through
// jatts_conv_desc.variant and compares bits).  Tiles, by variant number (JATTS_CONV_EMUL16_VARIANT or jatts_conv_desc.variant force one; 0 = the rules below):
//   6: 384 n x 64 t, eight waves of 3 x 4 fragments, ONE-step weight ring (226 registers), one workgroup per CU
//   3: 256 n x 64 t, eight waves of 2 x 4 fragments, one workgroup per CU
//   2: 128 n x 128 t, eight waves of 2 x 4 fragments (anti-phase staging), one workgroup per CU
//   1: 128 n x 64 t, four waves of 2 x 4 fragments, two workgroups per CU
//   9: 128 n x 32 t, four waves of 2 x 2 fragments, two workgroups per CU
template <typename T>
static int conv1d_emul16(const jatts_conv_desc& d, hipStream_t s) {
  static const int env_variant = [] { const char* e = getenv("JATTS_CONV_EMUL16_VARIANT"); return e ? atoi(e) : 0; }();
  const int variant = env_variant ? env_variant : d.variant;
  if (d.n_in > 1) return launch_conv_emul16<T, 4, 4, 2, 2, 3, 64, 1>(d, s);           // summed inputs (rare): 128 n x 128 t, four waves, one workgroup per CU
  if (d.n_out <= 64) return launch_conv_emul16<T, 2, 2, 2, 2, 1, 64, 2>(d, s);        // 64 n x 64 t, four waves, two workgroups per CU: the HBM-bound last upsampling conv
  const int64_t maxL = (int64_t)d.rg.max_len * d.rg.len_mul;
  const int64_t t64 = ((maxL + 63) / 64) * d.rg.n_seq;                                // 64-row time tiles of the launch
  const int64_t wgs128 = ((maxL + 127) / 128) * d.rg.n_seq * ((d.n_out + 127) / 128);
  // WIDE tiles (round 6, late).  Staging -- the activation loads, the split arithmetic, the LDS writes -- is paid per n tile and is the piece the k = 1 DIAG
  // table prices highest (profiles/r06_conv16_diag.txt: with every non-MFMA piece compiled out the 128 x 64 tile runs at 230 - 240 TFLOP/s, 0.9 of the LDS-fed
  // ceiling; staging +23 %, weight / B refills +17 %, epilogue +10 % ADD instead of overlapping, and the clock sits at 1.36 - 1.5 GHz against 1.85 in the probe).
  // A staged x chunk that serves 384 / 256 output channels halves / thirds it: 384 -> 384 k1 119 -> 142, 1536 -> 384 k3 174 -> 200 (its input staged once
  // instead of three times), 384 -> 1536 k3 184 -> 194, 2048 -> 512 k1 127 -> 146 TFLOP/s (profiles/r06_conv16_384_tile.txt, r06_conv16_wide_tiles.txt).
  // Conditions: n_out in whole tiles (a 384-wide output on 256-wide tiles leaves a quarter of the second one idle: -16 .. -21 %), at least one round of such
  // workgroups (fewer lose 13 - 15 %), and -- a two-output Q | K | V launch splits between workgroups -- n_split in whole tiles (jatts_conv1d checks whole 256s).
  const bool ok384 = d.n_split % 384 == 0;
  if (ok384 && (variant == 6 || (variant == 0 && d.n_out % 384 == 0 && t64 * (d.n_out / 384) >= 256))) return launch_conv_emul16<T, 3, 4, 8, 1, 1, 64, 1, 32, 1>(d, s);
  if (variant == 3 || (variant == 0 && d.n_out % 256 == 0 && t64 * (d.n_out / 256) >= 256)) return launch_conv_emul16<T, 2, 4, 8, 1, 1, 64, 1>(d, s);
  // (measured and dropped, evidence under profiles/: 192 n x 64 t two workgroups per CU -- r06_conv16_192_tile.txt; 512 n x 32 t -- r06_conv16_512_tile.txt: twice
  //  the weight bytes per MFMA cost what half the staging saved; the 256-wide tile with the one-step ring: +-2 %; four-wave tiles of 4 x 4 fragments, one wave per
  //  SIMD, 256 n x 64 t and 128 n x 128 t: -8 .. -25 % -- r06_conv16_wide_tiles.txt; 64 n x 32 t for one utterance: behind 128 n x 32 t -- r06_conv16_b1_tiles.txt)
  // ONE-UTTERANCE launches (the B = 1 drop-in path: 768 rows): even the 128 n x 64 t tile leaves most CUs idle -- 36 workgroups for a 1536 -> 384 conv, each
  // walking 144 K-steps.  128 n x 32 t when the 64-row tiling gives at most one workgroup per CU: 1536 -> 384 k3 110 -> 67 us, 384 -> 1536 k3 35 -> 29 us,
  // 384 -> 384 k1 16 -> 12 us; b1_latency 8.6 -> 7.55 ms (profiles/r06_conv16_b1_tiles.txt)
  if (variant == 9 || (variant == 0 && t64 * ((d.n_out + 127) / 128) <= 256)) return launch_conv_emul16<T, 2, 2, 4, 1, 1, 64, 2>(d, s);
  // 128 n x 64 t, two workgroups per CU: k = 1 (the eight-wave 128 x 128 tile halves the weight traffic -- 1.56 - 1.76 GHz -- but converts with the pipe idle; the
  // anti-phase staging gives it +3 - 5 % at k = 1, still behind this tile), and launches that cannot give every CU a 128 x 128 workgroup
  if (variant == 1 || (variant == 0 && (d.k_w == 1 || wgs128 <= 128))) return launch_conv_emul16<T, 2, 4, 4, 1, 1, 64, 2>(d, s);
  return launch_conv_emul16<T, 2, 4, 4, 2, 1, 64, 1>(d, s);                           // 128 n x 128 t, eight waves, one workgroup per CU
}

int jatts_conv1d_emul(const jatts_conv_desc& d, hipStream_t s) {
  if (d.w_layout == 1) return d.dtype == JATTS_F32E6 ? conv1d_emul16<bf3f>(d, s) : conv1d_emul16<bf3>(d, s);
  return d.dtype == JATTS_F32E6 ? conv1d_emul<bf3f>(d, s) : conv1d_emul<bf3>(d, s);
}
