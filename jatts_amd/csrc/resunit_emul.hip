// Fused HiFi-GAN dilation unit, f32 activations in HBM, f32-EQUIVALENT emulated MFMA operands (three bf16 terms per value):
// JATTS_F32E (seven partial products per product) and JATTS_F32E6 (six).
//
// The LDS tile holds 6 bytes per element, so the windows are narrower than the f32 / split kernels' at the same channel count;
// a K-step is 7 (6) x 32 pipe cycles per fragment pair against 48 B of each operand: the loop is matrix-pipe bound with room to spare
// on operand delivery, and what the tile choice trades is the halo overhead (tt_out / WGCOLS) against workgroups per CU.
#include "resunit_emul16_impl.h"

// Tile choice measured on the box (profiles/r05_emul_units.txt, 64 x 768 frames): C = 128 / 256 run best as ONE 4-wave workgroup per CU
// with NF = 2 x NT = 2 fragments per wave (24 MFMAs per K-step between operand fetches), C = 64 as two 4-wave workgroups per CU
// (one per CU with the 256-column window at k = 11), C = 32 as two 256-column workgroups.
template <typename T>
static int resunit_emul(const jatts_resunit_desc& d, hipStream_t s) {
  static const int variant = [] { const char* e = getenv("JATTS_RESUNIT_EMUL_VARIANT"); return e ? atoi(e) : 0; }();
  const int halo = (d.k_w - 1) * d.dil;   // x-tile rows beyond the workgroup's columns
  // Round 6, the B = 1 drop-in path: ONE utterance is 6 144 rows at C = 256 and 49 152 at C = 128 -- 114 and 417 workgroups of the throughput windows on 256
  // CUs, each walking both convs' whole contraction (100-180 us).  A launch that cannot hand every CU a workgroup (C = 256) or ends in a half-empty
  // second round (C = 128) takes the next narrower window: more halo recomputation, twice the workgroups, ~0.6x the time.  Bit-identical (nothing in this
  // arithmetic depends on the tile).
  const int64_t cols = (int64_t)d.rg.max_len * d.rg.len_mul;
  auto wgs = [&](int window) { const int64_t tt = window - (d.k_w - 1); return tt > 0 ? ((cols + tt - 1) / tt) * d.rg.n_seq : (int64_t)1 << 40; };
  const bool auto_tile = variant == 0;
  switch (d.channels) {
    // C <= 64: the residual from registers (RREG; variant 9 = the round-5 form that re-reads x in the store pass, for A/B)
    case 32:
      if (variant == 1) return launch_resunit_emul<T, 32, 128, 1, 2, 2, 2>(d, s);          // 2 waves per workgroup
      if (variant == 9) return launch_resunit_emul<T, 32, 256, 1, 2, 2, 2>(d, s);
      return launch_resunit_emul<T, 32, 256, 1, 2, 2, 2, false, true>(d, s);
    case 64:
      if (variant == 2) return launch_resunit_emul<T, 64, 256, 2, 2, 2, 1>(d, s);          // 8 waves NF = 1 NT = 2
      if (variant == 9) return d.k_w >= 11 ? launch_resunit_emul<T, 64, 256, 1, 2, 2, 1>(d, s) : launch_resunit_emul<T, 64, 128, 2, 2, 2, 2>(d, s);
      if (variant == 1 || d.k_w >= 11) return launch_resunit_emul<T, 64, 256, 1, 2, 2, 1, false, true>(d, s);   // 4 waves NF = 2 NT = 2, one workgroup per CU
      return launch_resunit_emul<T, 64, 128, 2, 2, 2, 2, false, true>(d, s);               // 4 waves NF = 1 NT = 2, two workgroups per CU
    case 128:
      if (variant == 1) return launch_resunit_emul<T, 128, 128, 4, 2, 2, 1>(d, s);         // 8 waves NF = 1 NT = 2, one workgroup per CU
      if (variant == 3) return launch_resunit_emul<T, 128, 128, 2, 1, 2, 1>(d, s);         // 8 waves NF = 2 NT = 1
      if ((variant == 4 || (auto_tile && wgs(128) <= 448)) && (64 + halo) * 784 + 1024 <= 80 * 1024) return launch_resunit_emul<T, 128, 64, 2, 1, 2, 2>(d, s);   // two per CU
      return launch_resunit_emul<T, 128, 128, 2, 2, 2, 1>(d, s);                           // 4 waves NF = 2 NT = 2, one workgroup per CU
    case 256:
      if (auto_tile && wgs(64) <= 160) return launch_resunit_emul<T, 256, 32, 4, 1, 2, 1>(d, s);   // small launch: 32-column window
      if ((64 + halo) * 1552 + 2048 <= 160 * 1024) {
        if (variant == 1) return launch_resunit_emul<T, 256, 64, 8, 2, 2, 1>(d, s);        // 8 waves NF = 1 NT = 2
        return launch_resunit_emul<T, 256, 64, 4, 2, 2, 1>(d, s);                          // 4 waves NF = 2 NT = 2
      }
      if (variant == 2) return launch_resunit_emul<T, 256, 32, 4, 1, 2, 1>(d, s);          // 32-column window, 82-row tile
      return launch_resunit_emul<T, 256, 64, 4, 2, 2, 1, true>(d, s);                      // k = 11, dilation 5: the 114-row x tile one channel half at a time
  }
  return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "resunit: unsupported channels for JATTS_F32E (32 / 64 / 128 / 256)");
}

// The same tiles on v_mfma_f32_16x16x32_bf16 (w_layout = 1; resunit_emul16_impl.h): waves as (WN, WT), a wave's tile 64 channels x 64 columns wherever the
// channel count allows (C = 32: 32 x 64, C = 64 with two workgroups per CU: 32 x 64).
template <typename T>
static int resunit_emul16(const jatts_resunit_desc& d, hipStream_t s) {
  const int halo = (d.k_w - 1) * d.dil;
  switch (d.channels) {
    // (round 6, measured and dropped -- profiles/r06_unit16_c64_c32_wide.txt: C = 32 with 512-column windows, one workgroup per CU: 9 - 22 % slower; the
    //  one-workgroup-per-CU C = 64 unit at k = 3 / 7: 10 - 12 % / 4 % slower.  Small-channel units live on the second resident workgroup.)
    case 32: return launch_resunit_emul16<T, 32, 256, 1, 4, 2, false, true>(d, s);
    case 64:
      if (d.k_w >= 11) return launch_resunit_emul16<T, 64, 256, 1, 4, 1, false, true>(d, s);
      return launch_resunit_emul16<T, 64, 128, 2, 2, 2, false, true>(d, s);
    case 128: return launch_resunit_emul16<T, 128, 128, 2, 2, 1>(d, s);
    case 256:
      if ((64 + halo) * 1552 + 2048 <= 160 * 1024) return launch_resunit_emul16<T, 256, 64, 4, 1, 1>(d, s);
      return launch_resunit_emul16<T, 256, 64, 4, 1, 1, true>(d, s);
  }
  return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "resunit: unsupported channels for JATTS_F32E (32 / 64 / 128 / 256)");
}

int jatts_resunit_emul(const jatts_resunit_desc& d, hipStream_t s) {
  if (d.w_layout == 1) return d.dtype == JATTS_F32E6 ? resunit_emul16<bf3f>(d, s) : resunit_emul16<bf3>(d, s);
  return d.dtype == JATTS_F32E6 ? resunit_emul<bf3f>(d, s) : resunit_emul<bf3>(d, s);
}
