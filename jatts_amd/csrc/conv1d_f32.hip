// jatts_conv1d, f32 operands (v_mfma_f32_32x32x2_f32: exact f32 fma chains, the reference's arithmetic).
#include <stdlib.h>

#include "conv1d_impl.h"

int jatts_conv1d_f32_direct(const jatts_conv_desc& d, int variant, hipStream_t s);   // conv1d_f32_direct.hip

int jatts_conv1d_f32(const jatts_conv_desc& d, hipStream_t s) {
  // JATTS_CONV_F32_TILE (process-wide tuning override) / jatts_conv_desc.variant (per call): see include/jatts_hip.h
  static const int env_tile = [] { const char* e = getenv("JATTS_CONV_F32_TILE"); return e ? atoi(e) : 0; }();
  const int tile = d.variant ? d.variant : env_tile;
  // default: the register-streamed kernel wherever it applies (one plain zero-padded input: every projection / FFN / postnet conv of the
  // acoustic models): 2-20 % faster than the LDS-staged tiles on every shape of tools/bench_conv.py (profiles/r03_notes.md)
  if (tile == 0 || tile >= 3) {
    const int rc = jatts_conv1d_f32_direct(d, tile, s);
    if (rc != 1) return rc;
  }
  if (d.n_out <= 64) return launch_conv<float, 2, 2, 1, 4>(d, s);
  // The 128 x 128 tile runs two workgroups per CU (512 slots).  A launch of <= ~1.1 x that many workgroups spends its second round
  // nearly empty; the 64-step tile (half the work per workgroup, 5-10 % less efficient per FLOP) fills the chip better there:
  // 4-14 % faster on the training-size and half-rate shapes (384->384, 1536->384 k3 at 24 576 rows; 512->512, 1024->512 at 12 288),
  // slower everywhere else (profiles/r02_notes.md).
  const int64_t maxL = (int64_t)d.rg.max_len * d.rg.len_mul;
  const int64_t wgs = ((maxL + 127) / 128) * d.rg.n_seq * ((d.n_out + 127) / 128);
  if (tile == 1 || (tile == 0 && wgs <= 600)) return launch_conv<float, 2, 1, 2, 2>(d, s);     // 128 n x 64 t
  return launch_conv<float, 2, 2, 2, 2>(d, s);
}
