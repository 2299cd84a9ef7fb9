// jatts_conv1d, f32 operands (v_mfma_f32_32x32x2_f32: exact f32 fma chains, the reference's arithmetic).
#include <stdlib.h>

#include "conv1d_impl.h"

int jatts_conv1d_f32(const jatts_conv_desc& d, hipStream_t s) {
  // JATTS_CONV_F32_TILE: 0 = heuristic (default), 1 = always 128 n x 64 t, 2 = always 128 n x 128 t
  static const int tile = [] { const char* e = getenv("JATTS_CONV_F32_TILE"); return e ? atoi(e) : 0; }();
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
