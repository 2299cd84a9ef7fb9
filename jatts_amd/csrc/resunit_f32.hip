// Fused HiFi-GAN dilation unit, f32 operands (exact-f32 MFMA: the reference's arithmetic).
//
// In f32 every unit shape is MFMA-bound (arithmetic intensity C*k/2 >= 48 FLOP/B against a ridge of ~20), and one
// v_mfma_f32_32x32x2_f32 occupies its SIMD for 64 cycles, so loads hide trivially.  What decides the rate is (a) the halo
// recompute, tt_out / WGCOLS of the issued MFMAs are useful -> widest tile the 160 KiB LDS admits, and (b) every
// SIMD holding a wave: 256-thread (or 512-thread) workgroups only.
#include "resunit_impl.h"

int jatts_resunit_f32(const jatts_resunit_desc& d, hipStream_t s) {
  static const int variant = [] { const char* e = getenv("JATTS_RESUNIT_F32_VARIANT"); return e ? atoi(e) : 0; }();
  const int halo = (d.k_w - 1) * d.dil;   // x-tile rows beyond the workgroup's columns
  if (variant == 9) {   // round-1 tiles (kept for A/B runs)
    switch (d.channels) {
      case 32: return launch_resunit<float, 32, 256, 1, 2>(d, s);
      case 64: return launch_resunit<float, 64, 128, 1, 2>(d, s);
      case 128: return launch_resunit<float, 128, 64, 2, 2>(d, s);
      case 256: return launch_resunit<float, 256, 32, 4, 1>(d, s);
    }
  }
  // last template argument: residual kept in registers (x fetched once); variant 2 = the re-reading kernels, for A/B runs
  if (variant == 2) {
    switch (d.channels) {
      case 32: return launch_resunit<float, 32, 512, 1, 4, 8, 2>(d, s);
      case 64: return launch_resunit<float, 64, 256, 1, 2, 8, 2>(d, s);
      case 128:
        if ((128 + halo) * 528 + 1024 <= 80 * 1024) return launch_resunit<float, 128, 128, 2, 2, 8, 2>(d, s);
        return launch_resunit<float, 128, 256, 2, 2, 8, 2>(d, s);
    }
  }
  switch (d.channels) {
    case 32:
      if (variant == 1) return launch_resunit<float, 32, 256, 1, 2, 8, 2>(d, s);
      return launch_resunit<float, 32, 512, 1, 4, 8, 2, true>(d, s);
    case 64:
      if (variant == 1) return launch_resunit<float, 64, 512, 1, 4, 8, 1>(d, s);
      return launch_resunit<float, 64, 256, 1, 2, 8, 2, true>(d, s);
    case 128:
      // 128 columns (4 waves): two workgroups per CU while (128 + halo) x 528 B fits twice in 160 KiB
      // (register-resident residual: neutral up to k = 7 on the 128-column tile, 3-5 % slower at k = 11 and on the 512-thread tile:
      //  those keep the re-read -- profiles/r02_notes.md)
      if (variant != 1 && (128 + halo) * 528 + 1024 <= 80 * 1024)
        return d.k_w <= 7 ? launch_resunit<float, 128, 128, 2, 2, 8, 2, true>(d, s) : launch_resunit<float, 128, 128, 2, 2, 8, 2>(d, s);
      return launch_resunit<float, 128, 256, 2, 2, 8, 2>(d, s);   // 8 waves, one workgroup per CU
    case 256:   // 416 / 344 registers already: no room for 128 / 96 more
      if ((128 + halo) * 1040 + 2048 <= 160 * 1024) return launch_resunit<float, 256, 128, 4, 4, 8, 1>(d, s);
      return launch_resunit<float, 256, 96, 4, 3, 8, 1, true>(d, s);
  }
  return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "resunit: unsupported channels/dtype (use jatts_conv1d)");
}
