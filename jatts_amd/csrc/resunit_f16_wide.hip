// Fused HiFi-GAN dilation unit, f16 operands, 128 / 256 / 512 channels (the MFMA-bound early stages).
#include "resunit_impl.h"

int jatts_resunit_f16_wide(const jatts_resunit_desc& d, hipStream_t s) {
  static const int variant = [] { const char* e = getenv("JATTS_RESUNIT_VARIANT"); return e ? atoi(e) : 0; }();
  // Default (variant 0) = the fastest tiling measured per shape (profiles/r01_notes.md): 4 time fragments
  // per wave where the accumulators still allow 2 waves/SIMD -- every weight fragment fetched through the
  // 64 B/clk vector L1 then feeds 4 MFMAs instead of 2, which is what bounds the 128/256-channel units.
  switch (d.channels * 10 + variant) {
    case 1280:
      // 2 x (256 + 2*p1) rows x 272 B must fit in 160 KiB for 2 workgroups/CU: k=11, d=5 misses by 5 rows -> 3-fragment tile
      if (((256 + (d.k_w - 1) * d.dil) * 272 + 1024) * 2 > 160 * 1024) return launch_resunit<f16, 128, 192, 2, 3, 4>(d, s);
      return launch_resunit<f16, 128, 256, 2, 4, 4>(d, s);
    case 1283: return launch_resunit<f16, 128, 256, 2, 4, 4>(d, s);
    case 1281: return launch_resunit<f16, 128, 128, 2, 2, 8>(d, s);
    case 1284: return launch_resunit<f16, 128, 128, 2, 4, 4>(d, s);
    case 1285: return launch_resunit<f16, 128, 192, 2, 3, 4>(d, s);
    case 2560: case 2563: return launch_resunit<f16, 256, 128, 4, 4, 4>(d, s);
    case 2561: return launch_resunit<f16, 256, 64, 4, 2, 8>(d, s);
    case 2564: return launch_resunit<f16, 256, 128, 4, 2, 4>(d, s);
    case 5120: return launch_resunit<f16, 512, 32, 4, 1>(d, s);
  }
  return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "resunit: unsupported channels/dtype (use jatts_conv1d)");
}
