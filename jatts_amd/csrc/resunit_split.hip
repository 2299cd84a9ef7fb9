// Fused HiFi-GAN dilation unit, f32 activations in HBM, split-precision (f16 hi/lo) MFMA operands: JATTS_F32S.
//
// The LDS tile holds 4 bytes per element like the f32 kernel's (hi and lo planes), so the window geometry follows
// resunit_f32.hip: the widest window whose tile still leaves two workgroups (or eight waves) per CU.  What changes is the
// balance: a K-step is 3 x 32 pipe cycles per fragment pair instead of 8 x 64, so operand delivery (weight fragments from
// L1 / L2, activation fragments from LDS) and the epilogues matter again, as in the f16 kernels.
#include "resunit_split_impl.h"

int jatts_resunit_split(const jatts_resunit_desc& d, hipStream_t s) {
  static const int variant = [] { const char* e = getenv("JATTS_RESUNIT_SPLIT_VARIANT"); return e ? atoi(e) : 0; }();
  const int halo = (d.k_w - 1) * d.dil;   // x-tile rows beyond the workgroup's columns
  switch (d.channels) {
    case 32:
      if (variant == 1) return launch_resunit_split<32, 512, 1, 4, 2, 2>(d, s);
      return launch_resunit_split<32, 256, 1, 2, 2, 2>(d, s);
    case 64:
      if (variant == 1) return launch_resunit_split<64, 256, 1, 4, 2, 2>(d, s);
      if (variant == 2) return launch_resunit_split<64, 256, 1, 2, 4, 2>(d, s);
      return launch_resunit_split<64, 256, 1, 2, 2, 2>(d, s);
    case 128:
      if (variant == 1) return launch_resunit_split<128, 256, 2, 2, 2, 1>(d, s);        // 8 waves, one workgroup per CU
      if (variant == 2) return launch_resunit_split<128, 128, 4, 4, 2, 2>(d, s);        // NF = 1, NT = 4
      if (variant == 3) return launch_resunit_split<128, 128, 2, 4, 2, 1>(d, s);        // NF = 2, NT = 4: two waves per workgroup
      if ((128 + halo) * 528 + 2048 + 64 <= 80 * 1024) return launch_resunit_split<128, 128, 2, 2, 4, 2>(d, s);
      return launch_resunit_split<128, 256, 2, 2, 4, 1>(d, s);
    case 256:
      if (variant == 1) return launch_resunit_split<256, 64, 4, 2, 2, 1>(d, s);
      if ((128 + halo) * 1040 + 4096 + 64 <= 160 * 1024) return launch_resunit_split<256, 128, 4, 4, 2, 1>(d, s);
      return launch_resunit_split<256, 96, 4, 3, 2, 1>(d, s);
  }
  return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "resunit: unsupported channels for JATTS_F32S (32 / 64 / 128 / 256)");
}
