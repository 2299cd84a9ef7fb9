// Fused HiFi-GAN ResBlock (all dilation units in one launch), JATTS_F32S: f32 x / y, split-precision MFMA operands (resblock_split_impl.h).
// The HBM-bound small-channel blocks only; wider receptive fields / channel counts go unit by unit (jatts_hifigan_resunit).
#include "resblock_split_impl.h"

int jatts_resblock_split(const jatts_resblock_desc& d, hipStream_t s) {
  static const int variant = [] { const char* e = getenv("JATTS_RESBLOCK_SPLIT_VARIANT"); return e ? atoi(e) : 0; }();
  switch (d.channels) {
    case 32:
      if (variant == 1) return launch_resblock_split<32, 256, 1, 2, 2, 2>(d, s);
      return launch_resblock_split<32, 512, 1, 4, 2, 2>(d, s);   // 4 waves x (32 ch x 128 columns), two workgroups per CU
    case 64:   // (tools/bench_unit.py --resblock --dtype split: 3.31 ms against 3.83 ms for the 8-wave form and 3.73 ms as three unit launches)
      if (variant == 1) return launch_resblock_split<64, 256, 2, 2, 2, 2>(d, s);   // 8 waves x (32 ch x 64 columns)
      return launch_resblock_split<64, 256, 1, 2, 2, 2>(d, s);   // 4 waves x (64 ch x 64 columns): 256 registers (one 8-byte spill outside the loops)
  }
  return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "resblock (split): 32 / 64 channels only (use jatts_hifigan_resunit)");
}
