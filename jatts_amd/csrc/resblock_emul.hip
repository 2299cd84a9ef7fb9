// Fused HiFi-GAN ResBlock (all dilation units in one launch), JATTS_F32E / JATTS_F32E6: f32 x / y, emulated MFMA operands (three exact bf16
// terms, seven / six partial products; resblock_emul_impl.h).  The HBM-bound small-channel blocks only; wider receptive fields / channel
// counts go unit by unit (jatts_hifigan_resunit).
#include "resblock_emul_impl.h"

template <typename T>
static int resblock_emul(const jatts_resblock_desc& d, hipStream_t s) {
  static const int variant = [] { const char* e = getenv("JATTS_RESBLOCK_EMUL_VARIANT"); return e ? atoi(e) : 0; }();
  switch (d.channels) {
    case 32:
      if (variant == 1) return launch_resblock_emul<T, 32, 512, 1, 4, 2, 1>(d, s);   // 4 waves x (32 ch x 128 columns), one workgroup per CU
      return launch_resblock_emul<T, 32, 256, 1, 2, 2, 2>(d, s);                     // 4 waves x (32 ch x 64 columns), two workgroups per CU
    case 64:
      if (variant == 1) return launch_resblock_emul<T, 64, 256, 1, 2, 2, 1>(d, s);   // 4 waves x (64 ch x 64 columns), one workgroup per CU
      if (variant == 2) return launch_resblock_emul<T, 64, 256, 2, 2, 2, 1>(d, s);   // 8 waves x (32 ch x 64 columns)
      return launch_resblock_emul<T, 64, 128, 2, 2, 2, 2>(d, s);                     // 4 waves x (32 ch x 64 columns), two workgroups per CU
  }
  return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "resblock (emulated): 32 / 64 channels only (use jatts_hifigan_resunit)");
}

int jatts_resblock_emul(const jatts_resblock_desc& d, hipStream_t s) { return d.dtype == JATTS_F32E6 ? resblock_emul<bf3f>(d, s) : resblock_emul<bf3>(d, s); }
