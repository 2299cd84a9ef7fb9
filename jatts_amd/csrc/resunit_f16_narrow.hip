// Fused HiFi-GAN dilation unit, f16 operands, 32 / 64 channels (the HBM-bound late stages).
#include "resunit_impl.h"

int jatts_resunit_f16_narrow(const jatts_resunit_desc& d, hipStream_t s) {
  // tile variants: <C, workgroup columns, waves along n, 32-col fragments per wave, weight-ring depth>.
  // JATTS_RESUNIT_VARIANT (tuning knob, read once) selects alternative tilings for sweeps.
  static const int variant = [] { const char* e = getenv("JATTS_RESUNIT_VARIANT"); return e ? atoi(e) : 0; }();
  const bool wide_k = d.k_w > 3;
  switch (d.channels * 10 + variant) {
    case 320: return wide_k && d.k_w > 7 ? launch_resunit<f16, 32, 512, 1, 4, 2>(d, s) : launch_resunit<f16, 32, 256, 1, 2>(d, s);
    case 321: return launch_resunit<f16, 32, 256, 1, 2>(d, s);
    case 323: return launch_resunit<f16, 32, 512, 1, 4, 2>(d, s);
    case 640: return wide_k ? launch_resunit<f16, 64, 512, 1, 4, 4>(d, s) : launch_resunit<f16, 64, 256, 1, 2>(d, s);
    case 641: return launch_resunit<f16, 64, 256, 1, 2>(d, s);
    case 643: return launch_resunit<f16, 64, 256, 1, 4, 4>(d, s);
    case 644: return launch_resunit<f16, 64, 512, 1, 4, 4>(d, s);
  }
  return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "resunit: unsupported channels/dtype (use jatts_conv1d)");
}
