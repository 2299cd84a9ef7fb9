// Fused HiFi-GAN ResBlock (all dilation units in one launch), f16 operands: tile selection.
#include "resblock_impl.h"

int jatts_resblock_f16(const jatts_resblock_desc& d, hipStream_t s) {
  // <C, window columns, waves along n, 32-column fragments per wave, weight-ring depth, min waves per SIMD>
  static const int variant = [] { const char* e = getenv("JATTS_RESBLOCK_VARIANT"); return e ? atoi(e) : 0; }();
  int H = 0;
  for (int u = 0; u < d.n_units; ++u) H += (d.k_w - 1) / 2 * (d.dil[u] + 1);
  if (2 * H > 128) return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "resblock: receptive field too wide to fuse (use jatts_hifigan_resunit)");
  switch (d.channels) {
    case 32:
      if (variant == 1) return launch_resblock<f16, 32, 256, 1, 2, 2, 2>(d, s);
      return launch_resblock<f16, 32, 512, 1, 4, 2, 2>(d, s);
    case 64:
      if (variant == 1 || (variant == 0 && 2 * H <= 32)) return launch_resblock<f16, 64, 256, 1, 2, 4, 2>(d, s);
      return launch_resblock<f16, 64, 512, 1, 4, 4, 1>(d, s);
    case 128:
      return launch_resblock<f16, 128, 256, 2, 4, 4, 1>(d, s);
  }
  return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "resblock: unsupported channels (use jatts_hifigan_resunit)");
}
