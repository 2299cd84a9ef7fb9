// Fused HiFi-GAN ResBlock (all dilation units in one launch), f32 operands: the small-channel k = 3 blocks only (resblock_impl.h).
#include "resblock_impl.h"

int jatts_resblock_f32(const jatts_resblock_desc& d, hipStream_t s) {
  int H = 0;
  for (int u = 0; u < d.n_units; ++u) H += (d.k_w - 1) / 2 * (d.dil[u] + 1);
  if (2 * H > 32) return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "resblock f32: receptive field too wide to fuse (use jatts_hifigan_resunit)");
  switch (d.channels) {
    case 32: return launch_resblock<float, 32, 512, 1, 4, 2, 2>(d, s);   // 4 waves x (32 ch x 128 columns), two workgroups per CU
    case 64: return launch_resblock<float, 64, 256, 1, 2, 2, 2>(d, s);   // 4 waves x (64 ch x 64 columns)
  }
  return jatts_set_error_msg(JATTS_ERR_UNSUPPORTED, "resblock f32: 32 / 64 channels only (use jatts_hifigan_resunit)");
}
