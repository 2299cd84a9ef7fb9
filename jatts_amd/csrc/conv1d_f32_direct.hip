// jatts_conv1d, f32 operands, register-streamed variants (conv1d_direct.h): no LDS tile, no barrier in the main loop.
#include "conv1d_direct.h"

// -> JATTS_OK / error, or 1 when the variant does not apply (the caller falls back to the LDS-staged kernel)
int jatts_conv1d_f32_direct(const jatts_conv_desc& d, int variant, hipStream_t s) {
  if (!conv_direct_ok(d) || d.n_out <= 64) return 1;
  switch (variant) {
    case 3: return launch_conv_direct<2, 2, 2, 2, 2>(d, s);      // 128n x 128t, 256 threads, ring 2: three workgroups per CU
    case 4: return launch_conv_direct<2, 2, 2, 2, 4>(d, s);      // ring 4 (two workgroups per CU)
    case 8: return launch_conv_direct<2, 2, 2, 2, 2, 1>(d, s);   // DIAGNOSIS, wrong results: nothing streamed in the main loop
    default: return 1;
  }
}
