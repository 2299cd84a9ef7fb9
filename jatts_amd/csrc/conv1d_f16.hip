// jatts_conv1d, f16 operands (v_mfma_f32_32x32x16_f16, f32 accumulate): 128n x 128t workgroup tile
// (64 accumulators per lane, 2-3 workgroups per CU).
#include "conv1d_impl.h"

int jatts_conv1d_f16_narrow(const jatts_conv_desc& d, hipStream_t s);

int jatts_conv1d_f16(const jatts_conv_desc& d, hipStream_t s) {
  if (d.n_out <= 64) return jatts_conv1d_f16_narrow(d, s);
  return launch_conv<f16, 2, 2, 2, 2>(d, s);
}
