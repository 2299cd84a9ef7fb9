// jatts_conv1d, f32 operands (v_mfma_f32_32x32x2_f32: exact f32 fma chains, the reference's arithmetic).
#include "conv1d_impl.h"

int jatts_conv1d_f32(const jatts_conv_desc& d, hipStream_t s) {
  return d.n_out <= 64 ? launch_conv<float, 2, 2, 1, 4>(d, s) : launch_conv<float, 2, 2, 2, 2>(d, s);
}
