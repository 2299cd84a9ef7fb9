// jatts_conv1d, f16 operands, n_out <= 64: 64n x 256t workgroup tile (all four waves along time).
#include "conv1d_impl.h"

int jatts_conv1d_f16_narrow(const jatts_conv_desc& d, hipStream_t s) { return launch_conv<f16, 2, 2, 1, 4>(d, s); }
