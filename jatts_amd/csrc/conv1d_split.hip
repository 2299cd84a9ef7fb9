// jatts_conv1d, JATTS_F32S: f32 activations in HBM, split-precision (f16 hi / lo) MFMA operands (conv1d_split.h).
#include <stdlib.h>

#include "conv1d_split.h"

// ONE tile geometry per (n_out class, input count): 128 time steps x 64-channel chunks everywhere.  The activation scale is the maximum of a
// (time tile, channel chunk) block, so the result depends on the block geometry -- and an utterance must come out bit-identical alone
// and inside any batch (tests/test_fullsize_gpu.py), which rules out choosing the tile by the size of the launch as the f32 kernels do.
int jatts_conv1d_split(const jatts_conv_desc& d, hipStream_t s) {
  if (d.n_in > 1)     // summed inputs (an unfused MRF mean in front of a HiFi-GAN upsampling conv; rare): 3x the staging registers, one workgroup per CU
    return launch_conv_split<2, 2, 2, 2, 3, 64, 1>(d, s);
  if (d.n_out <= 64) return launch_conv_split<2, 1, 1, 4, 1, 64, 2>(d, s);          // 64 n x 128 t, light on registers: the HBM-bound last upsampling conv
  // (measured and dropped: 128-channel chunks for k = 1 -- half the block-maximum exchanges and barriers per K, but the double-buffered tile is then
  //  135 KB of LDS = ONE workgroup per CU: 128-170 TFLOP/s against 165-213 with 64-channel chunks and two workgroups; profiles/r04_notes.md)
  return launch_conv_split<2, 2, 2, 2, 1, 64, 2>(d, s);                             // 128 n x 128 t
}
