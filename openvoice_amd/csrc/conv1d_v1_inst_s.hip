// Tile 128x128, 4-byte staging for rows that are not 16-byte aligned (frame-rate tensors with
// T % 4 != 0: WaveNet k5, 1x1 convs, conv_pre k7, ups.0 as a 3-tap conv).
#include "conv1d_mfma_v1.h"
namespace ovk {
namespace v1 {
// explicit kernel instantiations (both host and device passes see these)
template __global__ void conv1d_mfma_v1_kernel<1, 1, 2, 2, 2, 2, false>(const ov_conv1d_params);
template __global__ void conv1d_mfma_v1_kernel<3, 1, 2, 2, 2, 2, false>(const ov_conv1d_params);
template __global__ void conv1d_mfma_v1_kernel<5, 1, 2, 2, 2, 2, false>(const ov_conv1d_params);
template __global__ void conv1d_mfma_v1_kernel<7, 1, 2, 2, 2, 2, false>(const ov_conv1d_params);
#if !defined(__HIP_DEVICE_COMPILE__)
const ConvVariant kV1VariantsS[] = {
    {1, 1, TILE_128x128, 0, conv1d_v1_launch<1, 1, 2, 2, 2, 2, false>},
    {3, 1, TILE_128x128, 0, conv1d_v1_launch<3, 1, 2, 2, 2, 2, false>},
    {5, 1, TILE_128x128, 0, conv1d_v1_launch<5, 1, 2, 2, 2, 2, false>},
    {7, 1, TILE_128x128, 0, conv1d_v1_launch<7, 1, 2, 2, 2, 2, false>},
};
const int kV1NumVariantsS = sizeof(kV1VariantsS) / sizeof(kV1VariantsS[0]);
#endif
}  // namespace v1
}  // namespace ovk
