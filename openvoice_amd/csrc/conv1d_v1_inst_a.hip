// Tile 128x128 (2x2 waves, 64x64 per wave), 16-byte staging.  Rows >= 96: WaveNet, coupling,
// posterior, conv_pre, ups 0-2, ResBlock stages with C >= 128.
#include "conv1d_mfma_v1.h"
namespace ovk {
namespace v1 {
// explicit kernel instantiations (both host and device passes see these)
template __global__ void conv1d_mfma_v1_kernel<1, 1, 2, 2, 2, 2, true>(const ov_conv1d_params);
template __global__ void conv1d_mfma_v1_kernel<3, 1, 2, 2, 2, 2, true>(const ov_conv1d_params);
template __global__ void conv1d_mfma_v1_kernel<3, 3, 2, 2, 2, 2, true>(const ov_conv1d_params);
template __global__ void conv1d_mfma_v1_kernel<3, 5, 2, 2, 2, 2, true>(const ov_conv1d_params);
template __global__ void conv1d_mfma_v1_kernel<5, 1, 2, 2, 2, 2, true>(const ov_conv1d_params);
template __global__ void conv1d_mfma_v1_kernel<7, 1, 2, 2, 2, 2, true>(const ov_conv1d_params);
template __global__ void conv1d_mfma_v1_kernel<7, 3, 2, 2, 2, 2, true>(const ov_conv1d_params);
template __global__ void conv1d_mfma_v1_kernel<7, 5, 2, 2, 2, 2, true>(const ov_conv1d_params);
template __global__ void conv1d_mfma_v1_kernel<11, 1, 2, 2, 2, 2, true>(const ov_conv1d_params);
template __global__ void conv1d_mfma_v1_kernel<11, 3, 2, 2, 2, 2, true>(const ov_conv1d_params);
template __global__ void conv1d_mfma_v1_kernel<11, 5, 2, 2, 2, 2, true>(const ov_conv1d_params);
#if !defined(__HIP_DEVICE_COMPILE__)
const ConvVariant kV1VariantsA[] = {
    {1, 1, TILE_128x128, 1, conv1d_v1_launch<1, 1, 2, 2, 2, 2, true>},
    {3, 1, TILE_128x128, 1, conv1d_v1_launch<3, 1, 2, 2, 2, 2, true>},
    {3, 3, TILE_128x128, 1, conv1d_v1_launch<3, 3, 2, 2, 2, 2, true>},
    {3, 5, TILE_128x128, 1, conv1d_v1_launch<3, 5, 2, 2, 2, 2, true>},
    {5, 1, TILE_128x128, 1, conv1d_v1_launch<5, 1, 2, 2, 2, 2, true>},
    {7, 1, TILE_128x128, 1, conv1d_v1_launch<7, 1, 2, 2, 2, 2, true>},
    {7, 3, TILE_128x128, 1, conv1d_v1_launch<7, 3, 2, 2, 2, 2, true>},
    {7, 5, TILE_128x128, 1, conv1d_v1_launch<7, 5, 2, 2, 2, 2, true>},
    {11, 1, TILE_128x128, 1, conv1d_v1_launch<11, 1, 2, 2, 2, 2, true>},
    {11, 3, TILE_128x128, 1, conv1d_v1_launch<11, 3, 2, 2, 2, 2, true>},
    {11, 5, TILE_128x128, 1, conv1d_v1_launch<11, 5, 2, 2, 2, 2, true>},
};
const int kV1NumVariantsA = sizeof(kV1VariantsA) / sizeof(kV1VariantsA[0]);
#endif
}  // namespace v1
}  // namespace ovk
