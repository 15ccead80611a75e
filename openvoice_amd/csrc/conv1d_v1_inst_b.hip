// Tile 64x256 (1x4 waves, 64x64 per wave): 64-row convs (ResBlock stage 2, ups.3).
#include "conv1d_mfma_v1.h"
namespace ovk {
namespace v1 {
// explicit kernel instantiations (both host and device passes see these)
template __global__ void conv1d_mfma_v1_kernel<3, 1, 2, 2, 1, 4, true>(const ov_conv1d_params);
template __global__ void conv1d_mfma_v1_kernel<3, 3, 2, 2, 1, 4, true>(const ov_conv1d_params);
template __global__ void conv1d_mfma_v1_kernel<3, 5, 2, 2, 1, 4, true>(const ov_conv1d_params);
template __global__ void conv1d_mfma_v1_kernel<7, 1, 2, 2, 1, 4, true>(const ov_conv1d_params);
template __global__ void conv1d_mfma_v1_kernel<7, 3, 2, 2, 1, 4, true>(const ov_conv1d_params);
template __global__ void conv1d_mfma_v1_kernel<7, 5, 2, 2, 1, 4, true>(const ov_conv1d_params);
template __global__ void conv1d_mfma_v1_kernel<11, 1, 2, 2, 1, 4, true>(const ov_conv1d_params);
template __global__ void conv1d_mfma_v1_kernel<11, 3, 2, 2, 1, 4, true>(const ov_conv1d_params);
template __global__ void conv1d_mfma_v1_kernel<11, 5, 2, 2, 1, 4, true>(const ov_conv1d_params);
#if !defined(__HIP_DEVICE_COMPILE__)
const ConvVariant kV1VariantsB[] = {
    {3, 1, TILE_64x256, 1, conv1d_v1_launch<3, 1, 2, 2, 1, 4, true>},
    {3, 3, TILE_64x256, 1, conv1d_v1_launch<3, 3, 2, 2, 1, 4, true>},
    {3, 5, TILE_64x256, 1, conv1d_v1_launch<3, 5, 2, 2, 1, 4, true>},
    {7, 1, TILE_64x256, 1, conv1d_v1_launch<7, 1, 2, 2, 1, 4, true>},
    {7, 3, TILE_64x256, 1, conv1d_v1_launch<7, 3, 2, 2, 1, 4, true>},
    {7, 5, TILE_64x256, 1, conv1d_v1_launch<7, 5, 2, 2, 1, 4, true>},
    {11, 1, TILE_64x256, 1, conv1d_v1_launch<11, 1, 2, 2, 1, 4, true>},
    {11, 3, TILE_64x256, 1, conv1d_v1_launch<11, 3, 2, 2, 1, 4, true>},
    {11, 5, TILE_64x256, 1, conv1d_v1_launch<11, 5, 2, 2, 1, 4, true>},
};
const int kV1NumVariantsB = sizeof(kV1VariantsB) / sizeof(kV1VariantsB[0]);
#endif
}  // namespace v1
}  // namespace ovk
