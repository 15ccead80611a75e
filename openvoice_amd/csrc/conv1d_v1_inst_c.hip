// Tile 32x512 (1x4 waves, 32x128 per wave): 32-row convs (ResBlock stage 3).
#include "conv1d_mfma_v1.h"
namespace ovk {
namespace v1 {
// explicit kernel instantiations (both host and device passes see these)
template __global__ void conv1d_mfma_v1_kernel<3, 1, 1, 4, 1, 4, true>(const ov_conv1d_params);
template __global__ void conv1d_mfma_v1_kernel<3, 3, 1, 4, 1, 4, true>(const ov_conv1d_params);
template __global__ void conv1d_mfma_v1_kernel<3, 5, 1, 4, 1, 4, true>(const ov_conv1d_params);
template __global__ void conv1d_mfma_v1_kernel<7, 1, 1, 4, 1, 4, true>(const ov_conv1d_params);
template __global__ void conv1d_mfma_v1_kernel<7, 3, 1, 4, 1, 4, true>(const ov_conv1d_params);
template __global__ void conv1d_mfma_v1_kernel<7, 5, 1, 4, 1, 4, true>(const ov_conv1d_params);
template __global__ void conv1d_mfma_v1_kernel<11, 1, 1, 4, 1, 4, true>(const ov_conv1d_params);
template __global__ void conv1d_mfma_v1_kernel<11, 3, 1, 4, 1, 4, true>(const ov_conv1d_params);
template __global__ void conv1d_mfma_v1_kernel<11, 5, 1, 4, 1, 4, true>(const ov_conv1d_params);
#if !defined(__HIP_DEVICE_COMPILE__)
const ConvVariant kV1VariantsC[] = {
    {3, 1, TILE_32x512, 1, conv1d_v1_launch<3, 1, 1, 4, 1, 4, true>},
    {3, 3, TILE_32x512, 1, conv1d_v1_launch<3, 3, 1, 4, 1, 4, true>},
    {3, 5, TILE_32x512, 1, conv1d_v1_launch<3, 5, 1, 4, 1, 4, true>},
    {7, 1, TILE_32x512, 1, conv1d_v1_launch<7, 1, 1, 4, 1, 4, true>},
    {7, 3, TILE_32x512, 1, conv1d_v1_launch<7, 3, 1, 4, 1, 4, true>},
    {7, 5, TILE_32x512, 1, conv1d_v1_launch<7, 5, 1, 4, 1, 4, true>},
    {11, 1, TILE_32x512, 1, conv1d_v1_launch<11, 1, 1, 4, 1, 4, true>},
    {11, 3, TILE_32x512, 1, conv1d_v1_launch<11, 3, 1, 4, 1, 4, true>},
    {11, 5, TILE_32x512, 1, conv1d_v1_launch<11, 5, 1, 4, 1, 4, true>},
};
const int kV1NumVariantsC = sizeof(kV1VariantsC) / sizeof(kV1VariantsC[0]);
#endif
}  // namespace v1
}  // namespace ovk
