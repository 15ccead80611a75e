// Tile 128x128, 4-byte staging for rows that are not 16-byte aligned (frame-rate tensors with
// T % 4 != 0: 1x1 convs, conv_pre k7, ups.0 as a 3-tap conv).
#include "conv1d_mfma.h"
namespace ovk {
// explicit kernel instantiations (both host and device passes see these)
template __global__ void conv1d_mfma_kernel<1, 1, 2, 2, 2, 2, 32, false, OV_EPI_LINEAR>(const ov_conv1d_params, const int);
template __global__ void conv1d_mfma_kernel<3, 1, 2, 2, 2, 2, 16, false, OV_EPI_LINEAR>(const ov_conv1d_params, const int);
template __global__ void conv1d_mfma_kernel<5, 1, 2, 2, 2, 2, 16, false, OV_EPI_LINEAR>(const ov_conv1d_params, const int);
template __global__ void conv1d_mfma_kernel<7, 1, 2, 2, 2, 2, 16, false, OV_EPI_LINEAR>(const ov_conv1d_params, const int);
template __global__ void conv1d_mfma_kernel<3, 1, 2, 2, 2, 2, 16, false, OV_EPI_CONVT>(const ov_conv1d_params, const int);
#if !defined(__HIP_DEVICE_COMPILE__)
const ConvVariant kVariantsS[] = {
    {1, 1, TILE_128x128, 0, OV_EPI_LINEAR, conv1d_launch<1, 1, 2, 2, 2, 2, 32, false, OV_EPI_LINEAR>},
    {3, 1, TILE_128x128, 0, OV_EPI_LINEAR, conv1d_launch<3, 1, 2, 2, 2, 2, 16, false, OV_EPI_LINEAR>},
    {5, 1, TILE_128x128, 0, OV_EPI_LINEAR, conv1d_launch<5, 1, 2, 2, 2, 2, 16, false, OV_EPI_LINEAR>},
    {7, 1, TILE_128x128, 0, OV_EPI_LINEAR, conv1d_launch<7, 1, 2, 2, 2, 2, 16, false, OV_EPI_LINEAR>},
    {3, 1, TILE_128x128, 0, OV_EPI_CONVT, conv1d_launch<3, 1, 2, 2, 2, 2, 16, false, OV_EPI_CONVT>},
};
const int kNumVariantsS = sizeof(kVariantsS) / sizeof(kVariantsS[0]);
#endif
}  // namespace ovk
