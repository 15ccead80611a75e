// Tile 128x128 (2x2 matrix waves, 64x64 per wave), 16-byte staging: ResBlock stages with C >= 128,
// ups 1-2, and the frame-rate layers when T % 4 == 0.
#include "conv1d_mfma.h"
namespace ovk {
// explicit kernel instantiations (both host and device passes see these)
template __global__ void conv1d_mfma_kernel<3, 1, 2, 2, 2, 2, 16, true, OV_EPI_LINEAR>(const ov_conv1d_params, const int);
template __global__ void conv1d_mfma_kernel<3, 3, 2, 2, 2, 2, 16, true, OV_EPI_LINEAR>(const ov_conv1d_params, const int);
template __global__ void conv1d_mfma_kernel<3, 5, 2, 2, 2, 2, 16, true, OV_EPI_LINEAR>(const ov_conv1d_params, const int);
template __global__ void conv1d_mfma_kernel<7, 1, 2, 2, 2, 2, 16, true, OV_EPI_LINEAR>(const ov_conv1d_params, const int);
template __global__ void conv1d_mfma_kernel<7, 3, 2, 2, 2, 2, 16, true, OV_EPI_LINEAR>(const ov_conv1d_params, const int);
template __global__ void conv1d_mfma_kernel<7, 5, 2, 2, 2, 2, 16, true, OV_EPI_LINEAR>(const ov_conv1d_params, const int);
template __global__ void conv1d_mfma_kernel<11, 1, 2, 2, 2, 2, 16, true, OV_EPI_LINEAR>(const ov_conv1d_params, const int);
template __global__ void conv1d_mfma_kernel<11, 3, 2, 2, 2, 2, 16, true, OV_EPI_LINEAR>(const ov_conv1d_params, const int);
template __global__ void conv1d_mfma_kernel<11, 5, 2, 2, 2, 2, 16, true, OV_EPI_LINEAR>(const ov_conv1d_params, const int);
template __global__ void conv1d_mfma_kernel<1, 1, 2, 2, 2, 2, 32, true, OV_EPI_LINEAR>(const ov_conv1d_params, const int);
template __global__ void conv1d_mfma_kernel<5, 1, 2, 2, 2, 2, 16, true, OV_EPI_LINEAR>(const ov_conv1d_params, const int);
template __global__ void conv1d_mfma_kernel<3, 1, 2, 2, 2, 2, 16, true, OV_EPI_CONVT>(const ov_conv1d_params, const int);
#if !defined(__HIP_DEVICE_COMPILE__)
const ConvVariant kVariantsA[] = {
    {3, 1, TILE_128x128, 1, OV_EPI_LINEAR, conv1d_launch<3, 1, 2, 2, 2, 2, 16, true, OV_EPI_LINEAR>},
    {3, 3, TILE_128x128, 1, OV_EPI_LINEAR, conv1d_launch<3, 3, 2, 2, 2, 2, 16, true, OV_EPI_LINEAR>},
    {3, 5, TILE_128x128, 1, OV_EPI_LINEAR, conv1d_launch<3, 5, 2, 2, 2, 2, 16, true, OV_EPI_LINEAR>},
    {7, 1, TILE_128x128, 1, OV_EPI_LINEAR, conv1d_launch<7, 1, 2, 2, 2, 2, 16, true, OV_EPI_LINEAR>},
    {7, 3, TILE_128x128, 1, OV_EPI_LINEAR, conv1d_launch<7, 3, 2, 2, 2, 2, 16, true, OV_EPI_LINEAR>},
    {7, 5, TILE_128x128, 1, OV_EPI_LINEAR, conv1d_launch<7, 5, 2, 2, 2, 2, 16, true, OV_EPI_LINEAR>},
    {11, 1, TILE_128x128, 1, OV_EPI_LINEAR, conv1d_launch<11, 1, 2, 2, 2, 2, 16, true, OV_EPI_LINEAR>},
    {11, 3, TILE_128x128, 1, OV_EPI_LINEAR, conv1d_launch<11, 3, 2, 2, 2, 2, 16, true, OV_EPI_LINEAR>},
    {11, 5, TILE_128x128, 1, OV_EPI_LINEAR, conv1d_launch<11, 5, 2, 2, 2, 2, 16, true, OV_EPI_LINEAR>},
    {1, 1, TILE_128x128, 1, OV_EPI_LINEAR, conv1d_launch<1, 1, 2, 2, 2, 2, 32, true, OV_EPI_LINEAR>},
    {5, 1, TILE_128x128, 1, OV_EPI_LINEAR, conv1d_launch<5, 1, 2, 2, 2, 2, 16, true, OV_EPI_LINEAR>},
    {3, 1, TILE_128x128, 1, OV_EPI_CONVT, conv1d_launch<3, 1, 2, 2, 2, 2, 16, true, OV_EPI_CONVT>},
};
const int kNumVariantsA = sizeof(kVariantsA) / sizeof(kVariantsA[0]);
#endif
}  // namespace ovk
