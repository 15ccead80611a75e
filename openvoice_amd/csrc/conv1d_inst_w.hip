// Tile 128x128, WaveNet / coupling / posterior epilogues (16-byte and 4-byte staging).
#include "conv1d_mfma.h"
namespace ovk {
// explicit kernel instantiations (both host and device passes see these)
template __global__ void conv1d_mfma_kernel<5, 1, 2, 2, 2, 2, 16, true, OV_EPI_GATE>(const ov_conv1d_params, const int);
template __global__ void conv1d_mfma_kernel<5, 1, 2, 2, 2, 2, 16, false, OV_EPI_GATE>(const ov_conv1d_params, const int);
template __global__ void conv1d_mfma_kernel<1, 1, 2, 2, 2, 2, 32, true, OV_EPI_RESSKIP>(const ov_conv1d_params, const int);
template __global__ void conv1d_mfma_kernel<1, 1, 2, 2, 2, 2, 32, false, OV_EPI_RESSKIP>(const ov_conv1d_params, const int);
template __global__ void conv1d_mfma_kernel<1, 1, 2, 2, 2, 2, 32, true, OV_EPI_COUPLE>(const ov_conv1d_params, const int);
template __global__ void conv1d_mfma_kernel<1, 1, 2, 2, 2, 2, 32, false, OV_EPI_COUPLE>(const ov_conv1d_params, const int);
template __global__ void conv1d_mfma_kernel<1, 1, 2, 2, 2, 2, 32, true, OV_EPI_POSTERIOR>(const ov_conv1d_params, const int);
template __global__ void conv1d_mfma_kernel<1, 1, 2, 2, 2, 2, 32, false, OV_EPI_POSTERIOR>(const ov_conv1d_params, const int);
#if !defined(__HIP_DEVICE_COMPILE__)
const ConvVariant kVariantsW[] = {
    {5, 1, TILE_128x128, 1, OV_EPI_GATE, conv1d_launch<5, 1, 2, 2, 2, 2, 16, true, OV_EPI_GATE>},
    {5, 1, TILE_128x128, 0, OV_EPI_GATE, conv1d_launch<5, 1, 2, 2, 2, 2, 16, false, OV_EPI_GATE>},
    {1, 1, TILE_128x128, 1, OV_EPI_RESSKIP, conv1d_launch<1, 1, 2, 2, 2, 2, 32, true, OV_EPI_RESSKIP>},
    {1, 1, TILE_128x128, 0, OV_EPI_RESSKIP, conv1d_launch<1, 1, 2, 2, 2, 2, 32, false, OV_EPI_RESSKIP>},
    {1, 1, TILE_128x128, 1, OV_EPI_COUPLE, conv1d_launch<1, 1, 2, 2, 2, 2, 32, true, OV_EPI_COUPLE>},
    {1, 1, TILE_128x128, 0, OV_EPI_COUPLE, conv1d_launch<1, 1, 2, 2, 2, 2, 32, false, OV_EPI_COUPLE>},
    {1, 1, TILE_128x128, 1, OV_EPI_POSTERIOR, conv1d_launch<1, 1, 2, 2, 2, 2, 32, true, OV_EPI_POSTERIOR>},
    {1, 1, TILE_128x128, 0, OV_EPI_POSTERIOR, conv1d_launch<1, 1, 2, 2, 2, 2, 32, false, OV_EPI_POSTERIOR>},
};
const int kNumVariantsW = sizeof(kVariantsW) / sizeof(kVariantsW[0]);
#endif
}  // namespace ovk
