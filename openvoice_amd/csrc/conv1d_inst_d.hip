// Tile 32x256 (1x4 matrix waves, 32x64 per wave): MRF stage 3 (C = 32) with a smaller LDS
// footprint (4 workgroups per CU); 2 and 4 loader waves.
#include "conv1d_mfma.h"
namespace ovk {
#define LIST(X) \
  X(3, 1, 32x256, 16, 1, OV_EPI_LINEAR, 2) \
  X(3, 3, 32x256, 16, 1, OV_EPI_LINEAR, 2) \
  X(3, 5, 32x256, 16, 1, OV_EPI_LINEAR, 2) \
  X(7, 1, 32x256, 16, 1, OV_EPI_LINEAR, 2) \
  X(7, 3, 32x256, 16, 1, OV_EPI_LINEAR, 2) \
  X(7, 5, 32x256, 16, 1, OV_EPI_LINEAR, 2) \
  X(11, 1, 32x256, 16, 1, OV_EPI_LINEAR, 2) \
  X(11, 3, 32x256, 16, 1, OV_EPI_LINEAR, 2) \
  X(11, 5, 32x256, 16, 1, OV_EPI_LINEAR, 2) \
  X(3, 1, 32x256, 16, 1, OV_EPI_LINEAR, 4) \
  X(3, 3, 32x256, 16, 1, OV_EPI_LINEAR, 4) \
  X(3, 5, 32x256, 16, 1, OV_EPI_LINEAR, 4) \
  X(7, 1, 32x256, 16, 1, OV_EPI_LINEAR, 4) \
  X(7, 3, 32x256, 16, 1, OV_EPI_LINEAR, 4) \
  X(7, 5, 32x256, 16, 1, OV_EPI_LINEAR, 4) \
  X(11, 1, 32x256, 16, 1, OV_EPI_LINEAR, 4) \
  X(11, 3, 32x256, 16, 1, OV_EPI_LINEAR, 4) \
  X(11, 5, 32x256, 16, 1, OV_EPI_LINEAR, 4)
OV_DEFINE_VARIANTS(kVariantsD, LIST)
}  // namespace ovk
