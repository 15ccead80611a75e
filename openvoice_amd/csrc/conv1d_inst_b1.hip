// Tile 64x256 (1x4 matrix waves, 64x64 per wave): MRF stage 2 (C = 64); 1 and 2 loader waves.
#include "conv1d_mfma.h"
namespace ovk {
#define LIST(X) \
  X(3, 1, 64x256, 16, 1, OV_EPI_LINEAR, 1) \
  X(3, 3, 64x256, 16, 1, OV_EPI_LINEAR, 1) \
  X(3, 5, 64x256, 16, 1, OV_EPI_LINEAR, 1) \
  X(7, 1, 64x256, 16, 1, OV_EPI_LINEAR, 1) \
  X(7, 3, 64x256, 16, 1, OV_EPI_LINEAR, 1) \
  X(7, 5, 64x256, 16, 1, OV_EPI_LINEAR, 1) \
  X(11, 1, 64x256, 16, 1, OV_EPI_LINEAR, 1) \
  X(11, 3, 64x256, 16, 1, OV_EPI_LINEAR, 1) \
  X(11, 5, 64x256, 16, 1, OV_EPI_LINEAR, 1) \
  X(3, 1, 64x256, 16, 1, OV_EPI_LINEAR, 2) \
  X(3, 3, 64x256, 16, 1, OV_EPI_LINEAR, 2) \
  X(3, 5, 64x256, 16, 1, OV_EPI_LINEAR, 2) \
  X(7, 1, 64x256, 16, 1, OV_EPI_LINEAR, 2) \
  X(7, 3, 64x256, 16, 1, OV_EPI_LINEAR, 2) \
  X(7, 5, 64x256, 16, 1, OV_EPI_LINEAR, 2) \
  X(11, 1, 64x256, 16, 1, OV_EPI_LINEAR, 2) \
  X(11, 3, 64x256, 16, 1, OV_EPI_LINEAR, 2) \
  X(11, 5, 64x256, 16, 1, OV_EPI_LINEAR, 2)
OV_DEFINE_VARIANTS(kVariantsB1, LIST)
}  // namespace ovk
