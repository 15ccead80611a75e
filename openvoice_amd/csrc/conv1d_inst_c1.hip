// Tile 32x512 (1x4 matrix waves, 32x128 per wave): MRF stage 3 (C = 32); 1 and 2 loader waves.
#include "conv1d_mfma.h"
namespace ovk {
#define LIST(X) \
  X(3, 1, 32x512, 16, 1, OV_EPI_LINEAR, 1) \
  X(3, 3, 32x512, 16, 1, OV_EPI_LINEAR, 1) \
  X(3, 5, 32x512, 16, 1, OV_EPI_LINEAR, 1) \
  X(7, 1, 32x512, 16, 1, OV_EPI_LINEAR, 1) \
  X(7, 3, 32x512, 16, 1, OV_EPI_LINEAR, 1) \
  X(7, 5, 32x512, 16, 1, OV_EPI_LINEAR, 1) \
  X(11, 1, 32x512, 16, 1, OV_EPI_LINEAR, 1) \
  X(11, 3, 32x512, 16, 1, OV_EPI_LINEAR, 1) \
  X(11, 5, 32x512, 16, 1, OV_EPI_LINEAR, 1) \
  X(3, 1, 32x512, 16, 1, OV_EPI_LINEAR, 2) \
  X(3, 3, 32x512, 16, 1, OV_EPI_LINEAR, 2) \
  X(3, 5, 32x512, 16, 1, OV_EPI_LINEAR, 2) \
  X(7, 1, 32x512, 16, 1, OV_EPI_LINEAR, 2) \
  X(7, 3, 32x512, 16, 1, OV_EPI_LINEAR, 2) \
  X(7, 5, 32x512, 16, 1, OV_EPI_LINEAR, 2) \
  X(11, 1, 32x512, 16, 1, OV_EPI_LINEAR, 2) \
  X(11, 3, 32x512, 16, 1, OV_EPI_LINEAR, 2) \
  X(11, 5, 32x512, 16, 1, OV_EPI_LINEAR, 2)
OV_DEFINE_VARIANTS(kVariantsC1, LIST)
}  // namespace ovk
