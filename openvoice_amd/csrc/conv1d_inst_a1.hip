// Tile 128x128 (2x2 matrix waves, 64x64 per wave), 16-byte staging, 1 loader wave: the MRF
// (ResBlock) convs with C >= 128.
#include "conv1d_mfma.h"
namespace ovk {
#define LIST(X) \
  X(3, 1, 128x128, 16, 1, OV_EPI_LINEAR, 1) \
  X(3, 3, 128x128, 16, 1, OV_EPI_LINEAR, 1) \
  X(3, 5, 128x128, 16, 1, OV_EPI_LINEAR, 1) \
  X(7, 1, 128x128, 16, 1, OV_EPI_LINEAR, 1) \
  X(7, 3, 128x128, 16, 1, OV_EPI_LINEAR, 1) \
  X(7, 5, 128x128, 16, 1, OV_EPI_LINEAR, 1) \
  X(11, 1, 128x128, 16, 1, OV_EPI_LINEAR, 1) \
  X(11, 3, 128x128, 16, 1, OV_EPI_LINEAR, 1) \
  X(11, 5, 128x128, 16, 1, OV_EPI_LINEAR, 1)
OV_DEFINE_VARIANTS(kVariantsA1, LIST)
}  // namespace ovk
