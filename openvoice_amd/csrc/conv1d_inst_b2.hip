// Tile 64x256: MRF stage 2 (C = 64); 4 loader waves.
#include "conv1d_mfma.h"
namespace ovk {
#define LIST(X) \
  X(3, 1, 64x256, 16, 1, OV_EPI_LINEAR, 4) \
  X(3, 3, 64x256, 16, 1, OV_EPI_LINEAR, 4) \
  X(3, 5, 64x256, 16, 1, OV_EPI_LINEAR, 4) \
  X(7, 1, 64x256, 16, 1, OV_EPI_LINEAR, 4) \
  X(7, 3, 64x256, 16, 1, OV_EPI_LINEAR, 4) \
  X(7, 5, 64x256, 16, 1, OV_EPI_LINEAR, 4) \
  X(11, 1, 64x256, 16, 1, OV_EPI_LINEAR, 4) \
  X(11, 3, 64x256, 16, 1, OV_EPI_LINEAR, 4) \
  X(11, 5, 64x256, 16, 1, OV_EPI_LINEAR, 4)
OV_DEFINE_VARIANTS(kVariantsB2, LIST)
}  // namespace ovk
