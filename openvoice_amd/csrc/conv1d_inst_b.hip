// MRF convs with C = 64: tile 64x256 (1x4 matrix waves, 64x64 per wave), 32-channel chunks, 4 loader waves.
#include "conv1d_mfma.h"
namespace ovk {
#define LIST(X) \
  X(3, 1, 64x256, 32, 1, OV_EPI_LINEAR, 4) \
  X(3, 3, 64x256, 32, 1, OV_EPI_LINEAR, 4) \
  X(3, 5, 64x256, 32, 1, OV_EPI_LINEAR, 4) \
  X(7, 1, 64x256, 32, 1, OV_EPI_LINEAR, 4) \
  X(7, 3, 64x256, 32, 1, OV_EPI_LINEAR, 4) \
  X(7, 5, 64x256, 32, 1, OV_EPI_LINEAR, 4) \
  X(11, 1, 64x256, 32, 1, OV_EPI_LINEAR, 4) \
  X(11, 3, 64x256, 32, 1, OV_EPI_LINEAR, 4) \
  X(11, 5, 64x256, 32, 1, OV_EPI_LINEAR, 4)
OV_DEFINE_VARIANTS(kVariantsB, LIST)
}  // namespace ovk
