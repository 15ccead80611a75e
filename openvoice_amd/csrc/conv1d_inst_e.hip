// Alternates kept for A/B measurement: tile 64x256 with 16-channel chunks, tile 32x256.
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
  X(11, 5, 64x256, 16, 1, OV_EPI_LINEAR, 4) \
  X(3, 1, 32x256, 16, 1, OV_EPI_LINEAR, 4) \
  X(3, 3, 32x256, 16, 1, OV_EPI_LINEAR, 4) \
  X(3, 5, 32x256, 16, 1, OV_EPI_LINEAR, 4) \
  X(7, 1, 32x256, 16, 1, OV_EPI_LINEAR, 4) \
  X(7, 3, 32x256, 16, 1, OV_EPI_LINEAR, 4) \
  X(7, 5, 32x256, 16, 1, OV_EPI_LINEAR, 4) \
  X(11, 1, 32x256, 16, 1, OV_EPI_LINEAR, 4) \
  X(11, 3, 32x256, 16, 1, OV_EPI_LINEAR, 4) \
  X(11, 5, 32x256, 16, 1, OV_EPI_LINEAR, 4)
OV_DEFINE_VARIANTS(kVariantsE, LIST)
}  // namespace ovk
