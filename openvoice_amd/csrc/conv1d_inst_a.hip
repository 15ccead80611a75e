// MRF (ResBlock) convs with C >= 128: tile 128x128 (2x2 matrix waves, 64x64 per wave), 32-channel
// chunks, 2 loader waves -- the dispatcher's choice (profiles/: best of the (tile, chunk, loaders) sweep).
#include "conv1d_mfma.h"
namespace ovk {
#define LIST(X) \
  X(3, 1, 128x128, 32, 1, OV_EPI_LINEAR, 2) \
  X(3, 3, 128x128, 32, 1, OV_EPI_LINEAR, 2) \
  X(3, 5, 128x128, 32, 1, OV_EPI_LINEAR, 2) \
  X(7, 1, 128x128, 32, 1, OV_EPI_LINEAR, 2) \
  X(7, 3, 128x128, 32, 1, OV_EPI_LINEAR, 2) \
  X(7, 5, 128x128, 32, 1, OV_EPI_LINEAR, 2) \
  X(11, 1, 128x128, 32, 1, OV_EPI_LINEAR, 2) \
  X(11, 3, 128x128, 32, 1, OV_EPI_LINEAR, 2) \
  X(11, 5, 128x128, 32, 1, OV_EPI_LINEAR, 2)
OV_DEFINE_VARIANTS(kVariantsA, LIST)
}  // namespace ovk
