// MRF convs with C = 32: tile 32x512 (1x4 matrix waves, 32x128 per wave), 16-channel chunks (two per
// tile, 66 KB LDS, 2 workgroups per CU), 4 loader waves.
#include "conv1d_mfma.h"
namespace ovk {
#define LIST(X) \
  X(3, 1, 32x512, 16, 1, OV_EPI_LINEAR, 4) \
  X(3, 3, 32x512, 16, 1, OV_EPI_LINEAR, 4) \
  X(3, 5, 32x512, 16, 1, OV_EPI_LINEAR, 4) \
  X(7, 1, 32x512, 16, 1, OV_EPI_LINEAR, 4) \
  X(7, 3, 32x512, 16, 1, OV_EPI_LINEAR, 4) \
  X(7, 5, 32x512, 16, 1, OV_EPI_LINEAR, 4) \
  X(11, 1, 32x512, 16, 1, OV_EPI_LINEAR, 4) \
  X(11, 3, 32x512, 16, 1, OV_EPI_LINEAR, 4) \
  X(11, 5, 32x512, 16, 1, OV_EPI_LINEAR, 4)
OV_DEFINE_VARIANTS(kVariantsC, LIST)
}  // namespace ovk
