// k = 3 MRF convs with 32-channel chunks (half as many chunk hand-offs as the 16-channel default).
#include "conv1d_mfma.h"
namespace ovk {
#define LIST(X) \
  X(3, 1, 128x128, 32, 1, OV_EPI_LINEAR, 2) \
  X(3, 3, 128x128, 32, 1, OV_EPI_LINEAR, 2) \
  X(3, 5, 128x128, 32, 1, OV_EPI_LINEAR, 2) \
  X(3, 1, 64x256, 32, 1, OV_EPI_LINEAR, 4) \
  X(3, 3, 64x256, 32, 1, OV_EPI_LINEAR, 4) \
  X(3, 5, 64x256, 32, 1, OV_EPI_LINEAR, 4)
OV_DEFINE_VARIANTS(kVariantsE, LIST)
}  // namespace ovk
