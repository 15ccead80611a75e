// Alternates kept for A/B measurement (tools/bench_convs.py) and as fallbacks: tile 128x128 with
// 16-channel chunks.
#include "conv1d_mfma.h"
namespace ovk {
#define LIST(X) \
  X(3, 1, 128x128, 16, 1, OV_EPI_LINEAR, 2) \
  X(3, 3, 128x128, 16, 1, OV_EPI_LINEAR, 2) \
  X(3, 5, 128x128, 16, 1, OV_EPI_LINEAR, 2) \
  X(7, 1, 128x128, 16, 1, OV_EPI_LINEAR, 2) \
  X(7, 3, 128x128, 16, 1, OV_EPI_LINEAR, 2) \
  X(7, 5, 128x128, 16, 1, OV_EPI_LINEAR, 2) \
  X(11, 1, 128x128, 16, 1, OV_EPI_LINEAR, 2) \
  X(11, 3, 128x128, 16, 1, OV_EPI_LINEAR, 2) \
  X(11, 5, 128x128, 16, 1, OV_EPI_LINEAR, 2)
OV_DEFINE_VARIANTS(kVariantsD, LIST)
}  // namespace ovk
