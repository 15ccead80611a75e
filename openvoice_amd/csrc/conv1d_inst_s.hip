// Frame-rate and upsampling layers: 1x1 / k3 / k5 / k7 LINEAR and the 3-tap phase form of
// ConvTranspose1d (CONVT), 16-byte staging and the 4-byte fallback for unaligned rows.
#include "conv1d_mfma.h"
namespace ovk {
#define LIST(X) \
  X(1, 1, 128x128, 32, 1, OV_EPI_LINEAR, 4) \
  X(5, 1, 128x128, 16, 1, OV_EPI_LINEAR, 2) \
  X(3, 1, 128x128, 16, 1, OV_EPI_CONVT, 2) \
  X(3, 1, 128x128, 16, 1, EPI_CONVT_S8, 2) \
  X(3, 1, 128x128, 16, 1, EPI_CONVT_S2, 2) \
  X(1, 1, 128x128, 32, 0, OV_EPI_LINEAR, 4) \
  X(3, 1, 128x128, 16, 0, OV_EPI_LINEAR, 2) \
  X(5, 1, 128x128, 16, 0, OV_EPI_LINEAR, 2) \
  X(7, 1, 128x128, 16, 0, OV_EPI_LINEAR, 2) \
  X(3, 1, 128x128, 16, 0, OV_EPI_CONVT, 2) \
  X(3, 1, 128x128, 16, 0, EPI_CONVT_S8, 2) \
  X(3, 1, 128x128, 16, 0, EPI_CONVT_S2, 2) \
  X(3, 1, 128x128, 32, 1, OV_EPI_CONVT, 2) \
  X(3, 1, 128x128, 32, 1, EPI_CONVT_S8, 2) \
  X(3, 1, 128x128, 32, 1, EPI_CONVT_S2, 2) \
  X(3, 1, 64x256, 16, 1, OV_EPI_CONVT, 4) \
  X(3, 1, 64x256, 16, 1, EPI_CONVT_S8, 4) \
  X(3, 1, 64x256, 16, 1, EPI_CONVT_S2, 4) \
  X(3, 1, 64x256, 32, 1, OV_EPI_CONVT, 4) \
  X(3, 1, 64x256, 32, 1, EPI_CONVT_S8, 4) \
  X(3, 1, 64x256, 32, 1, EPI_CONVT_S2, 4) \
  X(4, 1, 128x128, 32, 1, OV_EPI_MAGNITUDE, 4) \
  X(4, 1, 128x128, 32, 0, OV_EPI_MAGNITUDE, 4)
OV_DEFINE_VARIANTS(kVariantsS, LIST)
}  // namespace ovk
