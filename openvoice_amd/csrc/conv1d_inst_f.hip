// MRF (ResBlock) convs with C >= 128 and the grouped ConvTranspose, tile 128x128, 32-channel chunks, FOUR loader waves
// -- one per SIMD, so the four matrix waves of a workgroup (which re-synchronise at every chunk barrier) all share
// their SIMD with the same company.  Measured against the two-loader instances of conv1d_inst_a.hip (same process,
// profiles/r02_s17_convs_loaders_2_vs_4.txt): k = 3 +2...5 %, k = 7 +0...1.5 %, k = 11 +-0.3 % -> the dispatcher's
// first choice for this tile (ov_conv1d_params.loaders = 2 selects the others).
#include "conv1d_mfma.h"
namespace ovk {
#define LIST(X) \
  X(3, 1, 128x128, 32, 1, OV_EPI_LINEAR, 4) \
  X(3, 3, 128x128, 32, 1, OV_EPI_LINEAR, 4) \
  X(3, 5, 128x128, 32, 1, OV_EPI_LINEAR, 4) \
  X(7, 1, 128x128, 32, 1, OV_EPI_LINEAR, 4) \
  X(7, 3, 128x128, 32, 1, OV_EPI_LINEAR, 4) \
  X(7, 5, 128x128, 32, 1, OV_EPI_LINEAR, 4) \
  X(11, 1, 128x128, 32, 1, OV_EPI_LINEAR, 4) \
  X(11, 3, 128x128, 32, 1, OV_EPI_LINEAR, 4) \
  X(11, 5, 128x128, 32, 1, OV_EPI_LINEAR, 4) \
  X(3, 1, 128x128, 32, 1, EPI_CONVT_S8, 4) \
  X(3, 1, 128x128, 32, 1, EPI_CONVT_S2, 4)
OV_DEFINE_VARIANTS(kVariantsF, LIST)
}  // namespace ovk
