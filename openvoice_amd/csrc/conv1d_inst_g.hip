// LDS-DMA staging variants of the MRF (ResBlock) convs: ONE loader wave issuing global_load_lds_dwordx4 (no staging
// registers), leaky-ReLU applied by the matrix waves after the ds_read -> 5-wave workgroups, three per CU instead of
// two (LDS: <= 35 KB per workgroup with these chunk sizes).  Selected with ov_conv1d_params.loaders = -1.
#include "conv1d_mfma.h"
namespace ovk {
#define LIST(X) \
  X(3, 1, 128x128, 32, 2, OV_EPI_LINEAR, 1) \
  X(3, 3, 128x128, 32, 2, OV_EPI_LINEAR, 1) \
  X(3, 5, 128x128, 32, 2, OV_EPI_LINEAR, 1) \
  X(7, 1, 128x128, 32, 2, OV_EPI_LINEAR, 1) \
  X(7, 3, 128x128, 32, 2, OV_EPI_LINEAR, 1) \
  X(7, 5, 128x128, 32, 2, OV_EPI_LINEAR, 1) \
  X(11, 1, 128x128, 32, 2, OV_EPI_LINEAR, 1) \
  X(11, 3, 128x128, 32, 2, OV_EPI_LINEAR, 1) \
  X(11, 5, 128x128, 32, 2, OV_EPI_LINEAR, 1) \
  X(3, 1, 64x256, 16, 2, OV_EPI_LINEAR, 1) \
  X(3, 3, 64x256, 16, 2, OV_EPI_LINEAR, 1) \
  X(3, 5, 64x256, 16, 2, OV_EPI_LINEAR, 1) \
  X(7, 1, 64x256, 16, 2, OV_EPI_LINEAR, 1) \
  X(7, 3, 64x256, 16, 2, OV_EPI_LINEAR, 1) \
  X(7, 5, 64x256, 16, 2, OV_EPI_LINEAR, 1) \
  X(11, 1, 64x256, 16, 2, OV_EPI_LINEAR, 1) \
  X(11, 3, 64x256, 16, 2, OV_EPI_LINEAR, 1) \
  X(11, 5, 64x256, 16, 2, OV_EPI_LINEAR, 1) \
  X(3, 1, 32x256, 16, 2, OV_EPI_LINEAR, 1) \
  X(3, 3, 32x256, 16, 2, OV_EPI_LINEAR, 1) \
  X(3, 5, 32x256, 16, 2, OV_EPI_LINEAR, 1) \
  X(7, 1, 32x256, 16, 2, OV_EPI_LINEAR, 1) \
  X(7, 3, 32x256, 16, 2, OV_EPI_LINEAR, 1) \
  X(7, 5, 32x256, 16, 2, OV_EPI_LINEAR, 1) \
  X(11, 1, 32x256, 16, 2, OV_EPI_LINEAR, 1) \
  X(11, 3, 32x256, 16, 2, OV_EPI_LINEAR, 1) \
  X(11, 5, 32x256, 16, 2, OV_EPI_LINEAR, 1)
OV_DEFINE_VARIANTS(kVariantsG, LIST)
}  // namespace ovk
