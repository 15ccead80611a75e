// Tile 128x128, WaveNet / coupling / posterior epilogues (16-byte and 4-byte staging).
#include "conv1d_mfma.h"
namespace ovk {
#define LIST(X) \
  X(5, 1, 128x128, 16, 1, OV_EPI_GATE, 2) \
  X(5, 1, 128x128, 32, 1, OV_EPI_GATE, 2) \
  X(1, 1, 128x128, 32, 1, OV_EPI_RESSKIP, 4) \
  X(1, 1, 128x128, 32, 1, OV_EPI_COUPLE, 4) \
  X(1, 1, 128x128, 32, 1, OV_EPI_POSTERIOR, 4) \
  X(5, 1, 128x128, 16, 0, OV_EPI_GATE, 2) \
  X(5, 1, 128x128, 32, 0, OV_EPI_GATE, 2) \
  X(1, 1, 128x128, 32, 0, OV_EPI_RESSKIP, 4) \
  X(1, 1, 128x128, 32, 0, OV_EPI_COUPLE, 4) \
  X(1, 1, 128x128, 32, 0, OV_EPI_POSTERIOR, 4)
OV_DEFINE_VARIANTS(kVariantsW, LIST)
}  // namespace ovk
