// Split-precision (3 bf16 planes) Conv1d, kernel size 3, dilation 1: explicit instantiations (see conv1d_split3.h).
#include "conv1d_split3.h"
namespace ovks3 {
int split3_launch_k3d1(const ov_conv1d_split3_params* p, hipStream_t stream) { return launch_by_width<3, 1>(p, stream); }
}  // namespace ovks3
