// Split-precision (3 bf16 planes) Conv1d, kernel size 11, dilation 1: explicit instantiations (see conv1d_split3.h).
#include "conv1d_split3.h"
namespace ovks3 {
int split3_launch_k11d1(const ov_conv1d_split3_params* p, hipStream_t stream) { return launch_by_width<11, 1>(p, stream); }
}  // namespace ovks3
