// Second-generation fused bf16 ResBlock pair, kernel size 3: explicit instantiations (see conv1d_bf16_pair2.h).
#include "conv1d_bf16_pair2.h"
namespace ovk16q {
int pair2_launch_k3(const ov_respair2_bf16_params* p, hipStream_t stream) { return pair2_launch_by_dilation<3>(p, stream); }
}  // namespace ovk16q
