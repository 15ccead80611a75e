// C ABI of the second-generation fused bf16 ResBlock pair (kernel: conv1d_bf16_pair2.h; instantiated per kernel size in
// conv1d_bf16_pair2_k3 / k7 / k11.hip).
#include "conv1d_bf16_pair2.h"

using namespace ovk16q;
extern "C" {

int ov_resblock_pair2_bf16_supported(int C, int K, int dil) {
  const bool kd = (K == 3 || K == 7 || K == 11) && (dil == 1 || dil == 3 || dil == 5);
  return kd && (C == 32 || C == 64 || C == 128) ? 1 : 0;
}

int ov_resblock_pair2_bf16cl(const ov_respair2_bf16_params* p, ov_stream_t stream) {
  if (!p || !p->x || !p->w1 || !p->b1 || !p->w2 || !p->b2 || !p->out) return OV_E_BADARG;
  if (p->B <= 0 || p->L <= 0 || p->C <= 0 || p->nwg < 0) return OV_E_BADARG;
  if (p->out == p->x) return OV_E_BADARG;
  if (!(p->slope > 0.f && p->slope <= 1.f) || p->out_slope < 0.f || p->out_slope > 1.f) return OV_E_UNSUPPORTED;
  if (!ov_resblock_pair2_bf16_supported(p->C, p->K, p->dil)) return OV_E_UNSUPPORTED;
  // without `add` the epilogue has one copy per case: activated (scale 1), scaled (raw), plain
  if (!p->add && p->out_slope != 0.f && p->out_slope != 1.f && p->scale != 1.f) return OV_E_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(p->x) & 15) || (reinterpret_cast<uintptr_t>(p->w1) & 15) ||
      (reinterpret_cast<uintptr_t>(p->w2) & 15) || (reinterpret_cast<uintptr_t>(p->out) & 15) ||
      (p->add && (reinterpret_cast<uintptr_t>(p->add) & 15)))
    return OV_E_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (p->K == 3) return pair2_launch_k3(p, st);
  if (p->K == 7) return pair2_launch_k7(p, st);
  if (p->K == 11) return pair2_launch_k11(p, st);
  return OV_E_UNSUPPORTED;
}

}  // extern "C"
