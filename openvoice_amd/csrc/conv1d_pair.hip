// Fused ResBlock1 pair (conv1d_pair.h): instances for the HBM-bound generator stages + the C ABI entry points.
//   C = 32: every (K, dilation) of the MRF, NT = 256 columns per step, 16-channel chunks
//   C = 64: NT = 128 columns per step, 32-channel chunks
#include "conv1d_pair.h"

namespace ovk {

#define OV_PAIR32(K, D) template __global__ void respair_mfma_kernel<K, D, 32, 256, 16, 4>(const ov_respair_params);
#define OV_PAIR64(K, D) template __global__ void respair_mfma_kernel<K, D, 64, 128, 32, 4>(const ov_respair_params);
#define OV_PAIR_LIST32(X) X(3, 1) X(3, 3) X(3, 5) X(7, 1) X(7, 3) X(7, 5) X(11, 1) X(11, 3) X(11, 5)
#define OV_PAIR_LIST64(X) X(3, 1) X(3, 3) X(3, 5) X(7, 1) X(7, 3) X(7, 5)
OV_PAIR_LIST32(OV_PAIR32)
OV_PAIR_LIST64(OV_PAIR64)

#if !defined(__HIP_DEVICE_COMPILE__)
#define OV_ROW32(K, D) {K, D, 32, respair_launch<K, D, 32, 256, 16, 4>},
#define OV_ROW64(K, D) {K, D, 64, respair_launch<K, D, 64, 128, 32, 4>},
const PairVariant kPairVariants[] = {OV_PAIR_LIST32(OV_ROW32) OV_PAIR_LIST64(OV_ROW64)};
const int kPairVariantsCount = sizeof(kPairVariants) / sizeof(kPairVariants[0]);
#endif

}  // namespace ovk

using namespace ovk;

#if !defined(__HIP_DEVICE_COMPILE__)
static pair_launch_fn find_pair(int C, int K, int dil) {
  for (int i = 0; i < kPairVariantsCount; ++i)
    if (kPairVariants[i].C == C && kPairVariants[i].K == K && kPairVariants[i].dil == dil) return kPairVariants[i].fn;
  return nullptr;
}

extern "C" {

int ov_resblock_pair_supported(int C, int K, int dil) { return find_pair(C, K, dil) ? 1 : 0; }

int ov_resblock_pair_f32(const ov_respair_params* pin, ov_stream_t stream) {
  if (!pin || !pin->x || !pin->w1 || !pin->b1 || !pin->w2 || !pin->b2 || !pin->out) return OV_E_BADARG;
  ov_respair_params q = *pin;
  if (q.B <= 0 || q.C <= 0 || q.L <= 0 || q.K <= 0 || q.dil <= 0 || q.nwg < 0) return OV_E_BADARG;
  if (q.ld == 0) q.ld = q.L;
  if (q.ld < q.L) return OV_E_BADARG;
  if (q.out == q.x) return OV_E_BADARG;   // (add == out is fine: read and written at the same positions by one lane)
  if ((int64_t)q.C * q.ld > UINT32_MAX) return OV_E_BADARG;   // per-utterance offsets are 32-bit inside the kernel
  if ((int64_t)q.B * ((q.L + q.K) / 128 + 2) > INT32_MAX) return OV_E_BADARG;
  if ((q.ld % 4) || (q.x_bstride % 4) || (reinterpret_cast<uintptr_t>(q.x) & 15) ||
      (reinterpret_cast<uintptr_t>(q.w1) & 15) || (reinterpret_cast<uintptr_t>(q.w2) & 15))
    return OV_E_ALIGN;
  if (q.col_limit && (q.col_limit_scale <= 0 || q.B > LIMIT_MAX_BATCH || (reinterpret_cast<uintptr_t>(q.col_limit) & 3)))
    return OV_E_BADARG;
  pair_launch_fn fn = find_pair(q.C, q.K, q.dil);
  if (!fn) return OV_E_UNSUPPORTED;
  return fn(&q, static_cast<hipStream_t>(stream));
}

}  // extern "C"
#endif
