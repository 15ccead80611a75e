// C ABI of the second-generation fused bf16 ResBlock pair (kernel: conv1d_bf16_pair2.h; instantiated per kernel size in
// conv1d_bf16_pair2_k3 / k7 / k11.hip).
#include "conv1d_bf16_pair2.h"

#include <cstring>

using namespace ovk16q;
extern "C" {

// Weights of the fused pair in 16x16x32 A-fragment order: record ((nt * Cin/32 + c) * K + tap) * 2 + f = 64 lanes x 8
// bf16, lane (l15 = lane & 15, g = lane >> 4) holds W[32 nt + 16 f + l15][32 c + 8 g .. + 8][tap]; one trailing all-zero
// record (the loaders' source for rows outside [0, L)).  Same size as ov_conv1d_bf16_pack's stream.
int ov_conv1d_bf16_pack16(const float* w, int Cout, int Cin, int K, uint16_t* dst) {
  if (!w || !dst || Cout <= 0 || Cin <= 0 || K <= 0 || Cin % 32 != 0) return OV_E_BADARG;
  const int ntiles = (Cout + 31) / 32, nchunks = Cin / 32;
  std::memset(dst, 0, ((size_t)ntiles * nchunks * K * 2 + 1) * 64 * 8 * sizeof(uint16_t));
  auto to_bf16 = [](float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN
    u += 0x7fffu + ((u >> 16) & 1u);                                            // round to nearest even
    return (uint16_t)(u >> 16);
  };
  for (int nt = 0; nt < ntiles; ++nt)
    for (int c = 0; c < nchunks; ++c)
      for (int tap = 0; tap < K; ++tap)
        for (int f = 0; f < 2; ++f) {
          uint16_t* rec = dst + ((((size_t)nt * nchunks + c) * K + tap) * 2 + f) * 64 * 8;
          for (int lane = 0; lane < 64; ++lane)
            for (int i = 0; i < 8; ++i) {
              const int co = 32 * nt + 16 * f + (lane & 15);
              const int ci = 32 * c + 8 * (lane >> 4) + i;
              if (co < Cout) rec[lane * 8 + i] = to_bf16(w[((size_t)co * Cin + ci) * K + tap]);
            }
        }
  return OV_OK;
}

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
