// C ABI of the Winograd-domain fp32 Conv1d (kernel: conv1d_wino.h): weight packer, dispatcher, entry points.
#include "conv1d_wino.h"

#include <cstring>

using namespace ovkw;

namespace {

// F(4, 3), points 0, 1, -1, 2, -2, infinity: the weight transform G (6 x 3).  Bt and At live in the kernel.
const double kG[6][3] = {{1.0 / 4, 0, 0},           {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                         {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6},  {0, 0, 1}};

}  // namespace

extern "C" {

int ov_conv1d_wino_chunk(int K, int Cout) { return Cout > 0 && wino_mw(Cout) ? wino_ci(K, wino_mw(Cout)) : 0; }

int ov_conv1d_wino_supported(int Cin, int Cout, int K, int dil) {
  const int ci = ov_conv1d_wino_chunk(K, Cout);
  if (ci == 0 || (dil != 1 && dil != 3 && dil != 5)) return 0;
  return (Cin > 0 && Cin % ci == 0 && Cout <= MAX_COUT) ? 1 : 0;
}

size_t ov_conv1d_wino_pack_size(int Cout, int Cin, int K) {
  const int ci = ov_conv1d_wino_chunk(K, Cout);
  if (ci == 0 || Cin <= 0 || Cin % ci != 0) return 0;
  return wino_pack_floats(Cout, Cin, K, ci);
}

int ov_conv1d_wino_pack_f32(const float* w, int Cout, int Cin, int K, float* dst) {
  const size_t nfl = ov_conv1d_wino_pack_size(Cout, Cin, K);
  if (!w || !dst || nfl == 0) return OV_E_BADARG;
  const int CI = ov_conv1d_wino_chunk(K, Cout), G = (K + 2) / 3, KR = CI * G, NPAIR = KR / 4, nchunks = Cin / CI;
  std::memset(dst, 0, nfl * sizeof(float));
  for (int mt = 0; mt < Cout / 32; ++mt)
    for (int c = 0; c < nchunks; ++c)
      for (int sp = 0; sp < NPAIR; ++sp)
        for (int lane = 0; lane < 64; ++lane)
          for (int e = 0; e < 12; ++e) {
            const int kstep = 2 * sp + e / 6, pt = e % 6;
            const int kk = 2 * kstep + (lane >> 5);
            const int g = kk / CI, ci = c * CI + kk % CI;
            const int co = 32 * mt + (lane & 31);
            double u = 0.0;
            for (int k = 0; k < 3; ++k) {
              const int tap = 3 * g + k - wino_lead(K);   // (zero taps pad the K real ones on both sides: conv1d_wino.h)
              if (tap >= 0 && tap < K) u += kG[pt][k] * (double)w[((size_t)co * Cin + ci) * K + tap];
            }
            const size_t sub = (((size_t)mt * nchunks + c) * NPAIR + sp) * 3 + e / 4;
            dst[sub * REC + (size_t)lane * 4 + e % 4] = (float)u;
          }
  return OV_OK;
}

int ov_conv1d_wino_f32(const ov_conv1d_wino_params* pin, ov_stream_t stream) {
  if (!pin) return OV_E_BADARG;
  ov_conv1d_wino_params q = *pin;
  if (!q.x || !q.w || !q.bias || !q.out) return OV_E_BADARG;
  if (q.B <= 0 || q.L <= 0 || q.nwg < 0) return OV_E_BADARG;
  if (q.col_limit && (q.col_limit_scale <= 0 || (reinterpret_cast<uintptr_t>(q.col_limit) & 3))) return OV_E_BADARG;
  if (q.col_limit && q.B > ovk::LIMIT_MAX_BATCH) q.col_limit = nullptr;   // documented: whole tensors
  if (q.x_ld == 0) q.x_ld = q.L;
  if (q.out_ld == 0) q.out_ld = q.L;
  if (q.x_ld < q.L || q.out_ld < q.L) return OV_E_BADARG;
  if (!ov_conv1d_wino_supported(q.Cin, q.Cout, q.K, q.dil)) return OV_E_UNSUPPORTED;
  if (!(q.in_slope > 0.f && q.in_slope <= 1.f)) return OV_E_UNSUPPORTED;
  if (q.out_slope == 0.f) q.out_slope = 1.f;
  if (!(q.out_slope > 0.f && q.out_slope <= 1.f)) return OV_E_UNSUPPORTED;
  if (q.out_slope != 1.f && (q.res || q.add)) return OV_E_UNSUPPORTED;   // the activated hand-over is the first conv's
  if (q.out == q.x) return OV_E_BADARG;   // (out may BE res or add: a lane reads exactly the 16 bytes it then writes)
  auto mis = [](const void* ptr) { return (reinterpret_cast<uintptr_t>(ptr) & 15) != 0; };
  if (mis(q.x) || mis(q.w) || mis(q.out) || (q.res && mis(q.res)) || (q.add && mis(q.add)) || (q.L & 3) || (q.x_ld & 3) ||
      (q.out_ld & 3) || (q.x_bstride & 3) || (q.out_bstride & 3) || (q.res && (q.res_bstride & 3)) ||
      (q.add && (q.add_bstride & 3)))
    return OV_E_ALIGN;
  // 32-bit element offsets inside one utterance
  if ((int64_t)q.Cin * q.x_ld >= (1LL << 31) || (int64_t)q.Cout * q.out_ld >= (1LL << 31)) return OV_E_BADARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (q.frags < 0 || q.frags > 2) return OV_E_BADARG;
  const int nf = q.frags ? q.frags : 2;
  if (q.dil != 1 && nf != 2) return OV_E_UNSUPPORTED;       // the dilated instances run two fragments per wave
  switch (q.K) {
    case 3: return wino_dispatch_k3(&q, nf, st);
    case 7: return wino_dispatch_k7(&q, nf, st);
    case 11: return wino_dispatch_k11(&q, nf, st);
  }
  return OV_E_UNSUPPORTED;
}

}  // extern "C"
