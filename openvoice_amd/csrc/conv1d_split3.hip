// C ABI of the split-precision Conv1d (kernel: conv1d_split3.h; instantiated per kernel size in conv1d_split3_k3 / k7 /
// k11.hip) and of the two layout kernels around an MRF stage that runs on it.
#include "conv1d_split3.h"

#include <cstring>


namespace ovks3 {

// fp32 [B][C][L] (time contiguous: the fp32 engine's layout) -> planes [3][B][L][C] of lrelu(x, slope): 32 time columns
// x all C channels per workgroup, transposed through LDS; reads are 128-byte row segments, writes whole 2 C-byte rows.
__global__ __launch_bounds__(256) void split3_from_f32_kernel(const float* __restrict__ x, uint16_t* __restrict__ planes,
                                                              int64_t plane_stride, int C, int L, float slope) {
  extern __shared__ float tile[];                    // [C][33]
  const int b = blockIdx.y, t0 = blockIdx.x * 32, tid = threadIdx.x;
  const float* xb = x + (int64_t)b * C * L;
  for (int e = tid; e < C * 32; e += 256) {
    const int ch = e >> 5, t = e & 31;
    float v = t0 + t < L ? xb[(int64_t)ch * L + t0 + t] : 0.f;
    v = v >= 0.f ? v : v * slope;
    tile[ch * 33 + t] = v;
  }
  __syncthreads();
  const int half = C / 2;                            // channel pairs per row
  for (int e = tid; e < 32 * half; e += 256) {
    const int t = e / half, cp = e - t * half;
    if (t0 + t >= L) continue;
    float v0 = tile[(2 * cp) * 33 + t], v1 = tile[(2 * cp + 1) * 33 + t];
    uint32_t* dst = reinterpret_cast<uint32_t*>(planes + ((int64_t)b * L + t0 + t) * C) + cp;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      const uint32_t u = pack2(v0, v1);
      *reinterpret_cast<uint32_t*>(reinterpret_cast<uint16_t*>(dst) + pl * plane_stride) = u;
      const f32x2 r = unpack2(u);
      v0 -= r[0];
      v1 -= r[1];
    }
  }
}

// out [B][C][L] fp32 = (sum over the given plane tensors of their value, each de-activated) * scale
__global__ __launch_bounds__(256) void split3_to_f32_kernel(const uint16_t* __restrict__ a, const uint16_t* __restrict__ b2,
                                                            const uint16_t* __restrict__ c3, int64_t plane_stride,
                                                            float* __restrict__ out, int C, int L, float in_slope, float scale) {
  extern __shared__ float tile[];                    // [C][33]
  const int b = blockIdx.y, t0 = blockIdx.x * 32, tid = threadIdx.x;
  const int half = C / 2;
  const float inv = 1.0f / in_slope;
  const uint16_t* srcs[3] = {a, b2, c3};
  for (int e = tid; e < 32 * half; e += 256) {
    const int t = e / half, cp = e - t * half;
    f32x2 sum = {0.f, 0.f};
    if (t0 + t < L) {
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        if (!srcs[i]) continue;
        const uint16_t* src = srcs[i] + ((int64_t)b * L + t0 + t) * C + 2 * cp;
        f32x2 v = (unpack2(*reinterpret_cast<const uint32_t*>(src)) + unpack2(*reinterpret_cast<const uint32_t*>(src + plane_stride))) +
                  unpack2(*reinterpret_cast<const uint32_t*>(src + 2 * plane_stride));
        if (in_slope != 1.0f) {
          const f32x2 m = v * inv;
          v = f32x2{v[0] < m[0] ? v[0] : m[0], v[1] < m[1] ? v[1] : m[1]};
        }
        sum = i == 0 ? v : sum + v;
      }
    }
    tile[(2 * cp) * 33 + t] = sum[0] * scale;
    tile[(2 * cp + 1) * 33 + t] = sum[1] * scale;
  }
  __syncthreads();
  float* ob = out + (int64_t)b * C * L;
  for (int e = tid; e < C * 32; e += 256) {
    const int ch = e >> 5, t = e & 31;
    if (t0 + t < L) ob[(int64_t)ch * L + t0 + t] = tile[ch * 33 + t];
  }
}

}  // namespace ovks3

using namespace ovks3;

extern "C" {

size_t ov_conv1d_split3_pack_size(int Cout, int Cin, int K) {
  if (Cout <= 0 || Cin <= 0 || K <= 0 || Cout % 32 != 0 || Cin % 32 != 0) return 0;
  return ((size_t)(Cout / 32) * (Cin / 32) * K * 6 + 1) * 512;
}

// Weights in v_mfma_f32_16x16x32_bf16 A-fragment order, three planes: record (((ct * Cin/32 + c) * K + tap) * 3 + plane)
// * 2 + f = 64 lanes x 8 bf16; lane (l15 = lane & 15, g = lane >> 4) holds plane `plane` of
// W[32 ct + 16 f + l15][32 c + 8 g .. + 8][tap]; planes: hi = bf16(w), mid = bf16(w - hi), lo = bf16(w - hi - mid)
// (round to nearest even; hi + mid + lo == w exactly).  One trailing all-zero record (DMA source outside [0, L)).
int ov_conv1d_split3_pack(const float* w, int Cout, int Cin, int K, uint16_t* dst) {
  const size_t n = ov_conv1d_split3_pack_size(Cout, Cin, K);
  if (!w || !dst || n == 0) return OV_E_BADARG;
  std::memset(dst, 0, n * sizeof(uint16_t));
  auto to_bf16 = [](float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN
    u += 0x7fffu + ((u >> 16) & 1u);                                            // round to nearest even
    return (uint16_t)(u >> 16);
  };
  auto from_bf16 = [](uint16_t h) {
    const uint32_t u = (uint32_t)h << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
  };
  const int nchunks = Cin / 32;
  for (int ct = 0; ct < Cout / 32; ++ct)
    for (int c = 0; c < nchunks; ++c)
      for (int tap = 0; tap < K; ++tap)
        for (int f = 0; f < 2; ++f)
          for (int lane = 0; lane < 64; ++lane)
            for (int i = 0; i < 8; ++i) {
              const int co = 32 * ct + 16 * f + (lane & 15);
              const int ci = 32 * c + 8 * (lane >> 4) + i;
              float v = w[((size_t)co * Cin + ci) * K + tap];
              for (int pl = 0; pl < 3; ++pl) {
                const uint16_t h = to_bf16(v);
                dst[((((((size_t)ct * nchunks + c) * K + tap) * 3 + pl) * 2 + f) * 64 + lane) * 8 + i] = h;
                v -= from_bf16(h);                                              // exact in fp32
              }
            }
  return OV_OK;
}

int ov_conv1d_split3_supported(int Cin, int Cout, int K, int dil) {
  const bool kd = (K == 3 || K == 7 || K == 11) && (dil == 1 || dil == 3 || dil == 5);
  return kd && (Cin == 64 || Cin == 128 || Cin == 256) && Cout == Cin ? 1 : 0;
}

int ov_conv1d_split3(const ov_conv1d_split3_params* pin, ov_stream_t stream) {
  if (!pin) return OV_E_BADARG;
  ov_conv1d_split3_params q = *pin;                  // (normalised copy: the limit is dropped where the table cannot hold it)
  const ov_conv1d_split3_params* p = &q;
  if (!p->x || !p->w || !p->bias || !p->out) return OV_E_BADARG;
  if (p->col_limit && (p->col_limit_scale <= 0 || (reinterpret_cast<uintptr_t>(p->col_limit) & 3))) return OV_E_BADARG;
  if (p->col_limit && p->B > ovks3::LIMIT_MAX_BATCH) q.col_limit = nullptr;   // documented: whole tensors
  if (p->B <= 0 || p->L <= 0 || p->nwg < 0 || p->x_plane <= 0 || p->out_plane <= 0) return OV_E_BADARG;
  if (p->res && p->res_plane <= 0) return OV_E_BADARG;
  if (p->out == p->x || p->out == p->res) return OV_E_BADARG;
  if (!(p->res_slope > 0.f && p->res_slope <= 1.f) || !(p->out_slope > 0.f && p->out_slope <= 1.f)) return OV_E_UNSUPPORTED;
  if (p->products != 6 && p->products != 3) return OV_E_UNSUPPORTED;
  if (!ov_conv1d_split3_supported(p->Cin, p->Cout, p->K, p->dil)) return OV_E_UNSUPPORTED;
  if (p->res && p->dil != 1) return OV_E_UNSUPPORTED;      // (the residual form keeps both output halves in LDS)
  if ((int64_t)p->B * p->L * p->Cin * 2 > INT64_MAX / 4) return OV_E_BADARG;
  if ((reinterpret_cast<uintptr_t>(p->x) & 15) || (reinterpret_cast<uintptr_t>(p->w) & 15) ||
      (reinterpret_cast<uintptr_t>(p->out) & 15) || (p->res && (reinterpret_cast<uintptr_t>(p->res) & 15)) ||
      (p->x_plane % 8) || (p->out_plane % 8) || (p->res && (p->res_plane % 8)))
    return OV_E_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
#define OV_SPLIT3_CASE(KK, DD) if (p->K == KK && p->dil == DD) return split3_launch_k##KK##d##DD(p, st);
  OV_SPLIT3_CASE(3, 1) OV_SPLIT3_CASE(3, 3) OV_SPLIT3_CASE(3, 5)
  OV_SPLIT3_CASE(7, 1) OV_SPLIT3_CASE(7, 3) OV_SPLIT3_CASE(7, 5)
  OV_SPLIT3_CASE(11, 1) OV_SPLIT3_CASE(11, 3) OV_SPLIT3_CASE(11, 5)
#undef OV_SPLIT3_CASE
  return OV_E_UNSUPPORTED;
}

int ov_split3_from_f32(const float* x, uint16_t* planes, int64_t plane_stride, int B, int C, int L, float slope,
                       ov_stream_t stream) {
  if (!x || !planes || B <= 0 || C <= 0 || L <= 0 || B > 65535 || C % 2 != 0 || C > 512) return OV_E_BADARG;
  if (plane_stride < (int64_t)B * L * C || (plane_stride & 1)) return OV_E_BADARG;
  if (!(slope > 0.f && slope <= 1.f)) return OV_E_UNSUPPORTED;
  if (reinterpret_cast<uintptr_t>(planes) & 3) return OV_E_ALIGN;
  hipLaunchKernelGGL(split3_from_f32_kernel, dim3((L + 31) / 32, B), dim3(256), (size_t)C * 33 * sizeof(float),
                     static_cast<hipStream_t>(stream), x, planes, plane_stride, C, L, slope);
  return hipGetLastError() == hipSuccess ? OV_OK : OV_E_LAUNCH;
}

int ov_split3_to_f32(const uint16_t* a, const uint16_t* b, const uint16_t* c, int64_t plane_stride, float* out, int B,
                     int C, int L, float in_slope, float scale, ov_stream_t stream) {
  if (!a || !out || B <= 0 || C <= 0 || L <= 0 || B > 65535 || C % 2 != 0 || C > 512) return OV_E_BADARG;
  if (plane_stride < (int64_t)B * L * C || (plane_stride & 1)) return OV_E_BADARG;
  if (!(in_slope > 0.f && in_slope <= 1.f)) return OV_E_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(a) & 3) || (b && (reinterpret_cast<uintptr_t>(b) & 3)) || (c && (reinterpret_cast<uintptr_t>(c) & 3)))
    return OV_E_ALIGN;
  hipLaunchKernelGGL(split3_to_f32_kernel, dim3((L + 31) / 32, B), dim3(256), (size_t)C * 33 * sizeof(float),
                     static_cast<hipStream_t>(stream), a, b, c, plane_stride, out, C, L, in_slope, scale);
  return hipGetLastError() == hipSuccess ? OV_OK : OV_E_LAUNCH;
}

}  // extern "C"
