// ReferenceEncoder kernels (reference: openvoice/models.py:339-359): LayerNorm over the frequency
// axis, six 3x3 stride-2 conv2d + ReLU, GRU recurrence.  1.09 GFLOP per 10 s clip -- 0.2 % of a
// conversion -- and it runs in extract_se, not in convert, so these are plain VALU kernels laid out
// for coalescing rather than MFMA tiles.
//
// Layout: every tensor keeps TIME as the contiguous axis, [N][C][F][T].  The spectrogram arrives
// as [N][F][T] already (that is C = 1), and after the conv stack [N][128][9][T'] flattens to
// [N][1152][T'] with channel index c*9 + f -- exactly the feature order the reference builds with
// transpose(1,2).view(N, T', 128*9) (models.py:351-354) -- so the GRU input projection is a 1x1
// conv in the (B, C, T) layout of the MFMA conv kernel and no transpose is ever materialised.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "openvoice_amd.h"

namespace ovk {

// y[n][f][t] = (x[n][f][t] - mean_f) * rstd_f * gamma[f] + beta[f]; one thread per (n, t), lanes
// along t so every pass over f is a coalesced row read.  Two-pass variance (as ATen's CPU kernel).
__global__ __launch_bounds__(256) void layernorm_freq_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             float* __restrict__ y, int F, int T, float eps) {
  const int n = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  const float* xr = x + (int64_t)n * F * T + t;
  float* yr = y + (int64_t)n * F * T + t;
  float s = 0.f;
  for (int f = 0; f < F; ++f) s += xr[(int64_t)f * T];
  const float mean = s / F;
  float v = 0.f;
  for (int f = 0; f < F; ++f) {
    const float d = xr[(int64_t)f * T] - mean;
    v = fmaf(d, d, v);
  }
  const float rstd = 1.f / sqrtf(v / F + eps);
  for (int f = 0; f < F; ++f) yr[(int64_t)f * T] = (xr[(int64_t)f * T] - mean) * rstd * gamma[f] + beta[f];
}

// 3x3, stride 2, pad 1 conv + ReLU.  w is the reference's [Cout][Cin][kh][kw] with kh along TIME and
// kw along FREQUENCY (the reference's image is [N][C][Ty][F], models.py:342-349).  One thread =
// one output position (fo, to) x COB output channels; the weight index is wave-uniform, so the
// compiler fetches weights with scalar loads and the 9 taps of an input channel feed 9*COB FMAs.
template <int COB>
__global__ __launch_bounds__(256) void conv2d_s2_relu_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ w,
                                                             const float* __restrict__ bias,
                                                             float* __restrict__ y, int Cin, int Cout, int Fi,
                                                             int Ti, int Fo, int To) {
  const int n = blockIdx.z;
  const int co0 = blockIdx.y * COB;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= Fo * To) return;
  const int fo = p / To, to = p - fo * To;
  int64_t off[9];
  bool ok[9];
#pragma unroll
  for (int kh = 0; kh < 3; ++kh)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int ti = 2 * to + kh - 1, fi = 2 * fo + kw - 1;
      ok[kh * 3 + kw] = ti >= 0 && ti < Ti && fi >= 0 && fi < Fi;
      off[kh * 3 + kw] = (int64_t)fi * Ti + ti;
    }
  float acc[COB];
#pragma unroll
  for (int c = 0; c < COB; ++c) acc[c] = bias[co0 + c];
  const float* xn = x + (int64_t)n * Cin * Fi * Ti;
  for (int ci = 0; ci < Cin; ++ci) {
    const float* xc = xn + (int64_t)ci * Fi * Ti;
    float v[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) v[k] = ok[k] ? xc[off[k]] : 0.f;
    const float* wc = w + ((int64_t)co0 * Cin + ci) * 9;
#pragma unroll
    for (int c = 0; c < COB; ++c)
#pragma unroll
      for (int k = 0; k < 9; ++k) acc[c] = fmaf(wc[(int64_t)c * Cin * 9 + k], v[k], acc[c]);
  }
  float* yn = y + ((int64_t)n * Cout + co0) * Fo * To + p;
#pragma unroll
  for (int c = 0; c < COB; ++c) yn[(int64_t)c * Fo * To] = fmaxf(acc[c], 0.f);
}

// GRU recurrence (torch.nn.GRU cell, gate order r, z, n; reference: models.py:356-357).
// gi[n][3H][T] holds W_ih x_t + b_ih for every step (computed by the 1x1 MFMA conv); one workgroup
// per sequence keeps h in LDS and walks the T steps.  whh_t is W_hh transposed to [H][3H] so lane j
// reads whh_t[k][j]: coalesced and L2-resident across steps.
template <int H>
__global__ __launch_bounds__(3 * H) void gru_kernel(const float* __restrict__ gi, const float* __restrict__ whh_t,
                                                    const float* __restrict__ bhh, float* __restrict__ h_out,
                                                    int T) {
  __shared__ float h[H];
  __shared__ float gh[3 * H];
  const int n = blockIdx.x, j = threadIdx.x;
  if (j < H) h[j] = 0.f;
  __syncthreads();
  const float* gin = gi + (int64_t)n * 3 * H * T;
  const float bj = bhh[j];
  for (int t = 0; t < T; ++t) {
    float acc = bj;
#pragma unroll 8
    for (int k = 0; k < H; ++k) acc = fmaf(whh_t[k * 3 * H + j], h[k], acc);
    gh[j] = acc;
    __syncthreads();
    if (j < H) {
      const float r = 1.f / (1.f + expf(-(gin[(int64_t)j * T + t] + gh[j])));
      const float z = 1.f / (1.f + expf(-(gin[(int64_t)(H + j) * T + t] + gh[H + j])));
      const float c = tanhf(gin[(int64_t)(2 * H + j) * T + t] + r * gh[2 * H + j]);
      h[j] = (1.f - z) * c + z * h[j];
    }
    __syncthreads();
  }
  if (j < H) h_out[(int64_t)n * H + j] = h[j];
}

}  // namespace ovk

using namespace ovk;

extern "C" {

int ov_layernorm_freq_f32(const float* x, const float* gamma, const float* beta, float* y, int N, int F, int T,
                          float eps, ov_stream_t stream) {
  if (!x || !gamma || !beta || !y || N <= 0 || F <= 0 || T <= 0 || N > 65535) return OV_E_BADARG;
  dim3 grid((T + 255) / 256, N);
  hipLaunchKernelGGL(layernorm_freq_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), x, gamma, beta,
                     y, F, T, eps);
  return hipGetLastError() == hipSuccess ? OV_OK : OV_E_LAUNCH;
}

int ov_conv2d_s2_relu_f32(const float* x, const float* w, const float* bias, float* y, int N, int Cin, int Cout,
                          int Fi, int Ti, ov_stream_t stream) {
  if (!x || !w || !bias || !y || N <= 0 || Cin <= 0 || Cout <= 0 || Fi <= 0 || Ti <= 0 || N > 65535)
    return OV_E_BADARG;
  if (Cout % 16 != 0) return OV_E_UNSUPPORTED;
  const int Fo = (Fi - 1) / 2 + 1, To = (Ti - 1) / 2 + 1;
  dim3 grid((Fo * To + 255) / 256, Cout / 16, N);
  hipLaunchKernelGGL(conv2d_s2_relu_kernel<16>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), x, w, bias, y,
                     Cin, Cout, Fi, Ti, Fo, To);
  return hipGetLastError() == hipSuccess ? OV_OK : OV_E_LAUNCH;
}

int ov_gru_f32(const float* gi, const float* whh_t, const float* bhh, float* h_out, int N, int H, int T,
               ov_stream_t stream) {
  if (!gi || !whh_t || !bhh || !h_out || N <= 0 || T <= 0) return OV_E_BADARG;
  if (H != 128) return OV_E_UNSUPPORTED;
  hipLaunchKernelGGL(gru_kernel<128>, dim3(N), dim3(384), 0, static_cast<hipStream_t>(stream), gi, whh_t, bhh,
                     h_out, T);
  return hipGetLastError() == hipSuccess ? OV_OK : OV_E_LAUNCH;
}

}  // extern "C"
