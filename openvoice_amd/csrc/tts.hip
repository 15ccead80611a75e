// Front-end kernels of the V1 base-speaker TTS path (SynthesizerTrn.infer, reference:
// openvoice/models.py:467-490): token embedding, channel LayerNorm (+ residual / ReLU / GELU / mask),
// windowed relative-position attention, depthwise dilated conv, the 1 -> C expansion of ConvFlow.pre,
// the inverse rational-quadratic spline, duration arithmetic and the duration-driven expansion of the
// prior.  Everything here runs at TOKEN rate (T_x ~ 10^2 per utterance): a few GFLOP per batch against
// ~280 GFLOP per utterance in the flow + generator that follow, so these are latency-bound VALU kernels
// laid out for coalescing ((B, C, T) with time contiguous, rows `ld` floats apart); the dense 1x1 / k3
// convs between them run on the MFMA conv kernel (conv1d_mfma.h).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "openvoice_amd.h"

namespace ovk {

// out[b][h][t] = t < len[b] ? emb[tok[b][t]][h] * scale : 0        (models.py:49-53)
__global__ __launch_bounds__(256) void embed_kernel(const int64_t* __restrict__ tok, const float* __restrict__ emb,
                                                    const int64_t* __restrict__ len, float* __restrict__ out,
                                                    int T, int H, int V, int ld, float scale) {
  const int b = blockIdx.z, h = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  int64_t id = tok[(int64_t)b * T + t];
  id = id < 0 ? 0 : (id >= V ? V - 1 : id);   // range is checked on the host; never read out of bounds
  out[((int64_t)b * H + h) * ld + t] = t < len[b] ? emb[id * H + h] * scale : 0.f;
}

// LayerNorm over channels of (B, C, T):
//   v = x (+ res);  v = relu(v) if PRE_RELU;  y = (v - mean) * rstd * gamma[c] + beta[c];
//   y = gelu_erf(y) if POST_GELU;  y += res2 if res2;  y *= mask[b][t] if mask.
// A workgroup = 32 time columns x 8 channel groups (256 threads): lanes along t (coalesced rows), each thread
// reduces C/8 channels, the 8 partial sums meet in LDS.  Token-rate tensors have only ~10^3 columns, so one thread
// per column (the first version) left the GPU 99 % idle and cost ~130 us per call; the second and third pass
// re-read L2-hot data.  Two-pass variance, as ATen's CPU kernel.
// reference: openvoice/modules.py:17-29 (LayerNorm), attentions.py:114-119, models.py:91-98, modules.py:121-129.
__global__ __launch_bounds__(256) void layernorm_ch_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           const float* __restrict__ res2,
                                                           const float* __restrict__ mask, float* __restrict__ out,
                                                           int C, int T, int ld, float eps, int flags) {
  __shared__ float part[8][33];
  const int b = blockIdx.y;
  const int tl = threadIdx.x & 31, cg = threadIdx.x >> 5;          // column within the tile, channel group
  const int t = blockIdx.x * 32 + tl;
  const bool live = t < T;
  const int64_t base = (int64_t)b * C * ld + (live ? t : 0);
  const bool relu = flags & OV_LN_PRE_RELU;
  auto value = [&](int c) {
    float v = x[base + (int64_t)c * ld];
    if (res) v += res[base + (int64_t)c * ld];
    return relu ? fmaxf(v, 0.f) : v;
  };
  // C <= 256 (every LayerNorm of the model: hidden 192, duration-predictor filter 256): the thread's C / 8 values are
  // read ONCE, 32 independent loads in flight, and stay in registers for the variance and the output pass (the first
  // version re-read them twice through a serial loop: ~22 us per call on a 1.2 MB tensor, 38 calls per infer()).
  // Same operations in the same order either way: bit-identical.
  constexpr int MAXV = 32;
  const bool cached = C <= 8 * MAXV;
  float v[MAXV];
  if (cached) {
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = cg + 8 * i;
      v[i] = (live && c < C) ? value(c) : 0.f;
    }
  }
  float s = 0.f;
  if (live) {
    if (cached) {
#pragma unroll
      for (int i = 0; i < MAXV; ++i)
        if (cg + 8 * i < C) s += v[i];
    } else {
      for (int c = cg; c < C; c += 8) s += value(c);
    }
  }
  part[cg][tl] = s;
  __syncthreads();
  float mean = 0.f;
#pragma unroll
  for (int g = 0; g < 8; ++g) mean += part[g][tl];
  mean /= C;
  __syncthreads();
  float var = 0.f;
  if (live) {
    if (cached) {
#pragma unroll
      for (int i = 0; i < MAXV; ++i)
        if (cg + 8 * i < C) {
          const float d = v[i] - mean;
          var = fmaf(d, d, var);
        }
    } else {
      for (int c = cg; c < C; c += 8) {
        const float d = value(c) - mean;
        var = fmaf(d, d, var);
      }
    }
  }
  part[cg][tl] = var;
  __syncthreads();
  var = 0.f;
#pragma unroll
  for (int g = 0; g < 8; ++g) var += part[g][tl];
  if (!live) return;
  const float rstd = 1.f / sqrtf(var / C + eps);
  const float mk = mask ? mask[(int64_t)b * ld + t] : 1.f;
  auto finish = [&](int c, float xv) {
    float y = (xv - mean) * rstd * gamma[c] + beta[c];
    if (flags & OV_LN_POST_GELU) y = 0.5f * y * (1.f + erff(y * 0.70710678118654752440f));
    if (res2) y += res2[base + (int64_t)c * ld];
    out[base + (int64_t)c * ld] = y * mk;
  };
  if (cached) {
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
      if (cg + 8 * i < C) finish(cg + 8 * i, v[i]);
  } else {
    for (int c = cg; c < C; c += 8) finish(c, value(c));
  }
}

// Self-attention with a +-window relative-position band (reference: openvoice/attentions.py:264-329).
// One workgroup = 8 query rows of one (utterance, head).  Scores of the 8 rows against all T keys live in
// LDS (8 x T floats), K and V stream through LDS in 64-key tiles.  The relative-key / relative-value terms
// touch only the 2w+1 diagonals that are non-zero in the reference's padded [T, 2T-1] tensors and are
// evaluated directly on that band.
template <int DK>
__global__ __launch_bounds__(256) void rel_attention_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                            const float* __restrict__ v,
                                                            const float* __restrict__ emb_k,
                                                            const float* __restrict__ emb_v,
                                                            const float* __restrict__ mask, float* __restrict__ out,
                                                            int64_t qkv_bstride, int64_t out_bstride, int T, int ld,
                                                            int window, float scale) {
  constexpr int QB = 8, KT = 64, KS = KT + 1;
  extern __shared__ float smem[];
  float* Qs = smem;                 // [QB][DK]
  float* KVs = Qs + QB * DK;        // [DK][KS]
  float* S = KVs + DK * KS;         // [QB][T]
  const int tid = threadIdx.x;
  const int b = blockIdx.z, h = blockIdx.y, t0 = blockIdx.x * QB;
  const int64_t rowbase = (int64_t)b * qkv_bstride + (int64_t)h * DK * ld;
  const int64_t obase = (int64_t)b * out_bstride + (int64_t)h * DK * ld;
  const float* mrow = mask + (int64_t)b * ld;

  for (int idx = tid; idx < QB * DK; idx += 256) {
    const int i = idx / DK, d = idx - i * DK;
    Qs[idx] = (t0 + i < T) ? q[rowbase + (int64_t)d * ld + t0 + i] * scale : 0.f;
  }
  // ---- scores = (q / sqrt(dk)) . k ------------------------------------------------------------------
  const int qi = tid >> 5, lj = tid & 31;
  for (int s0 = 0; s0 < T; s0 += KT) {
    __syncthreads();
    for (int idx = tid; idx < DK * KT; idx += 256) {
      const int d = idx / KT, j = idx - d * KT;
      KVs[d * KS + j] = (s0 + j < T) ? k[rowbase + (int64_t)d * ld + s0 + j] : 0.f;
    }
    __syncthreads();
    float a0 = 0.f, a1 = 0.f;
    for (int d = 0; d < DK; ++d) {
      const float qv = Qs[qi * DK + d];
      a0 = fmaf(qv, KVs[d * KS + lj], a0);
      a1 = fmaf(qv, KVs[d * KS + lj + 32], a1);
    }
    if (s0 + lj < T) S[qi * T + s0 + lj] = a0;
    if (s0 + lj + 32 < T) S[qi * T + s0 + lj + 32] = a1;
  }
  __syncthreads();
  // ---- + relative-key logits on the band ------------------------------------------------------------
  const int nrel = 2 * window + 1;
  if (tid < QB * nrel) {
    const int i = tid / nrel, r = tid - i * nrel;
    const int t = t0 + i, s = t + r - window;
    if (t < T && s >= 0 && s < T) {
      float a = 0.f;
      for (int d = 0; d < DK; ++d) a = fmaf(Qs[i * DK + d], emb_k[r * DK + d], a);
      S[i * T + s] += a;
    }
  }
  __syncthreads();
  // ---- mask (masked_fill(mask == 0, -1e4)) + softmax over keys: 32 lanes per query row --------------
  {
    const int t = t0 + qi;
    if (t < T) {
      const float mq = mrow[t];
      float mx = -3.0e38f;
      for (int s = lj; s < T; s += 32) {
        float val = S[qi * T + s];
        if (mq * mrow[s] == 0.f) val = -1e4f;
        S[qi * T + s] = val;
        mx = fmaxf(mx, val);
      }
#pragma unroll
      for (int off = 16; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
      float sum = 0.f;
      for (int s = lj; s < T; s += 32) {
        const float e = expf(S[qi * T + s] - mx);
        S[qi * T + s] = e;
        sum += e;
      }
#pragma unroll
      for (int off = 16; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
      const float inv = 1.f / sum;
      for (int s = lj; s < T; s += 32) S[qi * T + s] *= inv;
    }
  }
  // ---- out = p . v  (+ relative values on the band) --------------------------------------------------
  constexpr int ND = (DK + 31) / 32;
  float acc[ND];
#pragma unroll
  for (int m = 0; m < ND; ++m) acc[m] = 0.f;
  for (int s0 = 0; s0 < T; s0 += KT) {
    __syncthreads();
    for (int idx = tid; idx < DK * KT; idx += 256) {
      const int d = idx / KT, j = idx - d * KT;
      KVs[d * KS + j] = (s0 + j < T) ? v[rowbase + (int64_t)d * ld + s0 + j] : 0.f;
    }
    __syncthreads();
    const int jmax = min(KT, T - s0);
    for (int j = 0; j < jmax; ++j) {
      const float pv = S[qi * T + s0 + j];
#pragma unroll
      for (int m = 0; m < ND; ++m) {
        const int d = lj + 32 * m;
        if (d < DK) acc[m] = fmaf(pv, KVs[d * KS + j], acc[m]);
      }
    }
  }
  const int t = t0 + qi;
  if (t < T) {
    for (int r = 0; r < nrel; ++r) {
      const int s = t + r - window;
      if (s < 0 || s >= T) continue;
      const float pv = S[qi * T + s];
#pragma unroll
      for (int m = 0; m < ND; ++m) {
        const int d = lj + 32 * m;
        if (d < DK) acc[m] = fmaf(pv, emb_v[r * DK + d], acc[m]);
      }
    }
#pragma unroll
    for (int m = 0; m < ND; ++m) {
      const int d = lj + 32 * m;
      if (d < DK) out[obase + (int64_t)d * ld + t] = acc[m];
    }
  }
}

// Depthwise dilated conv over the masked input (DDSConv.convs_sep, reference: openvoice/modules.py:102-112,
// :121): out[b][c][t] = bias[c] + sum_j w[c][j] * (x * mask)[b][c][t + (j - (K-1)/2) * dil].
__global__ __launch_bounds__(256) void dwconv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ bias, const float* __restrict__ mask,
                                                     float* __restrict__ out, int C, int T, int ld, int K, int dil) {
  const int b = blockIdx.z, c = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  const float* xr = x + ((int64_t)b * C + c) * ld;
  const float* mr = mask + (int64_t)b * ld;
  float acc = bias[c];
  const int half = (K - 1) / 2;
  for (int j = 0; j < K; ++j) {
    const int s = t + (j - half) * dil;
    if (s >= 0 && s < T) acc = fmaf(w[c * K + j], xr[s] * mr[s], acc);
  }
  out[((int64_t)b * C + c) * ld + t] = acc;
}

// ConvFlow.pre (a 1 -> C pointwise conv) fused with DDSConv's `x + g` (reference: openvoice/modules.py:487-488,
// :118-119): out[b][c][t] = w[c] * x0[b][t] + bias[c] + g[b][c][t].
__global__ __launch_bounds__(256) void expand1_kernel(const float* __restrict__ x0, int64_t x0_bstride,
                                                      const float* __restrict__ w, const float* __restrict__ bias,
                                                      const float* __restrict__ g, float* __restrict__ out, int C,
                                                      int T, int ld) {
  const int b = blockIdx.z, c = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  const int64_t o = ((int64_t)b * C + c) * ld + t;
  out[o] = fmaf(w[c], x0[(int64_t)b * x0_bstride + t], bias[c]) + (g ? g[o] : 0.f);
}

// out[b][c][t] = (x[b][c][t] + bias_b[b][c]) * mask[b][t] -- DurationPredictor's `x + cond(g)` followed by the
// `x * x_mask` of its first conv (reference: openvoice/models.py:88-90).
__global__ __launch_bounds__(256) void add_bias_mask_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ bias_b,
                                                            const float* __restrict__ mask, float* __restrict__ out,
                                                            int C, int T, int ld) {
  const int b = blockIdx.z, c = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  const int64_t o = ((int64_t)b * C + c) * ld + t;
  out[o] = (x[o] + bias_b[(int64_t)b * C + c]) * mask[(int64_t)b * ld + t];
}

__device__ __forceinline__ float softplus_f(float v) { return v > 20.f ? v : log1pf(expf(v)); }

// Inverse of the unconstrained rational-quadratic spline with linear tails, one thread per (b, t)
// (reference: openvoice/transforms.py:50-97 tails, :100-188 knots + inverse branch), then ConvFlow's
// `cat([x0, x1]) * mask` (modules.py:511).  `h` holds the 3*NB-1 unnormalised parameters as rows of a
// (B, >=3*NB-1, T) tensor: NB widths, NB heights (both divided by sqrt(filter_channels) here), NB-1
// derivatives.  z is (B, 2, T); channel c1 is transformed in place, channel c0 only masked.
template <int NB>
__global__ __launch_bounds__(256) void rq_spline_inverse_kernel(float* __restrict__ z, int64_t z_bstride, int c0,
                                                                int c1, const float* __restrict__ h,
                                                                int64_t h_bstride, const float* __restrict__ mask,
                                                                int T, int ld, float inv_sqrt_filt, float tb,
                                                                float edge_const) {
  const int b = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  const float mk = mask[(int64_t)b * ld + t];
  float* zb = z + (int64_t)b * z_bstride;
  const float y = zb[(int64_t)c1 * ld + t];
  float x = y;
  if (y >= -tb && y <= tb) {
    const float* hb = h + (int64_t)b * h_bstride + t;
    constexpr float MINW = 1e-3f, MINH = 1e-3f, MIND = 1e-3f;
    float cw[NB + 1], chh[NB + 1], dv[NB + 1];
    // widths / heights: softmax -> floor -> cumulative knots on [-tb, tb] with the ends pinned
    for (int pass = 0; pass < 2; ++pass) {
      float u[NB];
      float mx = -3.0e38f;
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        u[i] = hb[(int64_t)(pass * NB + i) * ld] * inv_sqrt_filt;
        mx = fmaxf(mx, u[i]);
      }
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        u[i] = expf(u[i] - mx);
        sum += u[i];
      }
      float* kn = pass == 0 ? cw : chh;
      const float minv = pass == 0 ? MINW : MINH;
      float run = 0.f;
      kn[0] = -tb;
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        run += minv + (1.f - minv * NB) * (u[i] / sum);
        kn[i + 1] = 2.f * tb * run - tb;
      }
      kn[NB] = tb;
    }
    dv[0] = MIND + softplus_f(edge_const);
    dv[NB] = dv[0];
#pragma unroll
    for (int i = 1; i < NB; ++i) dv[i] = MIND + softplus_f(hb[(int64_t)(2 * NB + i - 1) * ld]);
    // bin: number of knots <= y, minus one (the reference nudges the last knot by 1e-6)
    int bin = -1;
#pragma unroll
    for (int i = 0; i <= NB; ++i) bin += (y >= (i == NB ? chh[i] + 1e-6f : chh[i])) ? 1 : 0;
    bin = bin < 0 ? 0 : (bin > NB - 1 ? NB - 1 : bin);
    float cwl = cw[0], bw = cw[1] - cw[0], chl = chh[0], bh = chh[1] - chh[0], d0 = dv[0], d1 = dv[1];
#pragma unroll
    for (int i = 1; i < NB; ++i) {
      if (bin == i) {
        cwl = cw[i]; bw = cw[i + 1] - cw[i]; chl = chh[i]; bh = chh[i + 1] - chh[i]; d0 = dv[i]; d1 = dv[i + 1];
      }
    }
    const float delta = bh / bw;
    const float dy = y - chl;
    const float s = d0 + d1 - 2.f * delta;
    const float a = dy * s + bh * (delta - d0);
    const float bq = bh * d0 - dy * s;
    const float c = -delta * dy;
    const float root = (2.f * c) / (-bq - sqrtf(bq * bq - 4.f * a * c));
    x = root * bw + cwl;
  }
  zb[(int64_t)c1 * ld + t] = x * mk;
  zb[(int64_t)c0 * ld + t] *= mk;
}

// Durations of one utterance (reference: openvoice/models.py:474-479, ElementwiseAffine reverse modules.py:397-399):
//   logw = ((z - m) * exp(-logs) * mask) * sdp_ratio + dp * (1 - sdp_ratio)
//   w_ceil = ceil(exp(logw) * mask * length_scale);  cum = inclusive prefix sum;  y_len = max(1, sum).
// One workgroup per utterance; token counts are ~10^2, so the scan is done by one lane.
__global__ __launch_bounds__(256) void duration_kernel(const float* __restrict__ z_sdp, int64_t z_bstride,
                                                       float ea_m, float ea_logs, const float* __restrict__ dp,
                                                       int64_t dp_bstride, const float* __restrict__ mask,
                                                       float* __restrict__ logw, int32_t* __restrict__ cum,
                                                       int64_t* __restrict__ y_len, int T, int ld, float sdp_ratio,
                                                       float length_scale) {
  const int b = blockIdx.x;
  const float* mr = mask + (int64_t)b * ld;
  const float einv = expf(-ea_logs);
  for (int t = threadIdx.x; t < T; t += 256) {
    const float mk = mr[t];
    const float ls = (z_sdp[(int64_t)b * z_bstride + t] - ea_m) * einv * mk;
    logw[(int64_t)b * ld + t] = ls * sdp_ratio + dp[(int64_t)b * dp_bstride + t] * (1.f - sdp_ratio);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int32_t run = 0;
    for (int t = 0; t < T; ++t) {
      const float w = expf(logw[(int64_t)b * ld + t]) * mr[t] * length_scale;
      run += (int32_t)ceilf(w);
      cum[(int64_t)b * ld + t] = run;
    }
    y_len[b] = run < 1 ? 1 : run;
  }
}

// Duration-driven expansion + prior sample (reference: openvoice/models.py:480-487, commons.py:128-142):
// frame t' of utterance b belongs to token j iff cum[j-1] <= t' < cum[j] (j < x_len, t' < y_len); then
//   m_p = m_tok[:, j], logs_p = logs_tok[:, j]   (both 0 outside),  z_p = m_p + noise * exp(logs_p) * noise_scale.
// The hard alignment attn[b][t'][j] is written when requested (the reference returns it).
__global__ __launch_bounds__(256) void expand_prior_kernel(const float* __restrict__ m_tok,
                                                           const float* __restrict__ logs_tok, int64_t tok_bstride,
                                                           int ldx,
                                                           const int32_t* __restrict__ cum,
                                                           const int64_t* __restrict__ x_len,
                                                           const int64_t* __restrict__ y_len,
                                                           const float* __restrict__ noise, int64_t noise_bstride,
                                                           int ldn, float* __restrict__ z_p, float* __restrict__ m_p,
                                                           float* __restrict__ logs_p, float* __restrict__ attn,
                                                           int C, int Tx, int Ty, int ldy, float noise_scale) {
  const int b = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= Ty) return;
  const int32_t* cb = cum + (int64_t)b * ldx;
  const int xl = (int)min((int64_t)Tx, x_len[b]);
  int j = -1;
  if (t < y_len[b] && xl > 0 && t < cb[xl - 1]) {
    int lo = 0, hi = xl - 1;            // first j with cum[j] > t
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cb[mid] > t) hi = mid; else lo = mid + 1;
    }
    j = lo;
  }
  if (attn) {
    float* ar = attn + ((int64_t)b * Ty + t) * Tx;
    for (int i = 0; i < Tx; ++i) ar[i] = i == j ? 1.f : 0.f;
  }
  for (int c = 0; c < C; ++c) {
    const int64_t o = ((int64_t)b * C + c) * ldy + t;
    const float m = j >= 0 ? m_tok[(int64_t)b * tok_bstride + (int64_t)c * ldx + j] : 0.f;
    const float lg = j >= 0 ? logs_tok[(int64_t)b * tok_bstride + (int64_t)c * ldx + j] : 0.f;
    if (m_p) m_p[o] = m;
    if (logs_p) logs_p[o] = lg;
    z_p[o] = m + noise[(int64_t)b * noise_bstride + (int64_t)c * ldn + t] * expf(lg) * noise_scale;
  }
}

}  // namespace ovk

using namespace ovk;

#define OV_LAUNCH_OK() (hipGetLastError() == hipSuccess ? OV_OK : OV_E_LAUNCH)

extern "C" {

int ov_embed_f32(const int64_t* tokens, const float* emb, const int64_t* lengths, float* out, int B, int T, int H,
                 int V, int ld, float scale, ov_stream_t stream) {
  if (!tokens || !emb || !lengths || !out || B <= 0 || T <= 0 || H <= 0 || V <= 0 || ld < T || B > 65535 || H > 65535)
    return OV_E_BADARG;
  dim3 grid((T + 255) / 256, H, B);
  hipLaunchKernelGGL(embed_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), tokens, emb, lengths, out, T,
                     H, V, ld, scale);
  return OV_LAUNCH_OK();
}

int ov_layernorm_ch_f32(const float* x, const float* res, const float* gamma, const float* beta, const float* res2,
                        const float* mask, float* out, int B, int C, int T, int ld, float eps, int flags,
                        ov_stream_t stream) {
  if (!x || !gamma || !beta || !out || B <= 0 || C <= 0 || T <= 0 || ld < T || B > 65535) return OV_E_BADARG;
  dim3 grid((T + 31) / 32, B);
  hipLaunchKernelGGL(layernorm_ch_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), x, res, gamma, beta,
                     res2, mask, out, C, T, ld, eps, flags);
  return OV_LAUNCH_OK();
}

int ov_rel_attention_f32(const float* q, const float* k, const float* v, const float* emb_k, const float* emb_v,
                         const float* mask, float* out, int64_t qkv_bstride, int64_t out_bstride, int B, int n_heads,
                         int dk, int T, int ld, int window, ov_stream_t stream) {
  if (!q || !k || !v || !emb_k || !emb_v || !mask || !out || B <= 0 || n_heads <= 0 || T <= 0 || ld < T ||
      window < 0 || B > 65535 || n_heads > 65535)
    return OV_E_BADARG;
  if (dk != 96) return OV_E_UNSUPPORTED;                   // hidden 192 / 2 heads (models.py:39-45)
  if (8 * (2 * window + 1) > 256) return OV_E_UNSUPPORTED;
  const size_t smem = (size_t)(8 * 96 + 96 * 65 + 8 * (size_t)T) * sizeof(float);
  if (smem > 64 * 1024) return OV_E_UNSUPPORTED;           // T <= 1199 tokens per utterance
  dim3 grid((T + 7) / 8, n_heads, B);
  if (qkv_bstride < (int64_t)n_heads * dk * ld || out_bstride < (int64_t)n_heads * dk * ld) return OV_E_BADARG;
  hipLaunchKernelGGL(rel_attention_kernel<96>, grid, dim3(256), smem, static_cast<hipStream_t>(stream), q, k, v, emb_k,
                     emb_v, mask, out, qkv_bstride, out_bstride, T, ld, window, 1.0f / sqrtf((float)dk));
  return OV_LAUNCH_OK();
}

int ov_dwconv1d_f32(const float* x, const float* w, const float* bias, const float* mask, float* out, int B, int C,
                    int T, int ld, int K, int dil, ov_stream_t stream) {
  if (!x || !w || !bias || !mask || !out || B <= 0 || C <= 0 || T <= 0 || ld < T || K <= 0 || (K & 1) == 0 ||
      dil <= 0 || B > 65535 || C > 65535)
    return OV_E_BADARG;
  dim3 grid((T + 255) / 256, C, B);
  hipLaunchKernelGGL(dwconv_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), x, w, bias, mask, out, C, T,
                     ld, K, dil);
  return OV_LAUNCH_OK();
}

int ov_expand1_f32(const float* x0, int64_t x0_bstride, const float* w, const float* bias, const float* g, float* out,
                   int B, int C, int T, int ld, ov_stream_t stream) {
  if (!x0 || !w || !bias || !out || B <= 0 || C <= 0 || T <= 0 || ld < T || B > 65535 || C > 65535)
    return OV_E_BADARG;
  dim3 grid((T + 255) / 256, C, B);
  hipLaunchKernelGGL(expand1_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), x0, x0_bstride, w, bias, g,
                     out, C, T, ld);
  return OV_LAUNCH_OK();
}

int ov_add_bias_mask_f32(const float* x, const float* bias_b, const float* mask, float* out, int B, int C, int T,
                         int ld, ov_stream_t stream) {
  if (!x || !bias_b || !mask || !out || B <= 0 || C <= 0 || T <= 0 || ld < T || B > 65535 || C > 65535)
    return OV_E_BADARG;
  dim3 grid((T + 255) / 256, C, B);
  hipLaunchKernelGGL(add_bias_mask_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), x, bias_b, mask, out,
                     C, T, ld);
  return OV_LAUNCH_OK();
}

int ov_rq_spline_inverse_f32(float* z, int64_t z_bstride, int c0, int c1, const float* h, int64_t h_bstride,
                             const float* mask, int B, int T, int ld, int num_bins, int filter_channels,
                             float tail_bound, ov_stream_t stream) {
  if (!z || !h || !mask || B <= 0 || T <= 0 || ld < T || filter_channels <= 0 || tail_bound <= 0.f || B > 65535 ||
      c0 == c1 || c0 < 0 || c1 < 0 || c0 > 1 || c1 > 1)
    return OV_E_BADARG;
  if (num_bins != 10) return OV_E_UNSUPPORTED;             // ConvFlow default (modules.py:466)
  // log(exp(1 - min_derivative) - 1), evaluated in double as numpy does (transforms.py:71), then cast
  const float edge = (float)log(exp(1.0 - 1e-3) - 1.0);
  dim3 grid((T + 255) / 256, B);
  hipLaunchKernelGGL(rq_spline_inverse_kernel<10>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), z, z_bstride,
                     c0, c1, h, h_bstride, mask, T, ld, 1.0f / sqrtf((float)filter_channels), tail_bound, edge);
  return OV_LAUNCH_OK();
}

int ov_duration_f32(const float* z_sdp, int64_t z_bstride, float ea_m, float ea_logs, const float* dp,
                    int64_t dp_bstride, const float* mask, float* logw, int32_t* cum, int64_t* y_len, int B, int T,
                    int ld, float sdp_ratio, float length_scale, ov_stream_t stream) {
  if (!z_sdp || !dp || !mask || !logw || !cum || !y_len || B <= 0 || T <= 0 || ld < T) return OV_E_BADARG;
  hipLaunchKernelGGL(duration_kernel, dim3(B), dim3(256), 0, static_cast<hipStream_t>(stream), z_sdp, z_bstride, ea_m,
                     ea_logs, dp, dp_bstride, mask, logw, cum, y_len, T, ld, sdp_ratio, length_scale);
  return OV_LAUNCH_OK();
}

int ov_expand_prior_f32(const float* m_tok, const float* logs_tok, int64_t tok_bstride, int ldx, const int32_t* cum,
                        const int64_t* x_len,
                        const int64_t* y_len, const float* noise, int64_t noise_bstride, int ldn, float* z_p,
                        float* m_p, float* logs_p, float* attn, int B, int C, int Tx, int Ty, int ldy,
                        float noise_scale, ov_stream_t stream) {
  if (!m_tok || !logs_tok || !cum || !x_len || !y_len || !noise || !z_p || B <= 0 || C <= 0 || Tx <= 0 || Ty <= 0 ||
      ldx < Tx || ldy < Ty || ldn < Ty || B > 65535 || tok_bstride < (int64_t)C * ldx)
    return OV_E_BADARG;
  dim3 grid((Ty + 255) / 256, B);
  hipLaunchKernelGGL(expand_prior_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), m_tok, logs_tok,
                     tok_bstride, ldx, cum, x_len, y_len, noise, noise_bstride, ldn, z_p, m_p, logs_p, attn, C, Tx, Ty, ldy, noise_scale);
  return OV_LAUNCH_OK();
}

}  // extern "C"
