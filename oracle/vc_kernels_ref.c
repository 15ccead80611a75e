/* ORACLE (test infrastructure, never the product path): plain-C, loop-level restatement of the operations that
 * carry the arithmetic of the path (convs: 98 % of the converter's FLOPs; the WaveNet gate; and the two TTS-side kernels
 * north_star names: the inverse rational-quadratic spline and relative-position attention), as the reference evaluates them.  No SIMD, no blocking, fp32
 * accumulation in the textbook order -- the independent check of oracle/vc_oracle.py (which restates the path in
 * terms of torch CPU operators) and, through it, of the HIP kernels.  Built by oracle/Makefile into
 * oracle/_build/libvc_kernels_ref.so; loaded by oracle/c_kernels.py; used by tests/test_oracle_c_kernels.py only. */
#include <math.h>
#include <stddef.h>

static float lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }

/* y[b][co][t] = bias[co] + sum_{ci, j} w[co][ci][j] * lrelu(x[b][ci][t + j*dil - pad], slope),  pad = (K*dil - dil)/2
 * reference: ResBlock1.forward, openvoice/modules.py:296-306 (F.leaky_relu(x, LRELU_SLOPE) then c1 / c2),
 * "same" padding get_padding, openvoice/commons.py:12-13; also Generator.conv_pre (models.py:273, slope = 1),
 * WN.in_layers (modules.py:166-172) and every 1x1 conv of the path (K = 1).  Zero padding outside [0, L). */
void ref_conv1d_f32(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int Cout, int L, int K,
                    int dil, float slope) {
  const int pad = (K * dil - dil) / 2;
  for (int b = 0; b < B; ++b)
    for (int co = 0; co < Cout; ++co)
      for (int t = 0; t < L; ++t) {
        float acc = bias ? bias[co] : 0.f;
        for (int ci = 0; ci < Cin; ++ci)
          for (int j = 0; j < K; ++j) {
            const int tt = t + j * dil - pad;
            if (tt >= 0 && tt < L)
              acc += w[((size_t)co * Cin + ci) * K + j] * lrelu(x[((size_t)b * Cin + ci) * L + tt], slope);
          }
        y[((size_t)b * Cout + co) * L + t] = acc;
      }
}

/* torch.nn.ConvTranspose1d(Cin, Cout, K, stride, padding = pad) applied to lrelu(x, slope):
 * y[b][co][u] = bias[co] + sum_{ci, t, j : t*stride - pad + j == u} w[ci][co][j] * lrelu(x[b][ci][t]),  L_out = (L-1)*stride - 2*pad + K
 * reference: Generator.ups, openvoice/models.py:244-256 (k = 2 * stride, pad = (k - stride) / 2 so L_out = stride * L)
 * and the leaky_relu in front of it, models.py:278-279. */
void ref_conv_transpose1d_f32(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int Cout,
                              int L, int K, int stride, int pad, float slope) {
  const int Lout = (L - 1) * stride - 2 * pad + K;
  for (int b = 0; b < B; ++b)
    for (int co = 0; co < Cout; ++co) {
      float* row = y + ((size_t)b * Cout + co) * Lout;
      for (int u = 0; u < Lout; ++u) row[u] = bias ? bias[co] : 0.f;
      for (int ci = 0; ci < Cin; ++ci)
        for (int t = 0; t < L; ++t) {
          const float v = lrelu(x[((size_t)b * Cin + ci) * L + t], slope);
          for (int j = 0; j < K; ++j) {
            const int u = t * stride - pad + j;
            if (u >= 0 && u < Lout) row[u] += w[((size_t)ci * Cout + co) * K + j] * v;
          }
        }
    }
}

/* acts[b][c][t] = tanh(x_in[b][c][t] + g[b][c]) * sigmoid(x_in[b][H + c][t] + g[b][H + c]),  c < H
 * reference: fused_add_tanh_sigmoid_multiply, openvoice/commons.py:100-107, as called by WN.forward
 * (modules.py:194-200) with g_l the per-layer slice of cond_layer(g) (T = 1, broadcast over t). */
void ref_gate_f32(const float* x_in, const float* g, float* acts, int B, int H, int T) {
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < H; ++c)
      for (int t = 0; t < T; ++t) {
        const float a = x_in[((size_t)b * 2 * H + c) * T + t] + (g ? g[(size_t)b * 2 * H + c] : 0.f);
        const float s = x_in[((size_t)b * 2 * H + H + c) * T + t] + (g ? g[(size_t)b * 2 * H + H + c] : 0.f);
        acts[((size_t)b * H + c) * T + t] = tanhf(a) * (1.f / (1.f + expf(-s)));
      }
}

/* One WaveNet layer as WN.forward evaluates it (openvoice/modules.py:192-209), in place on x and output:
 *   x_in = in_layer(x); acts = gate(x_in, g_l); rs = res_skip(acts);
 *   not last: x = (x + rs[:H]) * mask; output += rs[H:]        last: output += rs
 * w_in [2H][H][K], w_rs [2H or H][H][1]; mask [B][T]; scratch holds 2H*T + H*T + 2H*T floats. */
void ref_wn_layer_f32(float* x, float* output, const float* w_in, const float* b_in, const float* w_rs, const float* b_rs,
                      const float* g, const float* mask, int B, int H, int T, int K, int dil, int last, float* scratch) {
  for (int b = 0; b < B; ++b) {
    float* x_in = scratch;
    float* acts = x_in + (size_t)2 * H * T;
    float* rs = acts + (size_t)H * T;
    ref_conv1d_f32(x + (size_t)b * H * T, w_in, b_in, x_in, 1, H, 2 * H, T, K, dil, 1.f);
    ref_gate_f32(x_in, g ? g + (size_t)b * 2 * H : NULL, acts, 1, H, T);
    const int R = last ? H : 2 * H;
    ref_conv1d_f32(acts, w_rs, b_rs, rs, 1, H, R, T, 1, 1, 1.f);
    for (int c = 0; c < H; ++c)
      for (int t = 0; t < T; ++t) {
        const size_t i = ((size_t)b * H + c) * T + t;
        if (last) output[i] += rs[(size_t)c * T + t];
        else {
          x[i] = (x[i] + rs[(size_t)c * T + t]) * mask[(size_t)b * T + t];
          output[i] += rs[(size_t)(H + c) * T + t];
        }
      }
  }
}

/* Inverse of the unconstrained rational-quadratic spline with linear tails, one element:
 *   reference: unconstrained_rational_quadratic_spline, openvoice/transforms.py:50-97 (tails: identity outside
 *   [-tail, tail]; derivative logits padded with log(exp(1 - min_d) - 1) so that both edge derivatives are 1),
 *   rational_quadratic_spline(inverse=True), openvoice/transforms.py:100-188 (softmax widths/heights floored at
 *   min_w / min_h, cumulative knots rescaled to [-tail, tail] with the two ends pinned, bin search on the heights
 *   with the last edge nudged by 1e-6, quadratic root 2c / (-b - sqrt(b^2 - 4ac))).
 * uw, uh: NB logits each (already divided by sqrt(filter_channels), modules.py:499-502); ud: NB - 1 logits. */
#define RQ_MAX_BINS 32
static void rq_knots(const float* u, int nb, float minv, float tail, float* cum /* nb + 1 */) {
  float mx = u[0], sum = 0.f, e[RQ_MAX_BINS];
  for (int i = 1; i < nb; ++i) mx = u[i] > mx ? u[i] : mx;
  for (int i = 0; i < nb; ++i) { e[i] = expf(u[i] - mx); sum += e[i]; }
  float run = 0.f;
  cum[0] = -tail;
  for (int i = 0; i < nb; ++i) {
    run += minv + (1.f - minv * nb) * (e[i] / sum);
    cum[i + 1] = run * 2.f * tail - tail;
  }
  cum[nb] = tail;
}
static float softplus(float v) { return v > 20.f ? v : log1pf(expf(v)); }   /* torch.nn.functional.softplus threshold */

void ref_rq_spline_inverse_f32(const float* y, const float* uw, const float* uh, const float* ud, float* x, long n, int nb,
                               float tail, float min_w, float min_h, float min_d) {
  const float edge = logf(expf(1.f - min_d) - 1.f);
  for (long e = 0; e < n; ++e) {
    const float yv = y[e];
    if (!(yv >= -tail && yv <= tail)) { x[e] = yv; continue; }          /* linear tails: identity */
    float cw[RQ_MAX_BINS + 1], ch[RQ_MAX_BINS + 1];
    rq_knots(uw + e * nb, nb, min_w, tail, cw);
    rq_knots(uh + e * nb, nb, min_h, tail, ch);
    int b = 0;                                                          /* searchsorted on the heights */
    for (int i = 0; i <= nb; ++i) b += yv >= (i == nb ? ch[i] + 1e-6f : ch[i]);
    b = b - 1 < 0 ? 0 : (b - 1 > nb - 1 ? nb - 1 : b - 1);
    const float* de = ud + e * (nb - 1);
    const float d0 = min_d + softplus(b == 0 ? edge : de[b - 1]);
    const float d1 = min_d + softplus(b == nb - 1 ? edge : de[b]);
    const float bw = cw[b + 1] - cw[b], bh = ch[b + 1] - ch[b], delta = bh / bw;
    const float dy = yv - ch[b], s = d0 + d1 - 2.f * delta;
    const float a = dy * s + bh * (delta - d0), bq = bh * d0 - dy * s, c = -delta * dy;
    const float root = (2.f * c) / (-bq - sqrtf(bq * bq - 4.f * a * c));
    x[e] = root * bw + cw[b];
  }
}

/* Self-attention core with a +-window band of relative-position keys and values, one (batch, head) at a time:
 *   scores[t][s] = (q[t] / sqrt(dk)) . k[s] + (|s - t| <= w ? (q[t] / sqrt(dk)) . ek[s - t + w] : 0)
 *   scores[t][s] = -1e4 where mask[t] * mask[s] == 0;  p = softmax_s(scores)
 *   out[t] = sum_s p[t][s] v[s] + sum_{|r| <= w, 0 <= t + r < T} p[t][t + r] ev[r + w]
 * reference: MultiHeadAttention.attention, openvoice/attentions.py:264-329 (the relative logits are built there by
 * padding / reshaping a [T, 2T-1] tensor, _relative_position_to_absolute_position :331-345 and back :347-360; only the
 * 2w + 1 diagonals are ever non-zero, which is what is evaluated here).
 * q, k, v, out: [B][heads * dk][T] (the conv_q / conv_k / conv_v outputs as the reference lays them out);
 * ek, ev: [2w + 1][dk] (heads_share = True); mask [B][T]; row: scratch of T floats. */
void ref_rel_attention_f32(const float* q, const float* k, const float* v, const float* ek, const float* ev,
                           const float* mask, float* out, int B, int heads, int dk, int T, int w, float* row) {
  const float scale = 1.f / sqrtf((float)dk);
  for (int b = 0; b < B; ++b)
    for (int h = 0; h < heads; ++h) {
      const size_t base = ((size_t)b * heads + h) * dk * T;
      for (int t = 0; t < T; ++t) {
        float mx = -INFINITY;
        for (int s = 0; s < T; ++s) {
          float acc = 0.f;
          for (int d = 0; d < dk; ++d) acc += q[base + (size_t)d * T + t] * scale * k[base + (size_t)d * T + s];
          const int r = s - t;
          if (r >= -w && r <= w) {
            float rel = 0.f;
            for (int d = 0; d < dk; ++d) rel += q[base + (size_t)d * T + t] * scale * ek[(size_t)(r + w) * dk + d];
            acc += rel;
          }
          if (mask && mask[(size_t)b * T + t] * mask[(size_t)b * T + s] == 0.f) acc = -1e4f;
          row[s] = acc;
          mx = acc > mx ? acc : mx;
        }
        float sum = 0.f;
        for (int s = 0; s < T; ++s) { row[s] = expf(row[s] - mx); sum += row[s]; }
        for (int d = 0; d < dk; ++d) {
          float acc = 0.f;
          for (int s = 0; s < T; ++s) acc += row[s] * v[base + (size_t)d * T + s];
          for (int r = -w; r <= w; ++r)
            if (t + r >= 0 && t + r < T) acc += row[t + r] * ev[(size_t)(r + w) * dk + d];
          out[base + (size_t)d * T + t] = acc / sum;
        }
      }
    }
}
