// C ABI of libopenvoice_amd.so: dispatch of the MFMA conv family + the small non-GEMM kernels.
// Signatures and reference citations: include/openvoice_amd.h.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "conv1d_mfma.h"
#include "openvoice_amd.h"

namespace ovk {

// ---------------------------------------------------------------------------------------------
// conv_post + tanh (C_out = 1): not GEMM-shaped (M = 1), HBM-bound.  Each thread produces four
// consecutive samples from three aligned 16-byte loads per input channel; the [C][K] weight sits
// in LDS.  reference: openvoice/models.py:287-289.
// ---------------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(256) void conv_post_tanh_vec_kernel(const float* __restrict__ x,
                                                                 const float* __restrict__ w,
                                                                 float* __restrict__ out, int C, int L,
                                                                 float slope, const int32_t* __restrict__ col_limit,
                                                                 int limit_scale) {
  constexpr int PAD = (K - 1) / 2;
  static_assert(PAD <= 4, "halo must fit one float4 on each side");
  extern __shared__ float wsm[];
  for (int i = threadIdx.x; i < C * K; i += 256) wsm[i] = w[i];
  __syncthreads();
  const int b = blockIdx.y;
  const int64_t t = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (t >= L) return;
  if (col_limit && t >= (int64_t)col_limit[b] * limit_scale) {   // beyond the utterance: silence, x is not read
    *reinterpret_cast<f32x4*>(out + (int64_t)b * L + t) = f32x4{0.f, 0.f, 0.f, 0.f};
    return;
  }
  const float* xb = x + (int64_t)b * C * L;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int c = 0; c < C; ++c) {
    const float* xr = xb + (int64_t)c * L + t;
    f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
    if (t >= 4) lo = *reinterpret_cast<const f32x4*>(xr - 4);
    const f32x4 mid = *reinterpret_cast<const f32x4*>(xr);
    if (t + 4 < L) hi = *reinterpret_cast<const f32x4*>(xr + 4);
    float v[12] = {lo[0], lo[1], lo[2], lo[3], mid[0], mid[1], mid[2], mid[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
    for (int i = 0; i < 12; ++i) v[i] = lrelu(v[i], slope);
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const float wj = wsm[c * K + j];
#pragma unroll
      for (int o = 0; o < 4; ++o) acc[o] = fmaf(wj, v[4 + o + j - PAD], acc[o]);
    }
  }
  f32x4 r = {tanhf(acc[0]), tanhf(acc[1]), tanhf(acc[2]), tanhf(acc[3])};
  *reinterpret_cast<f32x4*>(out + (int64_t)b * L + t) = r;
}

__global__ __launch_bounds__(256) void conv_post_tanh_scalar_kernel(const float* __restrict__ x,
                                                                    const float* __restrict__ w,
                                                                    float* __restrict__ out, int C, int L,
                                                                    int K, float slope,
                                                                    const int32_t* __restrict__ col_limit,
                                                                    int limit_scale) {
  extern __shared__ float wsm[];
  for (int i = threadIdx.x; i < C * K; i += 256) wsm[i] = w[i];
  __syncthreads();
  const int b = blockIdx.y;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= L) return;
  if (col_limit && t >= (int64_t)col_limit[b] * limit_scale) {
    out[(int64_t)b * L + t] = 0.f;
    return;
  }
  const int pad = (K - 1) / 2;
  const float* xb = x + (int64_t)b * C * L;
  float acc = 0.f;
  for (int c = 0; c < C; ++c)
    for (int j = 0; j < K; ++j) {
      const int64_t tt = t + j - pad;
      if (tt >= 0 && tt < L) acc = fmaf(wsm[c * K + j], lrelu(xb[(int64_t)c * L + tt], slope), acc);
    }
  out[(int64_t)b * L + t] = tanhf(acc);
}

// ---------------------------------------------------------------------------------------------
// y[b][m] = bias[m] + w[m][:] . x[b][:]   -- one wave64 per output, butterfly reduction.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void linear_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ bias, float* __restrict__ y,
                                                     int M, int Kdim) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int b = blockIdx.y;
  if (m >= M) return;
  const float* wr = w + (int64_t)m * Kdim;
  const float* xr = x + (int64_t)b * Kdim;
  float acc = 0.f;
  for (int k = lane; k < Kdim; k += 64) acc = fmaf(wr[k], xr[k], acc);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if (lane == 0) y[(int64_t)b * M + m] = acc + (bias ? bias[m] : 0.f);
}

__global__ void frame_limits_kernel(const int64_t* __restrict__ lengths, int32_t* __restrict__ limits, int B, int T,
                                    int margin) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  int64_t n = lengths[b];
  n = (n < 0 ? 0 : n) + margin;
  limits[b] = (int32_t)(n < T ? n : T);
}

__global__ void sequence_mask_kernel(const int64_t* __restrict__ lengths, float* __restrict__ mask, int T,
                                     int ld) {
  const int b = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t < T) mask[(int64_t)b * ld + t] = t < lengths[b] ? 1.f : 0.f;
}

// rows of `ld` floats -> dense rows of `T` floats: the engine keeps frame-rate tensors in 16-byte aligned rows
// (T = 861 -> 864), the model seam returns the reference's dense [B, C, T] (openvoice/models.py:499)
__global__ void unpad_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int T, int ld) {
  const int64_t row = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t < T) dst[row * T + t] = src[row * ld + t];
}

// Framing for the spectrogram (reference: openvoice/mel_processing.py:54-58 reflect pad, :61-72 framing inside
// torch.stft): hops[b][c][u] = ypad[hop*u + c], ypad = y reflect-padded by `pad` samples on both sides.  With
// n_fft = 4*hop, frame t is hops[:, t .. t+3], so the windowed DFT becomes a 4-tap conv over u with `hop` input
// channels.  A 32-hop x `hop`-sample tile is read coalesced along samples, transposed through LDS and written
// coalesced along u.
__global__ __launch_bounds__(256) void frame_hops_kernel(const float* __restrict__ wave, float* __restrict__ hops,
                                                         int N, int hop, int pad, int U, int ld) {
  extern __shared__ float tile[];   // [32][hop + 1]
  const int b = blockIdx.y, u0 = blockIdx.x * 32;
  const int HS = hop + 1;
  const float* y = wave + (int64_t)b * N;
  for (int idx = threadIdx.x; idx < 32 * hop; idx += 256) {
    const int du = idx / hop, c = idx - du * hop;
    const int64_t i = (int64_t)(u0 + du) * hop + c - pad;     // index into the un-padded signal
    float v = 0.f;
    if (u0 + du < U && i > -(int64_t)N && i < 2 * (int64_t)N - 1) {
      const int64_t k = i < 0 ? -i : (i >= N ? 2 * ((int64_t)N - 1) - i : i);
      if (i >= -(int64_t)pad && i < (int64_t)N + pad) v = y[k];
    }
    tile[du * HS + c] = v;
  }
  __syncthreads();
  const int lane = threadIdx.x & 31, grp = threadIdx.x >> 5;   // 8 groups of 32 lanes, lanes along u
  for (int c = grp; c < hop; c += 8)
    if (u0 + lane < U) hops[((int64_t)b * hop + c) * ld + u0 + lane] = tile[lane * HS + c];
}

// Polyphase FIR over a mono waveform (the resampler of the audio boundary, see ov_polyphase_fir_f32): output t reads
// 2 * taps input samples starting at (t * Q) / P - taps + 1 with the weights of phase t % P.  One thread per output
// sample, float64 accumulation (the host restatement this is checked against computes in float64); the 2 * taps
// weights of a phase and the overlapping input windows of neighbouring threads come out of L1 / L2 -- 57 MFLOP for a
// 10 s file, not worth a tile.
__global__ __launch_bounds__(256) void polyphase_fir_kernel(const float* __restrict__ x, const double* __restrict__ h,
                                                            float* __restrict__ y, int64_t n_in, int64_t n_out, int P,
                                                            int Q, int taps) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n_out) return;
  const int64_t n = (t * Q) / P;                  // input sample at or before the output's position
  const double* __restrict__ w = h + (int64_t)(t % P) * (2 * taps);
  const int64_t first = n - taps + 1;
  double acc = 0.0;
  for (int j = 0; j < 2 * taps; ++j) {
    const int64_t i = first + j;
    if (i >= 0 && i < n_in) acc += w[j] * (double)x[i];
  }
  y[t] = (float)acc;
}

static conv_launch_fn find_variant(int K, int dil, int tile, int chunk, int vec, int epi, int nld) {
  const ConvVariant* tabs[] = {kVariantsA, kVariantsB, kVariantsC, kVariantsD, kVariantsE, kVariantsF, kVariantsS, kVariantsW};
  const int ns[] = {kVariantsACount, kVariantsBCount, kVariantsCCount, kVariantsDCount,
                    kVariantsECount, kVariantsFCount, kVariantsSCount, kVariantsWCount};
  for (unsigned t = 0; t < sizeof(ns) / sizeof(ns[0]); ++t)
    for (int i = 0; i < ns[t]; ++i) {
      const ConvVariant& v = tabs[t][i];
      if (v.K == K && v.dil == dil && v.tile == tile && v.chunk == chunk && v.vec == vec && v.epi == epi &&
          v.nld == nld)
        return v.fn;
    }
  return nullptr;
}

// Tuning choices, from the per-shape sweeps under profiles/ (tools/bench_convs.py):
//  * loader waves: 4 for 1x1 convs and for the 64- and 32-row tiles (a chunk is consumed in
//    1.7-2.5 us of matrix work there), 2 for 128x128; 1 was never best once the kernels fit 4
//    waves per SIMD;
//  * channels per LDS chunk: 32 wherever the double buffer still allows >= 2 workgroups per CU
//    (fewer chunk hand-offs: +1...10 %), 16 for the 32-row tiles.
static void loader_preference(int tile, int K, int out[3]) {
  // four loader waves (one per SIMD) wherever that instance exists: every tile and K measured faster or equal with
  // them (profiles/r02_s17_convs_loaders_2_vs_4.txt; round 1's sweep, on an earlier kernel, had 2 ahead at 128x128)
  (void)tile; (void)K;
  out[0] = 4; out[1] = 2; out[2] = 1;
}
static void chunk_preference(int tile, int K, int out[2]) {
  if (K != 1 && (tile == TILE_32x512 || tile == TILE_32x256)) { out[0] = 16; out[1] = 32; return; }
  out[0] = 32; out[1] = 16;
}

}  // namespace ovk

using namespace ovk;

extern "C" {

int ov_version(void) { return 209; }

// 0 in every shippable build; the measurement builds of scripts/exp_sync.sh (OV_EXP = 1 / 2: staging loads and / or
// barriers compiled out, numerically meaningless) report their number so that the Python binding can refuse them.
int ov_build_experiment(void) { return OV_EXP; }

int ov_conv1d_pack_rows(int Cout) { return (Cout + 127) / 128 * 128; }

size_t ov_conv1d_pack_size(int Cout, int Cin, int K) {
  if (Cout <= 0 || Cin <= 0 || K <= 0) return 0;
  const size_t mtiles = ov_conv1d_pack_rows(Cout) / 32;
  return mtiles * ((size_t)packed_units(Cin) * K + 1) * REC;
}

int ov_conv1d_pack_f32(const float* w, int Cout, int Cin, int K, float* dst) {
  if (!w || !dst || Cout <= 0 || Cin <= 0 || K <= 0) return OV_E_BADARG;
  const int mtiles = ov_conv1d_pack_rows(Cout) / 32;
  const int nu = packed_units(Cin);
  const size_t recs = (size_t)nu * K + 1;
  std::memset(dst, 0, ov_conv1d_pack_size(Cout, Cin, K) * sizeof(float));
  for (int mt = 0; mt < mtiles; ++mt)
    for (int U = 0; U < nu; ++U)
      for (int g = 0; g < K; ++g) {
        float* rec = dst + ((size_t)mt * recs + (size_t)U * K + g) * REC;
        for (int lane = 0; lane < 64; ++lane)
          for (int u = 0; u < 4; ++u) {
            const int s = 4 * g + u, pp = s / K, tap = s - pp * K;
            const int ci = UNIT * U + 2 * pp + (lane >> 5);
            const int co = 32 * mt + (lane & 31);
            if (co < Cout && ci < Cin) rec[lane * 4 + u] = w[((size_t)co * Cin + ci) * K + tap];
          }
      }
  return OV_OK;
}

int ov_conv1d_f32(const ov_conv1d_params* pin, ov_stream_t stream) {
  if (!pin || !pin->x || !pin->w || !pin->out) return OV_E_BADARG;
  ov_conv1d_params q = *pin;   // defaults filled in below; the kernels see the normalised copy
  ov_conv1d_params* p = &q;
  if (p->B <= 0 || p->Cin <= 0 || p->L <= 0 || p->M <= 0 || p->Cout <= 0 || p->K <= 0 || p->dil <= 0)
    return OV_E_BADARG;
  if (p->B > 65535) return OV_E_BADARG;
  const int epi = p->epi;
  if (epi < OV_EPI_LINEAR || epi > OV_EPI_MAGNITUDE) return OV_E_BADARG;
  const bool paired = epi == OV_EPI_GATE || epi == OV_EPI_POSTERIOR || epi == OV_EPI_MAGNITUDE;
  if (paired && (p->M % 64 != 0)) return OV_E_BADARG;
  if (epi == OV_EPI_POSTERIOR && !p->res) return OV_E_BADARG;
  if (epi == OV_EPI_RESSKIP && (!p->out2 || p->split % 32 != 0)) return OV_E_BADARG;
  if (!p->bias) return OV_E_BADARG;   // layers without a bias pass a zero vector (M floats)
  if (!paired && ((int64_t)p->Cout * (epi == OV_EPI_CONVT ? p->phase_s : 1)) % 32 != 0)
    return OV_E_UNSUPPORTED;          // rows are stored in whole 32-row fragments
  if ((epi == OV_EPI_GATE || epi == OV_EPI_POSTERIOR) && p->Cout % 32 != 0) return OV_E_UNSUPPORTED;
  if (epi == OV_EPI_CONVT && (p->phase_s <= 0 || 32 % p->phase_s != 0)) return OV_E_BADARG;
  const int64_t lout = (int64_t)p->L * (epi == OV_EPI_CONVT ? p->phase_s : 1);
  if (lout > INT32_MAX) return OV_E_BADARG;
  if (p->x_ld == 0) p->x_ld = p->L;
  if (p->out_ld == 0) p->out_ld = (int32_t)lout;
  if (p->mask_bstride == 0) p->mask_bstride = p->L;
  const int64_t lin = (int64_t)p->L + ((p->K % 2 == 0) ? (int64_t)(p->K - 1) * p->dil : 0);   // input columns read
  if (p->x_ld < lin || p->out_ld < lout || (p->mask && p->mask_bstride < p->L)) return OV_E_BADARG;
  // per-utterance offsets are 32-bit inside the kernels
  if ((int64_t)p->Cin * p->x_ld > UINT32_MAX || (int64_t)(p->M + 32) * p->out_ld > UINT32_MAX) return OV_E_BADARG;
  if (epi == OV_EPI_CONVT && (p->flags & OV_F_CONVT_GROUPED)) {
    const int need = p->phase_s == 8 ? 4 : 2;   // 16- / 8-byte interleaving stores
    if ((reinterpret_cast<uintptr_t>(p->out) & (4 * need - 1)) || (p->out_bstride % need) || (p->out_ld % need))
      return OV_E_ALIGN;
  }
  if ((reinterpret_cast<uintptr_t>(p->w) & 15)) return OV_E_ALIGN;
  if (p->col_limit && (p->col_limit_scale <= 0 || (reinterpret_cast<uintptr_t>(p->col_limit) & 3))) return OV_E_BADARG;
  if (p->col_limit && p->B > ovk::LIMIT_MAX_BATCH) p->col_limit = nullptr;   // documented: whole tensor
  if (p->chunk != 0 && p->chunk != 16 && p->chunk != 32) return OV_E_BADARG;
  if (p->tile < 0 || p->tile > 4 ||
      (p->loaders != 0 && p->loaders != 1 && p->loaders != 2 && p->loaders != 4))
    return OV_E_BADARG;
  int tile = TILE_128x128;
  if (p->tile > 0) tile = p->tile - 1;
  else if (p->M <= 32 && !paired) tile = TILE_32x512;
  else if (p->M <= 64 && !paired) tile = TILE_64x256;
  else if (!paired && epi == OV_EPI_LINEAR) {
    // Small launches (batch 1: what the reference API issues, openvoice/api.py:141-160): when the 128 x 128 tiling
    // leaves compute units idle (batch 1, generator stage 0: 108 workgroups on 256 CUs), the 32 x 256 tile -- four
    // times the workgroups per launch -- is 1.4-1.8x faster (profiles/r04_s16_convs_small_batch_tiles.txt: C = 256,
    // batch 1, k = 11 0.164 -> 0.092 ms); with one workgroup per CU or more, 128 x 128 wins.  (Also measured: the
    // 32 x 256 tile on a mostly empty second round -- batch 2, stage 1: 862 tiles for 512 slots -- is ~10 % faster per
    // launch at k >= 7, invisible end to end; not a rule.)  Falls back to 128 x 128 where no 32 x 256 instance exists.
    const long t128 = (long)p->B * ((p->M + 127) / 128) * ((p->L + 127) / 128);
    if (t128 < ovk::compute_units()) tile = TILE_32x256;
  }
  const bool can_vec = (p->x_ld % 4 == 0) && !(reinterpret_cast<uintptr_t>(p->x) & 15) && (p->x_bstride % 4 == 0);
  int pref[3];
  conv_launch_fn fn = nullptr;
  // ConvTranspose: phase counts 8 and 2 have their own instances (compile-time store pattern)
  // (grouped row order only: those instances skip the all-zero tap of each phase group and store 16 / 8 bytes)
  const bool grouped = epi == OV_EPI_CONVT && (p->flags & OV_F_CONVT_GROUPED);
  if (grouped && ((p->phase_s != 8 && p->phase_s != 2) || p->M % 64 != 0 || p->K != 3)) return OV_E_BADARG;
  const int kernel_epi = !grouped ? epi : (p->phase_s == 8 ? EPI_CONVT_S8 : EPI_CONVT_S2);
  // forced tile / loader count / chunk: exact match or OV_E_UNSUPPORTED (measurement knobs must not silently
  // fall back); otherwise the preferred tile, then 128x128, 16-byte staging before 4-byte.
  const int tiles_try[2] = {tile, p->tile > 0 ? tile : (int)TILE_128x128};
  int cpref[2];
  for (int ti = 0; ti < 2 && !fn; ++ti) {
    if (p->loaders) { pref[0] = pref[1] = pref[2] = p->loaders; }
    else loader_preference(tiles_try[ti], p->K, pref);
    if (p->chunk) { cpref[0] = cpref[1] = p->chunk; }
    else chunk_preference(tiles_try[ti], p->K, cpref);
    for (int vec = can_vec ? 1 : 0; vec >= 0 && !fn; --vec)
      for (int ci = 0; ci < 2 && !fn; ++ci)
        for (int li = 0; li < 3 && !fn; ++li)
          fn = find_variant(p->K, p->dil, tiles_try[ti], cpref[ci], vec, kernel_epi, pref[li]);
  }
  if (!fn) return OV_E_UNSUPPORTED;
  return fn(p, static_cast<hipStream_t>(stream));
}

int ov_frame_hops_f32(const float* wave, float* hops, int B, int N, int hop, int pad, int U, int ld,
                      ov_stream_t stream) {
  if (!wave || !hops || B <= 0 || N <= 0 || hop <= 0 || hop > 1024 || pad < 0 || pad >= N || U <= 0 || ld < U ||
      B > 65535)
    return OV_E_BADARG;
  dim3 grid((U + 31) / 32, B);
  hipLaunchKernelGGL(frame_hops_kernel, grid, dim3(256), (size_t)32 * (hop + 1) * sizeof(float),
                     static_cast<hipStream_t>(stream), wave, hops, N, hop, pad, U, ld);
  return hipGetLastError() == hipSuccess ? OV_OK : OV_E_LAUNCH;
}

int ov_polyphase_fir_f32(const float* x, const double* h, float* y, int64_t n_in, int64_t n_out, int P, int Q, int taps,
                         ov_stream_t stream) {
  if (!x || !h || !y || n_in <= 0 || n_out <= 0 || P <= 0 || Q <= 0 || taps <= 0 || taps > (1 << 20)) return OV_E_BADARG;
  if ((n_out + 255) / 256 > INT32_MAX) return OV_E_BADARG;
  hipLaunchKernelGGL(polyphase_fir_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, h, y, n_in, n_out, P, Q, taps);
  return hipGetLastError() == hipSuccess ? OV_OK : OV_E_LAUNCH;
}

int ov_conv_post_tanh_f32(const float* x, const float* w, float* out, int B, int C, int L, int K,
                          float in_slope, ov_stream_t stream) {
  return ov_conv_post_tanh_limited_f32(x, w, out, B, C, L, K, in_slope, nullptr, 0, stream);
}

int ov_frame_limits_i32(const int64_t* lengths, int32_t* limits, int B, int T, int margin, ov_stream_t stream) {
  if (!lengths || !limits || B <= 0 || T <= 0 || margin < 0) return OV_E_BADARG;
  hipLaunchKernelGGL(frame_limits_kernel, dim3((B + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                     lengths, limits, B, T, margin);
  return hipGetLastError() == hipSuccess ? OV_OK : OV_E_LAUNCH;
}

int ov_conv_post_tanh_limited_f32(const float* x, const float* w, float* out, int B, int C, int L, int K,
                                  float in_slope, const int32_t* col_limit, int col_limit_scale, ov_stream_t stream) {
  if (!x || !w || !out || B <= 0 || C <= 0 || L <= 0 || K <= 0 || (K & 1) == 0 || B > 65535) return OV_E_BADARG;
  if (col_limit && col_limit_scale <= 0) return OV_E_BADARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t smem = (size_t)C * K * sizeof(float);
  // the vector path decides "kept or zero" per 4 consecutive samples: a limit that is not a multiple of 4 samples
  // takes the per-sample path instead (never up to 3 samples computed from columns the generator left unwritten)
  const bool vec = (K == 7) && (L % 4 == 0) && !(reinterpret_cast<uintptr_t>(x) & 15) &&
                   !(reinterpret_cast<uintptr_t>(out) & 15) && (!col_limit || col_limit_scale % 4 == 0);
  if (vec) {
    dim3 grid((L / 4 + 255) / 256, B);
    hipLaunchKernelGGL(conv_post_tanh_vec_kernel<7>, grid, dim3(256), smem, st, x, w, out, C, L, in_slope, col_limit,
                       col_limit_scale);
  } else {
    dim3 grid((L + 255) / 256, B);
    hipLaunchKernelGGL(conv_post_tanh_scalar_kernel, grid, dim3(256), smem, st, x, w, out, C, L, K, in_slope,
                       col_limit, col_limit_scale);
  }
  return hipGetLastError() == hipSuccess ? OV_OK : OV_E_LAUNCH;
}

int ov_linear_f32(const float* x, const float* w, const float* bias, float* y, int B, int M, int Kdim,
                  ov_stream_t stream) {
  if (!x || !w || !y || B <= 0 || M <= 0 || Kdim <= 0 || B > 65535) return OV_E_BADARG;
  dim3 grid((M + 3) / 4, B);
  hipLaunchKernelGGL(linear_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), x, w, bias, y, M, Kdim);
  return hipGetLastError() == hipSuccess ? OV_OK : OV_E_LAUNCH;
}

int ov_sequence_mask_f32(const int64_t* lengths, float* mask, int B, int T, int ld, ov_stream_t stream) {
  if (!lengths || !mask || B <= 0 || T <= 0 || B > 65535 || (ld != 0 && ld < T)) return OV_E_BADARG;
  dim3 grid((T + 255) / 256, B);
  hipLaunchKernelGGL(sequence_mask_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), lengths, mask, T,
                     ld ? ld : T);
  return hipGetLastError() == hipSuccess ? OV_OK : OV_E_LAUNCH;
}

int ov_unpad_rows_f32(const float* src, float* dst, int rows, int T, int ld, ov_stream_t stream) {
  if (!src || !dst || rows <= 0 || T <= 0 || ld < T || rows > 65535 * 1024) return OV_E_BADARG;
  for (int r0 = 0; r0 < rows; r0 += 65535) {          // (grid.y limit)
    const int n = rows - r0 < 65535 ? rows - r0 : 65535;
    dim3 grid((T + 255) / 256, n);
    hipLaunchKernelGGL(unpad_rows_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), src + (int64_t)r0 * ld,
                       dst + (int64_t)r0 * T, T, ld);
  }
  return hipGetLastError() == hipSuccess ? OV_OK : OV_E_LAUNCH;
}

}  // extern "C"
