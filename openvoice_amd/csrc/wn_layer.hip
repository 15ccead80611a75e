// One WaveNet layer in ONE launch (reference: openvoice/modules.py:192-209, commons.py:100-107):
//
//   x_in  = in_layer(h) + cond_layer(g)[layer]            Conv1d(H, 2H, K, 'same') + per-utterance bias
//   acts  = tanh(x_in[:H]) * sigmoid(x_in[H:])            fused_add_tanh_sigmoid_multiply
//   rs    = res_skip_layer(acts)                          Conv1d(H, 2H, 1)  (last layer: H rows, all skip)
//   h'    = (h + rs[:H]) * mask                           skip += rs[H:]
//
// The two-launch form (gate kernel -> `acts` in HBM -> res/skip kernel) leaves the chip badly filled at frame rate:
// 32 utterances x 861 frames are 672 tiles of 128 x 128 on 512 workgroup slots, the 1x1 launch is 54 us long and
// never reaches clock (profiles/r01_s37: 48 % / 74 % matrix-pipe busy).  Here one workgroup owns ALL 2H gate rows of
// a time tile, so `acts` never leaves the CU, and the tile width is a multiple of SIXTEEN columns chosen by the
// launcher so that utterances split into equal tiles that fill the CUs in whole rounds (861 frames x 32 = 8 tiles of
// 112 columns per utterance = 256 tiles on 256 CUs, 96 % useful columns, instead of 84 % at 32-column granularity).
//
// That is why this kernel uses v_mfma_f32_16x16x4_f32 (16-column fragments; 32 cycles per SIMD for 2 048 flop = the
// same 64 flop/cycle/SIMD as the 32x32x2 form) and a 768-thread workgroup, one per CU:
//   * 8 MATRIX waves (2 per SIMD, so one wave's LDS / barrier waits are covered by its SIMD partner).  Wave w owns 2H/8
//     (= 48) gate rows x all W columns in phase 1 and 48 res/skip rows x W columns in phase 2: RB x NB fragments of
//     16 x 16 = 4 accumulator VGPRs each (84 at W = 112).  A operands (weights) stream from L2 in fragment order,
//     one 1 KiB dwordx4 record per 4 k-steps per row block, requested one record ahead; no weight is read twice by a
//     workgroup.  B operands come from LDS, one conflict-free ds_read_b32 per fragment column per k-step, read one
//     k-step ahead.
//   * 4 LOADER waves stage h[32-channel chunk][t0-4 .. t0+W+4) through registers into a double-buffered LDS tile,
//     one s_barrier per chunk (conv1d_mfma.h explains why the staging loads must not share a vmcnt queue with the
//     weight stream).
//   Gate rows are packed so that a lane's 4 accumulator rows are (tanh a, tanh b, sigmoid a, sigmoid b) of two
//   channels: the gate is computed in registers and written to the `acts` LDS tile [H][W] that phase 2 reads as its B
//   operand.  LDS: 2 x 18 KB chunk buffers + 108 KB acts = 144 KB.
//
// h is read with a (K-1)/2 halo from neighbouring tiles, so h' goes to a SECOND buffer (the caller ping-pongs).
//
// ROW-SPLIT FORM for one or two utterances (MODE 1 + MODE 2, round 5).  One utterance of 861 frames is 54 tiles of 16
// columns: 54 of 256 CUs, each of them matrix-bound on all 2H gate rows (23 us of MFMA issue per layer, 34 us measured,
// 48 layers per conversion = a quarter of a batch-1 conversion).  There the launcher splits the ROWS three ways instead:
// launch 1 (MODE 1) = phase 1 + gate with one 16-row fragment per wave, workgroup (tile, r) owning fragments 8r .. 8r+7 (64
// gated channels) and writing them to the `acts` scratch in HBM; launch 2 (MODE 2) = `acts` tile -> LDS, phase 2 + epilogue,
// again one fragment per wave.  3 x the workgroups, a third of the k-loop each, the same weight records (a wave reads its
// fragment's 1 KiB out of the fused packing's 3 KiB record) and the same summation order per output element: the results
// are bit-identical to the fused launch (tests/test_gpu_wn_layer.py).  No cross-workgroup wait anywhere.  Measured
// (profiles/r05_s18): 11.4 + 7.2 us per layer against 34.4 fused (16.7 + 7.2 with the fused k-loop's prefetch distances).  (Keeping all six chunks of the 24-column tile resident so
// that the k-loop waits for one global-memory latency instead of six was measured too: no change -- the gate launch is bound
// by its k-loop, not by the staging.)  The gate launch's k-loop has its own form with deeper operand rings (below).
#include <hip/hip_runtime.h>

#include "openvoice_amd.h"

namespace ovk {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WNL_MW = 8;     // matrix waves
constexpr int WNL_NLD = 4;    // loader waves
constexpr int WNL_CH = 32;    // input channels per LDS chunk
constexpr int WNL_XS = 144;   // LDS row stride in floats: >= 128 + 8, and = 16 (mod 32) so that the two k-rows a
                              // half-wave reads (or the two channel rows it writes) fall into disjoint bank halves
constexpr int WNL_PADA = 4;   // staged halo on each side (16-byte aligned; >= (K-1)/2)

// tanh(t) * sigmoid(s) = sign(t) (1 - a) / ((1 + a)(1 + e^-s)),  a = e^-2|t|  -- two v_exp_f32 and one v_rcp_f32
// (1 ulp each); the libm forms cost ~10x the issue slots, and here nothing else runs on the CU during the gate.
__device__ __forceinline__ float wn_gate(float t, float s) {
  const float a = __builtin_amdgcn_exp2f(fabsf(t) * -2.8853900817779268f);   // e^-2|t|
  const float e = __builtin_amdgcn_exp2f(s * -1.4426950408889634f);          // e^-s  (inf for s << 0: result 0)
  const float r = __builtin_amdgcn_rcpf((1.f + a) * (1.f + e));
  return copysignf((1.f - a) * r, t);
}

// MODE 0: the fused layer.  MODE 1 / 2: the row-split form's gate launch / res-skip launch (blockIdx.y = row third).
template <int K, int H, int NB, int MODE>
__global__ __launch_bounds__(64 * (WNL_MW + WNL_NLD)) void wn_layer_kernel(const ov_wn_layer_params p) {
  static_assert((2 * H) % (16 * WNL_MW) == 0 && H % WNL_CH == 0, "H must be a multiple of 64");
  static_assert(K % 2 == 1 && (K - 1) / 2 <= WNL_PADA, "odd K <= 9");
  static_assert(NB >= 1 && NB <= 8, "tile width 16 .. 128 columns");
  constexpr int RBP = 2 * H / (16 * WNL_MW);  // 16-row fragments per wave in the weight packing (and of the fused layer)
  constexpr int RB = MODE == 0 ? RBP : 1;     // ... and per wave of this launch
  static_assert(MODE == 0 || NB == 1, "the row-split form exists for 16-column tiles");
  constexpr int NCH = H / WNL_CH;
  constexpr int PAD = (K - 1) / 2, PADA = WNL_PADA;
  constexpr int W = 16 * NB;
  constexpr int XS = WNL_XS, BUF = WNL_CH * XS;
  constexpr int ROWV = (W + 2 * PADA) / 4;     // 16-byte vectors staged per row
  constexpr int NITEM = WNL_CH * ROWV;
  constexpr int PER_LANE = (NITEM + 64 * WNL_NLD - 1) / (64 * WNL_NLD);
  constexpr int STEPS1 = 8 * K;                // k-steps (4 input channels x 1 tap) per chunk
  constexpr int RPC = 2 * K;                   // weight records (4 k-steps) per chunk
  constexpr int NR1 = NCH * RPC, NR2 = H / 16;

  __shared__ __attribute__((aligned(16))) float xs[MODE == 2 ? 4 : 2 * BUF];
  __shared__ __attribute__((aligned(16))) float acts[MODE == 1 ? 4 : H * XS];
  __shared__ __attribute__((aligned(16))) float msk[128];   // mask[b][t0 .. t0+W): read by the epilogue from LDS, so
                                                           // that no global load (vmcnt) sits between its stores

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int T = p.T;
  const uint32_t ld = (uint32_t)p.ld;
  const int b = (int)blockIdx.x / p.ntile;
  const int t0 = ((int)blockIdx.x - b * p.ntile) * W;

  bool is_loader = false;
#pragma unroll
  for (int i = 0; i < WNL_NLD; ++i) is_loader |= (wave == WNL_MW + i);
  if (is_loader) {
    // ================================ loader waves ===============================================================
    const int llane = (wave - WNL_MW) * 64 + lane;
    const float* __restrict__ xb = p.x + (int64_t)b * p.bstride;
    if (llane < W / 4) {                       // visible to the matrix waves after the first chunk barrier
      const int t = t0 + 4 * llane;
      f32x4 m = {0.f, 0.f, 0.f, 0.f};
      if (t < T) m = *reinterpret_cast<const f32x4*>(p.mask + (int64_t)b * p.mask_bstride + t);
#pragma unroll
      for (int r = 0; r < 4; ++r) m[r] = t + r < T ? m[r] : 0.f;
      *reinterpret_cast<f32x4*>(msk + 4 * llane) = m;
    }
    if constexpr (MODE == 2) {
      // the gated tile [H][W] of launch 1 -> LDS; columns >= T as zeros (they feed only columns that are never written)
      const float* __restrict__ ab = p.acts + (int64_t)b * p.bstride;
      constexpr int AV = W / 4, NA = H * AV;
#pragma unroll
      for (int i = 0; i < (NA + 64 * WNL_NLD - 1) / (64 * WNL_NLD); ++i) {
        const int idx = i * (64 * WNL_NLD) + llane;
        if (idx < NA) {
          const int row = idx / AV, c4 = idx - row * AV;
          const int t = t0 + 4 * c4;
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (t < T) v = *reinterpret_cast<const f32x4*>(ab + (uint32_t)row * ld + (uint32_t)t);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = t + r < T ? v[r] : 0.f;
          *reinterpret_cast<f32x4*>(acts + row * XS + 4 * c4) = v;
        }
      }
      __syncthreads();   // (acts in LDS)
      return;
    }
#pragma unroll 1
    for (int chunk = 0; chunk < NCH; ++chunk) {
      float* dst = xs + (chunk & 1) * BUF;
      f32x4 stg[PER_LANE];
      int nval[PER_LANE];
#pragma unroll
      for (int i = 0; i < PER_LANE; ++i) {
        const int idx = i * (64 * WNL_NLD) + llane;
        const int row = idx / ROWV, c4 = idx - row * ROWV;
        const int t = t0 - PADA + 4 * c4;            // multiple of 4: a vector is wholly < 0 or >= 0
        const bool ok = idx < NITEM && t >= 0 && t < T;
        const uint32_t goff = ok ? (uint32_t)(chunk * WNL_CH + row) * ld + (uint32_t)t : 0u;
        nval[i] = ok ? min(T - t, 4) : 0;            // ragged T: a vector may straddle the row end
        stg[i] = *reinterpret_cast<const f32x4*>(xb + goff);
      }
#pragma unroll
      for (int i = 0; i < PER_LANE; ++i) {
        const int idx = i * (64 * WNL_NLD) + llane;
        if (idx < NITEM) {
          const int row = idx / ROWV, c4 = idx - row * ROWV;
          f32x4 v = stg[i];
          const int n = nval[i];
          v[0] = n > 0 ? v[0] : 0.f;
          v[1] = n > 1 ? v[1] : 0.f;
          v[2] = n > 2 ? v[2] : 0.f;
          v[3] = n > 3 ? v[3] : 0.f;
          *reinterpret_cast<f32x4*>(dst + row * XS + 4 * c4) = v;
        }
      }
      __syncthreads();   // hand buffer (chunk & 1) to the matrix waves
    }
    if constexpr (MODE == 0) __syncthreads();   // (acts in LDS) -- the matrix waves' phase boundary
    return;
  }

  // ================================== matrix waves ==================================================================
  // phase timers (measurement only: p.dbg != NULL): shader-clock ticks per phase, per matrix wave
  const bool dbg = p.dbg != nullptr;
  unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast = dbg ? __builtin_readcyclecounter() : 0ull;
  const unsigned long long tstart = tlast;
  auto mark = [&](int q) {
    if (dbg) {
      const unsigned long long now = __builtin_readcyclecounter();
      tph[q] += now - tlast;
      tlast = now;
    }
  };
  const int g = lane >> 4, c = lane & 15;   // operand k-row / fragment column; accumulator rows 4g .. 4g+3
  // this wave's first 16-row fragment (both phases) and where its weight records start: the packing is
  // [wave of the fused layer][record][RBP fragments][lane]
  const int frag0 = MODE == 0 ? wave * RBP : WNL_MW * (int)blockIdx.y + wave;
  const int row0 = 16 * frag0;
  const int pw = frag0 / RBP, pi = frag0 - pw * RBP;
  f32x4 acc[RB][NB];
  if constexpr (MODE != 2) {
    const float* __restrict__ cb = p.cond ? p.cond + (int64_t)b * p.cond_bstride : nullptr;
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      f32x4 bv = *reinterpret_cast<const f32x4*>(p.b_in + row0 + 16 * i + 4 * g);
      if (cb) bv += *reinterpret_cast<const f32x4*>(cb + row0 + 16 * i + 4 * g);
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[i][j] = bv;
    }
  }

  // ---- phase 1: gate rows = W_in * h ------------------------------------------------------------------------------
  f32x4 a_cur[RB], a_nxt[RB];
  if constexpr (MODE != 2) {
  const f32x4* __restrict__ w1 =
      reinterpret_cast<const f32x4*>(p.w_in) + (size_t)pw * ((NR1 + 1) * RBP * 64) + (size_t)pi * 64;
  if constexpr (MODE == 0) {
#pragma unroll
    for (int i = 0; i < RB; ++i) a_cur[i] = w1[i * 64 + lane];
  }
  const int xl_off = g * XS + c + (PADA - PAD);
  if constexpr (MODE == 1) {
    // One MFMA per k-step and wave: the fused loop's prefetch distances (B operand one k-step, weights one record ahead) are
    // a third of what they are with three row blocks -- too short for the LDS and L2 latencies.  Same k-step order (the
    // summation order is what makes the pair bit-identical to the fused launch), deeper rings: the B operands of a whole
    // tap (8 k-steps) are read one tap ahead, WD - 1 weight records (4 k-steps each) are in flight.  Fully unrolled: every
    // ring slot is a compile-time register.
    constexpr int WD = 6;
    f32x4 wr[WD];
#pragma unroll
    for (int r = 0; r < WD - 1; ++r) wr[r] = w1[(size_t)r * (RBP * 64) + lane];
#pragma unroll
    for (int chunk = 0; chunk < NCH; ++chunk) {
      __syncthreads();   // loaders finished buffer (chunk & 1); we finished reading the other one
      const float* xl = xs + (chunk & 1) * BUF + xl_off;
      float bv[2][8];
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8) bv[0][s8] = xl[4 * s8 * XS];
#pragma unroll
      for (int tap = 0; tap < K; ++tap) {
        if (tap + 1 < K) {
#pragma unroll
          for (int s8 = 0; s8 < 8; ++s8) bv[(tap + 1) & 1][s8] = xl[4 * s8 * XS + tap + 1];
        }
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) {
          const int m = 8 * tap + s8, R = chunk * RPC + m / 4, u = m & 3;
          if (u == 0 && R + WD - 1 < NR1) wr[(R + WD - 1) % WD] = w1[(size_t)(R + WD - 1) * (RBP * 64) + lane];
          __builtin_amdgcn_sched_barrier(0);
          acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[R % WD][u], bv[tap & 1][s8], acc[0][0], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  } else {
#pragma unroll 1
  for (int chunk = 0; chunk < NCH; ++chunk) {
    __syncthreads();   // loaders finished buffer (chunk & 1); we finished reading the other one
    mark(chunk == 0 ? 0 : 2);
    const float* xl = xs + (chunk & 1) * BUF + xl_off;
    const f32x4* __restrict__ wc = w1 + (size_t)chunk * (RPC * RBP * 64);
    float bcur[NB], bnxt[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) bcur[j] = xl[16 * j];
#pragma unroll
    for (int m = 0; m < STEPS1; ++m) {      // k-step m: tap m / 8, channels 4 (m % 8) + g
      const int u = m & 3;
      if (u == 0) {                         // the record after the last one of the last chunk is zero padding
#pragma unroll
        for (int i = 0; i < RB; ++i) a_nxt[i] = (wc + (size_t)(m / 4 + 1) * (RBP * 64))[i * 64 + lane];
      }
      if (m + 1 < STEPS1) {
        const int tap = (m + 1) / 8, s = (m + 1) % 8;
#pragma unroll
        for (int j = 0; j < NB; ++j) bnxt[j] = xl[4 * s * XS + 16 * j + tap];
      }
      __builtin_amdgcn_sched_barrier(0);    // operand prefetches stay ahead of the MFMAs that hide them
#pragma unroll
      for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[i][u], bcur[j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (m + 1 < STEPS1) {
#pragma unroll
        for (int j = 0; j < NB; ++j) bcur[j] = bnxt[j];
      }
      if (u == 3) {
#pragma unroll
        for (int i = 0; i < RB; ++i) a_cur[i] = a_nxt[i];
      }
    }
    mark(1);
  }
  }
  }

  // ---- gate: rows 4g + {0, 1, 2, 3} of fragment q = tanh rows of channels 8q+g, 8q+g+4, then their sigmoid rows ---
  if constexpr (MODE == 1) {
    // row-split form, launch 1: the gated channels of this wave's fragment go to the `acts` scratch in HBM
    float* __restrict__ ag = p.acts + (int64_t)b * p.bstride + (uint32_t)(8 * frag0 + g) * ld;
    const int col = t0 + c;
    const f32x4 v = acc[0][0];
    if (col < T) {
      ag[col] = wn_gate(v[0], v[2]);
      ag[4 * ld + col] = wn_gate(v[1], v[3]);
    }
    return;
  }
  const f32x4* __restrict__ w2 =
      reinterpret_cast<const f32x4*>(p.w_rs) + (size_t)pw * ((NR2 + 1) * RBP * 64) + (size_t)pi * 64;
  const bool skip_rows = row0 >= H;                 // rows >= H of the res/skip conv
  const bool idle2 = p.last && !skip_rows;          // last layer: no residual rows (modules.py:203-207)
  f32x4 wr2[MODE == 2 ? NR2 : 1];
  if (!idle2) {
    if constexpr (MODE == 2) {
      // row-split form, launch 2: all NR2 records of this wave's fragment (12 KiB) requested now, while the loader waves
      // fetch the `acts` tile -- the k-loop below then waits for nothing
#pragma unroll
      for (int r = 0; r < NR2; ++r) wr2[r] = w2[(size_t)r * (RBP * 64) + lane];
    } else {
#pragma unroll
      for (int i = 0; i < RB; ++i) a_cur[i] = w2[i * 64 + lane];   // first res/skip record in flight during the gate
    }
  }
  // Phase 2 runs TRANSPOSED (operands swapped: D = acts^T W^T, same LDS reads and the same weight records): a lane
  // then holds 4 consecutive time columns of ONE row per fragment, so h / skip / out are touched with 16-byte
  // accesses, a quarter of the instructions of the row-per-register layout.  Its accumulators start at bias + h
  // (residual rows) / + skip (skip rows); fragment (i, j)'s operands are requested as soon as its gate is done, so
  // the loads fly under the remaining gates and the barrier.
  const int64_t boff = (int64_t)b * p.bstride;
  const float* __restrict__ src =
      (skip_rows ? p.skip + boff + (size_t)(row0 - H) * ld : p.x + boff + (size_t)row0 * ld) + (size_t)c * ld;
  const bool zero_src = skip_rows && p.first;       // first layer initialises the skip accumulator
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    const int ch = 8 * (frag0 + i) + g;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      if constexpr (MODE == 0) {
        const f32x4 v = acc[i][j];
        acts[ch * XS + 16 * j + c] = wn_gate(v[0], v[2]);
        acts[(ch + 4) * XS + 16 * j + c] = wn_gate(v[1], v[3]);
      }
      if (!idle2) {
        const int col = t0 + 16 * j + 4 * g;          // multiple of 4; col < T => col + 3 < ld (ld % 4 == 0)
        const uint32_t voff = (uint32_t)(16 * i) * ld + (uint32_t)(col < T ? col : 0);
        if (zero_src) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};     // (uniform branch: a select would wait for the load)
        else acc[i][j] = *reinterpret_cast<const f32x4*>(src + voff);
      }
    }
  }
  mark(3);
  __syncthreads();   // (acts in LDS)
  mark(4);
  if (idle2) return;
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    const float bv = p.b_rs[row0 + 16 * i + c];
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[i][j] += bv;
  }

  // ---- phase 2: res/skip rows = W_rs * acts -----------------------------------------------------------------------
  if constexpr (MODE == 2) {
    // one MFMA per k-step and wave (see the gate launch): every B operand of the 48 k-steps read up front, the weight
    // records already in registers; the same k-step order as the fused loop
    const float* al = acts + g * XS + c;
    float bv[4 * NR2];
#pragma unroll
    for (int q = 0; q < 4 * NR2; ++q) bv[q] = al[4 * q * XS];
#pragma unroll
    for (int q = 0; q < 4 * NR2; ++q) {
      __builtin_amdgcn_sched_barrier(0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[q], wr2[q / 4][q & 3], acc[0][0], 0, 0, 0);   // transposed
    }
  } else {
    const float* al = acts + g * XS + c;
    float bcur[NB], bnxt[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) bcur[j] = al[16 * j];
#pragma unroll 1
    for (int grp = 0; grp < NR2 / 4; ++grp) {       // 16 k-steps (64 channels) per iteration
      const float* ag = al + grp * 64 * XS;
      const f32x4* __restrict__ wg = w2 + (size_t)grp * (4 * RBP * 64);
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        const int u = m & 3;
        if (u == 0) {
#pragma unroll
          for (int i = 0; i < RB; ++i) a_nxt[i] = (wg + (size_t)(m / 4 + 1) * (RBP * 64))[i * 64 + lane];
        }
        // k-step m + 1 (the first of the next group when m = 15; past the end it re-reads row 4g of the last group)
        {
          const int s = (m + 1) % 16;
          const float* an = (m + 1 < 16) ? ag : (grp + 1 < NR2 / 4 ? ag + 64 * XS : ag);
#pragma unroll
          for (int j = 0; j < NB; ++j) bnxt[j] = an[4 * s * XS + 16 * j];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
          for (int j = 0; j < NB; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bcur[j], a_cur[i][u], acc[i][j], 0, 0, 0);   // transposed
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NB; ++j) bcur[j] = bnxt[j];
        if (u == 3) {
#pragma unroll
          for (int i = 0; i < RB; ++i) a_cur[i] = a_nxt[i];
        }
      }
    }
  }

  mark(5);
  // ---- epilogue: h' = (h + res) * mask -> out;  skip + rs -> skip -------------------------------------------------
  {
    float* __restrict__ dst =
        (skip_rows ? p.skip + boff + (size_t)(row0 - H) * ld : p.out + boff + (size_t)row0 * ld) + (size_t)c * ld;
    if (t0 + W <= T) {                               // whole tile inside the utterance: straight-line 16-byte stores
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        f32x4 mk = {1.f, 1.f, 1.f, 1.f};
        if (!skip_rows) mk = *reinterpret_cast<const f32x4*>(msk + 16 * j + 4 * g);
#pragma unroll
        for (int i = 0; i < RB; ++i)
          *reinterpret_cast<f32x4*>(dst + (uint32_t)(16 * i) * ld + (uint32_t)(t0 + 16 * j + 4 * g)) = acc[i][j] * mk;
      }
    } else {                                         // ragged last tile: columns >= T are never written
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int col = t0 + 16 * j + 4 * g;
        if (col >= T) continue;
        f32x4 mk = {1.f, 1.f, 1.f, 1.f};
        if (!skip_rows) mk = *reinterpret_cast<const f32x4*>(msk + 16 * j + 4 * g);
#pragma unroll
        for (int i = 0; i < RB; ++i) {
          const f32x4 v = acc[i][j] * mk;
          float* o = dst + (uint32_t)(16 * i) * ld + (uint32_t)col;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (col + r < T) o[r] = v[r];
        }
      }
    }
  }
  mark(6);
  if (dbg && lane == 0) {
    tph[7] = tstart;
#pragma unroll
    for (int q = 0; q < 8; ++q) p.dbg[(((size_t)blockIdx.y * gridDim.x + blockIdx.x) * WNL_MW + wave) * 8 + q] = tph[q];
  }
}

#define OV_WN_INST(NB) template __global__ void wn_layer_kernel<5, 192, NB, 0>(const ov_wn_layer_params);
OV_WN_INST(1) OV_WN_INST(2) OV_WN_INST(3) OV_WN_INST(4) OV_WN_INST(5) OV_WN_INST(6) OV_WN_INST(7) OV_WN_INST(8)
template __global__ void wn_layer_kernel<5, 192, 1, 1>(const ov_wn_layer_params);
template __global__ void wn_layer_kernel<5, 192, 1, 2>(const ov_wn_layer_params);

typedef void (*wn_kernel_fn)(const ov_wn_layer_params);

template <int K, int H>
static wn_kernel_fn wn_kernel_for(int nb) {
  switch (nb) {
    case 1: return wn_layer_kernel<K, H, 1, 0>;
    case 2: return wn_layer_kernel<K, H, 2, 0>;
    case 3: return wn_layer_kernel<K, H, 3, 0>;
    case 4: return wn_layer_kernel<K, H, 4, 0>;
    case 5: return wn_layer_kernel<K, H, 5, 0>;
    case 6: return wn_layer_kernel<K, H, 6, 0>;
    case 7: return wn_layer_kernel<K, H, 7, 0>;
    case 8: return wn_layer_kernel<K, H, 8, 0>;
  }
  return nullptr;
}

}  // namespace ovk

using namespace ovk;

#if !defined(__HIP_DEVICE_COMPILE__)
#include <atomic>

// Compute units of the current device (cached per ordinal).
static int wn_compute_units() {
  static std::atomic<int> cache[16];
  int dev = 0;
  const bool known = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 16;
  if (known) {
    const int v = cache[dev].load(std::memory_order_relaxed);
    if (v > 0) return v;
  }
  int cus = 256;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
  if (known) cache[dev].store(cus, std::memory_order_relaxed);
  return cus;
}

extern "C" {

int ov_wn_layer_supported(int H, int K) { return (H == 192 && K == 5) ? 1 : 0; }

size_t ov_wn_pack_size(int rows, int cin, int K) {
  if (rows <= 0 || cin <= 0 || K <= 0 || rows % (16 * WNL_MW) || cin % WNL_CH) return 0;
  const size_t rb = rows / (16 * WNL_MW), nr = (size_t)(cin / WNL_CH) * 2 * K;
  return (size_t)WNL_MW * (nr + 1) * rb * 64 * 4;
}

// [wave][record][row block][lane] x 4 k-steps; wave w = rows 16*rb*w ..; record = 4 k-steps of one 32-channel
// chunk, k-step m of a chunk = tap m / 8, channels 4 (m % 8) + (lane >> 4); lane & 15 = row in the block.
int ov_wn_pack_f32(const float* w, int rows, int cin, int K, float* dst) {
  const size_t n = ov_wn_pack_size(rows, cin, K);
  if (!w || !dst || n == 0) return OV_E_BADARG;
  const int rb = rows / (16 * WNL_MW), rpc = 2 * K, nr = (cin / WNL_CH) * rpc;
  for (size_t i = 0; i < n; ++i) dst[i] = 0.f;
  for (int wv = 0; wv < WNL_MW; ++wv)
    for (int R = 0; R < nr; ++R)
      for (int i = 0; i < rb; ++i)
        for (int l = 0; l < 64; ++l)
          for (int e = 0; e < 4; ++e) {
            const int chunk = R / rpc, m = (R % rpc) * 4 + e;
            const int tap = m / 8, ci = chunk * WNL_CH + 4 * (m % 8) + (l >> 4);
            const int row = (wv * rb + i) * 16 + (l & 15);
            dst[((((size_t)wv * (nr + 1) + R) * rb + i) * 64 + l) * 4 + e] = w[((size_t)row * cin + ci) * K + tap];
          }
  return OV_OK;
}

int ov_wn_layer_tile(int B, int T, int width) {
  if (width > 0) return (width % 16 == 0 && width <= 128) ? width : 0;
  // Equal tiles of 16*nb columns per utterance; the launch runs ceil(tiles / CUs) rounds of one tile per CU, a
  // tile costing its width plus a fixed part (chunk ramp, gate, epilogue: ~10 columns' worth).
  const int cus = wn_compute_units();
  long best = -1;
  int best_nb = 8;
  for (int nb = 8; nb >= 1; --nb) {
    const long tiles = (long)B * ((T + 16 * nb - 1) / (16 * nb));
    const long cost = ((tiles + cus - 1) / cus) * (16 * nb + 10);
    if (best < 0 || cost < best) { best = cost; best_nb = nb; }
  }
  return 16 * best_nb;
}

int ov_wn_layer_f32(const ov_wn_layer_params* pin, ov_stream_t stream) {
  if (!pin || !pin->x || !pin->out || !pin->skip || !pin->w_in || !pin->b_in || !pin->w_rs || !pin->b_rs || !pin->mask)
    return OV_E_BADARG;
  ov_wn_layer_params q = *pin;
  if (q.B <= 0 || q.T <= 0 || q.H <= 0 || q.K <= 0) return OV_E_BADARG;
  if (!ov_wn_layer_supported(q.H, q.K)) return OV_E_UNSUPPORTED;
  if (q.ld == 0) q.ld = q.T;
  if (q.ld < q.T || q.out == q.x) return OV_E_BADARG;
  if (q.mask_bstride == 0) q.mask_bstride = q.ld;
  if ((int64_t)q.H * q.ld > UINT32_MAX / 2) return OV_E_BADARG;   // per-utterance offsets are 32-bit in the kernel
  if ((q.ld % 4) || (q.bstride % 4) || (q.cond_bstride % 4) || (reinterpret_cast<uintptr_t>(q.x) & 15) ||
      (reinterpret_cast<uintptr_t>(q.w_in) & 15) || (reinterpret_cast<uintptr_t>(q.w_rs) & 15) ||
      (reinterpret_cast<uintptr_t>(q.b_in) & 15) || (reinterpret_cast<uintptr_t>(q.b_rs) & 15) ||
      (q.cond && (reinterpret_cast<uintptr_t>(q.cond) & 15)) || (reinterpret_cast<uintptr_t>(q.out) & 15) ||
      (reinterpret_cast<uintptr_t>(q.skip) & 15))
    return OV_E_ALIGN;
  // the loaders read the mask as 16-byte vectors of 4 columns starting at multiples of 4: every row must start
  // 16-byte aligned and own the whole last vector (columns T .. round_up(T, 4) - 1 are read and discarded)
  if ((reinterpret_cast<uintptr_t>(q.mask) & 15) || (q.mask_bstride % 4)) return OV_E_ALIGN;
  if (q.mask_bstride < (int64_t)((q.T + 3) / 4) * 4) return OV_E_BADARG;
  if (q.row_split != 0 && q.row_split != 1 && q.row_split != 3) return OV_E_BADARG;
  if (q.acts) {
    if (reinterpret_cast<uintptr_t>(q.acts) & 15) return OV_E_ALIGN;
    if (q.acts == q.x || q.acts == q.out || q.acts == q.skip) return OV_E_BADARG;
  } else if (q.row_split == 3) {
    return OV_E_BADARG;
  }
  if (q.row_split == 3 && q.dbg) return OV_E_BADARG;   // the phase timers (and their buffer's size) belong to the fused launch
  // Row-split pair (see the head of this file): worth it while three times the 16-column tiles fit the compute units in
  // little more than one round -- one or two utterances at frame rate (measured, profiles/r05_s18: batch 1 7.90 -> 7.2 ms
  // per conversion, batch 2 12.58 -> 12.42, batch 3 even, batch 4+ slower).
  const int64_t tiles16 = (int64_t)q.B * ((q.T + 15) / 16);
  const bool split = q.row_split == 3 || (q.row_split == 0 && q.acts && q.width == 0 && !q.dbg &&
                                          3 * tiles16 <= wn_compute_units() + wn_compute_units() / 3);
  if (split) {
    if (q.width != 0 && q.width != 16) return OV_E_BADARG;
    if (tiles16 > INT32_MAX) return OV_E_BADARG;
    q.width = 16;
    q.ntile = (q.T + 15) / 16;
    const dim3 grid((unsigned)tiles16, 3), block(64 * (WNL_MW + WNL_NLD));
    hipLaunchKernelGGL((wn_layer_kernel<5, 192, 1, 1>), grid, block, 0, static_cast<hipStream_t>(stream), q);
    hipLaunchKernelGGL((wn_layer_kernel<5, 192, 1, 2>), grid, block, 0, static_cast<hipStream_t>(stream), q);
    return hipGetLastError() == hipSuccess ? OV_OK : OV_E_LAUNCH;
  }
  const int width = ov_wn_layer_tile(q.B, q.T, q.width);
  if (width == 0) return OV_E_BADARG;
  q.width = width;
  q.ntile = (q.T + width - 1) / width;
  if ((int64_t)q.B * q.ntile > INT32_MAX) return OV_E_BADARG;
  wn_kernel_fn kernel = wn_kernel_for<5, 192>(width / 16);
  hipLaunchKernelGGL(kernel, dim3((unsigned)(q.B * q.ntile)), dim3(64 * (WNL_MW + WNL_NLD)), 0,
                     static_cast<hipStream_t>(stream), q);
  return hipGetLastError() == hipSuccess ? OV_OK : OV_E_LAUNCH;
}

}  // extern "C"
#endif
