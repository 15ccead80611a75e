// bf16 channels-last Conv1d for the HiFi-GAN generator (BASELINE.json configs[4]: "bf16 HiFi-GAN decoder",
// SURVEY.md section 8f item 3).  Activations live in HBM as bf16 (B, L, C) -- channels contiguous -- so that the
// 8 consecutive-k bf16 values a lane feeds to v_mfma_f32_32x32x16_bf16 are ONE aligned 16-byte LDS read; with
// the fp32 path's (B, C, L) layout the same operand would be 8 reads from 8 rows.  In bf16 the generator is
// HBM-bound (13.6 ms of matrix time against ~23 ms of tensor traffic at batch 64), so
// the kernel is organised around traffic: one workgroup produces ALL output channels of its time tile (x is
// read once, out written once; the residual and the MRF running sum start in the accumulators), fp32
// accumulation, fp32 bias, round-to-nearest-even on the way out.
//
// GEMM view: D[t][co] = sum_{tap, ci} X[t + tap*DIL - PAD][ci] * W[co][ci][tap];  M = time (A operand, from
// LDS), N = output channels (B operand: weights pre-packed in fragment order, streamed from L2), K = (tap, ci).
// Any consistent labelling of the 16 k-slots of one MFMA works because A and B use the same one: slot (h, i)
// of a lane (h = lane >> 5, i < 8) is input channel 16*kb + 8*h + i of the current chunk in both operands.
//
// Workgroup = 4 matrix waves + 2 loader waves, as in conv1d_mfma.h: the loaders stage 32-channel chunks of
// the time tile (+ halo, leaky-ReLU applied in fp32 on the way in, zero outside [0, L)) into a double-buffered
// LDS tile with an 80-byte row pitch (conflict-free ds_read_b128 for 32 consecutive rows).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

#include "openvoice_amd.h"

namespace ovk16 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

constexpr int CH = 32;          // input channels per LDS chunk = 2 MFMA k-blocks
constexpr int PITCH = 80;       // bytes per LDS row: 64 data + 16 pad

__device__ __forceinline__ float bf2f(uint16_t v) { return __uint_as_float((uint32_t)v << 16); }
__device__ __forceinline__ uint16_t f2bf(float f) {
  __bf16 h = (__bf16)f;         // v_cvt_pk_bf16_f32: round to nearest even
  uint16_t u;
  __builtin_memcpy(&u, &h, 2);
  return u;
}

// two floats -> packed bf16 pair (lo | hi << 16), round to nearest even: one v_cvt_pk_bf16_f32
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  typedef __bf16 b2 __attribute__((ext_vector_type(2)));
  const b2 r = __builtin_convertvector(f2{lo, hi}, b2);
  uint32_t u;
  __builtin_memcpy(&u, &r, 4);
  return u;
}

// leaky-ReLU of two packed bf16 values, evaluated in fp32 and rounded to nearest even -- the loaders' inner loop, 6
// instructions per pair: unpack (shift, and), v_pk_mul_f32 by the slope, max(v, slope * v) for 0 < slope < 1 (positive
// values keep their bits, negative ones take the product -- identical to `v > 0 ? v : v * slope`), v_cvt_pk_bf16_f32.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
template <bool LT1>
__device__ __forceinline__ uint32_t lrelu_bf16x2(uint32_t w, float slope) {
  f32x2 v = {__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)};
  const f32x2 sv = v * slope;
  if constexpr (LT1) {   // (asm: fmaxf() canonicalises both operands first -- three v_max_f32 per value)
    asm("v_max_f32 %0, %1, %2" : "=v"(v[0]) : "v"(v[0]), "v"(sv[0]));
    asm("v_max_f32 %0, %1, %2" : "=v"(v[1]) : "v"(v[1]), "v"(sv[1]));
  } else {
    v[0] = v[0] > 0.f ? v[0] : sv[0];
    v[1] = v[1] > 0.f ? v[1] : sv[1];
  }
  const bf16x2 r = __builtin_convertvector(v, bf16x2);
  uint32_t u;
  __builtin_memcpy(&u, &r, 4);
  return u;
}

// WM x WN fragments of 32x32 per matrix wave; WVT x WVC matrix waves (time x channel) per workgroup.
// DEEP: weight fragments are requested WDEPTH - 1 k-steps ahead instead of one (see the main loop); needs an even
// number of 32-channel chunks (Cin % 64 == 0).
// NLD: loader waves (2 or 4).
// SCH: 32-channel chunks per staging round (1 or 2): with 2 a round stages 64 channels -- half the chunk barriers
// (where the four matrix waves of a workgroup re-synchronise: ~1 500 cycles each, profiles/r02_s14) and half the
// identity rounds; needs Cin % 64 == 0.
template <int K, int DIL, int WM, int WN, int WVT, int WVC, bool DEEP, int NLD, int SCH>
__global__ __launch_bounds__(64 * (4 + NLD)) void conv1d_bf16cl_kernel(const ov_conv1d_bf16_params p) {
  static_assert(WVT * WVC == 4, "4 matrix waves");
  constexpr int TT = 32 * WM * WVT;             // time rows per workgroup
  constexpr int PAD = (K - 1) * DIL / 2;
  constexpr int R = TT + 2 * PAD;               // staged rows
  constexpr int PITCHV = SCH * 64 + 16;         // bytes per LDS row: 64 / 128 data + 16 pad (conflict-free b128 reads)
  constexpr int BUF = R * PITCHV;               // bytes per LDS buffer
  constexpr int VPR = 4 * SCH;                  // 16-byte vectors per staged row
  constexpr int NITEM = R * VPR;                // 16-byte vectors per round
  constexpr int PER_LANE = (NITEM + 64 * NLD - 1) / (64 * NLD);
  // one LDS block: the two chunk buffers during the main loop, the output staging tile in the epilogue
  constexpr int NBW = 32 * WN * WVC;                          // output columns per workgroup
  constexpr int SP = NBW * 2 + 16;                            // staging row pitch in bytes (see the epilogue)
  constexpr int SMEM = 2 * BUF > TT * SP ? 2 * BUF : TT * SP;
  __shared__ __attribute__((aligned(16))) unsigned char xs[SMEM];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * TT;
  const int L = p.L, Cin = p.Cin, Cout = p.Cout;
  const int nchunks = Cin / CH;

  bool is_loader = false;
#pragma unroll
  for (int i = 0; i < NLD; ++i) is_loader |= (wave == 4 + i);
  // rounds of the workgroup: the Cin/32 chunks of x (all taps), then -- instead of reading the residual and the
  // MRF running sum element-wise into the accumulators (2-byte loads, 128 memory instructions per wave and tile:
  // measured 5x slower than the conv itself) -- the Cout/32 chunks of `res` and of `add`, staged through the same
  // LDS tile and added on the matrix pipe with an identity B fragment (exact: bf16 x 1.0 into fp32).
  constexpr int NBT = WN * WVC;                             // 32-column tiles per workgroup (one N-block)
  const int nco_all = (Cout + 31) / 32;
  const int c_first = blockIdx.z * NBT;                     // first res / add chunk this N-block owns
  const int nco = min(NBT, nco_all - c_first);
  const int nco_r = (nco + SCH - 1) / SCH;                  // identity rounds per operand (SCH chunks each)
  const int rounds_x = nchunks / SCH, rounds_res = p.res ? nco_r : 0, rounds_add = p.add ? nco_r : 0;
  const int rounds = rounds_x + rounds_res + rounds_add;
  if (is_loader) {
    const int llane = (wave - 4) * 64 + lane;
    const float slope = p.in_slope;
    const bool slope_lt1 = slope > 0.f && slope < 1.f;
    u32x4 stg[PER_LANE];
    int ok[PER_LANE];
    // issue the 16-byte loads of round `rd` into registers (they stay in flight across the barrier)
    auto issue = [&](int rd) {
      const uint16_t* src;
      int cw, c, rlo, rhi;
      // c = first 32-channel chunk of the round (SCH chunks are staged side by side)
      if (rd < rounds_x) { src = p.x + (int64_t)b * L * Cin; cw = Cin; c = rd * SCH; rlo = 0; rhi = R; }
      else if (rd < rounds_x + rounds_res) { src = p.res + (int64_t)b * L * Cout; cw = Cout; c = c_first + (rd - rounds_x) * SCH; rlo = PAD; rhi = PAD + TT; }
      else { src = p.add + (int64_t)b * L * Cout; cw = Cout; c = c_first + (rd - rounds_x - rounds_res) * SCH; rlo = PAD; rhi = PAD + TT; }
#pragma unroll
      for (int i = 0; i < PER_LANE; ++i) {
        const int idx = i * (64 * NLD) + llane;
        const int row = idx / VPR, q = idx % VPR;
        const int t = t0 - PAD + row;
        const int ch = c * CH + q * 8;                        // (a partial last N-block: channels past the end read 0)
        ok[i] = (idx < NITEM && row >= rlo && row < rhi && t >= 0 && t < L && ch < cw) ? 1 : 0;
        const int64_t off = ok[i] ? ((int64_t)t * cw + ch) : 0;
        stg[i] = *reinterpret_cast<const u32x4*>(src + off);
      }
    };
    issue(0);
    for (int rd = 0; rd < rounds; ++rd) {
      unsigned char* dst = xs + (rd & 1) * BUF;
      const bool act = rd < rounds_x && slope != 1.f;
#pragma unroll
      for (int i = 0; i < PER_LANE; ++i) {
        const int idx = i * (64 * NLD) + llane;
        if (idx < NITEM) {
          u32x4 v = stg[i];
          if (!ok[i]) v = u32x4{0u, 0u, 0u, 0u};
          else if (act) {
            if (slope_lt1) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = lrelu_bf16x2<true>(v[e], slope);
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = lrelu_bf16x2<false>(v[e], slope);
            }
          }
          *reinterpret_cast<u32x4*>(dst + (idx / VPR) * PITCHV + (idx % VPR) * 16) = v;
        }
      }
      if (rd + 1 < rounds) issue(rd + 1);   // next round's loads fly while the matrix waves work on this one
      __syncthreads();
    }
    return;
  }

  // ---------------------------------- matrix waves ---------------------------------------------------
  // phase timers (measurement only: p.dbg != NULL): 0 set-up, 1 chunk barriers, 2 k-step loops, 3 identity rounds,
  // 4 staging the tile in LDS, 5 its barrier, 6 global stores; 7 = tiles (1)
  const bool dbg = p.dbg != nullptr;
  unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 1};
  unsigned long long tlast = dbg ? __builtin_readcyclecounter() : 0ull;
  auto mark = [&](int q) {
    if (dbg) {
      const unsigned long long now = __builtin_readcyclecounter();
      tph[q] += now - tlast;
      tlast = now;
    }
  };
  auto dump = [&]() {
    if (dbg && lane == 0) {
      unsigned long long* d =
          p.dbg + (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 32 + wave * 8;
#pragma unroll
      for (int q = 0; q < 8; ++q) d[q] = tph[q];
    }
  };
  const int wt = wave / WVC, wc = wave % WVC;
  const int half = lane >> 5, l31 = lane & 31;
  const int ntile0 = (blockIdx.z * WVC + wc) * WN;          // first 32-column output tile of this wave
  const int trow0 = wt * (32 * WM);                         // first time row (within the workgroup tile)
  const int ntiles_co = (Cout + 31) / 32;

  // The MFMAs run TRANSPOSED (operands swapped: D^T = W^T X^T, same products, same order), so a lane holds ONE time row
  // (trow0 + 32 i + lane & 31) and, per 32-column tile, 16 output channels in 4 groups of 4 consecutive ones:
  // channel 8 q + 4 (lane >> 5) + e for registers 4 q + e.  The epilogue packs a group into 8 bytes -- a quarter of the
  // LDS / memory instructions of the channel-per-lane layout (tile -> LDS was 3 500 of a tile's 17-33 k cycles).
  // Accumulators start at the bias; residual and running sum arrive as identity rounds (see above).
  f32x16 acc[WM][WN];
#pragma unroll
  for (int n = 0; n < WN; ++n) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int co = 32 * (ntile0 + n) + 8 * q + 4 * half;
      f32x4v bv = {0.f, 0.f, 0.f, 0.f};
      if (p.bias && co < Cout) bv = *reinterpret_cast<const f32x4v*>(p.bias + (int64_t)b * p.bias_bstride + co);
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][n][4 * q + e] = bv[e];
    }
  }

  // packed weights: record index ((nt * nchunks + c) * K + tap) * 2 + kb, 64 lanes x 16 bytes each
  const u32x4* __restrict__ wbase = reinterpret_cast<const u32x4*>(p.w);
  // Column tiles past Cout (a partial last N-block) read the packer's trailing zero record at every step: the weight
  // loads are UNCONDITIONAL.  (As `tile valid ? load : 0` each load sat in its own branch diamond, hipcc's s_waitcnt
  // insertion lost count at every join and waited vmcnt(0) -- for the request just issued, a full L2 round trip --
  // once per ring revolution: +5...11 % on the C >= 128, K >= 7 convs, profiles/r02_s14.)
  uint32_t widx[WN], wstep[WN];
#pragma unroll
  for (int n = 0; n < WN; ++n) {
    const bool valid = (ntile0 + n) < ntiles_co;
    widx[n] = (valid ? (uint32_t)(ntile0 + n) : (uint32_t)ntiles_co) * (uint32_t)(nchunks * K * 2) * 64u + (uint32_t)lane;
    wstep[n] = valid ? 64u : 0u;
  }

  // per-lane LDS byte offset of the A operand: row (trow0 + lane & 31), k-slot half
  const int xl_off = (trow0 + l31) * PITCHV + half * 16;

  mark(0);
  if constexpr (DEEP) {
    // Weight fragments are a continuous stream of 1 KiB records (chunk, tap, k-block), requested WDEPTH - 1 k-steps
    // ahead into a ring of WDEPTH register sets.  One k-step is only WM MFMAs (128-256 cycles of a SIMD's matrix
    // pipe) against a ~700-cycle L2 round trip: with the round-1 one-step-ahead request every k-step waited ~500
    // cycles for its fragment (C >= 128 ran at 32-37 % of the bf16 MFMA peak).  The loop is unrolled over chunk
    // PAIRS so that the ring slot of every step is a compile-time constant (2 * STEPS % WDEPTH == 0) and hipcc's
    // s_waitcnt insertion waits for the oldest request only (vmcnt(WDEPTH - 2)); a 4-deep queue with a run-time slot
    // (round 1) cost 24-40 VGPRs and a workgroup per CU.
    constexpr int STEPS = 2 * K, WDEPTH = 4;
    static_assert((2 * STEPS) % WDEPTH == 0, "ring slot must be static across a chunk pair");
    const int total = nchunks * STEPS;                          // records of one output tile; record `total` exists
    u32x4 bq[WDEPTH][WN];                                       // (next tile's first record, or the packer's zero record)
    auto wload = [&](int g, u32x4 (&dst)[WN]) {
      const int r = min(g, total);
#pragma unroll
      for (int n = 0; n < WN; ++n) dst[n] = wbase[(uint32_t)r * wstep[n] + widx[n]];
    };
#pragma unroll
    for (int q = 0; q < WDEPTH - 1; ++q) wload(q, bq[q]);
    for (int c = 0; c < nchunks; c += 2) {
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        if (SCH == 1 || cc == 0) {   // one hand-off per staging round (SCH chunks)
          __syncthreads();
          mark(1);
        }
        const unsigned char* xl = xs + (((c + cc) / SCH) & 1) * BUF + xl_off + ((c + cc) % SCH) * 64;
        u32x4 aq[2][WM];          // A operands of the current / next k-step, alternating by step parity (no copies)
#pragma unroll
        for (int i = 0; i < WM; ++i) aq[0][i] = *reinterpret_cast<const u32x4*>(xl + (32 * i) * PITCHV);   // tap 0, kb 0
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
          const int gs = cc * STEPS + s;                          // step inside the chunk pair: static
          wload((c + cc) * STEPS + s + WDEPTH - 1, bq[(gs + WDEPTH - 1) % WDEPTH]);
          if (s + 1 < STEPS) {
            const int tap = (s + 1) >> 1, kb = (s + 1) & 1;
#pragma unroll
            for (int i = 0; i < WM; ++i)
              aq[(s + 1) & 1][i] = *reinterpret_cast<const u32x4*>(xl + (32 * i + tap * DIL) * PITCHV + kb * 32);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int n = 0; n < WN; ++n) {
              bf16x8 av, bv;
              __builtin_memcpy(&av, &aq[s & 1][i], 16);
              __builtin_memcpy(&bv, &bq[gs % WDEPTH][n], 16);
              acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bv, av, acc[i][n], 0, 0, 0);   // transposed: D^T = W^T X^T
            }
          __builtin_amdgcn_sched_barrier(0);
        }
        mark(2);
      }
    }
  } else {
  // Weight fragments are requested one k-block ahead.
  u32x4 bcur[WN], bnxt[WN];
#pragma unroll
  for (int n = 0; n < WN; ++n) bcur[n] = wbase[widx[n]];

  int rec = 0;
  for (int c = 0; c < nchunks; ++c) {
    if (c % SCH == 0) {   // one hand-off per staging round (SCH chunks)
      __syncthreads();
      mark(1);
    }
    const unsigned char* xl = xs + ((c / SCH) & 1) * BUF + xl_off + (c % SCH) * 64;
    constexpr int STEPS = 2 * K;                            // (tap, k-block) pairs of one chunk
    u32x4 aq[2][WM];          // A operands of the current / next k-step, alternating by step parity (no copies)
#pragma unroll
    for (int i = 0; i < WM; ++i) aq[0][i] = *reinterpret_cast<const u32x4*>(xl + (32 * i) * PITCHV);   // tap 0, kb 0
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      ++rec;   // one zero record per output tile is appended by the packer for the final prefetch
#pragma unroll
      for (int n = 0; n < WN; ++n) bnxt[n] = wbase[(uint32_t)rec * wstep[n] + widx[n]];
      if (s + 1 < STEPS) {
        const int tap = (s + 1) >> 1, kb = (s + 1) & 1;
#pragma unroll
        for (int i = 0; i < WM; ++i)
          aq[(s + 1) & 1][i] = *reinterpret_cast<const u32x4*>(xl + (32 * i + tap * DIL) * PITCHV + kb * 32);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int n = 0; n < WN; ++n) {
          bf16x8 av, bv;
          __builtin_memcpy(&av, &aq[s & 1][i], 16);
          __builtin_memcpy(&bv, &bcur[n], 16);
          acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bv, av, acc[i][n], 0, 0, 0);   // transposed: D^T = W^T X^T
        }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int n = 0; n < WN; ++n) bcur[n] = bnxt[n];
    }
    mark(2);
  }
  }

  // ---- identity rounds: acc += res, acc += add ----------------------------------------------------------
  for (int rd = rounds_x; rd < rounds; ++rd) {
    __syncthreads();
#pragma unroll
    for (int sub = 0; sub < SCH; ++sub) {
    const int c = c_first + ((rd - rounds_x) % nco_r) * SCH + sub;   // 32-channel chunk of res / add in this round
    const unsigned char* xl = xs + (rd & 1) * BUF + xl_off + PAD * PITCHV + sub * 64;
#pragma unroll
    for (int n = 0; n < WN; ++n) {
      if (c != ntile0 + n) continue;                        // only the wave that owns these 32 output channels
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        u32x4 idw;                                          // B[k][n] = 1 iff k-slot (kb, half, e) is channel n
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int k0 = 16 * kb + 8 * half + 2 * e;
          idw[e] = (k0 == l31 ? 0x3f80u : 0u) | (k0 + 1 == l31 ? 0x3f800000u : 0u);
        }
        bf16x8 bv;
        __builtin_memcpy(&bv, &idw, 16);
#pragma unroll
        for (int i = 0; i < WM; ++i) {
          const u32x4 a = *reinterpret_cast<const u32x4*>(xl + (32 * i) * PITCHV + kb * 32);
          bf16x8 av;
          __builtin_memcpy(&av, &a, 16);
          acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bv, av, acc[i][n], 0, 0, 0);   // transposed: D^T = W^T X^T
        }
      }
    }
    }
  }

  // ---- epilogue: scale, round to bf16, store channels-last ------------------------------------------------
  // The workgroup's TT x NB output tile goes through LDS and leaves as 16-byte stores of
  // whole rows -- NB * 2 bytes contiguous per time row, 4 KiB per instruction round of the four matrix waves.  Storing
  // from the accumulator layout is 64 two-byte store instructions per lane and tile, each 2 x 64 bytes, and the CU's
  // store path is issue-bound (the regime profiles/r02_s13 measured on the fp32 kernel): staging took the k = 3 convs
  // 10-20 % down (profiles/r02_s14).  (MFMAs issued transposed + 8-byte stores from registers, tried earlier in round
  // 2, touch 32 rows x 16 B per instruction and were no faster.)
  mark(3);
  // ConvTranspose (phase_s > 1) takes the same path: its rows are packed phase-major (row = ph * C + co), so output
  // element (t * s + ph, co) of the [L * s][C] tensor sits at t * Cout + row -- exactly the channels-last address of a
  // plain conv with Cout = s * C rows.  (Until round 4 it left as 8-byte stores from the accumulator layout.)
  {
    const float scale = p.scale;
    const float oslope = p.out_slope > 0.f ? p.out_slope : 1.f;   // 0 (old callers) = none
    unsigned char* stg_out = xs;
    __syncthreads();   // every matrix wave is done reading the chunk buffers (the loader waves have exited; a barrier
                       // counts the live waves only)
    // a lane writes its time row's 4-channel groups as 8-byte pieces (ds_write_b64)
#pragma unroll
    for (int n = 0; n < WN; ++n) {
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        unsigned char* rowp = stg_out + (trow0 + 32 * i + l31) * SP + (32 * (wc * WN + n) + 4 * half) * 2;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = acc[i][n][4 * q + e] * scale;
            v[e] = v[e] > 0.f ? v[e] : v[e] * oslope;   // optional activation on the way out (its only consumer applies it)
          }
          *reinterpret_cast<u32x2*>(rowp + 16 * q) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        }
      }
    }
    mark(4);
    __syncthreads();
    mark(5);
    constexpr int V_ROW = NBW / 8;                            // 16-byte vectors per row
    constexpr int NVEC = TT * V_ROW;
    uint16_t* outb = p.out + ((int64_t)b * L * Cout + (int64_t)blockIdx.z * NBW);
    const int ncol = min(NBW, Cout - (int)blockIdx.z * NBW);   // a partial last N-block stores its real columns only
#pragma unroll
    for (int k = 0; k < NVEC / 256; ++k) {
      const int idx = k * 256 + tid;                          // tid < 256: the matrix waves
      const int row = idx / V_ROW, c8 = (idx % V_ROW) * 8;
      const int t = t0 + row;
      if (t < L && c8 < ncol) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(stg_out + row * SP + c8 * 2);
        *reinterpret_cast<u32x4*>(outb + (int64_t)t * Cout + c8) = v;
      }
    }
    mark(6);
    dump();
  }
}

template <int K, int DIL, int WM, int WN, int WVT, int WVC>
int launch(const ov_conv1d_bf16_params* p, hipStream_t stream) {
  constexpr int TT = 32 * WM * WVT, NB = 32 * WN * WVC;
  dim3 grid((p->L + TT - 1) / TT, p->B, (p->Cout + NB - 1) / NB);
  // Deep weight prefetch for Cout > 64 (C = 64 measured 5-20 % slower with it); layout 2 = the one-step-ahead request
  // of round 1 (measurement knob).  Four loader waves on every launch: one per SIMD, so that the four matrix waves of
  // a workgroup -- which re-synchronise at every chunk barrier -- all share their SIMD with the same company.  (With
  // the cheap packed leaky-ReLU pass, 4 loaders are 3-17 % faster than 2 on every plain K >= 7 shape as well:
  // profiles/r02_s14_bf16_conv_findings.txt; round 1's slower pass had it the other way round.)
  const bool deep = p->Cin % (2 * CH) == 0 && p->Cout > 64 && p->layout != 2;
  const dim3 grid2 = grid;
#define OV16_GO(DEEP_, NLD_, SCH_) \
  hipLaunchKernelGGL((conv1d_bf16cl_kernel<K, DIL, WM, WN, WVT, WVC, DEEP_, NLD_, SCH_>), grid2, dim3(64 * (4 + NLD_)), 0, stream, *p)
  // 64-channel staging rounds on the wide layout whenever Cin % 64 == 0 (layout 8: 32-channel rounds, A/B knob);
  // on the 64-column layout they measured 0-8 % slower (profiles/r02_s14) and are not instantiated
  const bool wide = p->Cin % (2 * CH) == 0 && p->layout != 8 && p->layout != 2;
  if constexpr (32 * WN * WVC > 64) {   // (the narrow layouts never take the deep path: not instantiated)
    if (deep && wide) { OV16_GO(true, 4, 2); goto launched; }
    if (deep) { OV16_GO(true, 4, 1); goto launched; }
  }
  OV16_GO(false, 4, 1);
launched:
#undef OV16_GO
  return hipGetLastError() == hipSuccess ? OV_OK : OV_E_LAUNCH;
}

// Wave layouts by output width.  Weights are streamed from L2 once per (wave, k-block) and reused for the WM time
// sub-tiles of the wave; from 128 output columns up a wave owns 128 time rows x 32 columns (WM = 4, WN = 1): half the
// weight traffic of a 64x64 wave tile, twice the (cheap) LDS operand reads.  >= 128 columns: 128 t x 128 co per
// workgroup (wider outputs in N-blocks); 64 -> 256 t x 64 co; 32 -> 256 t x 32 co (4 waves along time in both).
// Measured and not kept (profiles/r02_s14_bf16_conv_findings.txt): 64x64 and 128 t x 64 co wave tiles, 256-row wave
// tiles (half the weight traffic, one workgroup per CU), a weight ring of 8, LDS operand reads two k-steps ahead,
// persistent workgroups with the next tile's first chunk requested before the epilogue.
template <int K, int DIL>
int launch_by_width(const ov_conv1d_bf16_params* p, hipStream_t stream) {
  if (p->Cout > 64) return launch<K, DIL, 4, 1, 1, 4>(p, stream);
  if (p->Cout > 32) return launch<K, DIL, 2, 2, 4, 1>(p, stream);
  return launch<K, DIL, 2, 1, 4, 1>(p, stream);
}

// conv_post + tanh on the bf16 channels-last tensor (reference: openvoice/models.py:287-289):
// out[b][t] = tanh(sum_{c, j} w[c][j] * lrelu(x[b][t + j - (K-1)/2][c], slope)), fp32 out; HBM-bound (the last and
// largest tensor of the generator: 64 bytes in, 4 bytes out per sample).  A thread reads ITS row once (four 16-byte
// loads), forms the K per-tap dot products of that row against weights held in scalar registers (the weight index is
// a compile-time constant: s_load), leaves them in LDS, and a sample is the sum of K partials of K neighbouring rows.
// (Round 3's version read all K rows per thread -- 28 loads -- and its 224 weights one by one from LDS: 0.46 ms per
// batch-64 for a 0.16 ms tensor read.)
template <int C, int K>
__global__ __launch_bounds__(256) void conv_post_tanh_bf16_kernel(const uint16_t* __restrict__ x,
                                                                  const float* __restrict__ w,
                                                                  float* __restrict__ out, int L, float slope) {
  constexpr int H = (K - 1) / 2, NR = 256 + K - 1;
  __shared__ float ps[K][NR + 1];
  const int b = blockIdx.y, t0 = blockIdx.x * 256, tid = threadIdx.x;
  const uint16_t* xb = x + (int64_t)b * L * C;
  auto row_partials = [&](int i) {
    const int t = t0 - H + i;
    float p[K];
#pragma unroll
    for (int j = 0; j < K; ++j) p[j] = 0.f;
    if (t >= 0 && t < L) {
      const u32x4* row = reinterpret_cast<const u32x4*>(xb + (int64_t)t * C);
      u32x4 v[C / 8];
#pragma unroll
      for (int q = 0; q < C / 8; ++q) v[q] = row[q];
#pragma unroll
      for (int q = 0; q < C / 8; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float lo = __uint_as_float(v[q][e] << 16), hi = __uint_as_float(v[q][e] & 0xffff0000u);
          lo = lo > 0.f ? lo : lo * slope;
          hi = hi > 0.f ? hi : hi * slope;
#pragma unroll
          for (int j = 0; j < K; ++j) {
            p[j] = fmaf(w[(8 * q + 2 * e) * K + j], lo, p[j]);
            p[j] = fmaf(w[(8 * q + 2 * e + 1) * K + j], hi, p[j]);
          }
        }
    }
#pragma unroll
    for (int j = 0; j < K; ++j) ps[j][i] = p[j];
  };
  row_partials(tid);
  if (tid < K - 1) row_partials(256 + tid);
  __syncthreads();
  const int t = t0 + tid;
  if (t < L) {
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < K; ++j) acc += ps[j][tid + j];      // tap j of sample t reads row t + j - H = tile row tid + j
    out[(int64_t)b * L + t] = tanhf(acc);
  }
}

}  // namespace ovk16

using namespace ovk16;

extern "C" {

size_t ov_conv1d_bf16_pack_size(int Cout, int Cin, int K) {
  if (Cout <= 0 || Cin <= 0 || K <= 0 || Cin % CH != 0) return 0;
  const size_t ntiles = (Cout + 31) / 32;
  return (ntiles * (size_t)(Cin / CH) * K * 2 + 1) * 64 * 8;   // bf16 elements; + one zero record
}

int ov_conv1d_bf16_pack(const float* w, int Cout, int Cin, int K, uint16_t* dst) {
  if (!w || !dst || Cout <= 0 || Cin <= 0 || K <= 0 || Cin % CH != 0) return OV_E_BADARG;
  const int ntiles = (Cout + 31) / 32, nchunks = Cin / CH;
  std::memset(dst, 0, ov_conv1d_bf16_pack_size(Cout, Cin, K) * sizeof(uint16_t));
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
        for (int kb = 0; kb < 2; ++kb) {
          uint16_t* rec = dst + ((((size_t)nt * nchunks + c) * K + tap) * 2 + kb) * 64 * 8;
          for (int lane = 0; lane < 64; ++lane)
            for (int i = 0; i < 8; ++i) {
              const int co = 32 * nt + (lane & 31);
              const int ci = c * CH + 16 * kb + 8 * (lane >> 5) + i;
              if (co < Cout) rec[lane * 8 + i] = to_bf16(w[((size_t)co * Cin + ci) * K + tap]);
            }
        }
  return OV_OK;
}

int ov_conv_post_tanh_bf16(const uint16_t* x, const float* w, float* out, int B, int C, int L, int K, float in_slope,
                           ov_stream_t stream) {
  if (!x || !w || !out || B <= 0 || L <= 0 || B > 65535) return OV_E_BADARG;
  if (C != 32 || K != 7) return OV_E_UNSUPPORTED;          // dec.conv_post of every released config
  if (reinterpret_cast<uintptr_t>(x) & 15) return OV_E_ALIGN;
  dim3 grid((L + 255) / 256, B);
  hipLaunchKernelGGL((conv_post_tanh_bf16_kernel<32, 7>), grid, dim3(256), 0, static_cast<hipStream_t>(stream), x, w,
                     out, L, in_slope);
  return hipGetLastError() == hipSuccess ? OV_OK : OV_E_LAUNCH;
}

int ov_conv1d_bf16cl(const ov_conv1d_bf16_params* p, ov_stream_t stream) {
  if (!p || !p->x || !p->w || !p->out) return OV_E_BADARG;
  if (p->B <= 0 || p->L <= 0 || p->Cin <= 0 || p->Cout <= 0 || p->B > 65535) return OV_E_BADARG;
  if (p->Cin % CH != 0 || p->Cout % 32 != 0) return OV_E_UNSUPPORTED;
  if (p->phase_s > 1 && (p->Cout % p->phase_s != 0 || (p->Cout / p->phase_s) % 32 != 0 || p->res || p->add))
    return OV_E_UNSUPPORTED;

  if ((reinterpret_cast<uintptr_t>(p->x) & 15) || (reinterpret_cast<uintptr_t>(p->w) & 15) ||
      (reinterpret_cast<uintptr_t>(p->out) & 15) || (p->bias && ((reinterpret_cast<uintptr_t>(p->bias) & 15) || p->bias_bstride % 4)))
    return OV_E_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
#define OV16_CASE(KK, DD) if (p->K == KK && p->dil == DD) return launch_by_width<KK, DD>(p, st);
  OV16_CASE(3, 1) OV16_CASE(3, 3) OV16_CASE(3, 5)
  OV16_CASE(7, 1) OV16_CASE(7, 3) OV16_CASE(7, 5)
  OV16_CASE(11, 1) OV16_CASE(11, 3) OV16_CASE(11, 5)
#undef OV16_CASE
  return OV_E_UNSUPPORTED;
}

}  // extern "C"
