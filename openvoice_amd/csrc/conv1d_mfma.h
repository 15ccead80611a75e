// Implicit-GEMM Conv1d on the gfx950 fp32 matrix pipe (v_mfma_f32_32x32x2_f32), wave-specialised.
//
// GEMM view of y[b][co][t] = sum_{ci,j} W[co][ci][j] * act(x[b][ci][t + j*DIL - PAD]):
//   M = output rows (co), N = time, K_gemm = (ci, tap).
//
// One workgroup = four MATRIX waves (one per SIMD) that own an M_BLK x N_BLK output tile of one utterance + NLD
// LOADER waves (4 on every shipped instance: one per SIMD, so the four matrix waves -- which re-synchronise at every
// chunk barrier -- all share their SIMD with the same company; 8 wave64 in all).  Input channels are walked in
// chunks of CHUNK rows; the loaders bring chunk c+1 -- rows with their (K-1)*DIL receptive-field halo, leaky-ReLU
// applied on the way in, zero-filled outside [0, L), the staging items dealt round-robin over the loader lanes so a
// whole chunk is in flight at once -- from HBM through registers into one half of a double-buffered LDS tile while
// the matrix waves run the MFMAs of chunk c out of the other half; one s_barrier per chunk hands a buffer over.
// Every tap of every MFMA B operand is then a conflict-free ds_read_b32 at a compile-time offset.
//
// Why a separate loader wave: the A operand (weights) never touches LDS -- weights are pre-packed at
// load time into MFMA fragment order, so a wave fetches the fragments of 4 consecutive k-steps with
// one coalesced 1 KiB global_load_dwordx4 (L2-resident), prefetched one group ahead.  s_waitcnt
// vmcnt retires loads IN ORDER, so if the same wave also had the HBM staging loads in flight, every
// wait for an (L2-fast) weight fragment would drain the (HBM-slow) staging loads issued before it
// and expose the full HBM latency once per chunk; measured on MI355X that cost 25-60 % of the matrix
// pipe on the k=3 convs.  With the roles split, the matrix waves' vmcnt queue holds weight
// fragments only.
//
// Large launches are persistent: the launch puts as many workgroups on the chip as fit at once and each walks
// the flattened (utterance, time-tile, M-block) list -- dense per-utterance tile counts when a length-aware column
// limit is given (ov_conv1d_params.col_limit) -- in XCD-contiguous ranges.  The loaders run one chunk
// ahead across tile boundaries and the first weight record of the next tile is requested before the
// epilogue of the current one, so only the first tile of a workgroup pays the launch / first-round-trip
// cost (~9 us per workgroup, measured; 20-30 % of a k = 3 tile).
//
// fp32 MFMA is bit-exact fmaf-chain arithmetic (no TF32-style truncation on gfx950), so parity
// with the fp32 reference is limited only by summation order.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>

#include "openvoice_amd.h"

namespace ovk {

// Internal epilogue codes: OV_EPI_CONVT with the phase count known at compile time (the dispatcher maps
// phase_s = 8 / 2 to them; any other power of two takes the generic OV_EPI_CONVT instance).  Separate
// kernels rather than a runtime switch: the three store patterns inlined into one kernel cost 176-190 VGPRs
// (one workgroup per CU), each alone fits the 128-VGPR budget of the other epilogues.
constexpr int EPI_CONVT_S8 = 16;
constexpr int EPI_CONVT_S2 = 17;
constexpr bool is_convt(int epi) { return epi == OV_EPI_CONVT || epi == EPI_CONVT_S8 || epi == EPI_CONVT_S2; }

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int UNIT = 8;    // input channels per unrolled unit (4 ci-pairs x K taps)
constexpr int REC = 256;   // floats per packed-weight record (64 lanes x 4 k-steps)
constexpr int LB_MAX = 12;  // loader: at most this many 16-byte (or 4-byte) loads in flight per lane
constexpr int LIMIT_MAX_BATCH = 256;   // length-aware work lists (ov_conv1d_params.col_limit): utterances per launch

// units are padded to a multiple of 4 (the largest units-per-chunk of any kernel variant)
__host__ __device__ inline int packed_units(int cin) { return ((cin + UNIT - 1) / UNIT + 3) / 4 * 4; }

__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }

// ---- epilogue: accumulators (= bias + conv [+ res + add], see conv_preload) -> global ------------------
// C/D fragment: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
// Addressing: 64-bit per-utterance base pointers stay scalar; a lane carries ONE 32-bit offset
// (its column + its half's row offset) and the 16 per-register row offsets are scalar multiples of
// L, so loads/stores take the saddr + voffset form and nothing 64-bit lives in VGPRs.  The epilogue
// kind is a template parameter and every optional operand is tested once per 32x32 fragment, never
// per element, so the element loops are straight-line code with 16 loads in flight.
template <int EPI, int WM, int WN>
__device__ __forceinline__ void conv_epilogue(const ov_conv1d_params& p, f32x16 (&acc)[WM][WN], int b, int tcol0,
                                              int mtile0, int q, int lane, bool preloaded) {
  const uint32_t L = (uint32_t)p.L;        // valid columns
  const uint32_t LD = (uint32_t)p.out_ld;  // row stride of out / res / add / out2
  const uint32_t half = (uint32_t)lane >> 5;
  const uint32_t Cout = (uint32_t)p.Cout;
  const float scale = p.scale;
  const float* mrow = p.mask ? p.mask + (int64_t)b * p.mask_bstride : nullptr;
  float* outb = p.out + (int64_t)b * p.out_bstride;   // bias (+ bias_b) is already in acc: see conv_preload
  const uint32_t col0 = (uint32_t)tcol0 + ((uint32_t)lane & 31u);

  if constexpr (EPI == EPI_CONVT_S8 || EPI == EPI_CONVT_S2) {
    // Grouped row order (OV_F_CONVT_GROUPED, engine.convt_row_order): packed tile 2q holds the output phases
    // p < s/2 (taps x[t-1], x[t]), tile 2q+1 the phases p >= s/2 (taps x[t], x[t+1]) of the same output channels --
    // 8 channels x 4 phases per tile for s = 8, 32 channels x 1 phase for s = 2 -- so that the all-zero third tap
    // of each tile is skipped in the main loop.  q = pair index (mtile0 / 2).
    static_assert(WM == 2, "grouped ConvTranspose pairs two 32-row tiles per wave");
    const uint32_t Lout = LD;   // row stride of the upsampled output (>= L * s)
    if ((uint32_t)q * 64u >= Cout * (uint32_t)p.phase_s) return;
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const uint32_t col = col0 + 32u * j;
      if (col >= L) continue;
      if constexpr (EPI == EPI_CONVT_S8) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const f32x16 v = acc[i][j];
#pragma unroll
          for (int c = 0; c < 4; ++c) {       // channel 8q + 2c + half, phases 4i .. 4i+3: 16 contiguous bytes
            f32x4 o = {v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]};
            *reinterpret_cast<f32x4*>(outb + (size_t)((uint32_t)q * 8u + 2u * c + half) * Lout +
                                      (8u * col + 4u * i)) = o;
          }
        }
      } else {
        const f32x16 v0 = acc[0][j], v1 = acc[1][j];
#pragma unroll
        for (int r = 0; r < 16; ++r) {        // channel 32q + row-in-tile, phases 0 and 1: 8 contiguous bytes
          const uint32_t co = (uint32_t)q * 32u + 4u * half + (r & 3) + 8 * (r >> 2);
          f32x2 o = {v0[r], v1[r]};
          *reinterpret_cast<f32x2*>(outb + (co * Lout + 2u * col)) = o;
        }
      }
    }
    return;
  } else if constexpr (EPI == OV_EPI_MAGNITUDE) {
    static_assert(WM == 2, "magnitude pairs two 32-row tiles per wave");
    // q = pair index: packed tile 2q holds the real parts of output rows 32q.., tile 2q+1 the imaginary parts
    if ((uint32_t)q * 32u >= Cout) return;
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const uint32_t col = col0 + 32u * j;
      if (col >= L) continue;
      const uint32_t rbase = (uint32_t)q * 32u + 4u * half;
      const uint32_t voff = rbase * LD + col;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const uint32_t rr = (r & 3) + 8 * (r >> 2);
        const float re = acc[0][j][r], im = acc[1][j][r];
        if (rbase + rr < Cout) (outb + (size_t)rr * LD)[voff] = sqrtf(re * re + im * im + scale);
      }
    }
    return;
  } else if constexpr (EPI == OV_EPI_GATE || EPI == OV_EPI_POSTERIOR) {
    static_assert(WM == 2, "gate/posterior pair two 32-row tiles per wave");
    // q = pair index: packed tiles 2q (tanh | m), 2q+1 (sigmoid | logs)
    if ((uint32_t)q * 32u >= Cout) return;
    const float* resb = p.res ? p.res + (int64_t)b * p.res_bstride : nullptr;
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const uint32_t col = col0 + 32u * j;
      if (col >= L) continue;
      const float mk = mrow ? mrow[col] : 1.f;
      const uint32_t voff = ((uint32_t)q * 32u + 4u * half) * LD + col;
      const uint32_t rbase = (uint32_t)q * 64u + 4u * half;
      const f32x16 v0 = acc[0][j], v1 = acc[1][j];   // bias (+ bias_b) already inside (conv_preload)
      if constexpr (EPI == OV_EPI_GATE) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const uint32_t rr = (r & 3) + 8 * (r >> 2);
          (outb + (size_t)rr * LD)[voff] = tanhf(v0[r]) * (1.f / (1.f + expf(-v1[r])));
        }
      } else {
        f32x16 nz;
#pragma unroll
        for (int r = 0; r < 16; ++r) nz[r] = (resb + (size_t)((r & 3) + 8 * (r >> 2)) * LD)[voff];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const uint32_t rr = (r & 3) + 8 * (r >> 2);
          (outb + (size_t)rr * LD)[voff] = (v0[r] * mk + nz[r] * scale * expf(v1[r] * mk)) * mk;
        }
      }
    }
    return;
  } else {
    if constexpr (EPI == OV_EPI_LINEAR) {
      // Nothing to read back (every conv1 of the MRF, and every conv2: its residual is already in the accumulators):
      // stores only, back to back.  Kept apart from the general path on purpose -- there the optional operand loads
      // sit between the fragments' stores and the wait for a load (vmcnt counts in order) is a wait for every store
      // issued before it: one exposed write round trip per fragment (1 500 cycles per tile, phase timers r02 s13).
      if (!mrow && !(p.res && !preloaded) && !(p.add && !preloaded)) {
#pragma unroll
        for (int i = 0; i < WM; ++i) {
          const uint32_t mt = (uint32_t)(mtile0 + i);
          if (mt * 32u >= Cout) continue;
#pragma unroll
          for (int j = 0; j < WN; ++j) {
            const uint32_t col = col0 + 32u * j;
            if (col >= L) continue;
            const uint32_t voff = (mt * 32u + 4u * half) * LD + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) (outb + (size_t)((r & 3) + 8 * (r >> 2)) * LD)[voff] = acc[i][j][r] * scale;
          }
        }
        return;
      }
    }
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      const uint32_t mt = (uint32_t)(mtile0 + i);
      // rows are stored in whole 32-row fragments (the dispatcher checks Cout*phase_s % 32 == 0)
      if (mt * 32u >= (is_convt(EPI) ? Cout * (uint32_t)p.phase_s : Cout)) continue;
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const uint32_t col = col0 + 32u * j;
        if (col >= L) continue;
        const float mk = mrow ? mrow[col] : 1.f;
        const uint32_t rbase = mt * 32u + 4u * half;
        f32x16 v = acc[i][j];   // bias (+ bias_b) already inside (conv_preload)
        if constexpr (is_convt(EPI)) {
          const uint32_t s = (uint32_t)p.phase_s;
          const uint32_t Lout = LD;   // row stride of the upsampled output (>= L * s)
          {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const uint32_t row = rbase + (r & 3) + 8 * (r >> 2);
              const uint32_t co = row / s, ph = row - co * s;
              outb[co * Lout + s * col + ph] = v[r];
            }
          }
        } else {
          const uint32_t voff = rbase * LD + col;   // the lane's only per-element offset
          if constexpr (EPI == OV_EPI_LINEAR) {
            // `preloaded`: res and add already sit in the accumulators (conv_preload)
            const float* resb = (p.res && !preloaded) ? p.res + (int64_t)b * p.res_bstride : nullptr;
            const float* addb = (p.add && !preloaded) ? p.add + (int64_t)b * p.add_bstride : nullptr;
            const float mkv = (p.flags & OV_F_MASK_V) ? mk : 1.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] *= mkv;
            if (resb) {
#pragma unroll
              for (int r = 0; r < 16; ++r) v[r] += (resb + (size_t)((r & 3) + 8 * (r >> 2)) * LD)[voff];
            }
            if (addb) {
#pragma unroll
              for (int r = 0; r < 16; ++r) v[r] += (addb + (size_t)((r & 3) + 8 * (r >> 2)) * LD)[voff];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) (outb + (size_t)((r & 3) + 8 * (r >> 2)) * LD)[voff] = v[r] * scale;
          } else if constexpr (EPI == OV_EPI_RESSKIP) {
            // h / skip are already inside the accumulators (conv_preload)
            if (mt * 32u < (uint32_t)p.split) {        // residual rows: h = (h + v) * mask, in place
#pragma unroll
              for (int r = 0; r < 16; ++r) (outb + (size_t)((r & 3) + 8 * (r >> 2)) * LD)[voff] = v[r] * mk;
            } else {                                   // skip rows: accumulate (or initialise)
              float* o2 = p.out2 + (int64_t)b * p.out2_bstride;
              const uint32_t voff2 = voff - (uint32_t)p.split * LD;
#pragma unroll
              for (int r = 0; r < 16; ++r) (o2 + (size_t)((r & 3) + 8 * (r >> 2)) * LD)[voff2] = v[r];
            }
          } else {  // OV_EPI_COUPLE: out is x1, in place; the accumulators hold m + x1 (forward) / m - x1 (reverse)
            const float sg = scale > 0.f ? mk : -mk;
#pragma unroll
            for (int r = 0; r < 16; ++r) (outb + (size_t)((r & 3) + 8 * (r >> 2)) * LD)[voff] = v[r] * sg;
          }
        }
      }
    }
  }
}

// ---- accumulator initialisation -------------------------------------------------------------------
// The accumulators START at everything the epilogue would otherwise have to read back from memory and add:
//   every kind      bias[row] (+ bias_b[b][row])
//   LINEAR          + res (+ add)                 the ResBlock residual and the MRF running sum
//   RESSKIP         + h (residual rows) / + skip (skip rows, unless OV_F_OUT2_INIT)
//   COUPLE          + x1 (forward) / - x1 (reverse; the epilogue negates); x1 = res when given, else out (in place)
// so that the epilogue only scales / masks and stores.  All of these loads -- 16 x WM x WN per lane -- are issued
// back to back as ONE batch at the top of the tile, straight into the accumulator registers (no temporaries, no
// branches: out-of-range columns and padding rows are clamped to a valid address and simply never stored), where
// the wave is about to wait an HBM round trip for the loaders' first chunk anyway.  Round 1 issued them one 32x32
// fragment at a time behind per-fragment branches: 4-8 serialised HBM round trips per tile, which is what made
// every conv2 (residual) launch of the MRF 0.2 ms slower than its conv1 twin whatever the tap count.
// fp32 addition order changes (operands first instead of last): ~1 ulp.
// Returns true when the epilogue's operands are now in `acc` (uniform across the workgroup).
template <int EPI, int WM, int WN>
__device__ __forceinline__ bool conv_preload(const ov_conv1d_params& p, f32x16 (&acc)[WM][WN], int b, int tcol0,
                                             int mtile0, int lane) {
  const uint32_t half = (uint32_t)lane >> 5;
  const uint32_t L = (uint32_t)p.L, LD = (uint32_t)p.out_ld;
  const uint32_t col0 = (uint32_t)tcol0 + ((uint32_t)lane & 31u);
  constexpr bool kOperands = EPI == OV_EPI_LINEAR || EPI == OV_EPI_RESSKIP || EPI == OV_EPI_COUPLE;
  bool preloaded = false;
  if constexpr (kOperands) {
    const float* src = nullptr;          // batch base of the tensor the accumulators start from
    int64_t src_bs = 0;
    if constexpr (EPI == OV_EPI_LINEAR) {
      if (p.res && !(p.flags & OV_F_MASK_V)) { src = p.res; src_bs = p.res_bstride; }   // v*mask + res keeps the epilogue order
    } else if (EPI == OV_EPI_COUPLE && p.res) {
      src = p.res; src_bs = p.res_bstride;      // x1 read from another tensor, written to out (no latent copy)
    } else {
      src = p.out; src_bs = p.out_bstride;
    }
    preloaded = src != nullptr;
    if (preloaded) {
      const uint32_t nfrag = (uint32_t)p.Cout / 32u;     // whole fragments of real rows (dispatcher: Cout % 32 == 0)
      const float* srcb = src + (int64_t)b * src_bs;
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        const uint32_t mt = min((uint32_t)(mtile0 + i), nfrag - 1u);   // padding fragments read a real one, never stored
        const float* rowb = srcb;
        uint32_t mrow = mt * 32u;
        bool zero = false;
        if constexpr (EPI == OV_EPI_RESSKIP) {
          if (mrow >= (uint32_t)p.split) {               // skip rows live in out2, first row = split
            rowb = p.out2 + (int64_t)b * p.out2_bstride;
            mrow -= (uint32_t)p.split;
            zero = (p.flags & OV_F_OUT2_INIT) != 0;      // first layer: the accumulator is being initialised
          }
        }
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          const uint32_t col = min(col0 + 32u * j, L - 1u);
          const uint32_t voff = (mrow + 4u * half) * LD + col;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = (rowb + (size_t)((r & 3) + 8 * (r >> 2)) * LD)[voff];
          if constexpr (EPI == OV_EPI_RESSKIP) {
            if (zero) {
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            }
          }
          if constexpr (EPI == OV_EPI_COUPLE) {
            if (!(p.scale > 0.f)) {
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[i][j][r] = -acc[i][j][r];
            }
          }
        }
      }
    }
  }
  if (!preloaded) {
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  }
  // bias (+ per-utterance bias): L2-resident vectors, one value per accumulator row.  Rows of padding tiles get no
  // bias and are never stored.
  {
    const float* __restrict__ bias = p.bias;
    const float* __restrict__ bias_b = p.bias_b ? p.bias_b + (int64_t)b * p.bias_b_bstride : nullptr;
    const uint32_t row_limit = (EPI == OV_EPI_GATE || EPI == OV_EPI_POSTERIOR || EPI == OV_EPI_MAGNITUDE)
                                   ? 2u * (uint32_t)p.Cout
                                   : (is_convt(EPI) ? (uint32_t)p.Cout * (uint32_t)p.phase_s : (uint32_t)p.Cout);
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      const uint32_t rbase = (uint32_t)(mtile0 + i) * 32u + 4u * half;
      if (EPI != OV_EPI_MAGNITUDE && (uint32_t)(mtile0 + i) * 32u < row_limit) {
        f32x16 bv;
#pragma unroll
        for (int r = 0; r < 16; ++r) bv[r] = bias[rbase + (r & 3) + 8 * (r >> 2)];
        if (bias_b) {
#pragma unroll
          for (int r = 0; r < 16; ++r) bv[r] += bias_b[rbase + (r & 3) + 8 * (r >> 2)];
        }
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] += bv;
      }
    }
  }
  if constexpr (EPI == OV_EPI_LINEAR) {
    // second addend (the MRF running sum; 8 of the 72 MRF launches): one fragment's 16 loads at a time, so that
    // the temporaries stay at 16 registers
    if (preloaded && p.add) {
      const uint32_t nfrag = (uint32_t)p.Cout / 32u;
      const float* addb = p.add + (int64_t)b * p.add_bstride;
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        const uint32_t mt = min((uint32_t)(mtile0 + i), nfrag - 1u);
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          const uint32_t col = min(col0 + 32u * j, L - 1u);
          const uint32_t voff = (mt * 32u + 4u * half) * LD + col;
          f32x16 av;
#pragma unroll
          for (int r = 0; r < 16; ++r) av[r] = (addb + (size_t)((r & 3) + 8 * (r >> 2)) * LD)[voff];
          __builtin_amdgcn_sched_barrier(0);   // all 16 loads issued before the first add (hipcc otherwise pairs them off)
          acc[i][j] += av;
        }
      }
    }
  }
  return preloaded;
}

// K taps, dilation DIL; wave tile = (32*WM) x (32*WN); WVM x WVN matrix waves per workgroup;
// CHUNK input channels per LDS fill; VEC: 16-byte staging loads (needs x_ld % 4 == 0 and 16-byte
// aligned rows; L itself may be ragged); EPI: epilogue kind (OV_EPI_*); NLD: loader waves (the
// staging items of a chunk are dealt round-robin to them, so NLD x LB x 64 loads are in flight).
// The gate epilogue lands at 130 VGPRs on its own; it is held to 128 (4 waves per SIMD, 1-2 VGPRs
// spilled in the epilogue) because the third workgroup per CU is worth more than the spill costs.
// Staging kinds (template argument VEC): how the loader waves bring a chunk into LDS.
constexpr int STAGE_SCALAR = 0;   // 4-byte loads through registers (rows not 16-byte aligned)
constexpr int STAGE_VEC = 1;      // 16-byte loads through registers, leaky-ReLU applied on the way in

template <int K, int DIL, int WM, int WN, int WVM, int WVN, int CHUNK, int VEC, int EPI, int NLD>
__global__ __launch_bounds__(64 * (4 + NLD), (EPI == OV_EPI_GATE || EPI == EPI_CONVT_S8 || EPI == EPI_CONVT_S2) ? 4 : 1) void conv1d_mfma_kernel(const ov_conv1d_params p) {
  static_assert(WVM * WVN == 4, "4 matrix waves per workgroup");
  static_assert(VEC == STAGE_SCALAR || VEC == STAGE_VEC, "staging kind");
  static_assert(CHUNK % UNIT == 0 && (CHUNK / UNIT == 2 || CHUNK / UNIT == 4), "chunk = 2 or 4 units");
  constexpr int UPC = CHUNK / UNIT;
  constexpr int N_BLK = 32 * WN * WVN;
  // Odd K: 'same' padding, taps t - PAD ... t + PAD.  Even K (the framing conv of the spectrogram):
  // forward alignment, taps t ... t + (K-1)*DIL, i.e. no left halo.
  constexpr int PAD = (K % 2 == 1) ? (K - 1) * DIL / 2 : 0;
  constexpr int PADR = (K - 1) * DIL - PAD;
  constexpr int PADA = (PAD + 3) / 4 * 4;
  constexpr int XS = N_BLK + PADA + (PADR + 3) / 4 * 4;  // LDS row stride (floats), multiple of 4
  constexpr int XS4 = XS / 4;
  constexpr int NITEM = VEC ? CHUNK * XS4 : CHUNK * XS;
  constexpr int BUF = CHUNK * XS;       // floats per LDS buffer
  // loads in flight per loader lane: the whole chunk in one batch when that needs <= LB_MAX of them
  constexpr int PER_LANE = (NITEM + 64 * NLD - 1) / (64 * NLD);
  constexpr int LB = PER_LANE < LB_MAX ? PER_LANE : LB_MAX;
  constexpr int NBATCH = (PER_LANE + LB - 1) / LB;
  static_assert(NLD == 1 || NLD == 2 || NLD == 4, "1, 2 or 4 loader waves");

  __shared__ __attribute__((aligned(16))) float xs[2 * BUF];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = p.L, Cin = p.Cin;
  const int nunits = packed_units(Cin);
  const int nchunks = nunits / UPC;
  // work list: wid = (b * ntiles + tile) * mblocks + mblock -- the M-blocks of one time tile (same x) and the
  // neighbouring time tiles (shared halo lines) are neighbours in the list.
  const int ntiles = (L + N_BLK - 1) / N_BLK;
  const int mblocks = (p.M + 32 * WM * WVM - 1) / (32 * WM * WVM);
  int total = ntiles * mblocks * p.B;
  // Length-aware work list (ov_conv1d_params.col_limit): utterance b contributes only the time tiles that start
  // before its column limit; lim_pref[b] = tiles of the utterances before b, so the list stays dense and is dealt
  // evenly whatever the lengths are.  One wave computes the prefix sums (<= 256 utterances) once per workgroup.
  __shared__ int lim_pref[LIMIT_MAX_BATCH + 1];
  const bool limited = p.col_limit != nullptr;
  if (limited) {
    if (wave == 0) {
      int carry = 0;
      for (int base = 0; base < p.B; base += 64) {
        const int bb = base + lane;
        int n = 0;
        if (bb < p.B) {
          long long c = (long long)p.col_limit[bb] * p.col_limit_scale;
          c = c < 0 ? 0 : (c > L ? L : c);
          n = ((int)c + N_BLK - 1) / N_BLK;
        }
        int s = n;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const int v = __shfl_up(s, d, 64);
          if (lane >= d) s += v;
        }
        if (bb < p.B) lim_pref[bb + 1] = carry + s;
        carry += __shfl(s, 63, 64);
      }
      if (lane == 0) lim_pref[0] = 0;
    }
    __syncthreads();
    total = __builtin_amdgcn_readfirstlane(lim_pref[p.B]) * mblocks;
  }
  // work item -> (utterance, time tile, M-block)
  auto decode = [&](int w, int& ub, int& utile, int& umblk) {
    const int g = w / mblocks;
    umblk = w - g * mblocks;
    if (!limited) {
      ub = g / ntiles;
      utile = g - ub * ntiles;
      return;
    }
    int lo = 0, hi = p.B;                      // lim_pref[lo] <= g < lim_pref[hi]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (__builtin_amdgcn_readfirstlane(lim_pref[mid]) <= g) lo = mid; else hi = mid;
    }
    ub = lo;
    utile = g - __builtin_amdgcn_readfirstlane(lim_pref[lo]);
  };
  // XCD-contiguous order: block j runs on XCD j % 8 (observed placement, a speed assumption only), and every XCD has
  // its own L2.  XCD x therefore owns the contiguous eighth [x * total / 8, (x + 1) * total / 8) of the list and its
  // workgroups stride through it together, so that list neighbours -- which share the 128-byte lines at their tile
  // edges and, for several M-blocks, the whole x tile -- are in flight on the SAME L2 at the same time.  With the
  // round-robin order every tile edge line was fetched from the fabric twice (PMC: 1.5x the tensor bytes read).
  int wid0 = blockIdx.x, wend = total, wstride = gridDim.x;
  if (!(p.flags & OV_F_NO_XCD_MAP) && (gridDim.x & 7u) == 0) {
    const int xcd = blockIdx.x & 7;
    const int lo = (int)(((long long)xcd * total) >> 3);
    wend = (int)(((long long)(xcd + 1) * total) >> 3);
    wid0 = lo + (int)(blockIdx.x >> 3);
    wstride = gridDim.x >> 3;
  }
  if (wid0 >= wend) return;

  // Written as a chain of equalities on purpose: with `wave >= 4` hipcc (ROCm 7.2) allocates 12 more
  // VGPRs for the matrix path (132 instead of 120), which costs a wave per SIMD and 5-25 % on MI355X.
  bool is_loader = false;
#pragma unroll
  for (int i = 0; i < NLD; ++i) is_loader |= (wave == 4 + i);
#ifndef OV_EXP
#define OV_EXP 0   // measurement builds only (scripts/exp_sync.sh): 1 = no loads, no barriers; 2 = no loads;
#endif             // 3 = phase timers of the matrix waves written through p.out2 (scripts/conv_phases.sh)
  if (is_loader) {
    if (OV_EXP == 1) return;
    // ================================ loader waves ===============================================
    const float slope = p.in_slope;
    const uint32_t ldx = (uint32_t)p.x_ld;
    const int llane = (wave - 4) * 64 + lane;   // position among the NLD * 64 loader lanes
    // valid INPUT columns: L for 'same' convs; the forward-aligned even-K conv consumes (K-1)*DIL more
    const int Lin = L + ((K % 2 == 1) ? 0 : (K - 1) * DIL);
    int it = 0;
    for (int wid = wid0; wid < wend; wid += wstride) {
      int lb, ltile, lmblk;
      decode(wid, lb, ltile, lmblk);
      const int t0 = ltile * N_BLK;
      const float* __restrict__ xb = p.x + (int64_t)lb * p.x_bstride;
      for (int chunk = 0; chunk < nchunks; ++chunk, ++it) {
        float* dst = xs + (it & 1) * BUF;
#pragma unroll 1   // one batch of LB loads per lane in flight at a time: bounds the loader's VGPRs
        for (int bt = 0; bt < (OV_EXP ? 0 : NBATCH); ++bt) {
          if constexpr (VEC) {
            f32x4 stg[LB];
            int nval[LB];   // valid leading elements of each vector (0 = zero fill); a VGPR count,
                            // not lane masks, so nothing mask-shaped stays live across the loads
#pragma unroll
            for (int i = 0; i < LB; ++i) {
              const int idx = (bt * LB + i) * (64 * NLD) + llane;
              const int row = idx / XS4, c4 = idx - row * XS4;
              const int ci = chunk * CHUNK + row;
              const int t = t0 - PADA + 4 * c4;     // multiple of 4: a vector is wholly < 0 or >= 0
              const bool ok = idx < NITEM && ci < Cin && t >= 0 && t < Lin;
              const uint32_t goff = ok ? (uint32_t)ci * ldx + (uint32_t)t : 0u;   // always valid
              nval[i] = ok ? min(Lin - t, 4) : 0;   // ragged L: a vector may straddle the row end
              stg[i] = *reinterpret_cast<const f32x4*>(xb + goff);
            }
#pragma unroll
            for (int i = 0; i < LB; ++i) {
              const int idx = (bt * LB + i) * (64 * NLD) + llane;
              if (idx < NITEM) {
                f32x4 v = stg[i];
                const int n = nval[i];
                v[0] = n > 0 ? lrelu(v[0], slope) : 0.f;
                v[1] = n > 1 ? lrelu(v[1], slope) : 0.f;
                v[2] = n > 2 ? lrelu(v[2], slope) : 0.f;
                v[3] = n > 3 ? lrelu(v[3], slope) : 0.f;
                *reinterpret_cast<f32x4*>(dst + 4 * idx) = v;  // row*XS + 4*c4 == 4*idx
              }
            }
          } else {
            float stg[LB];
            int nval[LB];
#pragma unroll
            for (int i = 0; i < LB; ++i) {
              const int idx = (bt * LB + i) * (64 * NLD) + llane;
              const int row = idx / XS, c = idx - row * XS;
              const int ci = chunk * CHUNK + row;
              const int t = t0 - PADA + c;
              const bool ok = idx < NITEM && ci < Cin && t >= 0 && t < Lin;
              const uint32_t goff = ok ? (uint32_t)ci * ldx + (uint32_t)t : 0u;
              nval[i] = ok ? 1 : 0;
              stg[i] = xb[goff];
            }
#pragma unroll
            for (int i = 0; i < LB; ++i) {
              const int idx = (bt * LB + i) * (64 * NLD) + llane;
              if (idx < NITEM) dst[idx] = nval[i] ? lrelu(stg[i], slope) : 0.f;
            }
          }
        }
        __syncthreads();  // hand buffer (it & 1) to the matrix waves
      }
    }
    return;
  }

  // ================================== matrix waves ================================================
  const int wm = wave / WVN, wn = wave % WVN;
  const int recs_per_mtile = nunits * K + 1;
  int wid = wid0;
  int mblk, tile, b;
  decode(wid, b, tile, mblk);
  int mtile0 = (mblk * WVM + wm) * WM;

  // weight fragments: scalar base + per-lane 32-bit index (in 16-byte units), record stride 64
  const f32x4* __restrict__ wbase = reinterpret_cast<const f32x4*>(p.w);
  uint32_t widx[WM];
#pragma unroll
  for (int i = 0; i < WM; ++i) widx[i] = (uint32_t)(mtile0 + i) * (uint32_t)recs_per_mtile * 64u + (uint32_t)lane;

  // per-lane LDS offset of the B operand: row (lane>>5) of a ci pair, column n of this wave
  const int xl_off = (lane >> 5) * XS + wn * (32 * WN) + (lane & 31) + (PADA - PAD);

  f32x4 a_cur[WM], a_nxt[WM];
#pragma unroll
  for (int i = 0; i < WM; ++i) a_cur[i] = wbase[widx[i]];

#if OV_EXP == 3
  // phase timers (measurement build): ticks per matrix wave in 0 = tile set-up + accumulator preload, 1 = chunk
  // barriers, 2 = k-step loops, 3 = next-tile bookkeeping + epilogue; 4 = tiles
  unsigned long long tph[5] = {0, 0, 0, 0, 0};
  unsigned long long tlast = __builtin_readcyclecounter();
#define OV_MARK(q) { const unsigned long long now_ = __builtin_readcyclecounter(); tph[q] += now_ - tlast; tlast = now_; }
#else
#define OV_MARK(q)
#endif
  int it = 0;
  while (true) {
    f32x16 acc[WM][WN];
    const bool preloaded = conv_preload<EPI, WM, WN>(p, acc, b, tile * N_BLK + wn * (32 * WN), mtile0, lane);
    OV_MARK(0)

    int rec = 0;
    for (int chunk = 0; chunk < nchunks; ++chunk, ++it) {
      if (OV_EXP != 1) __syncthreads();  // loader finished buffer (it & 1); we finished reading the other one
      OV_MARK(1)
      const float* xl = xs + (it & 1) * BUF + xl_off;
      // One k-step = one ci pair x one tap = WM*WN MFMAs (256 cycles of matrix pipe).  The B operands
      // of k-step s+1 are read from LDS while the MFMAs of k-step s run (explicit double buffer,
      // pinned by sched_barrier): left to itself hipcc issues each ds_read right before the MFMA that
      // consumes it and the wave eats the LDS latency every k-step (-15...35 % on MI355X).
      constexpr int STEPS = UPC * 4 * K;
      float bcur[WN], bnxt[WN];
#pragma unroll
      for (int j = 0; j < WN; ++j) bcur[j] = xl[32 * j];   // k-step 0: unit 0, pair 0, tap 0
#pragma unroll
      for (int sa = 0; sa < STEPS; ++sa) {
        const int u = sa & 3;
        if (u == 0) {
          ++rec;  // the record after the last real one is zero padding written by the packer
#pragma unroll
          for (int i = 0; i < WM; ++i) a_nxt[i] = (wbase + (size_t)rec * 64)[widx[i]];
        }
        if (sa + 1 < STEPS) {
          const int uu = (sa + 1) / (4 * K), sn = (sa + 1) - uu * (4 * K);
          const int pp = sn / K, tap = sn - pp * K;
#pragma unroll
          for (int j = 0; j < WN; ++j) bnxt[j] = xl[(uu * UNIT + 2 * pp) * XS + 32 * j + tap * DIL];
        }
        // Pin the prefetches here: without this hipcc sinks the loads next to their first use.
        __builtin_amdgcn_sched_barrier(0);
        // tap of THIS k-step; grouped ConvTranspose: tile parity = phase group, whose third tap is all zeros
        const int tap_now = (sa % (4 * K)) % K;
#pragma unroll
        for (int m = 0; m < WM * WN; ++m) {
          const int i = m / WN, j = m % WN;
          if ((EPI == EPI_CONVT_S8 || EPI == EPI_CONVT_S2) && tap_now == ((i & 1) ? 0 : K - 1)) continue;
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[i][u], bcur[j], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (sa + 1 < STEPS) {
#pragma unroll
          for (int j = 0; j < WN; ++j) bcur[j] = bnxt[j];
        }
        if (u == 3) {
#pragma unroll
          for (int i = 0; i < WM; ++i) a_cur[i] = a_nxt[i];
        }
      }
      OV_MARK(2)
    }
    // next work item; its first weight record is in flight while the epilogue of this one runs
    const int nwid = wid + wstride;
    const bool more = nwid < wend;
    int nmblk = 0, ntile = 0, nb = 0;
    if (more) decode(nwid, nb, ntile, nmblk);
    const int nmtile0 = (nmblk * WVM + wm) * WM;
    uint32_t nwidx[WM];
#pragma unroll
    for (int i = 0; i < WM; ++i) nwidx[i] = (uint32_t)(nmtile0 + i) * (uint32_t)recs_per_mtile * 64u + (uint32_t)lane;
    if (more) {
#pragma unroll
      for (int i = 0; i < WM; ++i) a_cur[i] = wbase[nwidx[i]];
    }
    conv_epilogue<EPI, WM, WN>(p, acc, b, tile * N_BLK + wn * (32 * WN), mtile0, mblk * WVM + wm, lane, preloaded);
    OV_MARK(3)
#if OV_EXP == 3
    ++tph[4];
    if (!more && EPI == OV_EPI_LINEAR && p.out2 && lane == 0) {
      unsigned long long* dbg = reinterpret_cast<unsigned long long*>(p.out2) + ((size_t)blockIdx.x * 4 + wave) * 8;
#pragma unroll
      for (int q = 0; q < 5; ++q) dbg[q] = tph[q];
    }
#endif
    if (!more) break;
    wid = nwid; tile = ntile; mblk = nmblk; b = nb; mtile0 = nmtile0;
#pragma unroll
    for (int i = 0; i < WM; ++i) widx[i] = nwidx[i];
  }
}

// Workgroups of one kernel instance that fit on the chip at once (occupancy x CUs) on the CURRENT device.  Cached per
// (kernel instance, device ordinal): `cache` is that instance's zero-initialised array, so a process driving several
// GPUs sizes each launch for the device it is launched on, and concurrent first calls are benign (same value).
constexpr int OV_MAX_DEVICES = 16;
inline int resident_workgroups(const void* kernel, int block_threads, std::atomic<int>* cache) {
  int dev = 0;
  const bool known = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < OV_MAX_DEVICES;
  if (known) {
    const int v = cache[dev].load(std::memory_order_relaxed);
    if (v > 0) return v;
  }
  int per_cu = 0, slots = 512;   // 2 per CU on an MI355X
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) == hipSuccess &&
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block_threads, 0) == hipSuccess && per_cu > 0)
    slots = per_cu * prop.multiProcessorCount;
  if (known) cache[dev].store(slots, std::memory_order_relaxed);
  return slots;
}

// Compute units of the current device (cached per ordinal).
inline int compute_units() {
  static std::atomic<int> cache[OV_MAX_DEVICES];
  int dev = 0;
  const bool known = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < OV_MAX_DEVICES;
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

typedef int (*conv_launch_fn)(const ov_conv1d_params*, hipStream_t);

template <int K, int DIL, int WM, int WN, int WVM, int WVN, int CHUNK, int VEC, int EPI, int NLD>
int conv1d_launch(const ov_conv1d_params* p, hipStream_t stream) {
  constexpr int M_BLK = 32 * WM * WVM, N_BLK = 32 * WN * WVN;
  const int ntiles = (p->L + N_BLK - 1) / N_BLK;
  const int mblocks = (p->M + M_BLK - 1) / M_BLK;
  // Persistent launch: one workgroup per resident slot, each strides over the work list -- when every slot
  // gets >= 16 tiles, so that the +-1 tile imbalance stays under ~6 %, and the conv has no residual operand:
  // a persistent workgroup reads the residual fragment of its next tile with nothing else of its own to run,
  // while one-tile workgroups leave that wait to be covered by whichever workgroup the dispatcher starts next
  // (measured with the clocks ramped, profiles/r01_s39: conv2 launches 2-5 % faster non-persistent at C = 64 / 32,
  // +-1 % at C = 128; conv1 launches 0-4 % faster persistent).  Smaller launches (stage 0 of the generator: 6.75
  // tiles per slot, the frame-rate layers) use one workgroup per tile and leave the balancing to the hardware
  // dispatcher.  A positive tiles_per_wg forces ceil(total / tiles_per_wg) workgroups, a negative one the persistent
  // launch (tests, A/B measurements).
  const long total = (long)ntiles * mblocks * p->B;
  auto kernel = conv1d_mfma_kernel<K, DIL, WM, WN, WVM, WVN, CHUNK, VEC, EPI, NLD>;
  static std::atomic<int> slot_cache[OV_MAX_DEVICES];   // per kernel instance (one instantiation per variant)
  const int slots = resident_workgroups(reinterpret_cast<const void*>(kernel), 64 * (4 + NLD), slot_cache);
  const int tpw = p->tiles_per_wg > 0 ? p->tiles_per_wg : 1;
  const bool persistent = p->tiles_per_wg < 0 || (p->tiles_per_wg == 0 && p->res == nullptr && total >= 16L * slots);
  long nwg = persistent ? (long)slots : (total + tpw - 1) / tpw;
  if (nwg > total) nwg = total;
  // whole multiples of 8 workgroups: the kernel's XCD-contiguous work order needs the same count on every XCD
  // (workgroups whose share of the list is empty exit at once)
  if (!(p->flags & OV_F_NO_XCD_MAP) && nwg >= 8) nwg = (nwg + 7) / 8 * 8;
  hipLaunchKernelGGL(kernel, dim3((unsigned)nwg), dim3(64 * (4 + NLD)), 0, stream, *p);
  return hipGetLastError() == hipSuccess ? OV_OK : OV_E_LAUNCH;
}

// Tile ids used by the dispatcher (ov_conv1d_params.tile = id + 1 forces one).
enum { TILE_128x128 = 0, TILE_64x256 = 1, TILE_32x512 = 2, TILE_32x256 = 3 };

struct ConvVariant {
  int K, dil, tile, chunk, vec, epi, nld;
  conv_launch_fn fn;
};

// Wave-tile template arguments <WM, WN, WVM, WVN> of each tile id.
#define OV_TILE_128x128 2, 2, 2, 2
#define OV_TILE_64x256 2, 2, 1, 4
#define OV_TILE_32x512 1, 4, 1, 4
#define OV_TILE_32x256 1, 2, 1, 4

// An instantiation list is a macro LIST(X) expanding to X(K, DIL, TILE, CHUNK, VEC, EPI, NLD) items;
// OV_DEFINE_VARIANTS(table, LIST) emits the explicit kernel instantiations (seen by the host and the
// device pass) and the host-side dispatch table `table` / `table##Count`.
#define OV_X_INST(K, DIL, TILE, CHUNK, VEC, EPI, NLD)                                                        \
  template __global__ void conv1d_mfma_kernel<K, DIL, OV_TILE_##TILE, CHUNK, VEC, EPI, NLD>(                  \
      const ov_conv1d_params);
#define OV_X_ROW(K, DIL, TILE, CHUNK, VEC, EPI, NLD)                                                         \
  {K, DIL, TILE_##TILE, CHUNK, VEC, EPI, NLD,                                                                \
   conv1d_launch<K, DIL, OV_TILE_##TILE, CHUNK, VEC, EPI, NLD>},
#if defined(__HIP_DEVICE_COMPILE__)
#define OV_DEFINE_VARIANTS(table, LIST) LIST(OV_X_INST)
#else
#define OV_DEFINE_VARIANTS(table, LIST)                   \
  LIST(OV_X_INST)                                         \
  const ConvVariant table[] = {LIST(OV_X_ROW)};           \
  const int table##Count = sizeof(table) / sizeof(table[0]);
#endif

// Each conv1d_inst_*.hip translation unit exports one table.
#define OV_DECLARE_VARIANTS(table)  \
  extern const ConvVariant table[]; \
  extern const int table##Count;
OV_DECLARE_VARIANTS(kVariantsA)
OV_DECLARE_VARIANTS(kVariantsB)
OV_DECLARE_VARIANTS(kVariantsC)
OV_DECLARE_VARIANTS(kVariantsD)
OV_DECLARE_VARIANTS(kVariantsE)
OV_DECLARE_VARIANTS(kVariantsF)
OV_DECLARE_VARIANTS(kVariantsS)
OV_DECLARE_VARIANTS(kVariantsW)

}  // namespace ovk
