// Fused ResBlock1 pair of the bf16 generator's MRF stages (C = 32 / 64 / 128; BASELINE.json configs[4],
// SURVEY.md section 8f item 3), second generation.  reference: openvoice/modules.py:296-306 (loop body of
// ResBlock1.forward), models.py:280-286 (MRF sum / mean).
//
//   t   = bf16( lrelu( c1(xa) + b1 ) )                                   xa = lrelu(x): the input is stored ACTIVATED
//   out = bf16( act_out( (c2(t) + b2 + x~) * scale ) )                   x~ = xa >= 0 ? xa : xa / slope
//   with the MRF running sum:  out = bf16( (bf16(c2(t) + b2 + x~) + add) * scale )
//
// What bounded the first generation (conv1d_bf16.hip as two launches per pair; profiles/r02_s14, r03_s11): per 128-row
// tile a matrix wave spent 20 % of its time at the 32/64-channel chunk hand-offs, 17 % moving the tile out, 15 % in the
// identity-MFMA rounds that bring the residual / running sum in, while the loader waves unpacked, activated and
// re-packed every x vector on the VALU.  Here none of that is left:
//   * activations live in HBM ACTIVATED (the producer applies the leaky ReLU before it rounds; the only raw reader, the
//     residual add, inverts it exactly in fp32: xa * (1 / slope) for negative values -- same relative rounding error as
//     storing x itself).  Staging is therefore pure LDS-DMA (global_load_lds_dwordx4): no registers, no VALU pass.
//   * one persistent workgroup per CU walks a run of time tiles left to right; the WHOLE input tile (all channels +
//     the (K-1) DIL halo) is resident in LDS, double buffered, XOR-swizzled through the DMA source addresses so that
//     every ds_read_b128 of an MFMA operand is conflict-free.  No chunk rounds, three barriers per tile.
//   * t never leaves the CU (sliding window: each row of t is computed once, as in conv1d_bf16_pair.hip).
//   * the output tile is built IN PLACE over the consumed input tile (each lane overwrites exactly the 8-byte cells it
//     read its residual from) and leaves as whole 2C-byte rows, stored by the loader waves while the matrix waves are
//     already in the next tile.
//   * each matrix wave (one per SIMD, 128 time rows x 32 output channels, v_mfma_f32_16x16x32_bf16 fragments) streams
//     its own weight fragments through a static 8-deep register ring that runs seamlessly c1 -> c2 -> next tile's c1:
//     from L2 at C = 128, from an LDS copy of both (or c1's) streams where that fits beside the tiles (C <= 64).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <type_traits>

#include "openvoice_amd.h"

#ifndef OV_CONV1D_BF16_PAIR2_H
#define OV_CONV1D_BF16_PAIR2_H

#ifndef OV_EXP
#define OV_EXP 0   // measurement builds only (scripts/exp_pair2.sh; results meaningless, the binding refuses them):
#endif             // 10 = no weight requests in the k-loops, 11 = no LDS operand reads, 12 = neither, 13 = no MFMAs

namespace ovk16q {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int NMW = 4;       // matrix waves: one per SIMD
constexpr int NLD = 4;       // loader waves (one per SIMD, so every matrix wave has the same company): LDS-DMA in, whole-row stores out

__device__ __forceinline__ uint32_t pack2(float lo, float hi) {       // one v_cvt_pk_bf16_f32, round to nearest even
  const bf16x2 h = __builtin_convertvector(f32x2{lo, hi}, bf16x2);
  uint32_t u;
  __builtin_memcpy(&u, &h, 4);
  return u;
}

// max / min without fmaxf()'s canonicalisation of both operands (three instructions per value)
__device__ __forceinline__ float vmax(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float vmin(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// leaky ReLU for 0 < slope <= 1: max(v, slope v); its inverse for inv = 1 / slope >= 1: min(v, inv v)
__device__ __forceinline__ f32x2 lrelu2(f32x2 v, float slope) {
  const f32x2 sv = v * slope;              // v_pk_mul_f32
  return f32x2{vmax(v[0], sv[0]), vmax(v[1], sv[1])};
}
__device__ __forceinline__ f32x2 unlrelu2(f32x2 v, float inv) {
  const f32x2 sv = v * inv;
  return f32x2{vmin(v[0], sv[0]), vmin(v[1], sv[1])};
}
__device__ __forceinline__ f32x2 unpack2(uint32_t w) {
  return f32x2{__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)};
}

// compile-time loop: f(integral_constant<int, I>) for I in [I0, N).  The k-loops MUST be unrolled (ring slots, operand
// double buffers and the per-step schedule are compile-time indices); `#pragma unroll` is a request hipcc drops silently
// once the body grows: it then indexes registers at run time (s_set_gpr_idx) -- 10x slower.
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// (utterance, step) sequence of one workgroup with the warm-up pseudo-step at a mid-utterance start (the tile before
// the first real one: it only produces the t context rows; nothing of it is stored)
struct Seq {
  long left;
  int b, i, nsteps;
  bool warm;
  __device__ __forceinline__ Seq(long g0, long g1, int nsteps_)
      : left(g1 - g0), b((int)(g0 / nsteps_)), i((int)(g0 - (long)(g0 / nsteps_) * nsteps_)), nsteps(nsteps_), warm(false) {
    warm = left > 0 && i != 0;
  }
  __device__ __forceinline__ bool valid() const { return left > 0; }
  __device__ __forceinline__ int tile() const { return i - (warm ? 1 : 0); }
  __device__ __forceinline__ void advance() {
    if (warm) { warm = false; return; }
    --left;
    if (++i == nsteps) { i = 0; ++b; }
  }
};

// Swizzle of the input tile: rows of P = 2 C bytes = SPR 16-byte slots, slot s of row r is stored at slot s ^ g(r).
// 16 consecutive rows (what a 16-lane service group of ds_read_b128 touches at one logical slot) then cover all 16
// 16-byte positions of the 256-byte LDS bank row.
template <int SPR>
__device__ __forceinline__ int swz(int row) {
  return SPR >= 16 ? (row & 15) : (SPR == 8 ? ((row >> 1) & 7) : ((row >> 2) & 3));
}

// WD: depth of the weight-fragment ring -- requests run WD - 1 positions (of 8 MFMAs = 128 matrix cycles) ahead
template <int K, int DIL, int C, int WD>
struct Geo {
  static constexpr int NCT = C / 32;                 // 32-channel output tiles = matrix waves along channels
  static constexpr int NTG = NMW / NCT;              // matrix waves along time
  static constexpr int TT = 128 * NTG;               // time rows per step
  static constexpr int NCH = C / 32;                 // 32-channel input chunks
  static constexpr int P1 = (K - 1) * DIL / 2, P2 = (K - 1) / 2, DELTA = P1 - P2;
  static constexpr int P = 2 * C, SPR = P / 16;      // input-tile row pitch (bytes), slots per row
  static constexpr int RPB = 1024 / P;               // rows per 1 KiB DMA block
  static constexpr int R1 = TT + 2 * P1;             // rows of an input tile
  static constexpr int NBLK = (R1 + RPB - 1) / RPB;  // DMA blocks of an input tile
  static constexpr int XB = NBLK * 1024;             // bytes per input-tile buffer
  static constexpr int PH = 2 * C + 16;              // row pitch of the t tile (conflict-free b128 reads)
  static constexpr int RH = TT + 2 * P2;             // its rows: [2 P2 rows of left context | TT new rows]
  static constexpr int S = NCH * K * 2;              // positions of one conv's k-loop: (32 input channels, tap, 16-channel output fragment)
  static constexpr int NXB = 2;                      // input-tile buffers
  static constexpr int BASE = NXB * XB + RH * PH + 2 * C * 4;
  // Both convs' weight streams resident in LDS where they fit beside the tiles (C = 32 except K = 11 with dilation 5,
  // C = 64 with K = 3): the short k-loops of exactly these shapes otherwise wait for their weight requests behind the
  // loader waves' stores in the CU's memory pipe (c2 2 584 ticks for 768 of MFMA issue at C = 32, K = 3, profiles/r04_s34)
  // WMODE 2 = both convs' streams, 1 = c1's only (C = 64, K = 7 / 11: its requests share the window with the DMA), 0 = none
  static constexpr int WOFF = (BASE + 1023) / 1024 * 1024;
  static constexpr int WMODE = C > 64 ? 0 : (WOFF + 2 * NCT * S * 1024 <= 160 * 1024 ? 2 : (WOFF + NCT * S * 1024 <= 160 * 1024 ? 1 : 0));
  static constexpr int WBYTES = WMODE * NCT * S * 1024;   // [conv][channel tile][position] records of 1 KiB
  static constexpr int SMEM = WMODE ? WOFF + WBYTES : BASE;
  static_assert(NCT * NTG == NMW && (C == 32 || C == 64 || C == 128), "4 matrix waves of 128 x 32");
  static_assert((2 * S) % WD == 0, "the weight ring slot of every position must be static");
  static_assert(SMEM <= 160 * 1024, "LDS");
};

template <int K, int DIL, int C, int WD>
__global__ __launch_bounds__(64 * (NMW + NLD)) void respair2_bf16_kernel(const ov_respair2_bf16_params p) {
  using G = Geo<K, DIL, C, WD>;
  constexpr int NXB = G::NXB;
  constexpr int TT = G::TT, NCH = G::NCH, P1 = G::P1, P2 = G::P2, DELTA = G::DELTA, P = G::P, SPR = G::SPR;
  constexpr int R1 = G::R1, NBLK = G::NBLK, XB = G::XB, PH = G::PH, RH = G::RH, S = G::S, NCT = G::NCT;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[G::SMEM];
  unsigned char* const xs = smem;                    // NXB input tiles
  unsigned char* const hb = smem + NXB * XB;         // the t tile
  float* const bsm = reinterpret_cast<float*>(smem + NXB * XB + RH * PH);
  constexpr int WMODE = G::WMODE;
  unsigned char* const wl = smem + G::WOFF;          // (WLDS) both convs' weight streams

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = p.L;
  const int nsteps = (L + P2 + TT - 1) / TT;
  const long SS = (long)p.B * nsteps;
  const long g0 = SS * blockIdx.x / gridDim.x, g1 = SS * (blockIdx.x + 1) / gridDim.x;
  if (g0 >= g1) return;
  if constexpr (WMODE > 0) {                          // all 8 waves copy; the matrix waves' ring prefill below reads
    constexpr int HALF = NCT * S * 64;                // records other waves wrote: own barrier (waits lgkmcnt(0) first)
    for (int e = tid; e < WMODE * HALF; e += 64 * (NMW + NLD)) {
      const u32x4* src = reinterpret_cast<const u32x4*>(e < HALF ? p.w1 : p.w2);
      reinterpret_cast<u32x4*>(wl)[e] = src[e < HALF ? e : e - HALF];
    }
    __syncthreads();
  }

  if (wave >= NMW) {
    // ================================ loader waves ===============================================
    // Per pseudo-step q (after barrier A of q): store the finished output tile of q - 1 out of its input buffer, then
    // start the DMA of the input tile of q + 1 into that same buffer.  Both passes deal the buffer's 1 KiB blocks to
    // the loader waves the same way, so a wave only ever overwrites blocks it has itself finished reading.
    const int lw = wave - NMW;
    const bool idle = (p.exp_flags & 1) != 0;   // MEASUREMENT ONLY (wrong results): no staging after the first tile
    // the packer's trailing all-zero record: source of every vector outside [0, L)
    const unsigned char* zsrc = reinterpret_cast<const unsigned char*>(p.w1) + (size_t)NCT * NCH * K * 2 * 1024;
    // Block blk of a tile buffer = rows [blk RPB, (blk + 1) RPB); lane -> (row lrow of the block, physical slot sp).
    // Byte offset of the lane's vector from the tile's first row IN THE TENSOR (source of the DMA, destination of the
    // store): blk * 1024 + lrow * P + 16 * (sp ^ g(row)), and g(row) only depends on blk % NPH = lw % NPH: one per-lane constant.
    // (The loaders share their SIMDs with the matrix waves: every VALU instruction here is taken from a matrix
    // wave's issue slots -- profiles/r04_s3: ~1 000 per step cost its SIMD-mates 3 000 ticks of c1.)
    constexpr int RPB = G::RPB, NPH = 16 / RPB;
    const int lrow = lane / SPR, sp = lane % SPR;
    static_assert(NLD % NPH == 0, "a loader wave's blocks (blk % NLD == lw) all share one swizzle phase");
    const uint32_t dof = (uint32_t)(lrow * P + 16 * (sp ^ swz<SPR>((lw % NPH) * RPB + lrow)));
    const uint32_t lds_lane = (uint32_t)(lane * 16);
    // Tiles whose rows all lie inside [0, L) -- all but the first and last one or two of an utterance -- take a path
    // WITHOUT vector ALU instructions: wave-uniform base (SALU) + the constant per-lane offset `dof`.  The matrix wave
    // on this SIMD runs at a higher priority and always has an MFMA waiting for the pipe, i.e. a claim on the VALU
    // issue slot: every VALU instruction of a loader waits for a gap (profiles/r04_s7: 80 loader instructions took
    // 2 400-10 800 ticks and the matrix waves then waited for the loaders at barrier B).
    typedef const __attribute__((address_space(1))) unsigned char* gc_ptr;
    typedef __attribute__((address_space(1))) unsigned char* gm_ptr;
    typedef __attribute__((address_space(3))) unsigned char* lds_ptr;
    // C = 128, K = 3: a pause of 256 cycles after every DMA instruction.  The 46 requests of a tile otherwise enter the CU's
    // memory pipe as one burst ahead of the matrix waves' weight requests, and a k-loop of 24 positions cannot ride that
    // out: c1 4 985 -> 4 489 ticks per step, 0.757 -> 0.731 ms (profiles/r04_s36; no effect at K >= 7, too slow at 512).
    constexpr bool PACE_DMA = C == 128 && K == 3;
    auto dma = [&](int buf, int b, int tile) {
      const int tbase = tile * TT - P1;
      const gc_ptr xb = (gc_ptr)(p.x) + ((int64_t)b * L + tbase) * P + lw * 1024;                          // (uniform)
      const lds_ptr lb = (lds_ptr)(xs) + buf * XB + lw * 1024;                                            // (uniform)
      if (tbase >= 0 && tbase + NBLK * RPB <= L) {
#pragma unroll
        for (int j = 0; j < (NBLK + NLD - 1) / NLD; ++j)
          if (j * NLD + lw < NBLK) {
            gc_ptr bj = xb + j * (NLD * 1024);
            asm volatile("" : "+s"(bj));     // opaque per block: scalar base + per-lane offset, no 64-bit vector adds
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bj + dof),
                                             (__attribute__((address_space(3))) void*)(lb + j * (NLD * 1024)), 16, 0, 0);
            if constexpr (PACE_DMA) __builtin_amdgcn_s_sleep(4);
          }
        return;
      }
#pragma unroll
      for (int j = 0; j < (NBLK + NLD - 1) / NLD; ++j) {
        const int blk = j * NLD + lw;
        if (blk < NBLK) {
          const int row = blk * RPB + lrow;
          const int t = tbase + row;
          const bool ok = row < R1 && t >= 0 && t < L;
          const gc_ptr src = ok ? xb + j * (NLD * 1024) + dof : (gc_ptr)zsrc;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                           (__attribute__((address_space(3))) void*)(lb + j * (NLD * 1024)), 16, 0, 0);
        }
      }
    };
    constexpr int SB0 = DELTA * SPR / 64, SB1 = ((DELTA + TT) * SPR + 63) / 64;   // blocks that hold output rows
    constexpr int J0 = SB0 / NLD, J1 = (SB1 + NLD - 1) / NLD, NSB = J1 - J0;      // ... of one loader wave: j in [J0, J1)
    // The MRF running sum (`add`) is applied HERE, on the way out: the matrix waves leave bf16(c2 + b2 + x~) in the
    // tile, the loaders read the same rows of `add` as whole 16-byte vectors (requested a step earlier, right after
    // the next tile's DMA) and store bf16((tile + add) * scale).  Read in the accumulator layout by the matrix waves
    // it was 16 eight-byte loads per lane of 16 contiguous bytes per row, sitting in the same in-order vmcnt queue as
    // the weight ring: launches with `add` ran 0.2-0.3 ms longer (profiles/r04_s4).
    const bool has_add = p.add != nullptr;
    const float scale = p.scale;
    const float add_oslope = p.out_slope > 0.f ? p.out_slope : 1.f;   // (the sum's consumer may want it activated)
    u32x4 addq[NSB];
    // rows of the lane that belong to the output window [DELTA, DELTA + TT): all of them except in the first / last block
    auto row_in_window = [&](int blk) { const int row = blk * RPB + lrow; return row >= DELTA && row < DELTA + TT; };
    auto add_request = [&](int b, int tile) {
      const int tbase = tile * TT - P2 - DELTA;
      const gc_ptr ab = (gc_ptr)(p.add) + ((int64_t)b * L + tbase) * P + lw * 1024;                       // (uniform)
      const bool interior = tbase + DELTA >= 0 && tbase + DELTA + TT <= L;
#pragma unroll
      for (int j = J0; j < J1; ++j) {
        const int blk = j * NLD + lw;
        bool ok = blk >= SB0 && blk < SB1;
        if (!interior || blk == SB0 || blk == SB1 - 1) {
          const int t = tbase + blk * RPB + lrow;
          ok = ok && row_in_window(blk) && t >= 0 && t < L;
          addq[j - J0] = *reinterpret_cast<const __attribute__((address_space(1))) u32x4*>(ok ? ab + j * (NLD * 1024) + dof : (gc_ptr)zsrc);
        } else if (ok) {
          addq[j - J0] = *reinterpret_cast<const __attribute__((address_space(1))) u32x4*>(ab + j * (NLD * 1024) + dof);
        }
      }
    };
    // The output tile leaves in two halves: LDS -> registers right after barrier A (that frees the buffer for the
    // next DMA at once), registers -> HBM after barrier B.  A 1 KiB store instruction holds the issuing wave for ~300
    // cycles (the CU's memory pipe drains at ~10 B / cycle, MI355X guide): issued before the DMA and before barrier B,
    // the eight stores of a loader wave delayed both (profiles/r04_s9: store pass 2 400 ticks per step at any priority).
    u32x4 ov[NSB];
    auto fetch_tile = [&](int buf) {
      const unsigned char* lb = xs + buf * XB + lw * 1024 + lds_lane;
#pragma unroll
      for (int j = J0; j < J1; ++j) {                                            // blk % NLD == lw, as in dma()
        const int blk = j * NLD + lw;
        if (blk >= SB0 && blk < SB1) ov[j - J0] = *reinterpret_cast<const u32x4*>(lb + j * (NLD * 1024));
      }
    };
    auto store = [&](int b, int tile) {
      const int tbase = tile * TT - P2 - DELTA;
      const gm_ptr ob = (gm_ptr)(p.out) + ((int64_t)b * L + tbase) * P + lw * 1024;                        // (uniform)
      const bool interior = tbase + DELTA >= 0 && tbase + DELTA + TT <= L;
      if (has_add) {
#pragma unroll
        for (int j = J0; j < J1; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const f32x2 r = lrelu2((unpack2(ov[j - J0][e]) + unpack2(addq[j - J0][e])) * scale, add_oslope);
            ov[j - J0][e] = pack2(r[0], r[1]);
          }
      }
#pragma unroll
      for (int j = J0; j < J1; ++j) {
        const int blk = j * NLD + lw;
        if (blk >= SB0 && blk < SB1) {
          gm_ptr bj = ob + j * (NLD * 1024);
          asm volatile("" : "+s"(bj));
          __attribute__((address_space(1))) u32x4* dst = reinterpret_cast<__attribute__((address_space(1))) u32x4*>(bj + dof);
          if (interior && blk != SB0 && blk != SB1 - 1) {
            *dst = ov[j - J0];
          } else {
            const int t = tbase + blk * RPB + lrow;
            if (row_in_window(blk) && t >= 0 && t < L) *dst = ov[j - J0];
          }
        }
      }
    };
    // measurement only (p.dbg): ticks 0 waiting for the DMA / stores, 1 at barrier A, 2 tile -> registers, 3 DMA issue,
    // 4 at barrier B, 6 stores (+ add requests), 5 at barrier C
    const bool ldbg = p.dbg != nullptr;
    unsigned long long lt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, llast = 0;
    auto lmark = [&](int ph) {
      if (ldbg) {
        const unsigned long long now = __builtin_readcyclecounter();
        lt[ph] += now - llast;
        llast = now;
      }
    };
    Seq cur(g0, g1, nsteps);                 // the pseudo-step whose barrier A comes next
    int prev_b = 0, prev_tile = 0, q = 0;
    bool prev_real = false;
    Seq nxt = cur;                           // the one after it
    nxt.advance();
    dma(0, cur.b, cur.tile());
    __builtin_amdgcn_s_barrier();                            // (init: the matrix waves have zeroed the t tile)
    if (ldbg) llast = __builtin_readcyclecounter();
    for (; cur.valid(); ++q) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's part of tile q has landed (and its stores left)
      lmark(0);
      __builtin_amdgcn_s_barrier();                          // A(q)
      lmark(1);
      const bool out_now = prev_real && !idle;
      if (out_now) fetch_tile((q + 1) & 1);                  // output tile of q - 1, built in place in ITS input buffer
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // every LDS read of that buffer has returned
      if (p.exp_flags & 8) {                                 // MEASUREMENT ONLY: ~500 VALU instructions of busy work
        f32x2 d0 = {1.0f + lane, 2.0f}, d1 = {3.0f, 4.0f + lane};
        for (int it = 0; it < 64; ++it) {
#pragma unroll
          for (int u = 0; u < 4; ++u) { d0 = d0 * d1 + d0; d1 = d1 * d0 + d1; }
        }
        if (d0[0] + d1[1] == 12345.678f) ov[0][0] = 1u;      // (keeps the loop alive)
      }
      lmark(2);
      if (nxt.valid() && !idle) dma((q + 1) & 1, nxt.b, nxt.tile());
      lmark(3);
      __builtin_amdgcn_s_barrier();                          // B(q)
      lmark(4);
      if (out_now) store(prev_b, prev_tile);
      if (has_add && !cur.warm) add_request(cur.b, cur.tile());   // consumed after B(q + 1)
      lmark(6);
      __builtin_amdgcn_s_barrier();                          // C(q)
      lmark(5);
      prev_real = !cur.warm; prev_b = cur.b; prev_tile = cur.tile();
      cur = nxt;
      nxt.advance();
    }
    __builtin_amdgcn_s_barrier();                            // A(end): the last output tile is complete
    if (prev_real) { fetch_tile((q + 1) & 1); store(prev_b, prev_tile); }
    if (ldbg && lane == 0) {
      lt[7] = (unsigned long long)q;
#pragma unroll
      for (int k = 0; k < 8; ++k) p.dbg[((size_t)blockIdx.x * (NMW + NLD) + wave) * 8 + k] = lt[k];
    }
    return;
  }

  // ================================== matrix waves ================================================
  // v_mfma_f32_16x16x32_bf16, D[channel][time] = W (16 channels x 32 k) * x (32 k x 16 time rows): the chip's bf16 matrix
  // pipe is POWER-limited on real operands, and this shape does 12 % more matrix work per watt than 32x32x16 (a 16 x 16
  // tile moves a quarter of the accumulator per instruction; profiles/r03_s5_mfma_sustained_ceilings_fp32_bf16.txt, and
  // in this kernel profiles/r04_s27: 1.57 -> 1.88 GHz at the same flops).  A wave's 128 x 32 tile is 8 (time) x 2
  // (channel) fragments; per 32 input channels of one tap it reads 8 x operands (16 bytes per lane each) and 2 weight
  // records -- the same operand bytes per flop as the 32x32x16 tiling had.  A lane holds time row 16 j + (lane & 15)
  // and channels 16 f + 4 (lane >> 4) + {0..3} of fragment (f, j): one 8-byte cell, as before.
  __builtin_amdgcn_s_setprio(2);                     // ahead of the loader wave on the same SIMD at every issue
  const int g4 = lane >> 4, l15 = lane & 15;
  const int nt = wave % NCT, tg = wave / NCT;        // this wave's output-channel tile / time group
  const int trow0 = 128 * tg;
  for (int e = tid; e < RH * PH / 4; e += 64 * NMW) reinterpret_cast<uint32_t*>(hb)[e] = 0u;
  if (tid < C) { bsm[tid] = p.b1[tid]; bsm[C + tid] = p.b2[tid]; }

  // packed weights (ov_conv1d_bf16_pack16): record ((nt * NCH + c) * K + tap) * 2 + f, 64 lanes x 16 bytes.
  // Weight stream of a step: positions [0, S) = c1's records, [S, 2 S) = c2's, then the next step's c1 again; the
  // request for position pos + WD - 1 is issued at position pos into ring slot (pos + WD - 1) % WD -- a compile-time
  // constant everywhere because 2 S % WD == 0.  `wp` is the (wave-uniform) running pointer of the request stream: a
  // loop-carried scalar, so the record addresses are two SALU adds per position (as `base + constant` they are loop
  // invariants and hipcc hoists all 2 S of them out of the step loop: hundreds of spilled registers).
  typedef const __attribute__((address_space(1))) u32x4* gw_ptr;    // explicitly GLOBAL: behind the asm below hipcc would
  const gw_ptr wg1 = (gw_ptr)(p.w1) + (size_t)nt * S * 64;          // otherwise fall back to flat loads, which also count
  const gw_ptr wg2 = (gw_ptr)(p.w2) + (size_t)nt * S * 64;          // on lgkmcnt and drain the LDS queue at every wait
  gw_ptr wp = wg1;
  u32x4 wq[WD];
  const unsigned char* const wlane = wl + nt * S * 1024 + lane * 16;
  auto wnext = [&](int pos) -> u32x4 {     // `pos` (the position being requested) is a compile-time constant at every call
    const int q = pos % (2 * S);
    if (WMODE == 2 || (WMODE == 1 && q < S)) return *reinterpret_cast<const u32x4*>(wlane + ((q / S) * NCT * S + q % S) * 1024);
    if (q == 0) wp = wg1;
    if (q == S) wp = wg2;
    asm volatile("" : "+s"(wp));          // opaque: keeps the request address a scalar base + lane offset
    const u32x4 v = wp[lane];
    wp += 64;
    return v;
  };
#pragma unroll
  for (int q = 0; q < WD - 1; ++q) wq[q] = wnext(q);
  wq[WD - 1] = wq[0];

  // per-lane LDS offsets of c1's x operand (input tile): row trow0 + l15 + tap DIL (+ 16 j), logical slot 4 c + g4
  //   ->  byte (row * P + 16 * (g4 ^ g(row)))  ^  64 c      (one register per tap: K; computing g(row) per (c, tap) in the
  //   k-loop instead costs five VALU instructions in one MFMA gap, +8 % on c1 -- profiles/r04_s28)
  uint32_t xl_tap[K];
#pragma unroll
  for (int tap = 0; tap < K; ++tap) {
    const int row = trow0 + l15 + tap * DIL;
    xl_tap[tap] = (uint32_t)(row * P + 16 * (g4 ^ swz<SPR>(row)));
  }
  // c2's x operand (t tile): row trow0 + l15 + tap (+ 16 j), byte 64 c + 16 g4
  const uint32_t hl_off = (uint32_t)((trow0 + l15) * PH + g4 * 16);
  // epilogue cells: cell c = 2 j + f of the lane = input-tile row trow0 + l15 + DELTA + 16 j, channels
  // 32 nt + 16 f + 4 g4 + {0..3}: logical slot 4 nt + 2 f + (g4 >> 1), byte 8 (g4 & 1) inside it
  const int erow = trow0 + l15 + DELTA;
  uint32_t ecell[2];
#pragma unroll
  for (int f = 0; f < 2; ++f)
    ecell[f] = (uint32_t)(erow * P + 16 * ((4 * nt + 2 * f + (g4 >> 1)) ^ swz<SPR>(erow)) + 8 * (g4 & 1));

  // measurement only (p.dbg != NULL): shader-clock ticks per phase, summed over the steps of this wave
  // 0 barrier A, 1 c1 k-loop, 2 t -> LDS, 3 barrier B, 4 c2 k-loop, 5 barrier C, 6 epilogue, 7 steps
  unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast = 0;
  const bool dbg = p.dbg != nullptr;
  auto mark = [&](int ph) {
    if (dbg) {
      const unsigned long long now = __builtin_readcyclecounter();
      tph[ph] += now - tlast;
      tlast = now;
    }
  };
  const bool has_add = p.add != nullptr;              // then the loader waves add it and scale (see above)
  const float slope = p.slope, inv_slope = 1.0f / p.slope, scale = has_add ? 1.0f : p.scale;
  const float oslope = (p.out_slope > 0.f && !has_add) ? p.out_slope : 1.f;
  const bool scaled = scale != 1.0f, act_out = oslope != 1.0f;    // (uniform; the epilogue skips what is an identity)

  // cell c = 2 j + f of the lane (its 16 output cells of 8 bytes)
  auto cell_addr = [&](int c, uint32_t boff) -> unsigned char* { return xs + boff + ecell[c & 1] + (c >> 1) * 16 * P; };

  __syncthreads();                                    // (init: t tile zeroed, biases in LDS)
  if (dbg) tlast = __builtin_readcyclecounter();
  int pstep = 0;
  for (Seq tk(g0, g1, nsteps); tk.valid(); ++pstep) {
    const int b = tk.b;
    const bool warm = tk.warm;
    const int t0 = tk.tile() * TT;
    Seq nx = tk;
    nx.advance();
    const bool next_valid = nx.valid();
    const bool next_fresh = next_valid && !nx.warm && nx.i == 0;

    // the per-lane operand offsets of this step's input buffer; opaque to the optimiser, or it hoists every
    // (offset ^ constant) of the unrolled k-loop out of the step loop and spills them
    uint32_t xlb[K];
    const uint32_t bufoff = (uint32_t)((pstep % NXB) * XB);
#pragma unroll
    for (int tap = 0; tap < K; ++tap) {
      xlb[tap] = xl_tap[tap] + bufoff;     // (XB is a multiple of 1024, the XOR constants are < 256: they commute)
      asm volatile("" : "+v"(xlb[tap]));
    }
    f32x4 acc[2][8];
    auto bias_init = [&](f32x4 (&a)[2][8], const float* bvec) {
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(bvec + 32 * nt + 16 * f + 4 * g4);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[f][j] = v;
      }
    };
    bias_init(acc, bsm);
    __syncthreads();                                  // A: input tile of this step is in LDS
    mark(0);
    // One k-loop for both convs.  Pair p = (c, tap): positions 2 p (f = 0) and 2 p + 1 (f = 1), each one weight record
    // against the SAME 8 x operands, 8 MFMAs of 16 matrix cycles.  Everything else a pair has to issue -- the 8 operand
    // reads of pair p + 1 (into the other buffer set), 2 weight requests, the next operand address -- sits BETWEEN its
    // MFMAs, one instruction per gap: a 16-cycle MFMA covers one ~5-cycle issue slot, not five of them (as a clump
    // at the head of a position, the way the 32x32x16 loop had them behind a 32-cycle MFMA, they left the pipe idle
    // ~18 of every 146 cycles, profiles/r04_s28).  Two lgkmcnt waits per pair: operands 0-3 before its first MFMA
    // (requested a pair ago), 4-7 before the fifth (requested half a pair + four MFMAs ago).
    auto kloop = [&](f32x4 (&a)[2][8], auto is_c1, auto&& obase, auto ostride) {
      constexpr bool C1L = decltype(is_c1)::value;
      constexpr int POS0 = C1L ? 0 : S;                // position of this conv's first record in the weight stream
      constexpr int NP = S / 2, STRIDE = decltype(ostride)::value;
      u32x4 xq[2][8];
      auto oread = [&](const unsigned char* base, int j) -> u32x4 {
        if constexpr (OV_EXP == 11 || OV_EXP == 12) return u32x4{0u, 0u, 0u, 0u};
        else return *reinterpret_cast<const u32x4*>(base + j * STRIDE);
      };
      const unsigned char* pbn = obase(0, 0);
#pragma unroll
      for (int j = 0; j < 8; ++j) xq[0][j] = oread(pbn, j);
#pragma unroll
      for (int j = 0; j < 8; ++j) xq[1][j] = u32x4{0u, 0u, 0u, 0u};
      pbn = obase(NP > 1 ? 1 / K : 0, NP > 1 ? 1 % K : 0);
      static_for<0, NP>([&](auto pc) {
        constexpr int pr = decltype(pc)::value, cb = pr & 1, nb = cb ^ 1;
        constexpr bool more = pr + 1 < NP;
        static_for<0, 2>([&](auto fc) {
          constexpr int f = decltype(fc)::value, s = 2 * pr + f;
          __builtin_amdgcn_sched_barrier(0);
          // (WL: one weight read per position sits in the same in-order queue, issued after MFMA 4)
          constexpr bool WL = WMODE == 2 || (WMODE == 1 && C1L);
          if constexpr (f == 0) __builtin_amdgcn_s_waitcnt(0xC07F | ((WL ? 6 : 4) << 8));     // operands 0-3 of this pair
          static_for<0, 8>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr (f == 0 && j == 4) __builtin_amdgcn_s_waitcnt(0xC07F | (((more ? 4 : 0) + (WL ? 1 : 0)) << 8));   // operands 4-7
            bf16x8 av, bv;
            __builtin_memcpy(&av, &wq[(POS0 + s) % WD], 16);
            __builtin_memcpy(&bv, &xq[cb][j], 16);
            if constexpr (OV_EXP == 13) { asm volatile("" :: "v"(av), "v"(bv)); }
            else a[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, a[f][j], 0, 0, 0);   // D[channel][time]
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (j < 4) {
              if constexpr (more) xq[nb][4 * f + j] = oread(pbn, 4 * f + j);
            } else if constexpr (j == 4) {
              // (overwrites the ring slot of position s - 1, whose MFMAs have all been issued)
              if constexpr (OV_EXP != 10 && OV_EXP != 12) wq[(POS0 + s + WD - 1) % WD] = wnext(POS0 + s + WD - 1);
            } else if constexpr (j == 5 && f == 1) {
              if constexpr (pr + 2 < NP) pbn = obase((pr + 2) / K, (pr + 2) % K);
            }
            __builtin_amdgcn_sched_barrier(0);
          });
        });
      });
    };
    // ---- c1: t = b1 + W1 * xa ------------------------------------------------------------------------------------
    kloop(acc, std::true_type{},
          [&](int c, int tap) -> const unsigned char* { return xs + (xlb[tap] ^ (uint32_t)(64 * c)); },
          std::integral_constant<int, 16 * P>{});
    mark(1);
    // t: activated in fp32, rounded once, zero outside [0, L) (c2 pads t, not x); 4 consecutive channels per store.
    // (Two copies under a uniform branch: written as one, hipcc turns the row test into a select per value.)
    {
      auto t_write = [&](auto edge) {
        constexpr bool EDGE = decltype(edge)::value;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          unsigned char* hrow = hb + (2 * P2 + trow0 + 16 * j + l15) * PH + (32 * nt + 4 * g4) * 2;
          const bool inside = !EDGE || t0 + trow0 + 16 * j + l15 < L;
#pragma unroll
          for (int f = 0; f < 2; ++f) {
            const f32x2 a = lrelu2(f32x2{acc[f][j][0], acc[f][j][1]}, slope);
            const f32x2 c = lrelu2(f32x2{acc[f][j][2], acc[f][j][3]}, slope);
            u32x2 o = {pack2(a[0], a[1]), pack2(c[0], c[1])};
            if (EDGE && !inside) o = u32x2{0u, 0u};
            *reinterpret_cast<u32x2*>(hrow + 32 * f) = o;
          }
        }
      };
      if (__builtin_expect(t0 + TT <= L, 1)) t_write(std::false_type{});   // every tile but an utterance's last one or two
      else t_write(std::true_type{});
    }
    mark(2);
    __syncthreads();                                  // B: t in LDS
    mark(3);
    // ---- c2 out of the t tile: output row o (global t0 - P2 + o) needs t rows [o, o + K - 1] of the tile ----
    bias_init(acc, bsm + C);
    kloop(acc, std::false_type{},
          [&](int c, int tap) -> const unsigned char* { return hb + hl_off + tap * PH + 64 * c; },
          std::integral_constant<int, 16 * PH>{});
    mark(4);
    __syncthreads();                                  // C: every wave is done reading the t tile
    mark(5);
    if (next_valid) {                                 // last 2 P2 rows of t -> left context of the next step
      for (int e = tid; e < 2 * P2 * (C / 2); e += 64 * NMW) {
        const int row = e / (C / 2), q = e - row * (C / 2);
        uint32_t* d = reinterpret_cast<uint32_t*>(hb + row * PH) + q;
        *d = next_fresh ? 0u : reinterpret_cast<const uint32_t*>(hb + (TT + row) * PH)[q];
      }
    }
    // ---- epilogue, in place over the input tile: cell = (acc + x~ [+ add]) * scale, activated for its consumer ----
    // All 16 cells of the lane are read first: as read-modify-write per cell the accesses may alias as far as hipcc
    // can tell, and the 16 LDS round trips run one after the other (5 000 ticks per step, profiles/r04_s2).
    if (!warm) {
      u32x2 xv[16];
      static_for<0, 16>([&](auto cc) { xv[decltype(cc)::value] = *reinterpret_cast<const u32x2*>(cell_addr(decltype(cc)::value, bufoff)); });
      // (one copy per case under uniform branches: as run-time flags hipcc evaluates both sides with selects)
      // Stage by stage over four cells, not cell by cell: one wave per SIMD has nobody to cover the ~8-cycle result
      // latency of a VALU instruction, so the seven dependent instructions of a cell must be interleaved with other
      // cells' (hipcc does not do it across the inline-asm min / max).
      auto finish = [&](auto act, auto scl) {
        constexpr bool ACT = decltype(act)::value, SCL = decltype(scl)::value;
        static_for<0, 4>([&](auto gc) {                      // four cells (eight independent chains) at a time
          constexpr int c0 = 4 * decltype(gc)::value;
          f32x2 v[4][2], m[4][2];
          static_for<0, 4>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            v[q][0] = unpack2(xv[c0 + q][0]);
            v[q][1] = unpack2(xv[c0 + q][1]);
          });
          static_for<0, 4>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            m[q][0] = v[q][0] * inv_slope;
            m[q][1] = v[q][1] * inv_slope;
          });
          static_for<0, 4>([&](auto qc) {
            constexpr int q = decltype(qc)::value, c = c0 + q, j = c >> 1, f = c & 1;
            v[q][0] = f32x2{vmin(v[q][0][0], m[q][0][0]), vmin(v[q][0][1], m[q][0][1])} + f32x2{acc[f][j][0], acc[f][j][1]};
            v[q][1] = f32x2{vmin(v[q][1][0], m[q][1][0]), vmin(v[q][1][1], m[q][1][1])} + f32x2{acc[f][j][2], acc[f][j][3]};
          });
          if constexpr (SCL) {
            static_for<0, 4>([&](auto qc) { constexpr int q = decltype(qc)::value; v[q][0] *= scale; v[q][1] *= scale; });
          }
          if constexpr (ACT) {
            static_for<0, 4>([&](auto qc) {
              constexpr int q = decltype(qc)::value;
              m[q][0] = v[q][0] * oslope;
              m[q][1] = v[q][1] * oslope;
            });
            static_for<0, 4>([&](auto qc) {
              constexpr int q = decltype(qc)::value;
              v[q][0] = f32x2{vmax(v[q][0][0], m[q][0][0]), vmax(v[q][0][1], m[q][0][1])};
              v[q][1] = f32x2{vmax(v[q][1][0], m[q][1][0]), vmax(v[q][1][1], m[q][1][1])};
            });
          }
          static_for<0, 4>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            *reinterpret_cast<u32x2*>(cell_addr(c0 + q, bufoff)) =
                u32x2{pack2(v[q][0][0], v[q][0][1]), pack2(v[q][1][0], v[q][1][1])};
          });
        });
      };
      if (act_out) finish(std::true_type{}, std::false_type{});          // an intermediate pair (its scale is 1)
      else if (scaled) finish(std::false_type{}, std::true_type{});      // the MRF mean without a running sum
      else finish(std::false_type{}, std::false_type{});
    }
    mark(6);
    tk = nx;
  }
  __syncthreads();                                    // A(end): the last output tile is complete
  if (dbg && lane == 0) {
    tph[7] = (unsigned long long)pstep;
#pragma unroll
    for (int q = 0; q < 8; ++q) p.dbg[((size_t)blockIdx.x * (NMW + NLD) + wave) * 8 + q] = tph[q];
  }
}

inline int cu_count(std::atomic<int>* cache) {
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

template <int K, int DIL, int C, int WD>
int launch(const ov_respair2_bf16_params* p, hipStream_t stream) {
  using G = Geo<K, DIL, C, WD>;
  static std::atomic<int> cache[16];
  const int slots = cu_count(cache);                  // one workgroup per CU (its LDS tile fills the CU)
  const long SS = (long)p->B * ((p->L + G::P2 + G::TT - 1) / G::TT);
  long nwg = p->nwg > 0 ? p->nwg : slots;
  if (nwg > SS) nwg = SS;
  hipLaunchKernelGGL((respair2_bf16_kernel<K, DIL, C, WD>), dim3((unsigned)nwg), dim3(64 * (NMW + NLD)), 0, stream, *p);
  return hipGetLastError() == hipSuccess ? OV_OK : OV_E_LAUNCH;
}

template <int K, int DIL>
int launch_by_width(const ov_respair2_bf16_params* p, hipStream_t stream) {
  // C = 32: 2 S = 4 K positions per step: a 4-deep ring; the 45 KB of both convs' weights are L1 / L2 hits for all waves
  if (p->C == 32) return launch<K, DIL, 32, 4>(p, stream);
  if (p->C == 64) return launch<K, DIL, 64, 8>(p, stream);
  if (p->C == 128) return launch<K, DIL, 128, 8>(p, stream);
  return OV_E_UNSUPPORTED;
}


// one translation unit per kernel size (conv1d_bf16_pair2_k3 / k7 / k11.hip: they compile in parallel -- the unrolled
// k-loops make this the slowest file of the library by far); each defines its dispatcher over (dilation, C)
int pair2_launch_k3(const ov_respair2_bf16_params* p, hipStream_t stream);
int pair2_launch_k7(const ov_respair2_bf16_params* p, hipStream_t stream);
int pair2_launch_k11(const ov_respair2_bf16_params* p, hipStream_t stream);

template <int K>
int pair2_launch_by_dilation(const ov_respair2_bf16_params* p, hipStream_t stream) {
  if (p->dil == 1) return launch_by_width<K, 1>(p, stream);
  if (p->dil == 3) return launch_by_width<K, 3>(p, stream);
  if (p->dil == 5) return launch_by_width<K, 5>(p, stream);
  return OV_E_UNSUPPORTED;
}

}  // namespace ovk16q
#endif
