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
//   * each matrix wave (one per SIMD, 128 time rows x 32 output channels) streams its own weight fragments from L2
//     through a static 8-deep register ring that runs seamlessly c1 -> c2 -> next tile's c1.
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
#ifndef OV_P2_LD
#define OV_P2_LD 2           // LDS operand buffers of the k-loops: the reads of k-step s + LD - 1 are issued at k-step s
#endif
constexpr int LD = OV_P2_LD;     // (3 measured: no faster without the deferred epilogue, spills with it -- profiles/r04_s22)
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
// once the body grows (the deferred epilogue): it then indexes registers at run time (s_set_gpr_idx) -- 10x slower.
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// cells (of the 16 a lane owns) finished before k-step st of an S-step k-loop
template <int S>
constexpr int cells_before(int st) { return st >= S ? 16 : (16 * st) / S; }

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

// WD: depth of the weight-fragment ring -- requests run WD - 1 k-steps (of 4 MFMAs = 128 matrix cycles) ahead
// DEFER: the epilogue of step i runs INSIDE the c1 k-loop of step i + 1 (its VALU work between the MFMAs), see the
// kernel; NXB input-tile buffers (3 where they fit: the DMA then runs a whole step ahead)
template <int K, int DIL, int C, int WD, bool DEFER>
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
  static constexpr int S = NCH * K * 2;              // k-steps (16 input channels x one tap) of one conv
  static constexpr int NXB = (DEFER && 3 * XB + RH * PH + 2 * C * 4 <= 160 * 1024) ? 3 : 2;
  static constexpr int SMEM = NXB * XB + RH * PH + 2 * C * 4;
  static_assert(NCT * NTG == NMW && (C == 32 || C == 64 || C == 128), "4 matrix waves of 128 x 32");
  static_assert((2 * S) % WD == 0, "the weight ring slot of every k-step must be static");
  static_assert(SMEM <= 160 * 1024, "LDS");
};

template <int K, int DIL, int C, int WD, bool DEFER>
__global__ __launch_bounds__(64 * (NMW + NLD)) void respair2_bf16_kernel(const ov_respair2_bf16_params p) {
  using G = Geo<K, DIL, C, WD, DEFER>;
  constexpr int NXB = G::NXB;
  constexpr int TT = G::TT, NCH = G::NCH, P1 = G::P1, P2 = G::P2, DELTA = G::DELTA, P = G::P, SPR = G::SPR;
  constexpr int R1 = G::R1, NBLK = G::NBLK, XB = G::XB, PH = G::PH, RH = G::RH, S = G::S, NCT = G::NCT;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[G::SMEM];
  unsigned char* const xs = smem;                    // NXB input tiles
  unsigned char* const hb = smem + NXB * XB;         // the t tile
  float* const bsm = reinterpret_cast<float*>(smem + NXB * XB + RH * PH);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = p.L;
  const int nsteps = (L + P2 + TT - 1) / TT;
  const long SS = (long)p.B * nsteps;
  const long g0 = SS * blockIdx.x / gridDim.x, g1 = SS * (blockIdx.x + 1) / gridDim.x;
  if (g0 >= g1) return;

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
    if constexpr (!DEFER) {
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
    } else {
      // Deferred epilogue: the output tile of step q - 1 is complete at barrier B(q) (its cells were finished inside
      // the c1 loop of step q).  After B(q): tile -> registers (frees the buffer), the running-sum rows of that tile are
      // requested, the DMA of step q + NXB - 1 goes into the freed buffer -- the LAST memory operations this wave issues
      // before barrier A(q + 1), so that a counted vmcnt can leave exactly them in flight when there are three buffers.
      // The stores (300 cycles of issue each) wait for the next A -> B window, where the loaders have nothing else to do.
      constexpr int NDMA0 = (NBLK + NLD - 1) / NLD;            // DMA instructions per tile of loader wave 0 (the others:
      const int ndma = (NBLK - lw + NLD - 1) / NLD;            // NDMA0 or NDMA0 - 1)
      Seq ahead = cur;                                         // the pseudo-step whose tile is requested next
#pragma unroll
      for (int k = 0; k < NXB - 1; ++k) {
        if (ahead.valid()) dma(k, ahead.b, ahead.tile());
        ahead.advance();
      }
      __builtin_amdgcn_s_barrier();                            // (init)
      if (ldbg) llast = __builtin_readcyclecounter();
      bool ovalid = false, dma_last = false;
      int ob = 0, otile = 0;
      for (; cur.valid(); ++q) {
        // x(q) has landed; with three buffers the DMA issued after B(q - 1) (tile q + 1) may stay in flight
        if (NXB == 3 && dma_last) {
          if (ndma == NDMA0) __builtin_amdgcn_s_waitcnt(0x0F70 | (NDMA0 & 15) | ((NDMA0 >> 4) << 14));
          else __builtin_amdgcn_s_waitcnt(0x0F70 | ((NDMA0 - 1) & 15) | (((NDMA0 - 1) >> 4) << 14));
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("" ::: "memory");
        lmark(0);
        __builtin_amdgcn_s_barrier();                          // A(q)
        lmark(1);
        if (ovalid && !idle) store(ob, otile);                 // output tile of q - 2 (in registers since B(q - 1))
        ovalid = false;
        lmark(6);
        __builtin_amdgcn_s_barrier();                          // B(q)
        lmark(4);
        if (prev_real && !idle) {
          fetch_tile((q + NXB - 1) % NXB);                     // output tile of q - 1
          ovalid = true; ob = prev_b; otile = prev_tile;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // every LDS read of that buffer has returned
        lmark(2);
        if (has_add && prev_real && !idle) add_request(prev_b, prev_tile);   // consumed after A(q + 1)
        dma_last = ahead.valid() && !idle;
        if (dma_last) dma((q + NXB - 1) % NXB, ahead.b, ahead.tile());
        lmark(3);
        __builtin_amdgcn_s_barrier();                          // C(q)
        lmark(5);
        prev_real = !cur.warm; prev_b = cur.b; prev_tile = cur.tile();
        cur.advance();
        ahead.advance();
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                            // A(end): the matrix waves have finished the last tile
      if (ovalid) store(ob, otile);
      if (prev_real) {
        fetch_tile((q + NXB - 1) % NXB);
        if (has_add) add_request(prev_b, prev_tile);
        store(prev_b, prev_tile);
      }
    }
    if (ldbg && lane == 0) {
      lt[7] = (unsigned long long)q;
#pragma unroll
      for (int k = 0; k < 8; ++k) p.dbg[((size_t)blockIdx.x * (NMW + NLD) + wave) * 8 + k] = lt[k];
    }
    return;
  }

  // ================================== matrix waves ================================================
  __builtin_amdgcn_s_setprio(2);                     // ahead of the loader wave on the same SIMD at every issue
  const int half = lane >> 5, l31 = lane & 31;
  const int nt = wave % NCT, tg = wave / NCT;        // this wave's output-channel tile / time group
  const int trow0 = 128 * tg;
  for (int e = tid; e < RH * PH / 4; e += 64 * NMW) reinterpret_cast<uint32_t*>(hb)[e] = 0u;
  if (tid < C) { bsm[tid] = p.b1[tid]; bsm[C + tid] = p.b2[tid]; }

  // packed weights (ov_conv1d_bf16_pack): record ((nt * NCH + c) * K + tap) * 2 + kb, 64 lanes x 16 bytes.
  // Weight stream of a step: positions [0, S) = c1's records, [S, 2 S) = c2's, then the next step's c1 again; the
  // request for position pos + WD - 1 is issued at position pos into ring slot (pos + WD - 1) % WD -- a compile-time
  // constant everywhere because 2 S % WD == 0.  `wp` is the (wave-uniform) running pointer of the request stream: a
  // loop-carried scalar, so the record addresses are two SALU adds per k-step (as `base + constant` they are loop
  // invariants and hipcc hoists all 2 S of them out of the step loop: hundreds of spilled registers).
  typedef const __attribute__((address_space(1))) u32x4* gw_ptr;    // explicitly GLOBAL: behind the asm below hipcc would
  const gw_ptr wg1 = (gw_ptr)(p.w1) + (size_t)nt * S * 64;          // otherwise fall back to flat loads, which also count
  const gw_ptr wg2 = (gw_ptr)(p.w2) + (size_t)nt * S * 64;          // on lgkmcnt and drain the LDS queue at every wait
  gw_ptr wp = wg1;
  u32x4 wq[WD];
  auto wnext = [&](int pos) -> u32x4 {     // `pos` (the position being requested) is a compile-time constant at every call
    const int q = pos % (2 * S);
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

  // per-lane LDS offsets of c1's B operand (input tile): row trow0 + l31 + tap DIL (+ 32 i), logical slot
  // 4 c + 2 kb + half  ->  byte (row * P + 16 * (half ^ g(row)))  ^  (64 c + 32 kb)
  uint32_t xl_tap[K];
#pragma unroll
  for (int tap = 0; tap < K; ++tap) {
    const int row = trow0 + l31 + tap * DIL;
    xl_tap[tap] = (uint32_t)(row * P + 16 * (half ^ swz<SPR>(row)));
  }
  // c2's B operand (t tile): row trow0 + l31 + tap (+ 32 i), byte 64 c + 32 kb + 16 half
  const uint32_t hl_off = (uint32_t)((trow0 + l31) * PH + half * 16);
  // epilogue cells: input-tile row trow0 + l31 + DELTA (+ 32 i), logical slot 4 nt + q, byte 8 half inside it
  const int erow = trow0 + l31 + DELTA;
  uint32_t ecell[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) ecell[q] = (uint32_t)(erow * P + 16 * ((4 * nt + q) ^ swz<SPR>(erow)) + 8 * half);

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

  // Deferred epilogue (DEFER): c2 accumulates into accE, which then waits through the next step's c1 loop, where its
  // 16 cells per lane are finished one by one -- LDS read a k-step ahead, four ~8-instruction stages placed after
  // the four MFMAs of a k-step (an MFMA occupies the matrix pipe for 32 cycles: ~5 more instructions issue for free),
  // LDS write.  As its own phase the epilogue is 2 500 ticks per step of pure VALU with the matrix pipe idle: 8 % of
  // a K = 11, C = 128 step, 20 % at K = 3 (profiles/r04_s12).
  f32x16 accE[4];
  bool pend = false;                                  // accE holds a step's c2 result whose cells are not finished yet
  uint32_t pbufoff = 0;                               // ... in this input-tile buffer
  // cell c = 4 i + q of the lane: rows trow0 + l31 + DELTA + 32 i, 4 channels 32 nt + 8 q + 4 half
  auto cell_addr = [&](int c, uint32_t boff) -> unsigned char* { return xs + boff + ecell[c & 3] + (c >> 2) * 32 * P; };
  auto cell_stage = [&](auto cc, auto sc, u32x2 xv, f32x2& v0, f32x2& v1, uint32_t boff) {
    constexpr int c = decltype(cc)::value, stage = decltype(sc)::value;
    constexpr int i = c >> 2, q = c & 3;
    if constexpr (stage == 0) {
      v0 = f32x2{accE[i][4 * q], accE[i][4 * q + 1]} + unlrelu2(unpack2(xv[0]), inv_slope);
    } else if constexpr (stage == 1) {
      v0 = lrelu2(v0 * scale, oslope);
      v1 = unlrelu2(unpack2(xv[1]), inv_slope);
    } else if constexpr (stage == 2) {
      v1 = lrelu2((f32x2{accE[i][4 * q + 2], accE[i][4 * q + 3]} + v1) * scale, oslope);
    } else {
      const u32x2 o = {pack2(v0[0], v0[1]), pack2(v1[0], v1[1])};
      if (pend) *reinterpret_cast<u32x2*>(cell_addr(c, boff)) = o;
    }
  };

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
    f32x16 acc[4];
    auto bias_init = [&](const float* bvec) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(bvec + 32 * nt + 8 * q + 4 * half);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][4 * q + e] = v[e];
      }
    };
    bias_init(bsm);
    __syncthreads();                                  // A: input tile of this step is in LDS
    mark(0);
    // ---- c1: t = b1 + W1 * xa  (+ the cells of the previous step's output, DEFER) ----------------------------------
    {
      u32x4 aq[LD][4];
      auto xread = [&](auto sc) {                      // B operands of k-step s' -> buffer s' % LD
        constexpr int sp = decltype(sc)::value;
        if constexpr (sp < S && OV_EXP != 11 && OV_EXP != 12) {
          constexpr int c = sp / (2 * K), tap = (sp / 2) % K, kb = sp & 1;
          const uint32_t a = xlb[tap] ^ (uint32_t)(64 * c + 32 * kb);
#pragma unroll
          for (int i = 0; i < 4; ++i) aq[sp % LD][i] = *reinterpret_cast<const u32x4*>(xs + a + i * 32 * P);
        }
      };
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int b = 0; b < LD; ++b) aq[b][i] = u32x4{0u, 0u, 0u, 0u};
      static_for<0, LD - 1>([&](auto sc) { xread(sc); });
      // cells [CB(s), CB(s + 1)) are finished in k-step s; their 8-byte reads are issued a k-step earlier
      constexpr int MAXC = (16 + S - 1) / S + 1;
      u32x2 cx[2][MAXC];
      f32x2 cv0[MAXC], cv1[MAXC];
      if constexpr (DEFER) {
        static_for<0, cells_before<S>(1)>([&](auto cc) {
          constexpr int c = decltype(cc)::value;
          cx[0][c] = *reinterpret_cast<const u32x2*>(cell_addr(c, pbufoff));
        });
      }
      static_for<0, S>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        constexpr int C0 = cells_before<S>(s), C1 = cells_before<S>(s + 1), C2 = cells_before<S>(s + 2);
        if constexpr (OV_EXP != 10 && OV_EXP != 12) wq[(s + WD - 1) % WD] = wnext(s + WD - 1);
        xread(std::integral_constant<int, s + LD - 1>{});
        constexpr int nread = DEFER ? C2 - C1 : 0;               // cell reads issued behind the operand reads
        if constexpr (DEFER) {
          static_for<C1, C2>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            cx[(s + 1) & 1][c - C1] = *reinterpret_cast<const u32x2*>(cell_addr(c, pbufoff));
          });
        }
        __builtin_amdgcn_sched_barrier(0);
        // ONE wait for the step's four operands (issued a step ago; what was issued in this step stays in flight):
        // hipcc's own staggered lgkmcnt(7 .. 4) puts an instruction between every two MFMAs, ~6 cycles each
        {
          // in flight behind the operands of this step: those of the next LD - 2 steps (issued earlier), of step
          // s + LD - 1 and the cell reads (issued just now); with DEFER the older cell reads / writes sit in between,
          // so the count is only exact without them -- a smaller count waits for more, never for less
          constexpr int ahead = (S - 1 - s < LD - 1 ? S - 1 - s : LD - 1);
          constexpr int left = DEFER ? ((s + LD - 1 < S ? 4 : 0) + nread) : 4 * ahead;
          static_assert(left <= 15, "lgkmcnt immediate");
          __builtin_amdgcn_s_waitcnt(0xC07F | (left << 8));
        }
        static_for<0, 4>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          bf16x8 av, bv;
          __builtin_memcpy(&av, &wq[s % WD], 16);
          __builtin_memcpy(&bv, &aq[s % LD][i], 16);
          if constexpr (OV_EXP == 13) { asm volatile("" :: "v"(av), "v"(bv)); }
          else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[i], 0, 0, 0);   // D[channel][time]
          if constexpr (DEFER && C1 > C0) {
            __builtin_amdgcn_sched_barrier(0);
            static_for<C0, C1>([&](auto cc) {
              constexpr int c = decltype(cc)::value;
              cell_stage(cc, ic, cx[s & 1][c - C0], cv0[c - C0], cv1[c - C0], pbufoff);
            });
            __builtin_amdgcn_sched_barrier(0);
          }
        });
        __builtin_amdgcn_sched_barrier(0);
      });
    }
    mark(1);
    // t: activated in fp32, rounded once, zero outside [0, L) (c2 pads t, not x); 4 consecutive channels per store.
    // (Two copies under a uniform branch: written as one, hipcc turns the row test into a select per value.)
    {
      auto t_write = [&](auto edge) {
        constexpr bool EDGE = decltype(edge)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          unsigned char* hrow = hb + (2 * P2 + trow0 + 32 * i + l31) * PH + (32 * nt + 4 * half) * 2;
          const bool inside = !EDGE || t0 + trow0 + 32 * i + l31 < L;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x2 a = lrelu2(f32x2{acc[i][4 * q], acc[i][4 * q + 1]}, slope);
            const f32x2 c = lrelu2(f32x2{acc[i][4 * q + 2], acc[i][4 * q + 3]}, slope);
            u32x2 o = {pack2(a[0], a[1]), pack2(c[0], c[1])};
            if (EDGE && !inside) o = u32x2{0u, 0u};
            *reinterpret_cast<u32x2*>(hrow + 16 * q) = o;
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
    // (DEFER: into accE, finished during the next step's c1)
    {
      f32x16 (&a2)[4] = DEFER ? accE : acc;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(bsm + C + 32 * nt + 8 * q + 4 * half);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) a2[i][4 * q + e] = v[e];
      }
      u32x4 aq[LD][4];
      auto hread = [&](auto sc) {
        constexpr int sp = decltype(sc)::value;
        if constexpr (sp < S && OV_EXP != 11 && OV_EXP != 12) {
          constexpr int c = sp / (2 * K), tap = (sp / 2) % K, kb = sp & 1;
#pragma unroll
          for (int i = 0; i < 4; ++i)
            aq[sp % LD][i] = *reinterpret_cast<const u32x4*>(hb + hl_off + (tap + 32 * i) * PH + 64 * c + 32 * kb);
        }
      };
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int b = 0; b < LD; ++b) aq[b][i] = u32x4{0u, 0u, 0u, 0u};
      static_for<0, LD - 1>([&](auto sc) { hread(sc); });
      static_for<0, S>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        if constexpr (OV_EXP != 10 && OV_EXP != 12) wq[(S + s + WD - 1) % WD] = wnext(S + s + WD - 1);
        hread(std::integral_constant<int, s + LD - 1>{});
        __builtin_amdgcn_sched_barrier(0);
        {
          constexpr int ahead = (S - 1 - s < LD - 1 ? S - 1 - s : LD - 1);
          __builtin_amdgcn_s_waitcnt(0xC07F | ((4 * ahead) << 8));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          bf16x8 av, bv;
          __builtin_memcpy(&av, &wq[(S + s) % WD], 16);
          __builtin_memcpy(&bv, &aq[s % LD][i], 16);
          if (OV_EXP == 13) { asm volatile("" :: "v"(av), "v"(bv)); continue; }
          a2[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, a2[i], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    }
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
    if constexpr (DEFER) {
      pend = !warm;
      pbufoff = bufoff;
    } else if (!warm) {
      unsigned char* xw = xs + bufoff;
      u32x2 xv[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) xv[i][q] = *reinterpret_cast<const u32x2*>(xw + ecell[q] + i * 32 * P);
      // (one copy per case under uniform branches: as run-time flags hipcc evaluates both sides with selects)
      // Stage by stage over four cells, not cell by cell: one wave per SIMD has nobody to cover the ~8-cycle result
      // latency of a VALU instruction, so the seven dependent instructions of a cell must be interleaved with other
      // cells' (hipcc does not do it across the inline-asm min / max).
      auto finish = [&](auto act, auto scl) {
        constexpr bool ACT = decltype(act)::value, SCL = decltype(scl)::value;
        static_for<0, 4>([&](auto ic) {                      // four cells (eight independent chains) at a time
          constexpr int i = decltype(ic)::value;
          f32x2 v[4][2], m[4][2];
          static_for<0, 4>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            v[q][0] = unpack2(xv[i][q][0]);
            v[q][1] = unpack2(xv[i][q][1]);
          });
          static_for<0, 4>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            m[q][0] = v[q][0] * inv_slope;
            m[q][1] = v[q][1] * inv_slope;
          });
          static_for<0, 4>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            v[q][0] = f32x2{vmin(v[q][0][0], m[q][0][0]), vmin(v[q][0][1], m[q][0][1])} + f32x2{acc[i][4 * q], acc[i][4 * q + 1]};
            v[q][1] = f32x2{vmin(v[q][1][0], m[q][1][0]), vmin(v[q][1][1], m[q][1][1])} + f32x2{acc[i][4 * q + 2], acc[i][4 * q + 3]};
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
            *reinterpret_cast<u32x2*>(xw + ecell[q] + i * 32 * P) =
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
  if constexpr (DEFER) {                              // the last step's cells: nothing left to hide them behind
    if (pend) {
      u32x2 xv[16];
      static_for<0, 16>([&](auto cc) { xv[decltype(cc)::value] = *reinterpret_cast<const u32x2*>(cell_addr(decltype(cc)::value, pbufoff)); });
      static_for<0, 16>([&](auto cc) {
        f32x2 v0, v1;
        static_for<0, 4>([&](auto st) { cell_stage(cc, st, xv[decltype(cc)::value], v0, v1, pbufoff); });
      });
    }
    mark(6);
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

template <int K, int DIL, int C, int WD, bool DEFER>
int launch(const ov_respair2_bf16_params* p, hipStream_t stream) {
  using G = Geo<K, DIL, C, WD, DEFER>;
  static std::atomic<int> cache[16];
  const int slots = cu_count(cache);                  // one workgroup per CU (its LDS tile fills the CU)
  const long SS = (long)p->B * ((p->L + G::P2 + G::TT - 1) / G::TT);
  long nwg = p->nwg > 0 ? p->nwg : slots;
  if (nwg > SS) nwg = SS;
  hipLaunchKernelGGL((respair2_bf16_kernel<K, DIL, C, WD, DEFER>), dim3((unsigned)nwg), dim3(64 * (NMW + NLD)), 0, stream, *p);
  return hipGetLastError() == hipSuccess ? OV_OK : OV_E_LAUNCH;
}

template <int K, int DIL>
int launch_by_width(const ov_respair2_bf16_params* p, hipStream_t stream) {
  const bool defer = !(p->exp_flags & 2);     // (bit 1: the epilogue as its own phase -- A/B measurements)
  // C = 32: 2 S = 4 K k-steps per step: a 4-deep ring; the 45 KB of both convs' weights are L1 / L2 hits for all waves
  if (p->C == 32) return defer ? launch<K, DIL, 32, 4, true>(p, stream) : launch<K, DIL, 32, 4, false>(p, stream);
  if (p->C == 64) return defer ? launch<K, DIL, 64, 8, true>(p, stream) : launch<K, DIL, 64, 8, false>(p, stream);
  if (p->C == 128) return defer ? launch<K, DIL, 128, 8, true>(p, stream) : launch<K, DIL, 128, 8, false>(p, stream);
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
