// Fused ResBlock1 pair for the MATRIX-bound stages of the bf16 generator (C = 64 / 128; BASELINE.json configs[4],
// SURVEY.md section 8f item 3), second generation.  reference: openvoice/modules.py:296-306 (loop body of
// ResBlock1.forward), models.py:280-286 (MRF sum / mean).
//
//   t   = bf16( lrelu( c1(xa) + b1 ) )                                   xa = lrelu(x): the input is stored ACTIVATED
//   out = bf16( act_out( (c2(t) + b2 + x~ [+ add]) * scale ) )           x~ = xa >= 0 ? xa : xa / slope
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

namespace ovk16q {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int NMW = 4;       // matrix waves: one per SIMD
constexpr int NLD = 2;       // loader waves: LDS-DMA in, whole-row stores out
constexpr int WD = 8;        // weight-fragment ring: requests run WD - 1 k-steps (of 4 MFMAs) ahead

__device__ __forceinline__ uint32_t pack2(float lo, float hi) {       // one v_cvt_pk_bf16_f32, round to nearest even
  const bf16x2 h = __builtin_convertvector(f32x2{lo, hi}, bf16x2);
  uint32_t u;
  __builtin_memcpy(&u, &h, 4);
  return u;
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
  return SPR >= 16 ? (row & 15) : ((row >> 1) & 7);
}

template <int K, int DIL, int C>
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
  static constexpr int SMEM = 2 * XB + RH * PH + 2 * C * 4;
  static_assert(NCT * NTG == NMW && (C == 64 || C == 128), "4 matrix waves of 128 x 32");
  static_assert((2 * S) % WD == 0, "the weight ring slot of every k-step must be static");
  static_assert(SMEM <= 160 * 1024, "LDS");
};

template <int K, int DIL, int C>
__global__ __launch_bounds__(64 * (NMW + NLD)) void respair2_bf16_kernel(const ov_respair2_bf16_params p) {
  using G = Geo<K, DIL, C>;
  constexpr int TT = G::TT, NCH = G::NCH, P1 = G::P1, P2 = G::P2, DELTA = G::DELTA, P = G::P, SPR = G::SPR;
  constexpr int R1 = G::R1, NBLK = G::NBLK, XB = G::XB, PH = G::PH, RH = G::RH, S = G::S, NCT = G::NCT;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[G::SMEM];
  unsigned char* const xs = smem;                    // two input tiles
  unsigned char* const hb = smem + 2 * XB;           // the t tile
  float* const bsm = reinterpret_cast<float*>(smem + 2 * XB + RH * PH);

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
    // the packer's trailing all-zero record: source of every vector outside [0, L)
    const unsigned char* zsrc = reinterpret_cast<const unsigned char*>(p.w1) + (size_t)NCT * NCH * K * 2 * 1024;
    auto dma = [&](int buf, int b, int tile) {
      const unsigned char* xb = reinterpret_cast<const unsigned char*>(p.x) + (size_t)b * L * P;
      const int tbase = tile * TT - P1;
#pragma unroll
      for (int blk = lw; blk < NBLK; blk += NLD) {
        const int U = blk * 64 + lane;
        const int row = U / SPR, sp = U % SPR;
        const int t = tbase + row;
        const bool ok = row < R1 && t >= 0 && t < L;
        const unsigned char* src = ok ? xb + ((size_t)t * P + (size_t)((sp ^ swz<SPR>(row)) * 16)) : zsrc;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(xs + buf * XB + U * 16), 16, 0, 0);
      }
    };
    constexpr int SB0 = DELTA * SPR / 64, SB1 = ((DELTA + TT) * SPR + 63) / 64;   // blocks that hold output rows
    auto store = [&](int buf, int b, int tile) {
      unsigned char* ob = reinterpret_cast<unsigned char*>(p.out) + (size_t)b * L * P;
      const int tbase = tile * TT - P2 - DELTA;
#pragma unroll
      for (int blk = SB0 + ((lw - SB0) & (NLD - 1)); blk < SB1; blk += NLD) {   // blk % NLD == lw, as in dma()
        const int U = blk * 64 + lane;
        const int row = U / SPR, sp = U % SPR;
        const int t = tbase + row;
        const u32x4 v = *reinterpret_cast<const u32x4*>(xs + buf * XB + U * 16);
        if (row >= DELTA && row < DELTA + TT && t >= 0 && t < L)
          *reinterpret_cast<u32x4*>(ob + ((size_t)t * P + (size_t)((sp ^ swz<SPR>(row)) * 16))) = v;
      }
    };
    Seq cur(g0, g1, nsteps);                 // the pseudo-step whose barrier A comes next
    Seq nxt = cur;                           // the one after it
    nxt.advance();
    dma(0, cur.b, cur.tile());
    __builtin_amdgcn_s_barrier();                            // (init: the matrix waves have zeroed the t tile)
    int prev_b = 0, prev_tile = 0, q = 0;
    bool prev_real = false;
    for (; cur.valid(); ++q) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's part of tile q has landed (and its stores left)
      __builtin_amdgcn_s_barrier();                          // A(q)
      if (prev_real) store((q + 1) & 1, prev_b, prev_tile);  // output tile of q - 1, built in place in ITS input buffer
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // every LDS read of that buffer has returned
      if (nxt.valid()) dma((q + 1) & 1, nxt.b, nxt.tile());
      __builtin_amdgcn_s_barrier();                          // B(q)
      __builtin_amdgcn_s_barrier();                          // C(q)
      prev_real = !cur.warm; prev_b = cur.b; prev_tile = cur.tile();
      cur = nxt;
      nxt.advance();
    }
    __builtin_amdgcn_s_barrier();                            // A(end): the last output tile is complete
    if (prev_real) store((q + 1) & 1, prev_b, prev_tile);
    return;
  }

  // ================================== matrix waves ================================================
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
  const float slope = p.slope, inv_slope = 1.0f / p.slope, scale = p.scale;
  const float oslope = p.out_slope > 0.f ? p.out_slope : 1.f;
  const bool has_add = p.add != nullptr;

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
    const uint32_t bufoff = (uint32_t)((pstep & 1) * XB);
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
    // ---- c1: t = b1 + W1 * xa -------------------------------------------------------------------
    {
      u32x4 aq[2][4];
#pragma unroll
      for (int i = 0; i < 4; ++i) aq[0][i] = *reinterpret_cast<const u32x4*>(xs + xlb[0] + i * 32 * P);
#pragma unroll
      for (int s = 0; s < S; ++s) {
        wq[(s + WD - 1) % WD] = wnext(s + WD - 1);
        if (s + 1 < S) {
          const int c = (s + 1) / (2 * K), tap = ((s + 1) / 2) % K, kb = (s + 1) & 1;
          const uint32_t a = xlb[tap] ^ (uint32_t)(64 * c + 32 * kb);
#pragma unroll
          for (int i = 0; i < 4; ++i) aq[(s + 1) & 1][i] = *reinterpret_cast<const u32x4*>(xs + a + i * 32 * P);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          bf16x8 av, bv;
          __builtin_memcpy(&av, &wq[s % WD], 16);
          __builtin_memcpy(&bv, &aq[s & 1][i], 16);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[i], 0, 0, 0);   // D[channel][time]
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    mark(1);
    // t: activated in fp32, rounded once, zero outside [0, L) (c2 pads t, not x); 4 consecutive channels per store
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool inside = t0 + trow0 + 32 * i + l31 < L;
      unsigned char* hrow = hb + (2 * P2 + trow0 + 32 * i + l31) * PH + (32 * nt + 4 * half) * 2;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = acc[i][4 * q + e];
          v[e] = v[e] > 0.f ? v[e] : v[e] * slope;
        }
        u32x2 o = {pack2(v[0], v[1]), pack2(v[2], v[3])};
        if (!inside) o = u32x2{0u, 0u};
        *reinterpret_cast<u32x2*>(hrow + 16 * q) = o;
      }
    }
    mark(2);
    __syncthreads();                                  // B: t in LDS
    mark(3);
    // the MRF running sum of this wave's cells: requested now, consumed in the epilogue
    u32x2 addv[4][4];
    if (has_add && !warm) {
      const uint16_t* ab = p.add + (size_t)b * L * C + 32 * nt + 4 * half;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int t = t0 - P2 + trow0 + 32 * i + l31;
        const bool ok = t >= 0 && t < L;
        const uint16_t* arow = ab + (size_t)(ok ? t : 0) * C;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          addv[i][q] = *reinterpret_cast<const u32x2*>(arow + 8 * q);
          if (!ok) addv[i][q] = u32x2{0u, 0u};
        }
      }
    }
    // ---- c2 out of the t tile: output row o (global t0 - P2 + o) needs t rows [o, o + K - 1] of the tile ----
    bias_init(bsm + C);
    {
      u32x4 aq[2][4];
#pragma unroll
      for (int i = 0; i < 4; ++i) aq[0][i] = *reinterpret_cast<const u32x4*>(hb + hl_off + i * 32 * PH);
#pragma unroll
      for (int s = 0; s < S; ++s) {
        wq[(S + s + WD - 1) % WD] = wnext(S + s + WD - 1);
        if (s + 1 < S) {
          const int c = (s + 1) / (2 * K), tap = ((s + 1) / 2) % K, kb = (s + 1) & 1;
#pragma unroll
          for (int i = 0; i < 4; ++i)
            aq[(s + 1) & 1][i] = *reinterpret_cast<const u32x4*>(hb + hl_off + (tap + 32 * i) * PH + 64 * c + 32 * kb);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          bf16x8 av, bv;
          __builtin_memcpy(&av, &wq[(S + s) % WD], 16);
          __builtin_memcpy(&bv, &aq[s & 1][i], 16);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[i], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
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
    if (!warm) {
      unsigned char* xw = xs + (pstep & 1) * XB;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          unsigned char* cell = xw + ecell[q] + i * 32 * P;
          const u32x2 xv = *reinterpret_cast<const u32x2*>(cell);
          float r[4] = {__uint_as_float(xv[0] << 16), __uint_as_float(xv[0] & 0xffff0000u),
                        __uint_as_float(xv[1] << 16), __uint_as_float(xv[1] & 0xffff0000u)};
          float a[4] = {0.f, 0.f, 0.f, 0.f};
          if (has_add) {
            a[0] = __uint_as_float(addv[i][q][0] << 16); a[1] = __uint_as_float(addv[i][q][0] & 0xffff0000u);
            a[2] = __uint_as_float(addv[i][q][1] << 16); a[3] = __uint_as_float(addv[i][q][1] & 0xffff0000u);
          }
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float xr = r[e] >= 0.f ? r[e] : r[e] * inv_slope;
            v[e] = (acc[i][4 * q + e] + xr + a[e]) * scale;
            v[e] = v[e] > 0.f ? v[e] : v[e] * oslope;
          }
          *reinterpret_cast<u32x2*>(cell) = u32x2{pack2(v[0], v[1]), pack2(v[2], v[3])};
        }
      }
    }
    mark(6);
    tk = nx;
  }
  __syncthreads();                                    // A(end): the last output tile is complete
  if (dbg && lane == 0) {
    tph[7] = (unsigned long long)pstep;
#pragma unroll
    for (int q = 0; q < 8; ++q) p.dbg[((size_t)blockIdx.x * NMW + wave) * 8 + q] = tph[q];
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

template <int K, int DIL, int C>
int launch(const ov_respair2_bf16_params* p, hipStream_t stream) {
  using G = Geo<K, DIL, C>;
  static std::atomic<int> cache[16];
  const int slots = cu_count(cache);                  // one workgroup per CU (its LDS tile fills the CU)
  const long SS = (long)p->B * ((p->L + G::P2 + G::TT - 1) / G::TT);
  long nwg = p->nwg > 0 ? p->nwg : slots;
  if (nwg > SS) nwg = SS;
  hipLaunchKernelGGL((respair2_bf16_kernel<K, DIL, C>), dim3((unsigned)nwg), dim3(64 * (NMW + NLD)), 0, stream, *p);
  return hipGetLastError() == hipSuccess ? OV_OK : OV_E_LAUNCH;
}

template <int K, int DIL>
int launch_by_width(const ov_respair2_bf16_params* p, hipStream_t stream) {
  if (p->C == 64) return launch<K, DIL, 64>(p, stream);
  if (p->C == 128) return launch<K, DIL, 128>(p, stream);
  return OV_E_UNSUPPORTED;
}

}  // namespace ovk16q

using namespace ovk16q;

extern "C" {

int ov_resblock_pair2_bf16_supported(int C, int K, int dil) {
  const bool kd = (K == 3 || K == 7 || K == 11) && (dil == 1 || dil == 3 || dil == 5);
  return kd && (C == 64 || C == 128) ? 1 : 0;
}

int ov_resblock_pair2_bf16cl(const ov_respair2_bf16_params* p, ov_stream_t stream) {
  if (!p || !p->x || !p->w1 || !p->b1 || !p->w2 || !p->b2 || !p->out) return OV_E_BADARG;
  if (p->B <= 0 || p->L <= 0 || p->C <= 0 || p->nwg < 0) return OV_E_BADARG;
  if (p->out == p->x) return OV_E_BADARG;
  if (!(p->slope > 0.f && p->slope <= 1.f) || p->out_slope < 0.f || p->out_slope > 1.f) return OV_E_UNSUPPORTED;
  if (!ov_resblock_pair2_bf16_supported(p->C, p->K, p->dil)) return OV_E_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(p->x) & 15) || (reinterpret_cast<uintptr_t>(p->w1) & 15) ||
      (reinterpret_cast<uintptr_t>(p->w2) & 15) || (reinterpret_cast<uintptr_t>(p->out) & 15) ||
      (p->add && (reinterpret_cast<uintptr_t>(p->add) & 7)))
    return OV_E_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
#define OV16Q_CASE(KK, DD) if (p->K == KK && p->dil == DD) return launch_by_width<KK, DD>(p, st);
  OV16Q_CASE(3, 1) OV16Q_CASE(3, 3) OV16Q_CASE(3, 5)
  OV16Q_CASE(7, 1) OV16Q_CASE(7, 3) OV16Q_CASE(7, 5)
  OV16Q_CASE(11, 1) OV16Q_CASE(11, 3) OV16Q_CASE(11, 5)
#undef OV16Q_CASE
  return OV_E_UNSUPPORTED;
}

}  // extern "C"
