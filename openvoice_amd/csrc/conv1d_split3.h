// Split-precision Conv1d of the generator's MRF stages: fp32-level products on the bf16 matrix pipe (VERDICT r04 item 1,
// DESIGN.md section 10 item 5).  reference: openvoice/modules.py:296-309 (the convs of ResBlock1.forward).
//
// On gfx950 the fp32 MFMA runs at 1/16 of the bf16 rate, and the fp32 conv family sits at 0.83 of that roof.  Here every
// fp32 operand is carried as THREE bf16 planes
//     v = hi + mid + lo,   hi = bf16(v),  mid = bf16(v - hi),  lo = bf16(v - hi - mid)       (all subtractions exact)
// which is LOSSLESS for fp32 (3 x 8 significand bits + the sign of each residual cover the 24 bits; proof in DESIGN.md
// section 3.10), and a product x * w is evaluated as the six plane products of weight >= 2^-18
//     hi*hi + hi*mid + mid*hi + hi*lo + lo*hi + mid*mid                                      (dropped: 2^-26 and below)
// on v_mfma_f32_16x16x32_bf16 with fp32 accumulation: 6 bf16 MFMAs where the fp32 pipe needs 16 bf16-MFMA-times.
//
// Data layout: activations channels-last, plane-major: [3][B][L][C] bf16 (plane 0 alone is the bf16 generator's tensor).
// The producer splits in its epilogue (this kernel's out is already split, activated for its consumer); a residual
// stored activated is inverted exactly-to-rounding in fp32 on read (as conv1d_bf16_pair2.h does).
//
// Kernel shape (the pair2 skeleton): ONE persistent workgroup per CU, 4 matrix waves (one per SIMD, 128 time rows x 32
// output channels each = a 128 x 128 output tile per step) + 4 helper waves: two stream the input in -- per 32 input
// channels one chunk of [3 planes][rows + halo][64 B], LDS-DMA (global_load_lds_dwordx4), double buffered, XOR-swizzled
// through the DMA source addresses -- and two move the finished tile out (the matrix waves split it into planes IN LDS,
// in place over the residual planes when there is a residual; the helper waves store whole rows).  Six MFMAs per
// (operand, weight-record) pair mean 0.25 ds_read_b128 + 0.06 weight records per MFMA -- a quarter of what the bf16
// pair kernel moves per MFMA -- so the k-loop is MFMA-issue-bound by construction.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <type_traits>

#include "openvoice_amd.h"

#ifndef OV_CONV1D_SPLIT3_H
#define OV_CONV1D_SPLIT3_H

namespace ovks3 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int NMW = 4;       // matrix waves: one per SIMD
constexpr int NIN = 2;       // helper waves that stream the input chunks (and the residual tile) in
constexpr int NOUT = 2;      // helper waves that store the output tile
constexpr int TT = 128;      // time rows per step
// Output tile of a step: TT rows x COT channels, COT = 128 (4 matrix waves side by side along the channels, 128 rows
// each) or 64 (C = 64: 2 x 2 waves of 64 rows x 32 channels).  It leaves through LDS in NR rounds of 3 planes x 16 KiB:
// COT = 128: two half-tiles of 64 rows x 256 B; COT = 64: the whole tile, 128 rows x 128 B.
constexpr int OPL = 16 * 1024;           // bytes per plane of a round (16 blocks of 1 KiB)
constexpr int OH = 3 * OPL;              // bytes per round

__device__ __forceinline__ uint32_t pack2(float lo, float hi) {       // one v_cvt_pk_bf16_f32, round to nearest even
  const bf16x2 h = __builtin_convertvector(f32x2{lo, hi}, bf16x2);
  uint32_t u;
  __builtin_memcpy(&u, &h, 4);
  return u;
}
__device__ __forceinline__ f32x2 unpack2(uint32_t w) {
  return f32x2{__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)};
}
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

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// (utterance, time tile, output-channel block) of a flattened step index; the channel block runs fastest, so the two
// blocks of a 256-channel layer read the same input tile back to back (the second one from L2)
// With per-utterance column limits (ov_conv1d_split3_params.col_limit: length-aware work lists, as ov_conv1d_params has
// them) the list stays DENSE: `pref` (LDS, B + 1 entries) holds the prefix sums of the utterances' tile counts and the
// utterance of a step is found by bisection -- tiles at or beyond an utterance's limit simply do not exist.
constexpr int LIMIT_MAX_BATCH = 256;
struct Step {
  int b, tile, mb;
  __device__ __forceinline__ Step(long s, int ntiles, int nmb, const int* pref, int B) {
    const long bt = s / nmb;
    mb = (int)(s - bt * nmb);
    if (!pref) {
      b = (int)(bt / ntiles);
      tile = (int)(bt - (long)b * ntiles);
    } else {                                           // largest b with pref[b] <= bt  (pref[0] = 0, pref[B] > bt)
      int lo = 0, hi = B;
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (pref[mid] <= (int)bt) lo = mid;
        else hi = mid;
      }
      b = lo;
      tile = (int)bt - pref[lo];
    }
  }
};

template <int K, int DIL, int CIN, int COT, bool RES>
struct Geo {
  static constexpr int NCT = COT / 32;               // matrix waves along the channels
  static constexpr int NTG = NMW / NCT;              // ... along time
  static constexpr int JW = 8 / NTG;                 // 16-row time fragments per wave
  static constexpr int NR = COT == 128 ? 2 : 1;      // output rounds per step
  static constexpr int OP = 2 * COT;                 // row pitch of an output round (bytes)
  static constexpr int OROWS = OPL / OP;             // its rows: 64 (two rounds) or 128 (one)
  static constexpr int RPBO = 1024 / OP;             // rows per 1 KiB block of a round: 4 or 8
  static constexpr int NCH = CIN / 32;               // 32-channel input chunks
  static constexpr int P1 = (K - 1) * DIL / 2;       // halo rows on each side
  static constexpr int R1 = TT + 2 * P1;             // rows of an input chunk
  static constexpr int NBLK = (R1 + 15) / 16;        // 1 KiB DMA blocks (16 rows x 64 B) per plane of a chunk
  static constexpr int XPL = NBLK * 1024;            // bytes per plane of a chunk buffer
  static constexpr int XB = 3 * XPL;                 // bytes per chunk buffer
  static constexpr int NOH = RES ? NR : 1;           // round buffers (with a residual every round is resident)
  // Chunk buffers: a ring of three where LDS allows (chunks are requested TWO ahead: an LDS-DMA round trip under load is
  // several microseconds, longer than the k-loop of one chunk at C = 64 -- its matrix pipes were 0.43-0.51 busy at 2.2 GHz
  // with a ring of two, profiles/r05_s4_pmc_mfma_busy_split_path.txt), else two (the C = 128 residual form)
  static constexpr int NBUF = 3 * XB + NOH * OH + COT * 4 <= 160 * 1024 ? 3 : 2;
  static constexpr int OOFF = NBUF * XB;
  static constexpr int BOFF = OOFF + NOH * OH;
  static constexpr int POFF = BOFF + COT * 4;        // prefix sums of the utterances' tile counts (length-aware lists)
  static constexpr int SMEM = POFF + (LIMIT_MAX_BATCH + 4) * 4;
  static constexpr int NPAIR = NCH * K;              // (chunk, tap) pairs of one step = 6 weight records each
  static_assert(NCH % 2 == 0, "the chunk loop is unrolled by two (static weight-ring slots for odd K)");
  static_assert(COT == 128 || COT == 64, "4 matrix waves: 1 x 4 or 2 x 2");
  static_assert(SMEM <= 160 * 1024, "LDS");
};

template <int K, int DIL, int CIN, int COT, bool RES, int NPROD>
__global__ __launch_bounds__(64 * (NMW + NIN + NOUT)) void conv1d_split3_kernel(const ov_conv1d_split3_params p) {
#ifdef OV_HOST_ONLY_KERNELS   // `make sanitize`: ASan / UBSan instrument the HOST side only; the unrolled device body (minutes
  (void)p;                    // of compile time per translation unit) is not what that build tests
#else
  using G = Geo<K, DIL, CIN, COT, RES>;
  constexpr int NCH = G::NCH, P1 = G::P1, R1 = G::R1, NBLK = G::NBLK, XPL = G::XPL, XB = G::XB, NOH = G::NOH;
  constexpr int NCT = G::NCT, JW = G::JW, NR = G::NR, OP = G::OP, OROWS = G::OROWS, RPBO = G::RPBO, NBUF = G::NBUF;
  constexpr int NPL = NPROD == 6 ? 3 : 2;            // planes the k-loop reads (3 products: hi*hi + hi*mid + mid*hi)
  static_assert(NPROD == 6 || NPROD == 3, "6 (fp32-level) or 3 (16-bit operands) plane products");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[G::SMEM];
  unsigned char* const xs = smem;                    // 2 chunk buffers: [plane][row][64 B], swizzled
  unsigned char* const ob = smem + G::OOFF;          // NOH output rounds: [plane][row][OP bytes], swizzled
  float* const bsm = reinterpret_cast<float*>(smem + G::BOFF);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = p.L;
  const int ntiles = (L + TT - 1) / TT;
  const int nmb = p.Cout / COT;
  // length-aware work list: tiles per utterance from its column limit, exclusive prefix sums in LDS (one wave: up to four
  // utterances per lane, a shuffle scan across the lanes); every role then maps step -> (utterance, tile) through it
  int* const pref_lds = reinterpret_cast<int*>(smem + G::POFF);
  const int* const pref = p.col_limit ? pref_lds : nullptr;
  long SS = (long)p.B * ntiles * nmb;
  if (pref) {
    if (wave == 0) {
      int cnt[4], sum = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int b = 4 * lane + i;
        long cols = b < p.B ? (long)p.col_limit[b] * p.col_limit_scale : 0;
        cols = cols < 0 ? 0 : (cols > L ? L : cols);
        cnt[i] = (int)((cols + TT - 1) / TT);
        sum += cnt[i];
      }
      int incl = sum;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int up = __shfl_up(incl, d, 64);
        if (lane >= d) incl += up;
      }
      int run = incl - sum;                            // exclusive prefix of this lane's first utterance
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (4 * lane + i <= p.B) pref_lds[4 * lane + i] = run;
        run += cnt[i];
      }
      if (lane == 63 && p.B == 4 * 64) pref_lds[256] = run;
    }
    __syncthreads();
    SS = (long)pref_lds[p.B] * nmb;
  }
  const long g0 = SS * blockIdx.x / gridDim.x, g1 = SS * (blockIdx.x + 1) / gridDim.x;
  if (g0 >= g1) return;
  typedef const __attribute__((address_space(1))) unsigned char* gc_ptr;
  typedef __attribute__((address_space(1))) unsigned char* gm_ptr;
  typedef __attribute__((address_space(3))) unsigned char* lds_ptr;
  // the packer's trailing all-zero record: DMA source of every vector outside [0, L)
  const gc_ptr zsrc = (gc_ptr)(p.w) + (size_t)(p.Cout / 32) * G::NPAIR * 6 * 1024;
  const int64_t PGI = 2 * (int64_t)CIN, PGO = 2 * (int64_t)p.Cout;   // global row pitches (bytes)

  if (wave >= NMW && wave < NMW + NIN) {
    // ================================ input waves ================================================
    // Chunk n (global count over this workgroup's steps) lives in buffer n % NBUF.  After barrier A(n) -- chunk n landed,
    // and every matrix wave is past the k-loop of chunk n - 1 -- the DMA of chunk n + 1 goes into the other buffer.
    // Block (plane, blk) = rows [16 blk, 16 blk + 16) of one plane: lane -> (row lrow = lane / 4, physical 16-byte slot
    // sp = lane % 4), which holds logical slot sp ^ g(row), g(row) = (row >> 2) & 3 = (lrow >> 2) & 3.
    const int iw = wave - NMW;
    const int lrow = lane >> 2, sp = lane & 3;
    const uint32_t dof = (uint32_t)(lrow * PGI + 16 * (sp ^ ((lrow >> 2) & 3)));
    auto dma_chunk = [&](int buf, const Step& st, int c) {
      const int tbase = st.tile * TT - P1;
      const lds_ptr lb = (lds_ptr)(xs) + buf * XB;
      const bool interior = tbase >= 0 && tbase + NBLK * 16 <= L;
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
        const gc_ptr xb = (gc_ptr)(p.x) + (int64_t)pl * p.x_plane * 2 + ((int64_t)st.b * L + tbase) * PGI + 64 * c;   // (uniform)
#pragma unroll
        for (int q = 0; q < (NBLK + NIN - 1) / NIN; ++q) {
          const int blk = q * NIN + iw;
          if (blk < NBLK) {
            gc_ptr bj = xb + (int64_t)blk * 16 * PGI;
            asm volatile("" : "+s"(bj));
            gc_ptr src = bj + dof;
            if (!interior) {
              const int t = tbase + blk * 16 + lrow;
              if (t < 0 || t >= L) src = zsrc + lane * 16;
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(lb + pl * XPL + blk * 1024), 16, 0, 0);
          }
        }
      }
    };
    // Residual tile (RES): every round [plane][OROWS rows][OP bytes] into the output buffers, where the matrix waves
    // replace each cell by its result.  Block blk = RPBO rows; lane -> (row lrow = lane / SPRO, physical 16-byte slot
    // sp = lane % SPRO) holding logical slot sp ^ g(row); COT = 128: g(row) = row & 15 = 4 (blk & 3) + lrow, this wave
    // takes the blocks with blk & 3 in {2 iw, 2 iw + 1}; COT = 64: g(row) = (row >> 1) & 7 = 4 (blk & 1) + (lrow >> 1),
    // this wave takes blk & 1 = iw.  One per-lane constant per swizzle phase.
    constexpr int SPRO = OP / 16, NE = COT == 128 ? 2 : 1, NQ = 8 / NE;   // blocks of a plane per wave: NQ x NE = 8
    const int lrowo = lane / SPRO, spo = lane % SPRO;
    auto oblk = [&](int w, int q, int e) { return COT == 128 ? 4 * q + 2 * w + e : 2 * q + w; };
    auto ophase = [&](int w, int e) { return COT == 128 ? 4 * (2 * w + e) + lrowo : 4 * w + (lrowo >> 1); };
    uint32_t rdof[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) rdof[e] = (uint32_t)(lrowo * PGO + 16 * (spo ^ ophase(iw, e)));
    auto dma_residual = [&](const Step& st) {
      if constexpr (RES) {
        const int t0 = st.tile * TT;
        const bool interior = t0 + TT <= L;
#pragma unroll
        for (int h = 0; h < NR; ++h)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) {
            const gc_ptr rb = (gc_ptr)(p.res) + (int64_t)pl * p.res_plane * 2 + ((int64_t)st.b * L + t0 + OROWS * h) * PGO + OP * st.mb;
            const lds_ptr lb = (lds_ptr)(ob) + h * OH + pl * OPL;
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
              for (int e = 0; e < NE; ++e) {
                const int blk = oblk(iw, q, e);
                gc_ptr bj = rb + (int64_t)blk * RPBO * PGO;
                asm volatile("" : "+s"(bj));
                gc_ptr src = bj + rdof[e];
                if (!interior && t0 + OROWS * h + RPBO * blk + lrowo >= L) src = zsrc + lane * 16;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(lb + blk * 1024), 16, 0, 0);
              }
          }
      }
    };
    // Chunk n = (step g0 + n / NCH, channel chunk n % NCH).  Ring of NBUF buffers: chunks n + 1 ... n + NBUF - 2 are in
    // flight while chunk n is consumed, chunk n + NBUF - 1 is requested after barrier A(n) into the buffer chunk n - 1
    // occupied (every matrix wave is past its k-loop by then).  LDS-DMA loads complete in issue order, so "chunk n has
    // landed" = at most the DMA instructions of the younger chunks outstanding: s_waitcnt vmcnt(DPC) for a ring of
    // three (DPC = this wave's DMA instructions per chunk, a compile-time count), vmcnt(0) for a ring of two and at the
    // tail.  The residual tile of a step is requested at its chunk 0 BEFORE that iteration's chunk request, so the wait
    // in front of the step's last chunk barrier covers it (the epilogue follows that chunk's k-loop).
    constexpr int DPC0 = NPL * ((NBLK + NIN - 1) / NIN), DPC1 = NPL * (NBLK / NIN);   // input wave 0 / 1 (NIN = 2)
    static_assert(NIN == 2 && DPC0 < 48, "vmcnt immediates");
    const long total = (g1 - g0) * NCH;
    auto request = [&](long n) {
      if (n < total) {
        const Step st(g0 + n / NCH, ntiles, nmb, pref, p.B);
        dma_chunk((int)(n % NBUF), st, (int)(n % NCH));
      }
    };
    for (int q = 0; q < NBUF - 1; ++q) request(q);
    __builtin_amdgcn_s_barrier();                            // (init: biases in LDS)
    for (long n = 0; n < total; ++n) {
      const int c = (int)(n % NCH);
      if (NBUF == 3 && n + 1 < total) {
        if (iw == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPC0) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPC1) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();                          // A(n): chunk n (and, from c = 1 on, the residual tile) landed
      // the output buffers are free from A(s, 0) on: the output waves fetched step s - 1's last round before it
      if (c == 0) dma_residual(Step(g0 + n / NCH, ntiles, nmb, pref, p.B));
      request(n + NBUF - 1);
      if (c == NCH - 1) {
        __builtin_amdgcn_s_barrier();                          // E0
        if constexpr (NR == 2) {
          __builtin_amdgcn_s_barrier();                        // E1
          __builtin_amdgcn_s_barrier();                        // E2
        }
      }
    }
    return;
  }

  if (wave >= NMW + NIN) {
    // ================================ output waves ===============================================
    // Round h of step s: LDS -> registers after barrier E0 (h = 0) / E2 (h = 1), registers -> HBM right after (the
    // stores never gate a barrier: these waves wait for nothing but their own LDS reads).  Blocks and swizzle phases
    // as in the input waves' residual pass: this wave takes blk & 3 in {2 ow, 2 ow + 1} (COT = 128) / blk & 1 = ow (64).
    const int ow = wave - NMW - NIN;
    constexpr int SPRO = OP / 16, NE = COT == 128 ? 2 : 1, NQ = 8 / NE;
    const int lrowo = lane / SPRO, spo = lane % SPRO;
    auto oblk = [&](int w, int q, int e) { return COT == 128 ? 4 * q + 2 * w + e : 2 * q + w; };
    auto ophase = [&](int w, int e) { return COT == 128 ? 4 * (2 * w + e) + lrowo : 4 * w + (lrowo >> 1); };
    uint32_t odof[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) odof[e] = (uint32_t)(lrowo * PGO + 16 * (spo ^ ophase(ow, e)));
    u32x4 ov[3][NQ][NE];
    auto fetch = [&](int hbuf) {
      const unsigned char* lb = ob + hbuf * OH + lane * 16;
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
          for (int e = 0; e < NE; ++e) ov[pl][q][e] = *reinterpret_cast<const u32x4*>(lb + pl * OPL + oblk(ow, q, e) * 1024);
    };
    auto store = [&](const Step& st, int h) {
      const int t0 = st.tile * TT + OROWS * h;
      const bool interior = t0 + OROWS <= L;
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        const gm_ptr obase = (gm_ptr)(p.out) + (int64_t)pl * p.out_plane * 2 + ((int64_t)st.b * L + t0) * PGO + OP * st.mb;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
          for (int e = 0; e < NE; ++e) {
            const int blk = oblk(ow, q, e);
            gm_ptr bj = obase + (int64_t)blk * RPBO * PGO;
            asm volatile("" : "+s"(bj));
            __attribute__((address_space(1))) u32x4* dst = reinterpret_cast<__attribute__((address_space(1))) u32x4*>(bj + odof[e]);
            if (interior || t0 + RPBO * blk + lrowo < L) *dst = ov[pl][q][e];
          }
      }
    };
    __builtin_amdgcn_s_barrier();                            // (init)
    for (long s = g0; s < g1; ++s) {
      const Step st(s, ntiles, nmb, pref, p.B);
#pragma unroll
      for (int c = 0; c < NCH; ++c) __builtin_amdgcn_s_barrier();   // A(s, c)
      __builtin_amdgcn_s_barrier();                            // E0: round 0 is complete in its buffer
      fetch(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (read before E1 / A(s + 1, 0) lets anything overwrite it)
      if constexpr (NR == 2) __builtin_amdgcn_s_barrier();     // E1: buffer 0 may be overwritten (NOH = 1: by round 1)
      store(st, 0);
      if constexpr (NR == 2) {
        __builtin_amdgcn_s_barrier();                          // E2: round 1 is complete
        fetch(NOH - 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        store(st, 1);
      }
    }
    return;
  }

  // ================================== matrix waves ================================================
  // v_mfma_f32_16x16x32_bf16, D[channel][time] = W (16 channels x 32 k) * x (32 k x 16 time rows), fragments as in
  // conv1d_bf16_pair2.h: a wave's (16 JW) x 32 tile is JW (time, j) x 2 (channel, f) fragments; a lane holds time row
  // trow0 + 16 j + (lane & 15) and channels 32 ct + 16 f + 4 (lane >> 4) + {0..3} of fragment (f, j).
  __builtin_amdgcn_s_setprio(2);
  const int g4 = lane >> 4, l15 = lane & 15;
  const int ct = wave % NCT, trow0 = (wave / NCT) * 16 * JW;

  // per-lane LDS offset of the x operand of tap `tap`: row trow0 + l15 + tap DIL (+ 16 j: same swizzle), logical slot g4
  uint32_t xl_tap[K];
#pragma unroll
  for (int tap = 0; tap < K; ++tap) {
    const int row = trow0 + l15 + tap * DIL;
    xl_tap[tap] = (uint32_t)(row * 64 + 16 * (g4 ^ ((row >> 2) & 3)));
  }
  // epilogue cells: cell (f, jj) of a round = row erow0 + 16 jj + l15 of the round's buffer (COT = 128: two rounds of
  // the wave's rows 64 h + ..., erow0 = 0; COT = 64: one round, erow0 = trow0), channels 32 ct + 16 f + 4 g4 + {0..3}:
  // logical 16-byte slot 4 ct + 2 f + (g4 >> 1), 8 bytes at 8 (g4 & 1) inside it; the row's swizzle depends on l15 only
  const int erow0 = COT == 128 ? 0 : trow0;
  const int eswz = COT == 128 ? l15 : (l15 >> 1) & 7;
  uint32_t ecell[2];
#pragma unroll
  for (int f = 0; f < 2; ++f)
    ecell[f] = (uint32_t)((erow0 + l15) * OP + 16 * ((4 * ct + 2 * f + (g4 >> 1)) ^ eswz) + 8 * (g4 & 1));

  // packed weights (ov_conv1d_split3_pack): record (((ct * NCH + c) * K + tap) * 3 + plane) * 2 + f for the 32-channel
  // output tile NCT mb + ct: the 6 records of a (chunk, tap) pair are consecutive, a step's stream is sequential.
  typedef const __attribute__((address_space(1))) u32x4* gw_ptr;
  const gw_ptr wall = (gw_ptr)(p.w);
  gw_ptr wp = wall;
  // Weight ring: WR sets of a pair's six records.  Wave tiles of 128 rows (JW = 8): two sets, the next pair (96 MFMAs =
  // 1 540 cycles away) is requested while the current one runs; slot = pair parity, static because the chunk loop is
  // unrolled by two (2 K pairs).  Wave tiles of 64 rows (JW = 4, C = 64): a pair is only 48 MFMAs (770 cycles, about one L2
  // round trip under load: the k-loops ran at 1.30x their MFMA issue time, profiles/r05_s6), so FOUR sets and requests TWO
  // pairs ahead; slot = tap % 4 -- any three consecutive pairs then sit in distinct slots for K in {3, 7, 11} (K % 4 = 3:
  // ... K-2 -> 1, K-1 -> 2, next chunk's 0 -> 0, 1 -> 1), static without any unrolling.
  constexpr int WR = JW == 4 ? 4 : 2, WAHEAD = WR == 4 ? 2 : 1;
  u32x4 wq[WR][3][2];                      // [ring slot][plane][f]
  auto wrequest = [&](int par, int r) {    // record r (= 2 plane + f) of the pair being requested; r, par compile-time
    asm volatile("" : "+s"(wp));           // opaque: scalar base + lane offset, not hoisted out of the step loop
    if (NPL == 3 || r < 4) wq[par][r >> 1][r & 1] = wp[lane + 64 * r];
  };

  // measurement only (p.dbg != NULL): ticks 0 at barriers A, 1 k-loops, 2 epilogue arithmetic, 3 at barriers E, 7 steps
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
  const float scale = p.scale, oslope = p.out_slope, inv_rslope = 1.0f / p.res_slope;
  const bool scaled = scale != 1.0f, act_out = oslope != 1.0f, res_act = p.res_slope != 1.0f;

  __syncthreads();                                    // (init; the helper waves' matching s_barrier)
  if (dbg) tlast = __builtin_readcyclecounter();
  int n = 0, nstep = 0;
  {                                                   // first pair(s) of the first step
    const Step st(g0, ntiles, nmb, pref, p.B);
    wp = wall + (size_t)(NCT * st.mb + ct) * G::NPAIR * 6 * 64;
    static_for<0, WAHEAD>([&](auto qc) {
      static_for<0, 6>([&](auto rc) { wrequest(decltype(qc)::value, decltype(rc)::value); });
      wp += 6 * 64;
    });
  }
  for (long s = g0; s < g1; ++s, ++nstep) {
    const Step st(s, ntiles, nmb, pref, p.B);
    const Step nx(s + 1 < g1 ? s + 1 : s, ntiles, nmb, pref, p.B);
    if (tid < COT) bsm[tid] = p.bias[COT * st.mb + tid];   // (read after barrier A(s, 0); last read before E0 of s - 1)
    f32x4 acc[2][JW];
    // ---- k-loops: two chunks per iteration (static ring parity: 2 K pairs) ------------------------------------------
    for (int it = 0; it < NCH / 2; ++it) {
      static_for<0, 2>([&](auto cc) {
        constexpr int ci = decltype(cc)::value;
        __syncthreads();                              // A(s, c): chunk n is in LDS
        mark(0);
        if (ci == 0 && it == 0) {
          // (bias in LDS since the barrier above)
#pragma unroll
          for (int f = 0; f < 2; ++f) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(bsm + 32 * ct + 16 * f + 4 * g4);
#pragma unroll
            for (int j = 0; j < JW; ++j) acc[f][j] = v;
          }
        }
        const uint32_t bufoff = (uint32_t)(n * XB);
        n = n + 1 == NBUF ? 0 : n + 1;
        uint32_t xlb[K];
#pragma unroll
        for (int tap = 0; tap < K; ++tap) {
          xlb[tap] = xl_tap[tap] + bufoff;
          asm volatile("" : "+v"(xlb[tap]));
        }
        // One (chunk, tap) pair = JW / 2 blocks of 24 MFMAs: block jh covers time fragments j = 2 jh, 2 jh + 1 with all
        // six plane products against the pair's six weight records; the same accumulator recurs every 4th MFMA.  Between
        // the MFMAs, one instruction per gap: the 6 operand reads of the next block, weight requests of the next pair.
        u32x4 xq[2][3][2];                            // [block parity][plane][jj]
        auto oread = [&](uint32_t base, int pl, int j) -> u32x4 {
          return *reinterpret_cast<const u32x4*>(xs + base + pl * XPL + j * 1024);
        };
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) xq[0][pl][jj] = oread(xlb[0], pl, jj);
        static_for<0, K>([&](auto tc) {
          constexpr int tap = decltype(tc)::value;
          constexpr int par = WR == 4 ? tap % 4 : (ci * K + tap) & 1;      // ring slot of this pair
          constexpr int rtap = (tap + WAHEAD) % K;                           // tap of the pair requested during this one
          constexpr int rpar = WR == 4 ? rtap % 4 : par ^ 1;                 // ... and its slot
          if constexpr (ci == 1 && tap == K - WAHEAD) {
            // the pair requested now is pair 0 of the next iteration -- or of the next step (whose output tile may differ)
            if (it + 1 == NCH / 2) wp = wall + (size_t)(NCT * nx.mb + ct) * G::NPAIR * 6 * 64;
          }
          constexpr int JB = JW / 2;                  // blocks per pair
          constexpr int MPB = 4 * NPROD;              // MFMAs per block
          constexpr int RQ0 = 2 * NPL + (NPROD == 6 ? 2 : 0);   // first weight-request gap (after the operand reads), then every 2nd
          static_for<0, JB>([&](auto jc) {
            constexpr int jh = decltype(jc)::value, blk = JB * tap + jh, cb = blk & 1, nb = cb ^ 1;
            constexpr bool more = blk + 1 < JB * K;   // the first block of the next chunk is read after its barrier
            constexpr int ntap = (blk + 1) / JB, njh = (blk + 1) % JB;
            // products (weight plane, x plane), ordered by WEIGHT plane -- hi*lo, hi*mid, hi*hi, mid*hi, mid*mid, lo*hi -- so
            // that the records requested first (hi) are the ones needed first and the lo records have 20 more MFMAs to
            // arrive: at C = 64 a pair is only 48 MFMAs (770 cycles) and the k-loops ran at 0.61 of their MFMA issue time
            // waiting for weights (profiles/r05_s5_split3_table_3_deep_chunk_ring.txt); 3 products: hi*mid, hi*hi, mid*hi
            constexpr int WPL[6] = {0, 0, 0, 1, 1, 2}, XPLN[6] = {2, 1, 0, 0, 1, 0};
            constexpr int P0 = NPROD == 6 ? 0 : 1;
            static_for<P0, P0 + NPROD>([&](auto pc) {
              constexpr int pr = decltype(pc)::value;
              static_for<0, 4>([&](auto ac) {
                constexpr int a = decltype(ac)::value, jj = a >> 1, f = a & 1, m = (pr - P0) * 4 + a;   // m-th MFMA of the block
                bf16x8 av, bv;
                __builtin_memcpy(&av, &wq[par][WPL[pr]][f], 16);
                __builtin_memcpy(&bv, &xq[cb][XPLN[pr]][jj], 16);
                __builtin_amdgcn_sched_barrier(0);
                acc[f][2 * jh + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc[f][2 * jh + jj], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (m < 2 * NPL) {
                  if constexpr (more) xq[nb][m >> 1][m & 1] = oread(xlb[ntap < K ? ntap : 0], m >> 1, 2 * njh + (m & 1));
                } else if constexpr (jh == 0 && m >= RQ0 && (m - RQ0) % 2 == 0 && (m - RQ0) / 2 < 2 * NPL) {
                  wrequest(rpar, (m - RQ0) / 2);      // all of that pair's records in the first block: earliest possible
                } else if constexpr (m == MPB - 1 && jh == JB - 1) {
                  wp += 6 * 64;
                }
                __builtin_amdgcn_sched_barrier(0);
              });
            });
          });
        });
        mark(1);
      });
    }
    // ---- epilogue: NR rounds of 4 time fragments per wave, split into planes in LDS (in place over the residual) -----
    auto half = [&](auto hc) {
      constexpr int h = decltype(hc)::value;
      unsigned char* const obuf = ob + (h < NOH ? h : NOH - 1) * OH;
      u32x2 rv[RES ? 3 : 1][8];
      if constexpr (RES) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
          for (int c = 0; c < 8; ++c)
            rv[pl][c] = *reinterpret_cast<const u32x2*>(obuf + pl * OPL + ecell[c & 1] + (c >> 1) * 16 * OP);
      }
      static_for<0, 8>([&](auto cc) {
        constexpr int c = decltype(cc)::value, f = c & 1, jj = c >> 1, j = 4 * h + jj;
        f32x2 v[2] = {f32x2{acc[f][j][0], acc[f][j][1]}, f32x2{acc[f][j][2], acc[f][j][3]}};
        if constexpr (RES) {
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            f32x2 r = (unpack2(rv[0][c][e]) + unpack2(rv[1][c][e])) + unpack2(rv[2][c][e]);   // exact: the planes of an fp32 value
            if (res_act) {
              const f32x2 m = r * inv_rslope;
              r = f32x2{vmin(r[0], m[0]), vmin(r[1], m[1])};
            }
            v[e] += r;
          }
        }
        if (scaled) { v[0] *= scale; v[1] *= scale; }
        if (act_out) {
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const f32x2 m = v[e] * oslope;
            v[e] = f32x2{vmax(v[e][0], m[0]), vmax(v[e][1], m[1])};
          }
        }
        // split: hi = bf16(v), mid = bf16(v - hi), lo = bf16(v - hi - mid); the subtractions are exact in fp32
        u32x2 pl3[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          pl3[pl] = u32x2{pack2(v[0][0], v[0][1]), pack2(v[1][0], v[1][1])};
          if (pl < 2) {
            v[0] -= unpack2(pl3[pl][0]);
            v[1] -= unpack2(pl3[pl][1]);
          }
        }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          *reinterpret_cast<u32x2*>(obuf + pl * OPL + ecell[f] + jj * 16 * OP) = pl3[pl];
      });
    };
    half(std::integral_constant<int, 0>{});
    mark(2);
    __syncthreads();                                  // E0: round 0 is in its buffer
    if constexpr (NR == 2) {
      __syncthreads();                                // E1: the output waves hold it in registers
      mark(3);
      half(std::integral_constant<int, 1>{});
      mark(2);
      __syncthreads();                                // E2: round 1 is in its buffer
    }
    // (one round: the output waves fetch it before they arrive at A(s + 1, 0), and nothing writes the buffer before that)
    mark(3);
  }
  if (dbg && lane == 0) {
    tph[7] = (unsigned long long)nstep;
#pragma unroll
    for (int q = 0; q < 8; ++q) p.dbg[((size_t)blockIdx.x * NMW + wave) * 8 + q] = tph[q];
  }
#endif
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

template <int K, int DIL, int CIN, bool RES, int NPROD>
int launch(const ov_conv1d_split3_params* p, hipStream_t stream) {
  constexpr int COT = CIN == 64 ? 64 : 128;
  static std::atomic<int> cache[16];
  const int slots = cu_count(cache);                  // one workgroup per CU
  const long SS = (long)p->B * ((p->L + TT - 1) / TT) * (p->Cout / COT);
  long nwg = p->nwg > 0 ? p->nwg : slots;
  if (nwg > SS) nwg = SS;
  hipLaunchKernelGGL((conv1d_split3_kernel<K, DIL, CIN, COT, RES, NPROD>), dim3((unsigned)nwg), dim3(64 * (NMW + NIN + NOUT)), 0,
                     stream, *p);
  return hipGetLastError() == hipSuccess ? OV_OK : OV_E_LAUNCH;
}

template <int K, int DIL, int CIN>
int launch_by_form(const ov_conv1d_split3_params* p, hipStream_t stream) {
  if (p->products != 6 && p->products != 3) return OV_E_UNSUPPORTED;
  if (p->res) {                                       // the residual form (conv2 of a ResBlock pair: dilation 1)
    if constexpr (DIL == 1) return p->products == 6 ? launch<K, DIL, CIN, true, 6>(p, stream) : launch<K, DIL, CIN, true, 3>(p, stream);
    else return OV_E_UNSUPPORTED;
  }
  return p->products == 6 ? launch<K, DIL, CIN, false, 6>(p, stream) : launch<K, DIL, CIN, false, 3>(p, stream);
}

template <int K, int DIL>
int launch_by_width(const ov_conv1d_split3_params* p, hipStream_t stream) {
  if (p->Cin == 64) return launch_by_form<K, DIL, 64>(p, stream);
  if (p->Cin == 128) return launch_by_form<K, DIL, 128>(p, stream);
  if (p->Cin == 256) return launch_by_form<K, DIL, 256>(p, stream);
  return OV_E_UNSUPPORTED;
}

// one translation unit per (kernel size, dilation): conv1d_split3_k3d1.hip ... k11d5.hip compile in parallel (the
// unrolled k-loops make these the slowest files of the library)
#define OV_SPLIT3_DECLARE(K, D) int split3_launch_k##K##d##D(const ov_conv1d_split3_params* p, hipStream_t stream);
OV_SPLIT3_DECLARE(3, 1) OV_SPLIT3_DECLARE(3, 3) OV_SPLIT3_DECLARE(3, 5)
OV_SPLIT3_DECLARE(7, 1) OV_SPLIT3_DECLARE(7, 3) OV_SPLIT3_DECLARE(7, 5)
OV_SPLIT3_DECLARE(11, 1) OV_SPLIT3_DECLARE(11, 3) OV_SPLIT3_DECLARE(11, 5)
#undef OV_SPLIT3_DECLARE

}  // namespace ovks3
#endif
