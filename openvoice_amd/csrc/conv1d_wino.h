// Winograd-domain Conv1d on the gfx950 fp32 matrix pipe: the K-tap stride-1 'same' conv of a ResBlock
// (reference: openvoice/modules.py:296-309) computed as ceil(K/3) shifted 3-tap groups, each by the minimal-filtering
// algorithm F(4, 3) -- 6 products per 4 outputs per (co, ci, group) where the direct form needs 12 -- with the
// products of all groups and input channels accumulated in the TRANSFORM domain on v_mfma_f32_32x32x2_f32:
//
//   out[co][4j + i] = sum_p At[i][p] * Y_p[co][j],      Y_p[co][j] = sum_{g, ci} U_p[co][g][ci] * V_p[g][ci][j]
//   U_p[co][g][ci]  = sum_k G[p][k] * w[co][ci][3g + k]               (weights: float64 at pack time, once)
//   V_p[g][ci][j]   = sum_m Bt[p][m] * act(x[ci][4j + 3g + m - PAD])  (inputs: VALU, on the way into LDS)
//
// i.e. six independent GEMMs (one per interpolation point p: 0, +-1, +-2, infinity) with M = co, N = tile index j,
// K_gemm = (group, ci).  The 3 ceil(K/3) - K taps that pad the last group are exact zeros, and a zero tap at the END of a
// group makes its point-infinity weight zero (G row (0, 0, 1)), one at the START of a group its point-0 weight (G row
// (1/4, 0, 0)): those products are never issued.  K = 7 is laid out as (0 w0 w1)(w2 w3 w4)(w5 w6 0) -- one zero at each
// end: two products dropped -- and K = 11 as (w0 w1 w2) ... (w9 w10 0): one dropped (wino_lead / wino_zero_product below).
// Executed MACs per 4 outputs and (co, ci): K = 3: 6 (direct 12), K = 7: 16 (28), K = 11: 23 (44).
// fp32 throughout; the transforms round where the direct form does not, so the result carries ~4x the direct kernel's
// rounding error (measured against float64: profiles/r06_*), two orders of magnitude inside the path's 1e-3 bar.
//
// One workgroup = 4 MATRIX waves (wave w owns output rows 32w .. 32w + 31 of a 128-row M-block x 32 tiles = 128
// columns: 6 accumulator fragments = 96 VGPRs) + 2 HELPER waves; 2 workgroups per CU.  Input channels are walked in
// chunks of CI.  Per chunk the helpers (a) stage the raw rows (leaky ReLU applied, zero outside [0, L)) from HBM
// through registers into LDS, one chunk ahead, channel PAIRS interleaved (raw[pair][column][2]), and (b) transform the
// previous raw chunk into V[k-step = g * CI / 2 + pair][tile][channel of the pair][6], two channels per lane in packed
// fp32 (10 ds_read_b128 -> 48 v_pk_* -> 24 ds_write2_b32 per (pair, tile) at K = 11), double buffered; ONE s_barrier per
// chunk.  A matrix wave reads its B operands as three conflict-free ds_read_b64 per k-step (two points each) and streams
// its A operands (U, fragment order, 3 KiB per k-step pair) from L2 one pair ahead; the epilogue applies At, adds bias /
// residual / MRF running sum, scales, and stores 16 bytes per lane and row (a lane's 4 outputs are consecutive columns).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <type_traits>

#include "conv1d_mfma.h"
#include "openvoice_amd.h"

namespace ovkw {

using ovk::f32x16;
using ovk::f32x2;
using ovk::f32x4;

// LDS byte address of a __shared__ object (what the DS instructions written as inline assembly take)
__device__ __forceinline__ uint32_t lds_addr(const void* ptr) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)ptr;
}

// Workgroup barrier for data exchanged through LDS only: __syncthreads() also waits for every outstanding GLOBAL load of
// the wave (vmcnt(0): the release fence of a workgroup-scope barrier) -- for a matrix wave that is its weight prefetch (an L2
// round trip issued a k-step earlier), for a helper the staging vectors it requested two chunks ahead on purpose.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int OFF>
__device__ __forceinline__ void lds_write_b32(uint32_t addr, float v) {
  static_assert(OFF >= 0 && OFF < 65536, "DS immediate offset");
  asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
// two dwords from two registers that need not be neighbours, at addr + 256 O0 and addr + 256 O1 bytes
template <int O0, int O1>
__device__ __forceinline__ void lds_write2st64_b32(uint32_t addr, float a, float b) {
  static_assert(O0 >= 0 && O0 < 256 && O1 >= 0 && O1 < 256, "DS write2st64 offsets are 8-bit, in units of 64 dwords");
  asm volatile("ds_write2st64_b32 %0, %1, %2 offset0:%3 offset1:%4" ::"v"(addr), "v"(a), "v"(b), "n"(O0), "n"(O1) : "memory");
}
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

constexpr int REC = 256;        // floats per 1 KiB weight sub-record

// Zero taps in FRONT of the K real ones (the rest of the 3 ceil(K/3) - K padding taps follow them): the conv is computed as
// a (K + lead + trail)-tap conv with 'same' padding (K - 1) / 2 + lead on the left.
constexpr int wino_lead(int K) { return K == 7 ? 1 : 0; }
constexpr int wino_trail(int K) { return (K + 2) / 3 * 3 - K - wino_lead(K); }
// the product of group g at interpolation point pt (0: point 0, 5: infinity) is identically zero -- weight and all
constexpr bool wino_zero_product(int K, int g, int pt) {
  return (pt == 0 && g == 0 && wino_lead(K) >= 1) || (pt == 5 && g == (K + 2) / 3 - 1 && wino_trail(K) >= 1);
}
// products issued per 4 outputs and (co, ci)
constexpr int wino_products(int K) {
  int n = 0;
  for (int g = 0; g < (K + 2) / 3; ++g)
    for (int pt = 0; pt < 6; ++pt) n += wino_zero_product(K, g, pt) ? 0 : 1;
  return n;
}
static_assert(wino_products(3) == 6 && wino_products(7) == 16 && wino_products(11) == 23, "executed products per tile");
constexpr int MAX_COUT = 512;   // rows of the bias vector kept in LDS

// Dilation DIL > 1 (the first conv of each ResBlock pair, reference openvoice/modules.py:228-251): an output at column
// t reads x[t + (tap - PAD) DIL], so the columns of one residue class r = t mod DIL form a dilation-1 problem of their own.
// A Winograd tile is then 4 outputs DIL apart, r + DIL (4j + i), and an N-block of 32 NF tiles holds ALL DIL classes of a
// contiguous column range: tile n = r * J + j with J = (32 NF) / DIL tiles per class, 4 J DIL columns per block (the
// 32 NF - DIL J left-over tiles compute a duplicate of tile 0 and are not stored: 1.6 % at DIL = 3, 6.3 % at DIL = 5).
template <int K, int DIL, int NF>
struct GeoD {
  static constexpr int G = (K + 2) / 3;
  static constexpr int NT = 32 * NF;
  static constexpr int J = NT / DIL;                             // tiles per residue class
  static constexpr int NCOL = 4 * J * DIL;                       // output columns per N-block
  static constexpr int PADD = ((K - 1) / 2 + wino_lead(K)) * DIL;   // 'same' padding in columns (leading zero taps included)
  static constexpr int PADA = (PADD + 3) / 4 * 4;                // raw rows start at column t0 - PADA (16-byte aligned)
  static constexpr int NV = 3 * (G - 1) + 6;                     // inputs a tile reads, DIL apart
  static constexpr int RW = (PADA - PADD + DIL - 1 + DIL * (4 * (J - 1) + NV - 1) + 1 + 3) / 4 * 4;   // floats per raw row
  static_assert(DIL == 3 || DIL == 5, "dilated instances: 3 and 5");
};

// Geometry of a K-tap conv: group g holds taps 3g - lead .. 3g - lead + 2 (taps outside [0, K) are zero).
template <int K>
struct Geo {
  static constexpr int G = (K + 2) / 3;
  static constexpr int PAD = (K - 1) / 2 + wino_lead(K);
  static constexpr int OFF0 = 8 - PAD;                          // raw index of input m = 0 of (tile 0, group 0)
  static constexpr int WSTART = OFF0 / 4 * 4;                   // aligned start of a tile's input window
  static constexpr int WLEN = OFF0 - WSTART + 3 * (G - 1) + 6;  // floats of the window all groups read
  static constexpr int NB128 = (WLEN + 3) / 4;
  static_assert(PAD <= 8 && WSTART + 4 * NB128 <= 20, "input window exceeds the raw row (128 NF + 16 floats)");
};

// Packed weights: sub-record ((mt * nchunks + c) * NPAIR + sp) * 3 + j, 64 lanes x 4 floats; lane l holds elements
// 4j .. 4j + 3 of its 12-vector [k-step 2sp: p = 0..5][k-step 2sp + 1: p = 0..5] of row 32 mt + (l & 31), k-row
// kk = 2 * kstep + (l >> 5) of chunk c, kk = g * CI + ci_local.  Three zero sub-records close the stream.
__host__ __device__ inline size_t wino_pack_floats(int Cout, int Cin, int K, int CI) {
  const int G = (K + 2) / 3;
  return ((size_t)(Cout / 32) * (Cin / CI) * (CI * G / 4) * 3 + 3) * REC;
}

// phase timers (DBG instances only: the dispatcher selects them when ov_conv1d_wino_params.dbg is set)
#ifndef OVW_EXP
#define OVW_EXP 0   // measurement builds only (scripts/exp_wino.sh): 1 = MFMAs replaced by one FMA; 2 = helper transform skipped;
                    // 3 = no weight stream (the first fragments of an item are reused); 4 = no epilogue stores; 5 = no B-operand
                    // reads inside the k-loop (the chunk's first operands are reused); 6 = helpers idle (barriers only); 7 = helpers idle and NO chunk barriers at all; 8 = 7 + 3 + 5
                    // (the bare MFMA loop)
#endif
#define OVW_MARK(q)                                                \
  if constexpr (DBG) {                                             \
    const unsigned long long now_ = __builtin_readcyclecounter(); \
    tph[q] += now_ - tlast;                                        \
    tlast = now_;                                                  \
  }

// NF = MFMA N fragments (of 32 tiles = 128 columns) per matrix wave.  NF = 1: 96 accumulator registers, 2 helper waves, two
// workgroups per CU (they cover each other's barriers and epilogues).  NF = 2: 192 accumulator registers, 4 helper
// waves, ONE workgroup per CU -- every A fragment feeds two MFMAs, which halves the weight stream from L2 (at NF = 1 the
// 512 workgroups pull 12 TB/s of fragments, 3/4 of what the L2s deliver with nothing else running: profiles/r06_s4).
// MW = matrix waves along M.  MW = 4: the four waves own the four 32-row fragments of a 128-row M-block and share one
// column block.  MW = 2 (Cout a multiple of 64 only: the C = 64 stage): two 32-row fragments x TWO adjacent column
// sub-blocks per workgroup (wave w: rows 32 (w & 1), sub-block w >> 1), so the workgroup still runs four matrix waves on a
// 64-row layer; the waves of a sub-block pair stream the same A fragments.  MW = 1 (Cout a multiple of 32: the C = 32 stage,
// K = 11): ONE 32-row fragment x FOUR column sub-blocks (1 024 columns per workgroup), all four waves on the same A stream.
// The helpers' work per MFMA doubles with every halving of the rows (the transform of a tile is shared by fewer row
// fragments); what keeps them off the critical path is their instruction count -- see the helper section.
template <int K, int DIL, int CI, int NF, int MW, bool DBG>
__global__ __launch_bounds__(64 * (4 + 2 * NF), NF == 1 ? 3 : 2) void conv1d_wino_kernel(const ov_conv1d_wino_params p) {
  using Ge = Geo<K>;
  using Gd = GeoD<K, DIL == 1 ? 3 : DIL, NF>;   // (only read when DIL > 1)
  constexpr int G = Ge::G, KR = CI * G, NSTEP = KR / 2, NPAIR = NSTEP / 2;
  constexpr int NB = 4 / MW;           // column sub-blocks per workgroup
  constexpr int NTS = 32 * NF;         // Winograd tiles per sub-block (what one matrix wave owns)
  constexpr int NT = NTS * NB;         // Winograd tiles per N-block
  constexpr int NCOLS = DIL == 1 ? 4 * NTS : Gd::NCOL;   // output columns per sub-block
  constexpr int NCOL = NCOLS * NB;     // output columns per N-block
  constexpr int RW = (DIL == 1 ? NCOLS + 16 : Gd::RW) + (NB - 1) * NCOLS;   // floats per raw LDS row: column t0 - ORG + c at index c
  static_assert(MW == 4 || ((MW == 2 || MW == 1) && NF == 2), "two- and one-row-fragment workgroups run two fragments per wave");
  constexpr int ORG = DIL == 1 ? 8 : Gd::PADA;
  constexpr int RW4 = RW / 4;
  constexpr int NHELP = 2 * NF;        // helper waves
  static_assert(DIL == 1 || NF == 2, "dilated instances run two fragments per wave");
  static_assert(KR % 4 == 0 && CI % 2 == 0, "k-rows in whole k-step pairs");
  constexpr int NPR = CI / 2;            // input-channel pairs per chunk (a pair = one k-step of each group)
  constexpr int NITEM = NPR * RW4;       // staging items per chunk: (channel pair, 4 columns) = two 16-byte vectors
  constexpr int RAWBUF = CI * RW;        // floats per raw buffer: raw[pair][raw index][channel of the pair]
  constexpr int VBUF = KR * NT * 6;      // floats per V buffer: V[k-row = g * CI + channel][tile][6]
  __shared__ __attribute__((aligned(16))) float raw[2 * RAWBUF];
  __shared__ __attribute__((aligned(16))) float Vs[2 * VBUF];
  __shared__ __attribute__((aligned(16))) float bias_s[MAX_COUT];   // read back 16 bytes at a time at every item start
  // DIL > 1: a lane's four outputs are DIL columns apart; they leave through an 8-row x 256-column tile per matrix wave
  // (written 4 bytes at a time, read back as whole rows) so that the global stores are 16 bytes per lane as at DIL = 1
  constexpr int OST = 256;
  __shared__ __attribute__((aligned(16))) float ostage[DIL == 1 ? 4 : 4 * 8 * OST];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = p.L;
  const int nchunks = p.Cin / CI;
  const int ntiles = (L + NCOL - 1) / NCOL;
  const int mblocks = p.Cout / (32 * MW);
  int total = ntiles * mblocks * p.B;
  // Length-aware work list (ov_conv1d_wino_params.col_limit, as ov_conv1d_params.col_limit): utterance b contributes only
  // the N-blocks that start before its column limit; lim_pref[b] = blocks of the utterances before b, so the list stays
  // dense and is dealt evenly whatever the lengths are.  One wave computes the prefix sums once per workgroup.
  __shared__ int lim_pref[ovk::LIMIT_MAX_BATCH + 1];
  const bool limited = p.col_limit != nullptr;
  if (limited) {
    if (wave == 0) {
      int carry = 0;
      for (int base = 0; base < p.B; base += 64) {
        const int bb = base + lane;
        int nb = 0;
        if (bb < p.B) {
          long long c = (long long)p.col_limit[bb] * p.col_limit_scale;
          c = c < 0 ? 0 : (c > L ? L : c);
          nb = ((int)c + NCOL - 1) / NCOL;
        }
        int sc = nb;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const int v = __shfl_up(sc, d, 64);
          if (lane >= d) sc += v;
        }
        if (bb < p.B) lim_pref[bb + 1] = carry + sc;
        carry += __shfl(sc, 63, 64);
      }
      if (lane == 0) lim_pref[0] = 0;
    }
    __syncthreads();
    total = __builtin_amdgcn_readfirstlane(lim_pref[p.B]) * mblocks;
  }
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
  // XCD-contiguous order (see conv1d_mfma.h): XCD x = blockIdx.x % 8 owns a contiguous eighth of the list
  int wid0 = blockIdx.x, wend = total, wstride = gridDim.x;
  if ((gridDim.x & 7u) == 0) {
    const int xcd = blockIdx.x & 7;
    const int lo = (int)(((long long)xcd * total) >> 3);
    wend = (int)(((long long)(xcd + 1) * total) >> 3);
    wid0 = lo + (int)(blockIdx.x >> 3);
    wstride = gridDim.x >> 3;
  }
  if (wid0 >= wend) return;
  const int nitems = (wend - wid0 + wstride - 1) / wstride;
  const int nstream = nitems * nchunks;      // chunks this workgroup walks, across its items
  for (int i = tid; i < p.Cout; i += 64 * (4 + NHELP)) bias_s[i] = p.bias[i];   // visible after barrier (A)

  const bool is_helper = wave >= 4;
  if (is_helper) {
    // ================================ helper waves ================================================
    // A helper shares its SIMD with a matrix wave that always has an MFMA waiting for the pipe, i.e. a standing claim on
    // the vector-ALU issue slot: every VALU instruction of a helper waits for a gap (~40 cycles each at priority 3, ~64
    // without: phase timers, profiles/r06_s15).  What a helper costs is therefore its VALU INSTRUCTION COUNT, and the
    // code below is written against that: two input channels per lane in packed fp32 (v_pk_fma / v_pk_add / v_pk_mul_f32
    // -- the raw rows are channel-pair interleaved in LDS so that one ds_read_b128 delivers two columns of both channels
    // as aligned register pairs), ds_write2_b32 for everything stored (two registers that need not be neighbours, two
    // immediate offsets: no moves to build vectors), staging addresses as scalar base + loop-invariant lane offset.
    __builtin_amdgcn_s_setprio(3);
    const float slope = p.in_slope;
    const uint32_t ldx = (uint32_t)p.x_ld;
    const int hl = (wave - 4) * 64 + lane;
    constexpr int NLOAD = (NITEM + 64 * NHELP - 1) / (64 * NHELP);
    // TWO register sets: the vectors of chunk c travel in set c & 1 and are requested a whole chunk period before they are
    // written to LDS -- with one set, request and use sat in the same period, and a chunk could not be shorter than an
    // HBM round trip under load (~2.5 us: the floor of every instance whose k-loop is shorter, profiles/r06_s16)
    f32x4 sa[2][NLOAD], sb[2][NLOAD];    // the staged vector of the pair's first / second channel
    uint32_t sbyte[NLOAD];               // byte offset of (row 2 pr, raw index 4 c4) from (first row of the chunk, raw index 0)
    int scol[NLOAD];                     // 4 c4
    uint32_t hasbits = 0, okbits[2] = {0, 0};
    bool edge[2] = {false, false};
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
      const int idx = i * (64 * NHELP) + hl;
      const int pr = idx / RW4, c4 = idx - pr * RW4;
      const bool has = idx < NITEM;
      hasbits |= (has ? 1u : 0u) << i;
      scol[i] = 4 * c4;
      sbyte[i] = has ? 4u * (2u * (uint32_t)pr * ldx + 4u * (uint32_t)c4) : 0u;
    }
    const uint32_t raw_lds = lds_addr(raw) + 32u * (uint32_t)hl;   // this lane's item of staging round 0: 8 floats per item
    const uint32_t v_lds = lds_addr(Vs);
    int lwid = wid0, lchunk = 0, lpos = 0;                   // position of the staging stream
    int lb, ltile, lmblk;
    decode(lwid, lb, ltile, lmblk);

    auto issue_loads = [&](auto setc) {
      constexpr int S = decltype(setc)::value;
      const bool live = lpos < nstream;
      const int tb = ltile * NCOL - ORG;                     // column of raw index 0 (a multiple of 4)
      const float* __restrict__ xrow = p.x + (int64_t)lb * p.x_bstride + (int64_t)(lchunk * CI) * ldx;
      // interior blocks (all but the first and last of an utterance): every vector lies inside [0, L) -- no lane arithmetic
      edge[S] = !(live && tb >= 0 && tb + RW <= L);
      if (!edge[S]) {
        const char* b0 = reinterpret_cast<const char*>(xrow + tb);
        const char* b1 = reinterpret_cast<const char*>(xrow + tb + ldx);
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
          sa[S][i] = *reinterpret_cast<const f32x4*>(b0 + sbyte[i]);
          sb[S][i] = *reinterpret_cast<const f32x4*>(b1 + sbyte[i]);
        }
      } else {
        const char* b0 = reinterpret_cast<const char*>(xrow);
        const char* b1 = reinterpret_cast<const char*>(xrow + ldx);
        uint32_t okb = 0;
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
          const int t = tb + scol[i];                        // multiple of 4: a vector is wholly inside or outside
          const bool ok = live && ((hasbits >> i) & 1u) && t >= 0 && t < L;
          const uint32_t off = ok ? sbyte[i] + 4u * (uint32_t)tb : 0u;
          okb |= (ok ? 1u : 0u) << i;
          sa[S][i] = *reinterpret_cast<const f32x4*>(b0 + off);
          sb[S][i] = *reinterpret_cast<const f32x4*>(b1 + off);
        }
        okbits[S] = okb;
      }
      ++lpos;
      if (++lchunk == nchunks) {
        lchunk = 0;
        lwid += wstride;
        if (lwid < wend) decode(lwid, lb, ltile, lmblk);
      }
    };
    // leaky ReLU as max(v, slope v) (0 < slope <= 1: the same value as v > 0 ? v : slope v for every v, NaN included)
    auto act2 = [&](f32x2 v) {
      const f32x2 sv = v * f32x2{slope, slope};
      f32x2 r;
      asm("v_max_f32 %0, %1, %2" : "=v"(r[0]) : "v"(v[0]), "v"(sv[0]));
      asm("v_max_f32 %0, %1, %2" : "=v"(r[1]) : "v"(v[1]), "v"(sv[1]));
      return r;
    };
    auto write_raw = [&](int buf, auto setc) {
      constexpr int S = decltype(setc)::value;
      const uint32_t base = raw_lds + (uint32_t)buf * (RAWBUF * 4);
#pragma unroll
      for (int i = 0; i < NLOAD; ++i) {
        if ((hasbits >> i) & 1u) {
          f32x2 a01 = {sa[S][i][0], sa[S][i][1]}, a23 = {sa[S][i][2], sa[S][i][3]};
          f32x2 b01 = {sb[S][i][0], sb[S][i][1]}, b23 = {sb[S][i][2], sb[S][i][3]};
          if (slope != 1.f) {
            a01 = act2(a01); a23 = act2(a23); b01 = act2(b01); b23 = act2(b23);
          }
          if (edge[S] && !((okbits[S] >> i) & 1u)) a01 = a23 = b01 = b23 = f32x2{0.f, 0.f};
          const uint32_t ad = base + (uint32_t)i * (64 * NHELP * 32);   // raw[pair][4 c4 + e][channel]: item idx at 8 idx floats
          asm volatile("ds_write2_b32 %0, %1, %2 offset1:1" ::"v"(ad), "v"(a01[0]), "v"(b01[0]) : "memory");
          asm volatile("ds_write2_b32 %0, %1, %2 offset0:2 offset1:3" ::"v"(ad), "v"(a01[1]), "v"(b01[1]) : "memory");
          asm volatile("ds_write2_b32 %0, %1, %2 offset0:4 offset1:5" ::"v"(ad), "v"(a23[0]), "v"(b23[0]) : "memory");
          asm volatile("ds_write2_b32 %0, %1, %2 offset0:6 offset1:7" ::"v"(ad), "v"(a23[1]), "v"(b23[1]) : "memory");
        }
      }
    };
    // raw[buf] -> V[buf]: item = (channel pair, tile); V[k-row = g * CI + 2 pair + channel][tile][6]
    constexpr int ROUNDS = NPR * NT / (64 * NHELP);
    static_assert(NPR * NT % (64 * NHELP) == 0, "whole rounds of helper lanes");
    // loop-invariant byte offsets of a lane's items: the raw window, and -- per V buffer and point -- the LDS address of the
    // pair's first channel's record of group 0 (12 registers per round that a helper has to spare: the stores below then
    // need no address arithmetic at all)
    uint32_t tsrc[ROUNDS], vadr[2][ROUNDS][6];
    static_assert((NT * 24) % 256 == 0 && (CI * NT * 24) % 256 == 0, "channel / group record distances in 256-byte units");
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const int idx = r * (64 * NHELP) + hl;
      const int tile = idx & (NT - 1), pr = idx / NT;        // tile of the N-block = sub-block * NTS + tile of the sub-block
      if constexpr (DIL == 1) {
        tsrc[r] = 8u * (uint32_t)(pr * RW + Ge::WSTART + 4 * tile);
      } else {
        // tile -> (residue class, tile of the class); the left-over tiles duplicate tile 0 (never stored)
        const int sub = tile / NTS, tl = tile - sub * NTS;
        const int rc0 = tl / Gd::J, rc = rc0 < DIL ? rc0 : 0, jt = rc0 < DIL ? tl - rc0 * Gd::J : 0;
        tsrc[r] = 8u * (uint32_t)(pr * RW + (Gd::PADA - Gd::PADD) + sub * NCOLS + rc + 4 * DIL * jt);
      }
#pragma unroll
      for (int bf = 0; bf < 2; ++bf)
#pragma unroll
        for (int q = 0; q < 6; ++q) vadr[bf][r][q] = v_lds + (uint32_t)bf * (VBUF * 4) + 24u * (uint32_t)(2 * pr * NT + tile) + 4u * q;
    }
    auto transform = [&](auto bufc) {
      constexpr int buf = decltype(bufc)::value;
      const char* src = reinterpret_cast<const char*>(raw) + buf * (RAWBUF * 4);
#pragma unroll
      for (int r = 0; r < ROUNDS; ++r) {
        constexpr int NWIN = DIL == 1 ? 4 * Ge::NB128 : Gd::NV;
        f32x2 win[NWIN];                                     // (channel 2 pr, channel 2 pr + 1) at NWIN consecutive inputs
        if constexpr (DIL == 1) {
#pragma unroll
          for (int q = 0; q < 2 * Ge::NB128; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + tsrc[r] + 16 * q);
            win[2 * q] = f32x2{v[0], v[1]};
            win[2 * q + 1] = f32x2{v[2], v[3]};
          }
        } else {
#pragma unroll
          for (int u = 0; u < Gd::NV; ++u) win[u] = *reinterpret_cast<const f32x2*>(src + tsrc[r] + 8 * DIL * u);
        }
        static_for<0, G>([&](auto gc) {
          constexpr int g = decltype(gc)::value;
          constexpr int o = (DIL == 1 ? Ge::OFF0 - Ge::WSTART : 0) + 3 * g;
          const f32x2 d0 = win[o], d1 = win[o + 1], d2 = win[o + 2], d3 = win[o + 3], d4 = win[o + 4], d5 = win[o + 5];
          // Bt d, points 0, 1, -1, 2, -2, infinity: 12 packed operations for the 12 values of the channel pair
          const f32x2 c4 = {4.f, 4.f}, m4 = {-4.f, -4.f}, m5 = {-5.f, -5.f}, c2 = {2.f, 2.f}, m2 = {-2.f, -2.f};
          const f32x2 t1 = __builtin_elementwise_fma(m4, d2, d4), t2 = __builtin_elementwise_fma(m4, d1, d3);
          const f32x2 t3 = d4 - d2, t4 = d3 - d1;
          // (a point whose product is identically zero -- wino_zero_product -- is neither computed nor stored: its V slot is
          // never read into an MFMA)
          constexpr bool Z0 = wino_zero_product(K, g, 0), Z5 = wino_zero_product(K, g, 5);
          const f32x2 v1 = t1 + t2;
          const f32x2 v2 = t1 - t2;
          const f32x2 v3 = __builtin_elementwise_fma(c2, t4, t3);
          const f32x2 v4 = __builtin_elementwise_fma(m2, t4, t3);
          // V[k-row g CI + 2 pr + channel][tile][6]: the records of the pair's channels are NT * 24 bytes apart, those of
          // consecutive groups CI * NT * 24 -- both whole multiples of 256 bytes, i.e. the two offsets of a
          // ds_write2st64_b32: ONE store per point writes both channels' values from their (non-adjacent) registers
          constexpr int OA = g * CI * NT * 24 / 256, OB = OA + NT * 24 / 256;
          constexpr int bf = buf;
          if constexpr (!Z0) {
            const f32x2 v0 = __builtin_elementwise_fma(c4, d0, __builtin_elementwise_fma(m5, d2, d4));
            lds_write2st64_b32<OA, OB>(vadr[bf][r][0], v0[0], v0[1]);
          }
          lds_write2st64_b32<OA, OB>(vadr[bf][r][1], v1[0], v1[1]);
          lds_write2st64_b32<OA, OB>(vadr[bf][r][2], v2[0], v2[1]);
          lds_write2st64_b32<OA, OB>(vadr[bf][r][3], v3[0], v3[1]);
          lds_write2st64_b32<OA, OB>(vadr[bf][r][4], v4[0], v4[1]);
          if constexpr (!Z5) {
            const f32x2 v5 = __builtin_elementwise_fma(c4, d1, __builtin_elementwise_fma(m5, d3, d5));
            lds_write2st64_b32<OA, OB>(vadr[bf][r][5], v5[0], v5[1]);
          }
        });
      }
    };
    // the LDS stores above are inline assembly the compiler's counters do not see: drain them before every barrier
    auto hand_over = [&]() { lds_barrier(); };

    unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tlast = DBG ? __builtin_readcyclecounter() : 0ull;
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    issue_loads(S0{});
    write_raw(0, S0{});           // chunk 0
    issue_loads(S1{});            // chunk 1 in set 1
    hand_over();                  // (A) raw[0] complete
    transform(std::integral_constant<int, 0>{});
    write_raw(1, S1{});
    issue_loads(S0{});            // chunk 2 in set 0
    hand_over();                  // (B) V[0], raw[1] complete
    OVW_MARK(4)
    // period i (the matrix waves multiply V[i & 1]): request chunk i + 3 (set (i + 1) & 1), transform chunk i + 1, write
    // chunk i + 2 (set i & 1, requested a period ago) to raw[i & 1]; nothing is requested beyond the stream's end
    auto period = [&](int i, auto par) {
      constexpr int P = decltype(par)::value;
      if (OVW_EXP < 6) issue_loads(std::integral_constant<int, 1 - P>{});
      OVW_MARK(0)
      if (OVW_EXP != 2 && OVW_EXP < 6 && i + 1 < nstream) transform(std::integral_constant<int, 1 - P>{});
      OVW_MARK(1)
      if (OVW_EXP < 6) write_raw(P, par);
      OVW_MARK(2)
      if (OVW_EXP < 7) hand_over();                // V[(i + 1) & 1] and raw[i & 1] handed over; V[i & 1] free again
      OVW_MARK(3)
    };
    for (int i = 0; i < nstream; i += 2) {
      period(i, S0{});
      if (i + 1 < nstream) period(i + 1, S1{});
    }
    if constexpr (DBG) {
      if (lane == 0) {
        unsigned long long* d = p.dbg + ((size_t)blockIdx.x * 8 + wave) * 8;
        tph[7] = (unsigned long long)nstream;
#pragma unroll
        for (int q = 0; q < 8; ++q) d[q] = tph[q];
      }
    }
    return;
  }

  // ================================== matrix waves ================================================
  const int half = lane >> 5, n = lane & 31;
  const f32x4* __restrict__ wbase = reinterpret_cast<const f32x4*>(p.w);
  const uint32_t LD = (uint32_t)p.out_ld;
  int wid = wid0, b, tile, mblk;
  decode(wid, b, tile, mblk);
  const int wrow = MW == 4 ? wave : (MW == 2 ? (wave & 1) : 0), wsub = MW == 4 ? 0 : (MW == 2 ? (wave >> 1) : wave);
  int mtile = mblk * MW + wrow;
  // sub-record index of the first k-step pair of (mtile, chunk 0); the stream of an item is contiguous
  uint32_t rec = (uint32_t)mtile * (uint32_t)nchunks * (uint32_t)(NPAIR * 3);
  // (A ring of three pairs -- requests two pairs ahead -- was built for K = 7, where a chunk is a whole number of ring
  // revolutions: no gain with the helpers running, profiles/r06_s29; any form with several copies of the chunk body by ring
  // phase makes hipcc spill 400-600 registers.)
  f32x4 a_cur[3], a_nxt[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) a_cur[j] = wbase[(size_t)(rec + j) * 64 + lane];
  const int voffB = (half * NT + wsub * NTS + n) * 6;   // this lane's float offset inside a k-step's two k-rows (fragment 0)

  __syncthreads();   // (A)
  __syncthreads();   // (B)
  unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast = DBG ? __builtin_readcyclecounter() : 0ull;
  int it = 0;
  while (true) {
    f32x16 acc[6][NF];
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
      for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][f][r] = 0.f;
    // At column of point 1 is (1, 1, 1, 1): a bias there reaches all four outputs of a tile
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 bq = *reinterpret_cast<const f32x4*>(bias_s + mtile * 32 + 4 * half + 8 * i);
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        acc[1][f][4 * i] = bq[0]; acc[1][f][4 * i + 1] = bq[1]; acc[1][f][4 * i + 2] = bq[2]; acc[1][f][4 * i + 3] = bq[3];
      }
    }

    OVW_MARK(3)
    // B operands: ONE register set, refilled in place -- the two points of pair qq are read for the next k-step (of this chunk
    // or step 0 of the item's next one) right after the MFMAs that consumed them have issued (4 NF MFMAs = 256 NF cycles later
    // they are needed again).  Filled here for the item's first chunk (the set is not kept across the epilogue: its twelve
    // registers are part of the operand ring there).
    f32x2 bq[NF][3];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int qq = 0; qq < 3; ++qq) bq[f][qq] = *reinterpret_cast<const f32x2*>(Vs + (it & 1) * VBUF + voffB + f * (32 * 6) + 2 * qq);
    for (int chunk = 0; chunk < nchunks; ++chunk, ++it) {
      const float* vb = Vs + (it & 1) * VBUF + voffB;
      const float* vbn = Vs + ((it + 1) & 1) * VBUF + voffB;   // the item's next chunk
      const bool more_chunks = chunk + 1 < nchunks;
#pragma unroll
      for (int s = 0; s < NSTEP; ++s) {
        const int s2 = s & 1;
        if (s == NSTEP - 1) {
          // The chunk hand-over sits BEFORE the last k-step, not after it: that step's B operands are in registers already
          // (nothing reads V[it & 1] from LDS any more), so the helpers may start overwriting it, and the first operands of
          // the next chunk are read behind this step's MFMAs instead of in a bubble after the barrier -- measured with idle
          // helpers a chunk cost its MFMA issue time + ~1 100 cycles whatever its length (profiles/r06_s24)
          OVW_MARK(0)
          if (OVW_EXP < 7) lds_barrier();
          OVW_MARK(1)
        }
        if (s2 == 0) {
          rec += 3;   // the sub-records after the last real pair are zeros written by the packer
#pragma unroll
          for (int j = 0; j < 3; ++j) a_nxt[j] = (OVW_EXP == 3 || OVW_EXP == 8) ? a_cur[j] : wbase[(size_t)(rec + j) * 64 + lane];
        }
#pragma unroll
        for (int qq = 0; qq < 3; ++qq) {
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const int q = 2 * qq + t, e = s2 * 6 + q;
            if (wino_zero_product(K, 2 * s / CI, q)) continue;   // (s is a compile-time value: the loop is unrolled)
#pragma unroll
            for (int f = 0; f < NF; ++f) {
#if OVW_EXP == 1   // measurement build: no MFMAs (what do the helpers cost when the matrix pipe is idle?)
              acc[q][f][0] += a_cur[e >> 2][e & 3] * bq[f][qq][t];
#else
              acc[q][f] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[e >> 2][e & 3], bq[f][qq][t], acc[q][f], 0, 0, 0);
#endif
            }
          }
          __builtin_amdgcn_sched_barrier(0);
          if (OVW_EXP != 5 && OVW_EXP != 8 && (s + 1 < NSTEP || more_chunks)) {
#pragma unroll
            for (int f = 0; f < NF; ++f)
              bq[f][qq] = *reinterpret_cast<const f32x2*>((s + 1 < NSTEP ? vb + (s + 1) * (2 * NT * 6) : vbn) + f * (32 * 6) + 2 * qq);
          }
        }
        if (s2 == 1) {
#pragma unroll
          for (int j = 0; j < 3; ++j) a_cur[j] = a_nxt[j];
        }
      }
    }

    // next work item: its first weight pair and bias are in flight while this item's epilogue runs
    const int nwid = wid + wstride;
    const bool more = nwid < wend;
    int nb = 0, ntile = 0, nmblk = 0;
    if (more) decode(nwid, nb, ntile, nmblk);
    const int nmtile = nmblk * MW + wrow;
    const int ob = b, otile = tile, omtile = mtile;
    if (more) {
      rec = (uint32_t)nmtile * (uint32_t)nchunks * (uint32_t)(NPAIR * 3);
#pragma unroll
      for (int j = 0; j < 3; ++j) a_cur[j] = wbase[(size_t)(rec + j) * 64 + lane];
      wid = nwid; b = nb; tile = ntile; mblk = nmblk; mtile = nmtile;
    }

    // ---- epilogue: out = (At Y + res + add) * scale, 16 bytes per lane and row --------------------------------
    // Stores and loads share one in-order counter (vmcnt): a wait for an operand load issued AFTER a store is a wait for
    // that store's acknowledgement.  The operand rows are therefore requested one group AHEAD of the stores (two register
    // sets), and launches without operands (every first conv of a pair) take a variant with no loads at all.
    {
      const float scale = p.scale;
      const float oslope = p.out_slope;
      const bool act_out = oslope < 1.f;      // (launcher: 0 < out_slope <= 1, < 1 only without res / add)
      float* outb = p.out + (int64_t)ob * p.out_bstride;
      const float* resb = p.res ? p.res + (int64_t)ob * p.res_bstride : nullptr;
      const float* addb = p.add ? p.add + (int64_t)ob * p.add_bstride : nullptr;
      const uint32_t rbase = (uint32_t)omtile * 32u + 4u * (uint32_t)half;
      auto at4 = [&](int f, int r) {
        const float y0 = acc[0][f][r], y1 = acc[1][f][r], y2 = acc[2][f][r], y3 = acc[3][f][r], y4 = acc[4][f][r],
                    y5 = acc[5][f][r];
        const float s1 = y1 + y2, d1 = y1 - y2, s2 = y3 + y4, d2 = y3 - y4;
        f32x4 o;
        o[0] = (y0 + s1) + s2;
        o[1] = __builtin_fmaf(2.f, d2, d1);
        o[2] = __builtin_fmaf(4.f, s2, s1);
        o[3] = __builtin_fmaf(8.f, d2, d1) + y5;
        return o;
      };
      if constexpr (DIL == 1) {
        auto run = [&](auto has_res, auto has_add) {
          constexpr bool HR = decltype(has_res)::value, HA = decltype(has_add)::value;
          constexpr int NGRP = 16 * NF;                      // one accumulator row (4 columns) per group
          // addressing: 64-bit row bases stay scalar ((32 mtile + row of the register) * LD), a lane carries ONE 32-bit
          // offset per fragment (its half's 4 rows + its column): loads / stores take the saddr + voffset form
          uint32_t voff[NF];
          bool colok[NF];
#pragma unroll
          for (int f = 0; f < NF; ++f) {
            const uint32_t col = (uint32_t)otile * NCOL + (uint32_t)(wsub * NCOLS) + 128u * f + 4u * (uint32_t)n;
            colok[f] = col < (uint32_t)L;                    // L % 4 == 0: a lane's 4 columns are in or out together
            voff[f] = 4u * (uint32_t)half * LD + (colok[f] ? col : 0u);
          }
          const size_t mrow = (size_t)omtile * 32u;
          // operand ring: the rows of DEPTH - 1 groups are in flight while one is consumed (32 registers of operands).  A ramped
          // ring that grows into the accumulator registers each consumed group frees was tried (round 6): hipcc does not
          // reuse the freed elements of the 16-register accumulator tuples and spills 30-50 registers instead.
          constexpr int DEPTH = (HR && HA) ? 4 : 8;
          f32x4 rv[HR ? DEPTH : 1], av[HA ? DEPTH : 1];
          auto load = [&](int g) {                           // group g = (fragment, accumulator register r)
            const int f = g / 16, r = g % 16;
            const size_t rowb = (mrow + (size_t)((r & 3) + 8 * (r >> 2))) * LD;
            if constexpr (HR) rv[g % DEPTH] = *reinterpret_cast<const f32x4*>(resb + rowb + voff[f]);
            if constexpr (HA) av[g % DEPTH] = *reinterpret_cast<const f32x4*>(addb + rowb + voff[f]);
          };
          if constexpr (HR || HA) {
#pragma unroll
            for (int g = 0; g < DEPTH - 1; ++g) load(g);
          }
#pragma unroll
          for (int g = 0; g < NGRP; ++g) {
            if constexpr (HR || HA) {
              if (g + DEPTH - 1 < NGRP) load(g + DEPTH - 1);
            }
            const int f = g / 16, r = g % 16;
            f32x4 o = at4(f, r);
            if constexpr (HR) o += rv[g % DEPTH];
            if constexpr (HA) o += av[g % DEPTH];
            o *= scale;
            if constexpr (!HR && !HA) {
              if (act_out) {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = ovk::lrelu(o[e], oslope);
              }
            }
            if (colok[f] && (OVW_EXP != 4 || o[0] == 12345.f))
              *reinterpret_cast<f32x4*>(outb + (mrow + (size_t)((r & 3) + 8 * (r >> 2))) * LD + voff[f]) = o;
          }
        };
        using T = std::true_type;
        using F = std::false_type;
        if (resb && addb) run(T{}, T{});
        else if (resb) run(T{}, F{});
        else if (addb) run(F{}, T{});
        else run(F{}, F{});
      } else {
        float* ost = ostage + wave * (8 * OST);
        const uint32_t c0 = (uint32_t)otile * NCOL + (uint32_t)(wsub * NCOLS);   // first column of this wave's sub-block (multiple of 4)
        // this lane's tiles: fragment f, tile 32 f + n -> columns rc + DIL (4 jt + i) of the sub-block, or none
        int cbase[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          const int tl = 32 * f + n, rc0 = tl / Gd::J;
          cbase[f] = rc0 < DIL ? rc0 + 4 * DIL * (tl - rc0 * Gd::J) : -1;
        }
        const uint32_t col = c0 + 4u * (uint32_t)lane;         // the 16 bytes this lane stores of each row
        const bool colok = 4 * lane < NCOLS && col < (uint32_t)L;
        const uint32_t ccol = colok ? col : 0u;
        auto run = [&](auto has_res, auto has_add) {
          constexpr bool HR = decltype(has_res)::value, HA = decltype(has_add)::value;
          // group = 2 output rows of the 8-row stage; 16 groups per item: g = 4 rg + 2 hh + (pair of the half)
          constexpr int DEPTH = (HR && HA) ? 2 : 3;
          f32x4 rv[HR ? DEPTH : 1][2], av[HA ? DEPTH : 1][2];
          auto row0 = [&](int g) { return (uint32_t)omtile * 32u + 8u * (g / 4) + 2u * (g % 4); };   // rows 8 rg + 2 q, + 1
          auto load = [&](int g) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              if constexpr (HR) rv[g % DEPTH][j] = *reinterpret_cast<const f32x4*>(resb + (size_t)(row0(g) + j) * LD + ccol);
              if constexpr (HA) av[g % DEPTH][j] = *reinterpret_cast<const f32x4*>(addb + (size_t)(row0(g) + j) * LD + ccol);
            }
          };
          if constexpr (HR || HA) {
#pragma unroll
            for (int g = 0; g < DEPTH - 1; ++g) load(g);
          }
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            // the wave's rows 8 rg .. 8 rg + 7: lane (half, n) holds rows 4 half + j; (same wave writes and then reads the
            // stage: LDS operations of a wave complete in order)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
              for (int f = 0; f < NF; ++f) {
                const f32x4 o = at4(f, 4 * rg + j);
                if (cbase[f] >= 0) {
                  float* od = ost + (4 * half + j) * OST + cbase[f];
                  od[0] = o[0]; od[DIL] = o[1]; od[2 * DIL] = o[2]; od[3 * DIL] = o[3];
                }
              }
#pragma unroll
            for (int q = 0; q < 4; ++q) {                      // stage rows 2 q, 2 q + 1 = output rows 8 rg + 2 q, + 1
              const int g = 4 * rg + q;
              if constexpr (HR || HA) {
                if (g + DEPTH - 1 < 16) load(g + DEPTH - 1);
              }
#pragma unroll
              for (int j = 0; j < 2; ++j) {
                f32x4 o = *reinterpret_cast<const f32x4*>(ost + (2 * q + j) * OST + 4 * lane);
                if constexpr (HR) o += rv[g % DEPTH][j];
                if constexpr (HA) o += av[g % DEPTH][j];
                o *= scale;
                if constexpr (!HR && !HA) {
                  if (act_out) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = ovk::lrelu(o[e], oslope);
                  }
                }
                if (colok) *reinterpret_cast<f32x4*>(outb + (size_t)(row0(g) + j) * LD + col) = o;
              }
            }
          }
        };
        using T = std::true_type;
        using F = std::false_type;
        if (resb && addb) run(T{}, T{});
        else if (resb) run(T{}, F{});
        else if (addb) run(F{}, T{});
        else run(F{}, F{});
      }
    }
    OVW_MARK(2)
    if (!more) break;
  }
  if constexpr (DBG) {
    if (lane == 0) {
      unsigned long long* d = p.dbg + ((size_t)blockIdx.x * 8 + wave) * 8;
      tph[7] = (unsigned long long)it;
#pragma unroll
      for (int q = 0; q < 8; ++q) d[q] = tph[q];
    }
  }
}

template <int K, int DIL, int CI, int NF, int MW, bool DBG>
int wino_launch(const ov_conv1d_wino_params* p, hipStream_t stream) {
  constexpr int NCOL = (DIL == 1 ? 128 * NF : GeoD<K, DIL == 1 ? 3 : DIL, NF>::NCOL) * (4 / MW), NHELP = 2 * NF;
  const int ntiles = (p->L + NCOL - 1) / NCOL;
  const long total = (long)ntiles * (p->Cout / (32 * MW)) * p->B;   // (an upper bound under a column limit: idle workgroups exit)
  auto kernel = conv1d_wino_kernel<K, DIL, CI, NF, MW, DBG>;
  static std::atomic<int> slot_cache[ovk::OV_MAX_DEVICES];
  const int slots = ovk::resident_workgroups(reinterpret_cast<const void*>(kernel), 64 * (4 + NHELP), slot_cache);
  long nwg = p->nwg > 0 ? p->nwg : slots;
  if (nwg > total) nwg = total;
  if (nwg >= 8) nwg = (nwg + 7) / 8 * 8;
  hipLaunchKernelGGL(kernel, dim3((unsigned)nwg), dim3(64 * (4 + NHELP)), 0, stream, *p);
  return hipGetLastError() == hipSuccess ? OV_OK : OV_E_LAUNCH;
}

// Input channels per LDS fill: by kernel size and by the rows a workgroup covers (128: four row fragments; 64: two row
// fragments x two column sub-blocks, i.e. twice the V tile per channel; 32: one row fragment x four sub-blocks, K = 11
// only -- at K = 7 a chunk of two channels would be an odd number of k-steps) -- the packed weight stream is ordered by it.
constexpr int wino_ci(int K, int mw = 4) {
  return mw == 1 ? (K == 11 ? 2 : 0) : K == 3 ? (mw == 4 ? 16 : 8) : (K == 7 || K == 11) ? (mw == 4 ? 8 : 4) : 0;
}
constexpr int wino_mw(int Cout) { return Cout % 128 == 0 ? 4 : (Cout % 64 == 0 ? 2 : (Cout % 32 == 0 ? 1 : 0)); }

// One translation unit per kernel size (conv1d_wino_k3 / k7 / k11.hip) so that the instances compile in parallel.
int wino_dispatch_k3(const ov_conv1d_wino_params* p, int nf, hipStream_t stream);
int wino_dispatch_k7(const ov_conv1d_wino_params* p, int nf, hipStream_t stream);
int wino_dispatch_k11(const ov_conv1d_wino_params* p, int nf, hipStream_t stream);

#define OVW_DISPATCH_MW(KK, MWV)                                                                                        \
  {                                                                                                                      \
    constexpr int CI = ovkw::wino_ci(KK, MWV);                                                                           \
    if constexpr (MWV == 4) {                                                                                            \
      if (p->dil == 1 && nf == 1)                                                                                        \
        return dbg ? wino_launch<KK, 1, CI, 1, 4, true>(p, st) : wino_launch<KK, 1, CI, 1, 4, false>(p, st);              \
    }                                                                                                                    \
    if (p->dil == 1) return dbg ? wino_launch<KK, 1, CI, 2, MWV, true>(p, st) : wino_launch<KK, 1, CI, 2, MWV, false>(p, st); \
    if (p->dil == 3) return dbg ? wino_launch<KK, 3, CI, 2, MWV, true>(p, st) : wino_launch<KK, 3, CI, 2, MWV, false>(p, st); \
    if (p->dil == 5) return dbg ? wino_launch<KK, 5, CI, 2, MWV, true>(p, st) : wino_launch<KK, 5, CI, 2, MWV, false>(p, st); \
    return OV_E_UNSUPPORTED;                                                                                             \
  }
#define OVW_DEFINE_DISPATCH(KK)                                                                                          \
  int ovkw::wino_dispatch_k##KK(const ov_conv1d_wino_params* p, int nf, hipStream_t st) {                                \
    const bool dbg = p->dbg != nullptr;                                                                                  \
    if (ovkw::wino_mw(p->Cout) == 4) OVW_DISPATCH_MW(KK, 4)                                                              \
    if (nf != 2) return OV_E_UNSUPPORTED;                                                                                \
    if (ovkw::wino_mw(p->Cout) == 2) OVW_DISPATCH_MW(KK, 2)                                                              \
    if constexpr (ovkw::wino_ci(KK, 1) != 0) OVW_DISPATCH_MW(KK, 1)                                                      \
    return OV_E_UNSUPPORTED;                                                                                             \
  }

}  // namespace ovkw
