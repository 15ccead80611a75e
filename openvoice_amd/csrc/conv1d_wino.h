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
// K_gemm = (group, ci).  Executed MACs per 4 outputs and (co, ci): K = 3: 6 (direct 12), K = 7: 18 (28), K = 11: 24 (44).
// fp32 throughout; the transforms round where the direct form does not, so the result carries ~4x the direct kernel's
// rounding error (measured against float64: profiles/r06_*), two orders of magnitude inside the path's 1e-3 bar.
//
// One workgroup = 4 MATRIX waves (wave w owns output rows 32w .. 32w + 31 of a 128-row M-block x 32 tiles = 128
// columns: 6 accumulator fragments = 96 VGPRs) + 2 HELPER waves; 2 workgroups per CU.  Input channels are walked in
// chunks of CI.  Per chunk the helpers (a) stage the raw rows (leaky ReLU applied, zero outside [0, L)) from HBM
// through registers into LDS, one chunk ahead, and (b) transform the previous raw chunk into V[k = g*CI + ci][tile][6]
// (5 ds_read_b128 -> 48 VALU -> 12 ds_write_b64 per (ci, tile) at K = 11), double buffered; ONE s_barrier per chunk.
// A matrix wave reads its B operands as three conflict-free ds_read_b64 per k-step (two points each) and streams its A
// operands (U, fragment order, 3 KiB per k-step pair) from L2 one pair ahead; the epilogue applies At, adds bias /
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

constexpr int REC = 256;        // floats per 1 KiB weight sub-record
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
  static constexpr int PADD = (K - 1) / 2 * DIL;                 // 'same' padding in columns
  static constexpr int PADA = (PADD + 3) / 4 * 4;                // raw rows start at column t0 - PADA (16-byte aligned)
  static constexpr int NV = 3 * (G - 1) + 6;                     // inputs a tile reads, DIL apart
  static constexpr int RW = (PADA - PADD + DIL - 1 + DIL * (4 * (J - 1) + NV - 1) + 1 + 3) / 4 * 4;   // floats per raw row
  static_assert(DIL == 3 || DIL == 5, "dilated instances: 3 and 5");
};

// Geometry of a K-tap conv: group g holds taps 3g .. 3g + 2 (taps >= K are zero).
template <int K>
struct Geo {
  static constexpr int G = (K + 2) / 3;
  static constexpr int PAD = (K - 1) / 2;
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
                    // 3 = no weight stream (the first fragments of an item are reused); 4 = no epilogue stores
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
// 64-row layer; the waves of a sub-block pair stream the same A fragments.
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
  static_assert(MW == 4 || (MW == 2 && NF == 2), "two-row-fragment workgroups run two fragments per wave");
  constexpr int ORG = DIL == 1 ? 8 : Gd::PADA;
  constexpr int RW4 = RW / 4;
  constexpr int NHELP = 2 * NF;        // helper waves
  static_assert(DIL == 1 || NF == 2, "dilated instances run two fragments per wave");
  static_assert(KR % 4 == 0 && CI % 2 == 0, "k-rows in whole k-step pairs");
  constexpr int RAWBUF = CI * RW;        // floats per raw buffer
  constexpr int VBUF = KR * NT * 6;      // floats per V buffer
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
    // Above the matrix waves in the issue arbiter: a helper's ~200 instructions per chunk are a few per cent of its
    // SIMD's cycles, but at equal priority each of them queued behind a 64-cycle MFMA of the two matrix waves it shares
    // the SIMD with (phase timers, profiles/r06_s3: the transform alone took 63 % of a chunk period).
    __builtin_amdgcn_s_setprio(3);
    const float slope = p.in_slope;
    const uint32_t ldx = (uint32_t)p.x_ld;
    const int hl = (wave - 4) * 64 + lane;
    constexpr int NITEM = CI * RW4;                          // 16-byte staging vectors per chunk
    constexpr int NLOAD = (NITEM + 64 * NHELP - 1) / (64 * NHELP);
    f32x4 stg[NLOAD];
    int nval[NLOAD];
    int lwid = wid0, lchunk = 0, lpos = 0;                   // position of the staging stream
    int lb, ltile, lmblk;
    decode(lwid, lb, ltile, lmblk);

    auto issue_loads = [&]() {
      const bool live = lpos < nstream;
      const int t0 = ltile * NCOL;
      const float* __restrict__ xb = p.x + (int64_t)lb * p.x_bstride;
#pragma unroll
      for (int i = 0; i < NLOAD; ++i) {
        const int idx = i * (64 * NHELP) + hl;
        const int row = idx / RW4, c4 = idx - row * RW4;
        const int ci = lchunk * CI + row;
        const int t = t0 - ORG + 4 * c4;                     // multiple of 4: a vector is wholly inside or outside
        const bool ok = live && idx < NITEM && t >= 0 && t < L;
        const uint32_t goff = ok ? (uint32_t)ci * ldx + (uint32_t)t : 0u;
        nval[i] = ok ? min(L - t, 4) : 0;
        stg[i] = *reinterpret_cast<const f32x4*>(xb + goff);
      }
      ++lpos;
      if (++lchunk == nchunks) {
        lchunk = 0;
        lwid += wstride;
        if (lwid < wend) decode(lwid, lb, ltile, lmblk);
      }
    };
    auto write_raw = [&](int buf) {
      float* dst = raw + buf * RAWBUF;
#pragma unroll
      for (int i = 0; i < NLOAD; ++i) {
        const int idx = i * (64 * NHELP) + hl;
        if (idx < NITEM) {
          f32x4 v = stg[i];
          const int n = nval[i];
          v[0] = n > 0 ? ovk::lrelu(v[0], slope) : 0.f;
          v[1] = n > 1 ? ovk::lrelu(v[1], slope) : 0.f;
          v[2] = n > 2 ? ovk::lrelu(v[2], slope) : 0.f;
          v[3] = n > 3 ? ovk::lrelu(v[3], slope) : 0.f;
          *reinterpret_cast<f32x4*>(dst + 4 * idx) = v;       // row * RW + 4 * c4 == 4 * idx
        }
      }
    };
    // raw[buf] -> V[buf]: item = (ci_local, tile); V[(g * CI + ci_local) * NT + tile][6]
    auto transform = [&](int buf) {
      const float* src = raw + buf * RAWBUF;
      float* dst = Vs + buf * VBUF;
      constexpr int ROUNDS = CI * NT / (64 * NHELP);
      static_assert(CI * NT % (64 * NHELP) == 0, "whole rounds of helper lanes");
#pragma unroll
      for (int r = 0; r < ROUNDS; ++r) {
        const int idx = r * (64 * NHELP) + hl;
        const int tile = idx & (NT - 1), cil = idx / NT;     // tile of the N-block = sub-block * NTS + tile of the sub-block
        constexpr int NWIN = DIL == 1 ? 4 * Ge::NB128 : Gd::NV;
        float win[NWIN];
        if constexpr (DIL == 1) {
#pragma unroll
          for (int q = 0; q < Ge::NB128; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + cil * RW + Ge::WSTART + 4 * tile + 4 * q);
            win[4 * q] = v[0]; win[4 * q + 1] = v[1]; win[4 * q + 2] = v[2]; win[4 * q + 3] = v[3];
          }
        } else {
          // tile -> (residue class, tile of the class); the left-over tiles duplicate tile 0 (never stored)
          const int sub = tile / NTS, tl = tile - sub * NTS;
          const int rc0 = tl / Gd::J, rc = rc0 < DIL ? rc0 : 0, jt = rc0 < DIL ? tl - rc0 * Gd::J : 0;
          const float* s0 = src + cil * RW + (Gd::PADA - Gd::PADD) + sub * NCOLS + rc + 4 * DIL * jt;
#pragma unroll
          for (int u = 0; u < Gd::NV; ++u) win[u] = s0[DIL * u];
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const int o = (DIL == 1 ? Ge::OFF0 - Ge::WSTART : 0) + 3 * g;
          const float d0 = win[o], d1 = win[o + 1], d2 = win[o + 2], d3 = win[o + 3], d4 = win[o + 4], d5 = win[o + 5];
          // Bt d, points 0, 1, -1, 2, -2, infinity
          const float t1 = __builtin_fmaf(-4.f, d2, d4), t2 = __builtin_fmaf(-4.f, d1, d3);
          const float t3 = d4 - d2, t4 = d3 - d1;
          f32x2 v01, v23, v45;
          v01[0] = __builtin_fmaf(4.f, d0, __builtin_fmaf(-5.f, d2, d4));
          v01[1] = t1 + t2;
          v23[0] = t1 - t2;
          v23[1] = __builtin_fmaf(2.f, t4, t3);
          v45[0] = __builtin_fmaf(-2.f, t4, t3);
          v45[1] = __builtin_fmaf(4.f, d1, __builtin_fmaf(-5.f, d3, d5));
          float* o6 = dst + ((g * CI + cil) * NT + tile) * 6;
          *reinterpret_cast<f32x2*>(o6) = v01;
          *reinterpret_cast<f32x2*>(o6 + 2) = v23;
          *reinterpret_cast<f32x2*>(o6 + 4) = v45;
        }
      }
    };

    unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tlast = DBG ? __builtin_readcyclecounter() : 0ull;
    issue_loads();
    write_raw(0);                 // chunk 0
    issue_loads();                // chunk 1 in registers
    __syncthreads();              // (A) raw[0] complete
    transform(0);
    write_raw(1);
    __syncthreads();              // (B) V[0], raw[1] complete
    OVW_MARK(4)
    for (int i = 0; i < nstream; ++i) {
      issue_loads();              // chunk i + 2 (nothing beyond the stream's end)
      OVW_MARK(0)
      if (OVW_EXP != 2 && i + 1 < nstream) transform((i + 1) & 1);
      OVW_MARK(1)
      write_raw(i & 1);
      OVW_MARK(2)
      __syncthreads();            // V[(i + 1) & 1] and raw[i & 1] handed over; V[i & 1] free again
      OVW_MARK(3)
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
  const int wrow = MW == 4 ? wave : (wave & 1), wsub = MW == 4 ? 0 : (wave >> 1);
  int mtile = mblk * MW + wrow;
  // sub-record index of the first k-step pair of (mtile, chunk 0); the stream of an item is contiguous
  uint32_t rec = (uint32_t)mtile * (uint32_t)nchunks * (uint32_t)(NPAIR * 3);
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
    for (int chunk = 0; chunk < nchunks; ++chunk, ++it) {
      const float* vb = Vs + (it & 1) * VBUF + voffB;
      // B operands: ONE register set, refilled in place -- the two points of pair qq are read for k-step s + 1 right after
      // the MFMAs of k-step s that consumed them have issued (4 NF MFMAs = 256 NF cycles later they are needed again)
      f32x2 bq[NF][3];
#pragma unroll
      for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int qq = 0; qq < 3; ++qq) bq[f][qq] = *reinterpret_cast<const f32x2*>(vb + f * (32 * 6) + 2 * qq);
#pragma unroll
      for (int s = 0; s < NSTEP; ++s) {
        const int s2 = s & 1;
        if (s2 == 0) {
          rec += 3;   // the sub-records after the last real pair are zeros written by the packer
#pragma unroll
          for (int j = 0; j < 3; ++j) a_nxt[j] = OVW_EXP == 3 ? a_cur[j] : wbase[(size_t)(rec + j) * 64 + lane];
        }
#pragma unroll
        for (int qq = 0; qq < 3; ++qq) {
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const int q = 2 * qq + t, e = s2 * 6 + q;
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
          if (s + 1 < NSTEP) {
#pragma unroll
            for (int f = 0; f < NF; ++f)
              bq[f][qq] = *reinterpret_cast<const f32x2*>(vb + (s + 1) * (2 * NT * 6) + f * (32 * 6) + 2 * qq);
          }
        }
        if (s2 == 1) {
#pragma unroll
          for (int j = 0; j < 3; ++j) a_cur[j] = a_nxt[j];
        }
      }
      OVW_MARK(0)
      __syncthreads();
      OVW_MARK(1)
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
          // operand ring: the rows of DEPTH - 1 groups are in flight while one is consumed (32 registers of operands)
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
// fragments x two column sub-blocks, i.e. twice the V tile per channel) -- the packed weight stream is ordered by it.
constexpr int wino_ci(int K, int mw = 4) { return K == 3 ? (mw == 4 ? 16 : 8) : (K == 7 || K == 11) ? (mw == 4 ? 8 : 4) : 0; }
constexpr int wino_mw(int Cout) { return Cout % 128 == 0 ? 4 : (Cout % 64 == 0 ? 2 : 0); }

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
    OVW_DISPATCH_MW(KK, 2)                                                                                               \
  }

}  // namespace ovkw
