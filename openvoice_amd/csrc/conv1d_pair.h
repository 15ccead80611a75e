// Fused ResBlock1 pair on the gfx950 fp32 matrix pipe:  out = (c2(lrelu(c1(lrelu(x)))) + x [+ add]) * scale
//   c1 = Conv1d(C, C, K, dilation DIL), c2 = Conv1d(C, C, K, dilation 1), both 'same'-padded
//   reference: openvoice/modules.py:296-306 (one iteration of ResBlock1.forward's loop, x_mask = None) and the MRF sum /
//   mean of openvoice/models.py:280-286 through `add` / `scale`.
//
// Why: for C = 32 / 64 the two convs run as two launches are HBM-bound -- 5 tensor passes per pair (read x, write t,
// read t, read x again as the residual, write out) at 24-48 FLOP/B.  Here the intermediate t never leaves the CU:
// 2 passes (+1 for the residual re-read, which hits the Infinity Cache).
//
// How -- a sliding window along time instead of halo recomputation:
//   * A workgroup (4 matrix waves + NLD loader waves, as conv1d_mfma.h) owns a RUN of consecutive NT-column tiles of
//     one utterance and walks it left to right.  Step i computes h = lrelu(c1(lrelu(x)) + b1) for the columns
//     [i NT, (i+1) NT) into accumulators (input channels streamed through the double-buffered LDS chunks by the loader
//     waves, exactly like the single-conv kernel), writes it to an LDS tile `hb` BEHIND the last K-1 columns of the
//     previous step's h, and then runs c2 straight out of `hb`, producing the output columns
//     [i NT - P2, (i+1) NT - P2), P2 = (K-1)/2: every column of h is computed once, every MFMA of both convs is useful
//     work (a halo-recompute tiling at 32-column fragment granularity would waste 12.5 % of each conv at NT = 256).
//   * The flattened (utterance, step) list is cut into gridDim.x equal ranges (one per resident workgroup slot); a range
//     that starts in the middle of an utterance first runs c1 alone on the tile before it (one extra half-step per ~54
//     steps at the benchmark shape); h outside [0, L) is zero (c2 pads h, not x).
//   * Accumulators: C x NT / 4 waves = 32 registers, used for h and then reused for the output; the residual tile is
//     fetched into 32 more registers at the top of the step and waits there through c1, so c2's accumulators start at
//     x + b2 with no exposed load.  The running MRF sum `add` (2 of 9 pair launches per stage) is added in the epilogue.
//   * Arithmetic order is the single-conv kernel's (same chunk / unit / tap order, accumulators start at residual +
//     bias), so the result is bit-identical to the two-launch path -- the parity test checks exactly that.
//
// Synchronisation: per step the matrix waves pass C/CHUNK chunk barriers (loader hand-offs), one barrier after h is in
// LDS and one after c2 has finished reading it (then the last K-1 columns move to the front of `hb`).  The loader waves
// execute the same barriers and stage up to two chunks ahead: while c2 runs they fill both chunk buffers of the next
// step, so c1 never waits for HBM in steady state.
#pragma once
#include <type_traits>

#include "conv1d_mfma.h"

namespace ovk {

// (utterance, step) sequence of one workgroup, with the warm-up pseudo-step (c1 only, tile i-1) at a mid-utterance start.
// One 64-bit division at construction, increments afterwards.
struct PairSeq {
  long left;      // real steps left, including the current one
  int b, i, nsteps;
  bool warm;
  const int* pref;   // length-aware list: LDS prefix table of the utterances' step counts (else NULL: nsteps each)
  __device__ __forceinline__ int count(int u) const {
    return __builtin_amdgcn_readfirstlane(pref[u + 1]) - __builtin_amdgcn_readfirstlane(pref[u]);
  }
  __device__ __forceinline__ PairSeq(long g0, long g1, int nsteps_, const int* pref_, int B)
      : left(g1 - g0), b(0), i(0), nsteps(nsteps_), warm(false), pref(pref_) {
    if (left <= 0) return;
    if (!pref) {
      b = (int)(g0 / nsteps_);
      i = (int)(g0 - (long)b * nsteps_);
    } else {
      int lo = 0, hi = B;                     // pref[lo] <= g0 < pref[hi]
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (__builtin_amdgcn_readfirstlane(pref[mid]) <= (int)g0) lo = mid; else hi = mid;
      }
      b = lo;
      i = (int)g0 - __builtin_amdgcn_readfirstlane(pref[lo]);
      nsteps = count(lo);
    }
    warm = i != 0;
  }
  __device__ __forceinline__ bool valid() const { return left > 0; }
  __device__ __forceinline__ int batch() const { return b; }
  __device__ __forceinline__ int step() const { return i; }
  __device__ __forceinline__ int tile() const { return i - (warm ? 1 : 0); }   // the h tile this pseudo-step computes
  __device__ __forceinline__ void advance() {
    if (warm) { warm = false; return; }
    --left;
    if (++i == nsteps) {
      i = 0;
      ++b;
      if (pref && left > 0) {                 // utterances whose limit is 0 have no steps at all
        while ((nsteps = count(b)) == 0) ++b;
      }
    }
  }
};

template <int K, int DIL, int C, int NT, int CHUNK, int NLD>
__global__ __launch_bounds__(64 * (4 + NLD), 4) void respair_mfma_kernel(const ov_respair_params p) {
  static_assert(C == 32 || C == 64, "channel counts of the HBM-bound generator stages");
  static_assert(NT % 128 == 0 && CHUNK % UNIT == 0 && C % CHUNK == 0, "tile shape");
  static_assert(K % 2 == 1, "'same' padding");
  constexpr int PF = 1;   // B operands are read from LDS PF k-steps ahead of their MFMAs (2 measured: no faster)
  constexpr int WM = C / 32, WN = NT / 128;      // per-wave fragments: all C rows x NT/4 columns
  constexpr int P1 = (K - 1) * DIL / 2, P2 = (K - 1) / 2;
  constexpr int PADA = (P1 + 3) / 4 * 4;
  constexpr int XS = NT + 2 * PADA, XS4 = XS / 4;   // x chunk row stride
  constexpr int HS = (NT + 2 * P2 + 3) / 4 * 4;     // h tile row stride: [2 P2 tail columns | NT new columns]
  constexpr int NCH = C / CHUNK, UPC = CHUNK / UNIT;
  constexpr int BUF = CHUNK * XS;
  constexpr int NITEM = CHUNK * XS4;
  constexpr int PER_LANE = (NITEM + 64 * NLD - 1) / (64 * NLD);
  constexpr int LB = PER_LANE < LB_MAX ? PER_LANE : LB_MAX;
  constexpr int NBATCH = (PER_LANE + LB - 1) / LB;
  constexpr int UNITS = C / UNIT;                   // c2 walks all input channels out of hb
  constexpr int UG = UNITS < 4 ? UNITS : 4;         // ... in groups of <= 4 units (bounds the unrolled code)

  __shared__ __attribute__((aligned(16))) float xs[2 * BUF];
  __shared__ __attribute__((aligned(16))) float hb[C * HS];
  __shared__ float bsm[2 * C];   // b1 | b2: read back per step with immediate offsets (as global loads their 64 row
                                 // addresses are step-invariant, get hoisted into VGPR pairs and spill)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = p.L;
  const uint32_t ld = (uint32_t)p.ld;
  const int nsteps = (L + P2 + NT - 1) / NT;
  long S = (long)p.B * nsteps;
  // Length-aware work list (ov_respair_params.col_limit): utterance b walks only the steps that produce columns before
  // its limit -- ceil((limit + P2) / NT) of them, none when the limit is 0; the prefix sums of the step counts (<= 256
  // utterances) are built once per workgroup and the flattened list stays dense.
  __shared__ int lim_pref[LIMIT_MAX_BATCH + 1];
  const int* pref = nullptr;
  if (p.col_limit != nullptr) {
    if (wave == 0) {
      int carry = 0;
      for (int base = 0; base < p.B; base += 64) {
        const int bb = base + lane;
        int n = 0;
        if (bb < p.B) {
          long long c = (long long)p.col_limit[bb] * p.col_limit_scale;
          c = c < 0 ? 0 : (c > L ? L : c);
          n = c > 0 ? ((int)c + P2 + NT - 1) / NT : 0;
        }
        int sc = n;
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
    pref = lim_pref;
    S = __builtin_amdgcn_readfirstlane(lim_pref[p.B]);
  }
  const long g0 = S * blockIdx.x / gridDim.x, g1 = S * (blockIdx.x + 1) / gridDim.x;
  if (g0 >= g1) return;
  const PairSeq first(g0, g1, nsteps, pref, p.B);
  const long npseudo = (g1 - g0) + (first.warm ? 1 : 0);

  bool is_loader = false;
#pragma unroll
  for (int i = 0; i < NLD; ++i) is_loader |= (wave == 4 + i);

  if (is_loader) {
    // ================================ loader waves ===============================================
    const float slope = p.slope;
    const int llane = (wave - 4) * 64 + lane;
    const long total = npseudo * NCH;
    PairSeq st = first;
    int st_chunk = 0;
    long staged = 0, begun = 0;
    auto stage_one = [&]() {
      const float* __restrict__ xb = p.x + (int64_t)st.batch() * p.x_bstride;
      const int t0 = st.tile() * NT;
      float* dst = xs + (staged & 1) * BUF;
#pragma unroll 1
      for (int bt = 0; bt < NBATCH; ++bt) {
        f32x4 stg[LB];
        int nval[LB];
#pragma unroll
        for (int i = 0; i < LB; ++i) {
          const int idx = (bt * LB + i) * (64 * NLD) + llane;
          const int row = idx / XS4, c4 = idx - row * XS4;
          const int ci = st_chunk * CHUNK + row;
          const int t = t0 - PADA + 4 * c4;     // multiple of 4: a vector is wholly < 0 or >= 0
          const bool ok = idx < NITEM && t >= 0 && t < L;
          const uint32_t goff = ok ? (uint32_t)ci * ld + (uint32_t)t : 0u;
          nval[i] = ok ? min(L - t, 4) : 0;
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
            *reinterpret_cast<f32x4*>(dst + 4 * idx) = v;
          }
        }
      }
      ++staged;
      if (++st_chunk == NCH) { st_chunk = 0; st.advance(); }
    };
    stage_one();
    __syncthreads();                     // (init) the matrix waves have zeroed hb and filled bsm
    for (long ps = 0; ps < npseudo; ++ps) {
      for (int c = 0; c < NCH; ++c) {
        __syncthreads();                 // chunk `begun` handed over; the matrix waves finished chunk begun - 1
        ++begun;
        while (staged < total && staged <= begun) stage_one();
      }
      __syncthreads();                   // (h in LDS) -- every chunk begun so far has been consumed
      while (staged < total && staged <= begun + 1) stage_one();
      __syncthreads();                   // (c2 done with hb)
    }
    return;
  }

  // ================================== matrix waves ================================================
  const int wn = wave;                                // 4 waves side by side along time
  const uint32_t half = (uint32_t)lane >> 5;
  const int recs1 = packed_units(C) * K + 1;          // records per 32-row tile (incl. the zero record), both convs
  const f32x4* __restrict__ w1_ = reinterpret_cast<const f32x4*>(p.w1);
  const f32x4* __restrict__ w2_ = reinterpret_cast<const f32x4*>(p.w2);
  uint32_t widx[WM];
#pragma unroll
  for (int i = 0; i < WM; ++i) widx[i] = (uint32_t)i * (uint32_t)recs1 * 64u + (uint32_t)lane;
  const int col_w = wn * (32 * WN) + (lane & 31);     // this lane's first column inside the NT tile
  const int xl_off = (lane >> 5) * XS + col_w + (PADA - P1);
  const int hl_off = (lane >> 5) * HS + col_w;

  // hb starts zeroed: a run that begins at an utterance start sees h = 0 left of column 0
  for (int e = tid; e < C * HS; e += 256) hb[e] = 0.f;
  if (tid < C) { bsm[tid] = p.b1[tid]; bsm[C + tid] = p.b2[tid]; }
  __syncthreads();   // (init) -- paired with the loaders' first barrier

  f32x4 a_cur[WM], a_nxt[WM];
#pragma unroll
  for (int i = 0; i < WM; ++i) a_cur[i] = w1_[widx[i]];

  // one chunk / unit group of k-steps: B operands from LDS one k-step ahead (see conv1d_mfma.h)
  auto mma = [&](auto kk, auto dd, auto nu, auto rs, const float* xl, const f32x4* __restrict__ wb, int& rec,
                 f32x16 (&acc)[WM][WN]) {
    constexpr int KK = decltype(kk)::value, DD = decltype(dd)::value, NU = decltype(nu)::value,
                  RS = decltype(rs)::value;
    constexpr int STEPS = NU * 4 * KK;
    // ring of PF + 1 operand sets: set (sa % (PF+1)) feeds k-step sa, the reads for k-step sa + PF are issued
    // before its MFMAs (all indices are compile-time after unrolling)
    float bq[PF + 1][WN];
    auto boff = [](int s) {
      const int uu = s / (4 * KK), sn = s - uu * (4 * KK);
      const int pp = sn / KK, tap = sn - pp * KK;
      return (uu * UNIT + 2 * pp) * RS + tap * DD;
    };
#pragma unroll
    for (int q = 0; q < PF; ++q)
      if (q < STEPS) {
#pragma unroll
        for (int j = 0; j < WN; ++j) bq[q][j] = xl[boff(q) + 32 * j];
      }
#pragma unroll
    for (int sa = 0; sa < STEPS; ++sa) {
      const int u = sa & 3;
      if (u == 0) {
        ++rec;
#pragma unroll
        for (int i = 0; i < WM; ++i) a_nxt[i] = (wb + (size_t)rec * 64)[widx[i]];
      }
      if (sa + PF < STEPS) {
#pragma unroll
        for (int j = 0; j < WN; ++j) bq[(sa + PF) % (PF + 1)][j] = xl[boff(sa + PF) + 32 * j];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < WM * WN; ++m) {
        const int i = m / WN, j = m % WN;
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[i][u], bq[sa % (PF + 1)][j], acc[i][j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (u == 3) {
#pragma unroll
        for (int i = 0; i < WM; ++i) a_cur[i] = a_nxt[i];
      }
    }
  };
  using IK = std::integral_constant<int, K>;
  using ID = std::integral_constant<int, DIL>;
  using I1 = std::integral_constant<int, 1>;

  // measurement only (p.dbg != NULL): shader-clock ticks per phase, summed over the steps of this wave
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
  if (dbg) tlast = __builtin_readcyclecounter();
  long it = 0;
  for (PairSeq tk = first; tk.valid();) {
    const int b = tk.batch();
    const bool warm = tk.warm;
    const int t0 = tk.tile() * NT;
    const float* __restrict__ xb = p.x + (int64_t)b * p.x_bstride;
    float* __restrict__ ob = p.out + (int64_t)b * p.out_bstride;
    // The weight bases are made opaque once per step: the k-step loops are fully unrolled, so every record address
    // (base + constant) is loop-invariant and hipcc would otherwise hoist all ~50-90 of them out of the step loop
    // into VGPR pairs (248 VGPRs at C = 64, K = 3 instead of ~110).
    const f32x4* w1 = w1_; const f32x4* w2 = w2_;
    asm volatile("" : "+s"(w1), "+s"(w2));

    // ---- residual tile -> registers (columns t0 - P2 ...; out-of-range columns are clamped and never stored) ----
    f32x16 rv[WM][WN];
    if (!warm) {
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          const int gc = min(max(t0 - P2 + col_w + 32 * j, 0), L - 1);
          const uint32_t voff = ((uint32_t)i * 32u + 4u * half) * ld + (uint32_t)gc;
#pragma unroll
          for (int r = 0; r < 16; ++r) rv[i][j][r] = (xb + (size_t)((r & 3) + 8 * (r >> 2)) * ld)[voff];
        }
    }

    mark(0);   // residual loads issued
    // ---- c1: h = b1 + W1 * lrelu(x) ------------------------------------------------------------------
    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      f32x16 bv;
#pragma unroll
      for (int r = 0; r < 16; ++r) bv[r] = bsm[i * 32 + 4 * half + (r & 3) + 8 * (r >> 2)];
#pragma unroll
      for (int j = 0; j < WN; ++j) acc[i][j] = bv;
    }
    int rec = 0;
    for (int chunk = 0; chunk < NCH; ++chunk, ++it) {
      __syncthreads();
      mark(1);   // waiting for the loaders (chunk barrier)
      mma(IK{}, ID{}, std::integral_constant<int, UPC>{}, std::integral_constant<int, XS>{},
          xs + (it & 1) * BUF + xl_off, w1, rec, acc);
      mark(2);   // c1 k-steps
    }
    // first W2 record in flight while h goes to LDS
#pragma unroll
    for (int i = 0; i < WM; ++i) a_cur[i] = w2[widx[i]];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const bool inside = t0 + col_w + 32 * j < L;     // t0 >= 0: only the right end can stick out
        float* hrow = hb + (i * 32 + 4 * half) * HS + 2 * P2 + col_w + 32 * j;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          hrow[((r & 3) + 8 * (r >> 2)) * HS] = inside ? lrelu(acc[i][j][r], p.slope) : 0.f;
      }
    mark(3);                                           // h -> LDS
    __syncthreads();                                   // (h in LDS)
    mark(4);                                           // waiting at the h barrier

    // what comes next decides what happens to the tail columns
    PairSeq nx = tk;
    nx.advance();
    const bool next_valid = nx.valid();
    const bool next_fresh = next_valid && !nx.warm && nx.step() == 0;   // next step starts an utterance: h = 0 on its left

    if (!warm) {
      // ---- c2: out = (x + b2) + W2 * h ---------------------------------------------------------------
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        f32x16 bv;
#pragma unroll
        for (int r = 0; r < 16; ++r) bv[r] = bsm[C + i * 32 + 4 * half + (r & 3) + 8 * (r >> 2)];
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = rv[i][j] + bv;
      }
      rec = 0;
#pragma unroll 1
      for (int ug = 0; ug < UNITS / UG; ++ug)
        mma(IK{}, I1{}, std::integral_constant<int, UG>{}, std::integral_constant<int, HS>{},
            hb + hl_off + ug * UG * UNIT * HS, w2, rec, acc);
      mark(5);   // c2 k-steps (incl. accumulator init)
      // first W1 record of the next step in flight during the epilogue
      if (next_valid) {
#pragma unroll
        for (int i = 0; i < WM; ++i) a_cur[i] = w1[widx[i]];
      }
      const float* addb = p.add ? p.add + (int64_t)b * p.add_bstride : nullptr;
      const float scale = p.scale;
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          const int gc = t0 - P2 + col_w + 32 * j;
          if (gc < 0 || gc >= L) continue;
          const uint32_t voff = ((uint32_t)i * 32u + 4u * half) * ld + (uint32_t)gc;
          f32x16 v = acc[i][j];
          if (addb) {
            f32x16 av;
#pragma unroll
            for (int r = 0; r < 16; ++r) av[r] = (addb + (size_t)((r & 3) + 8 * (r >> 2)) * ld)[voff];
            v += av;
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) (ob + (size_t)((r & 3) + 8 * (r >> 2)) * ld)[voff] = v[r] * scale;
        }
    } else {
#pragma unroll
      for (int i = 0; i < WM; ++i) a_cur[i] = w1[widx[i]];
    }
    mark(6);                                           // output epilogue
    __syncthreads();                                   // (c2 done with hb)
    // the last 2 P2 columns of h become the left context of the next step (or zeros at an utterance start)
    if (next_valid) {
      for (int e = tid; e < C * 2 * P2; e += 256) {
        const int row = e / (2 * P2), k = e - row * (2 * P2);
        hb[row * HS + k] = next_fresh ? 0.f : hb[row * HS + NT + k];
      }
    }
    tk = nx;
    mark(7);                                           // barrier + tail copy
  }
  if (dbg && lane == 0) {
#pragma unroll
    for (int q = 0; q < 8; ++q) p.dbg[((size_t)blockIdx.x * 4 + wave) * 8 + q] = tph[q];
  }
}

typedef int (*pair_launch_fn)(const ov_respair_params*, hipStream_t);

template <int K, int DIL, int C, int NT, int CHUNK, int NLD>
int respair_launch(const ov_respair_params* p, hipStream_t stream) {
  auto kernel = respair_mfma_kernel<K, DIL, C, NT, CHUNK, NLD>;
  static std::atomic<int> slot_cache[OV_MAX_DEVICES];
  const int slots = resident_workgroups(reinterpret_cast<const void*>(kernel), 64 * (4 + NLD), slot_cache);
  constexpr int P2 = (K - 1) / 2;
  const long S = (long)p->B * ((p->L + P2 + NT - 1) / NT);
  long nwg = p->nwg > 0 ? p->nwg : slots;
  if (nwg > S) nwg = S;
  hipLaunchKernelGGL(kernel, dim3((unsigned)nwg), dim3(64 * (4 + NLD)), 0, stream, *p);
  return hipGetLastError() == hipSuccess ? OV_OK : OV_E_LAUNCH;
}

struct PairVariant {
  int K, dil, C;
  pair_launch_fn fn;
};
extern const PairVariant kPairVariants[];
extern const int kPairVariantsCount;

}  // namespace ovk
