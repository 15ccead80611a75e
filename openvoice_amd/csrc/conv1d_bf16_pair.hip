// Fused ResBlock1 pair, bf16 channels-last (BASELINE.json configs[4]; SURVEY.md section 8f item 3: "LDS-resident c1 -> c2
// fusion to beat the bf16 HBM bound"):
//   out = bf16( (c2(lrelu(bf16(c1(lrelu(x)) + b1))) + b2 + x [+ add]) * scale )
// reference: openvoice/modules.py:296-306 (loop body of ResBlock1.forward), models.py:280-286 (MRF sum / mean).
//
// In bf16 the C = 32 / 64 stages of the generator are HBM-bound: as two launches a pair moves 5 tensor passes (read x,
// write t, read t, read x again, write out) for ~25 us of matrix time.  Here t never leaves the CU -- it is rounded to
// bf16 exactly where the two-launch path rounds it (so results are bit-identical), activated, and written into an LDS
// tile from which c2 reads its A operands -- and a workgroup walks a RUN of consecutive time tiles left to right,
// keeping the last K-1 rows of the previous tile's t as left context (sliding window, as csrc/conv1d_pair.h does in
// fp32): every row of t is computed once.  The residual and the MRF running sum are added on the matrix pipe with an
// identity B fragment, staged through the same LDS chunk buffers as x (see conv1d_bf16.hip).
//
//
// One persistent workgroup per CU: 8 matrix waves (one 32 x 32 output fragment each: TT / 32 time fragments x C / 32
// channel tiles) + 4 loader waves.  Everything the k-step loops touch is in LDS:
//   wsm  both convs' packed weights, copied once per workgroup (it runs ~200 steps)             12 ... 48 KB
//   xs   double-buffered 32-channel chunks of lrelu(x) with the (K-1) DIL halo                    -> c1's B operands
//   xr   the same chunks RAW, output rows only, double-buffered per step                         -> residual
//   hb   the t tile: [K-1 rows of left context | TT new rows]                                    -> c2's B operands
// The loaders read every x vector ONCE and write both its activated and its raw copy; the next item's loads are in
// flight (in registers) while the current one is written and across the barriers.  The MFMAs are issued "transposed"
// (A = weights, B = activations) so that a lane ends up with 4 x 4 consecutive CHANNELS of one time row: t goes to LDS
// and the output to HBM with 8-byte stores (the time-major form needs four times as many 2-byte ones).
// Rounds of one step: C/32 x chunks -> c1;  t -> LDS;  c2 out of hb;  residual out of xr;  C/32 chunks of `add`
// (staged through xs) when the MRF running sum is given;  store.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <type_traits>

#include "openvoice_amd.h"

namespace ovk16p {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int CH = 32;       // channels per staged chunk = 2 MFMA k-blocks
constexpr int PITCH = 80;    // bytes per row of a staged chunk: 64 data + 16 pad (conflict-free ds_read_b128)
constexpr int NLD = 4;       // loader waves (8 measured no faster: the step is bound by the matrix waves' dependent chain)
constexpr int NMW = 8;       // matrix waves

__device__ __forceinline__ uint16_t f2bf(float f) {
  __bf16 h = (__bf16)f;      // round to nearest even
  uint16_t u;
  __builtin_memcpy(&u, &h, 2);
  return u;
}
__device__ __forceinline__ float bf2f(uint16_t v) { return __uint_as_float((uint32_t)v << 16); }
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {       // one v_cvt_pk_bf16_f32, round to nearest even
  const bf16x2_t h = __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t);
  uint32_t u;
  __builtin_memcpy(&u, &h, 4);
  return u;
}

// (utterance, step) sequence of one workgroup with the warm-up pseudo-step (c1 only) at a mid-utterance start
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

template <int K, int DIL, int C, int TT>
__global__ __launch_bounds__(64 * (NMW + NLD)) void respair_bf16cl_kernel(const ov_respair_bf16_params p) {
  static_assert(C % CH == 0 && TT % 32 == 0 && K % 2 == 1, "shape");
  constexpr int NTF = TT / 32, NCT = C / 32, NCH = C / CH;
  static_assert(NTF * NCT == NMW, "one 32 x 32 output fragment per matrix wave");
  constexpr int P1 = (K - 1) * DIL / 2, P2 = (K - 1) / 2;
  constexpr int R1 = TT + 2 * P1;                 // rows of a staged x chunk
  constexpr int BUF = R1 * PITCH;
  constexpr int RBUF = TT * PITCH;                // one raw chunk (output rows only)
  constexpr int PH = 2 * C + 16;                  // row pitch of the t tile (bytes)
  constexpr int RH = TT + 2 * P2;                 // its rows: [2 P2 rows of left context | TT new rows]
  constexpr int NITEM = R1 * 4;                   // 16-byte vectors of a chunk
  constexpr int PER_LANE = (NITEM + 64 * NLD - 1) / (64 * NLD);
  constexpr int DEPTH = 4;                        // items whose loads are in flight in the loaders' registers
  constexpr int WREC = NCT * NCH * K * 2;         // 1 KiB records of one conv's packed weight
  __shared__ __attribute__((aligned(16))) unsigned char xs[2 * BUF];
  __shared__ __attribute__((aligned(16))) unsigned char xr[2 * NCH * RBUF];
  __shared__ __attribute__((aligned(16))) unsigned char hb[RH * PH];
  __shared__ __attribute__((aligned(16))) unsigned char wsm[2 * WREC * 1024];
  __shared__ __attribute__((aligned(16))) float bsm[2 * C];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = p.L;
  const int nsteps = (L + P2 + TT - 1) / TT;
  const long S = (long)p.B * nsteps;
  const long g0 = S * blockIdx.x / gridDim.x, g1 = S * (blockIdx.x + 1) / gridDim.x;
  if (g0 >= g1) return;
  const bool has_add = p.add != nullptr;
  const int n_add = has_add ? NCH : 0;            // `add` rounds of a real step

  {   // weights -> LDS: [W1 records | W2 records]
    const u32x4* g1p = reinterpret_cast<const u32x4*>(p.w1);
    const u32x4* g2p = reinterpret_cast<const u32x4*>(p.w2);
    u32x4* wl = reinterpret_cast<u32x4*>(wsm);
    for (int e = tid; e < WREC * 64; e += 64 * (NMW + NLD)) { wl[e] = g1p[e]; wl[WREC * 64 + e] = g2p[e]; }
  }

  if (wave >= NMW) {
    // ================================ loader waves ===============================================
    // The NLD loader waves stage every item together (a quarter of its 16-byte vectors each) and keep the loads of
    // DEPTH further items in flight in registers -- ~60 KB per CU, what 6 TB/s x ~2.5 us of loaded HBM latency needs
    // (a step is SHORTER than that latency).  Static schedule: at the barrier where the matrix waves start consuming
    // item k, item k + 1 is written into the other chunk buffer (item k - 1's, consumed by then; its raw copy goes to
    // xr[pseudo-step & 1], never two steps ahead of the residual that still needs the other one) and the loads of
    // item k + 1 + DEPTH are issued into the register slot just freed.
    // The loop below is unrolled over the DEPTH register slots with a fixed number of loads per item (dummy loads past
    // the end of the sequence), so that hipcc's s_waitcnt insertion sees straight-line code and waits for the OLDEST
    // item only (vmcnt((DEPTH - 1) x loads per item)); with the slot chosen at run time it falls back to vmcnt(0),
    // which waits for the loads issued a moment ago and collapses the depth to one.
    const int llane = (wave - NMW) * 64 + lane;
    const float slope = p.slope;
    Seq st(g0, g1, nsteps);          // the item whose loads are issued next
    int st_round = 0, st_pstep = 0;
    bool more = true;
    long written = 0;
    u32x4 stg[DEPTH][PER_LANE];
    int okm[DEPTH];                                // bit i: vector i of the slot is inside the tensor
    int pd_kind[DEPTH], pd_c[DEPTH], pd_par[DEPTH];
    // per-lane constants of the PER_LANE vectors a lane stages of every item
    int vrow[PER_LANE];
    uint32_t voff[PER_LANE], loff[PER_LANE];
#pragma unroll
    for (int i = 0; i < PER_LANE; ++i) {
      const int idx = i * (64 * NLD) + llane;
      vrow[i] = idx >> 2;
      voff[i] = (uint32_t)(idx & 3) * 16u;         // byte offset inside a 64-byte chunk row
      loff[i] = (uint32_t)(idx >> 2) * PITCH + (uint32_t)(idx & 3) * 16u;
    }
    auto issue = [&](auto slot) {
      constexpr int SL = decltype(slot)::value;
      const int nrounds = NCH + (st.warm ? 0 : n_add);
      const int t0 = st.tile() * TT;
      const unsigned char* src = reinterpret_cast<const unsigned char*>(p.x + (int64_t)st.b * L * C);
      int tbase = t0 - P1, nrows = R1;
      pd_kind[SL] = 0; pd_c[SL] = st_round;
      if (st_round >= NCH) {
        src = reinterpret_cast<const unsigned char*>(p.add + (int64_t)st.b * L * C);
        pd_kind[SL] = 1; pd_c[SL] = st_round - NCH; nrows = TT; tbase = t0 - P2;
      }
      if (!more) { src = reinterpret_cast<const unsigned char*>(p.x); nrows = 0; pd_kind[SL] = 2; pd_c[SL] = 0; tbase = 0; }
      pd_par[SL] = st_pstep & 1;
      src += pd_c[SL] * (CH * 2);                  // uniform: chunk column of the utterance
      int m = 0;
#pragma unroll
      for (int i = 0; i < PER_LANE; ++i) {
        const int t = tbase + vrow[i];
        m |= (vrow[i] < nrows && t >= 0 && t < L) ? (1 << i) : 0;
        const uint32_t tc = (uint32_t)min(max(t, 0), L - 1);       // always a valid row; zeroed at write when outside
        stg[SL][i] = *reinterpret_cast<const u32x4*>(src + (tc * (uint32_t)(C * 2) + voff[i]));
      }
      okm[SL] = m;
      if (more && ++st_round == nrounds) { st_round = 0; st.advance(); ++st_pstep; more = st.valid(); }
    };
    // `part`: 0 / 1 = the even / odd vectors of the lane (the write of one item is split around a barrier, see the
    // main loop), 2 = all of them
    auto write = [&](auto slot, auto part) {
      constexpr int SL = decltype(slot)::value;
      constexpr int PART = decltype(part)::value;
      unsigned char* dst = xs + (written & 1) * BUF;
      unsigned char* raw = xr + (pd_par[SL] * NCH + pd_c[SL]) * RBUF - (P1 - P2) * PITCH;   // output-window rows
      const int nrows = pd_kind[SL] == 0 ? R1 : (pd_kind[SL] == 1 ? TT : 0);
      const bool is_x = pd_kind[SL] == 0;
#pragma unroll
      for (int i = 0; i < PER_LANE; ++i) {
        if (PART != 2 && (i & 1) != PART) continue;
        if (vrow[i] < nrows) {
          u32x4 v = stg[SL][i];
          if (!((okm[SL] >> i) & 1)) v = u32x4{0u, 0u, 0u, 0u};
          if (is_x) {
            if (vrow[i] >= P1 - P2 && vrow[i] < P1 - P2 + TT) *reinterpret_cast<u32x4*>(raw + loff[i]) = v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              // leaky ReLU on a packed bf16 pair, evaluated in fp32 and rounded to nearest even: for 0 < slope <= 1
              // (checked by the dispatcher) it is max(x, slope x).  6 instructions per pair -- shift, and,
              // v_pk_mul_f32, two v_max_f32 (as asm: fmaxf() canonicalises both operands first, three v_max_f32 per
              // value), v_cvt_pk_bf16_f32 -- the loaders' pass is what a step waits for (profiles/r02_s8)
              f32x2_t w = {__uint_as_float(v[e] << 16), __uint_as_float(v[e] & 0xffff0000u)};
              const f32x2_t sw = w * slope;
              asm("v_max_f32 %0, %1, %2" : "=v"(w[0]) : "v"(w[0]), "v"(sw[0]));
              asm("v_max_f32 %0, %1, %2" : "=v"(w[1]) : "v"(w[1]), "v"(sw[1]));
              v[e] = pack2(w[0], w[1]);
            }
          }
          *reinterpret_cast<u32x4*>(dst + loff[i]) = v;
        }
      }
      if (PART != 0) ++written;
    };
    // The barrier sequence of the matrix waves, replayed: next_item_tick() passes every barrier up to and including
    // the one at which the next item starts being consumed (false when the sequence is over).
    Seq tk(g0, g1, nsteps);
    int pos = 0;                                   // position inside the pseudo-step: x ticks, (t), add ticks, (end)
    auto next_item_tick = [&]() -> bool {
      while (tk.valid()) {
        const int n_a = tk.warm ? 0 : n_add;
        if (pos < NCH) { __syncthreads(); ++pos; return true; }              // x chunk tick
        if (pos == NCH) { __syncthreads(); ++pos; continue; }               // (t in LDS)
        if (pos < NCH + 1 + n_a) { __syncthreads(); ++pos; return true; }   // `add` chunk tick
        __syncthreads();                                                     // (step done with hb / xs / xr)
        pos = 0;
        tk.advance();
      }
      return false;
    };
    unsigned long long lt[3] = {0, 0, 0}, llast = 0;     // measurement only (p.dbg): ticks in barriers / write / issue
    const bool ldbg = p.dbg != nullptr;
    auto lmark = [&](int ph) {
      if (ldbg) {
        const unsigned long long now = __builtin_readcyclecounter();
        lt[ph] += now - llast;
        llast = now;
      }
    };
    // If the NEXT barrier of the sequence is not an item tick (the "t in LDS" barrier after the x chunks, the end-of-step
    // barrier), pass it.  The write of item k + 1 is split around it: per step the matrix waves run c1 | barrier | c2 +
    // residual + epilogue, and a loader that does its whole pass (write + issue, ~3 300 ticks at C = 32, K = 3:
    // profiles/r02_s8) in front of that barrier leaves them waiting there for ~3 000 ticks and then waits itself
    // through c2 -- the two were running one after the other.  Both halves go to the chunk buffer (and raw buffer) that
    // nothing reads before item k + 1's own tick, so the split needs no extra synchronisation.
    auto pass_nontick = [&]() {
      if (!tk.valid()) return;
      const int n_a = tk.warm ? 0 : n_add;
      if (pos == NCH) { __syncthreads(); ++pos; }
      else if (pos == NCH + 1 + n_a) { __syncthreads(); pos = 0; tk.advance(); }
    };
    static_assert(DEPTH == 4, "the loop below is unrolled over four register slots");
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>;
    using S3 = std::integral_constant<int, 3>;
    using PA = std::integral_constant<int, 0>;
    using PB = std::integral_constant<int, 1>;
    using PALL = std::integral_constant<int, 2>;
    issue(S0{});
    issue(S1{});
    issue(S2{});
    issue(S3{});
    write(S0{}, PALL{});                           // item 0 goes straight into the (empty) chunk buffer 0
    issue(S0{});
    __syncthreads();                               // (init)
    // at the tick of item k: item k + 1 (slot (k + 1) % DEPTH) is written -- half before, half after the next barrier
    // when that is not item k + 1's own tick -- and item k + 1 + DEPTH issued into the same slot
    if (ldbg) llast = __builtin_readcyclecounter();
    for (;;) {
      if (!next_item_tick()) break;
      lmark(0); write(S1{}, PA{}); lmark(1); pass_nontick(); lmark(0); write(S1{}, PB{}); lmark(1); issue(S1{}); lmark(2);
      if (!next_item_tick()) break;
      lmark(0); write(S2{}, PA{}); lmark(1); pass_nontick(); lmark(0); write(S2{}, PB{}); lmark(1); issue(S2{}); lmark(2);
      if (!next_item_tick()) break;
      lmark(0); write(S3{}, PA{}); lmark(1); pass_nontick(); lmark(0); write(S3{}, PB{}); lmark(1); issue(S3{}); lmark(2);
      if (!next_item_tick()) break;
      lmark(0); write(S0{}, PA{}); lmark(1); pass_nontick(); lmark(0); write(S0{}, PB{}); lmark(1); issue(S0{}); lmark(2);
    }
    if (ldbg && lane == 0) {
#pragma unroll
      for (int q = 0; q < 3; ++q) p.dbg[((size_t)blockIdx.x * (NMW + NLD) + wave) * 9 + q] = lt[q];
    }
    return;
  }

  // ================================== matrix waves ================================================
  const int half = lane >> 5, l31 = lane & 31;
  const int ti = wave % NTF, nt = wave / NTF;       // this wave's time fragment / channel tile
  const int trow0 = 32 * ti;
  for (int e = tid; e < RH * PH / 4; e += 64 * NMW) reinterpret_cast<uint32_t*>(hb)[e] = 0u;
  if (tid < C) { bsm[tid] = p.b1[tid]; bsm[C + tid] = p.b2[tid]; }
  __syncthreads();                                  // (init)

  const u32x4* wl1 = reinterpret_cast<const u32x4*>(wsm) + (size_t)nt * (NCH * K * 2) * 64 + lane;
  const u32x4* wl2 = wl1 + (size_t)WREC * 64;
  // Where both convs' fragments of this wave's channel tile fit in <= 112 VGPRs they live in registers for the whole
  // kernel: no weight traffic at all inside the step loop (C = 32: K = 3, 7; C = 64: K = 3), half the LDS reads.
  constexpr int WR = NCH * K * 2;                   // records of one conv of one channel tile
  constexpr bool WREG = 2 * WR * 4 <= 112;
  u32x4 wr1[WREG ? WR : 1], wr2[WREG ? WR : 1];
  if constexpr (WREG) {
#pragma unroll
    for (int r = 0; r < WR; ++r) { wr1[r] = wl1[(size_t)r * 64]; wr2[r] = wl2[(size_t)r * 64]; }
  }
  const int xl_off = (trow0 + l31) * PITCH + half * 16;
  const int hl_off = (trow0 + l31) * PH + half * 16;
  // identity A fragments: A[m][k] = 1 iff k-slot (kb, half, e) is channel m = l31 of the chunk
  u32x4 idw[2];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k0 = 16 * kb + 8 * half + 2 * e;
      idw[kb][e] = (k0 == l31 ? 0x3f80u : 0u) | (k0 + 1 == l31 ? 0x3f800000u : 0u);
    }

  // one 32-channel chunk: 2 K k-steps, D[channel][time] += W[channel][k] * X[k][time]; both operands from LDS, read
  // one k-step ahead (`wl` = the chunk's first weight record of this wave's channel tile)
  auto mma = [&](auto dd, auto rp, auto cc, const unsigned char* xl, const u32x4* wl, const u32x4* wr, f32x16& acc) {
    constexpr int DD = decltype(dd)::value, RP = decltype(rp)::value, CC = decltype(cc)::value;
    constexpr int STEPS = 2 * K;
    u32x4 acur = WREG ? wr[CC * STEPS] : wl[(size_t)CC * STEPS * 64];
    u32x4 bcur = *reinterpret_cast<const u32x4*>(xl), anxt = acur, bnxt = bcur;
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      if (s + 1 < STEPS) {
        const int tap = (s + 1) >> 1, kb = (s + 1) & 1;
        anxt = WREG ? wr[CC * STEPS + s + 1] : wl[(size_t)(CC * STEPS + s + 1) * 64];
        bnxt = *reinterpret_cast<const u32x4*>(xl + (tap * DD) * RP + kb * 32);
      }
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 av, bv;
      __builtin_memcpy(&av, &acur, 16);
      __builtin_memcpy(&bv, &bcur, 16);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      acur = anxt;
      bcur = bnxt;
    }
  };
  auto identity = [&](const unsigned char* xl, f32x16& acc) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const u32x4 b = *reinterpret_cast<const u32x4*>(xl + kb * 32);
      bf16x8 av, bv;
      __builtin_memcpy(&av, &idw[kb], 16);
      __builtin_memcpy(&bv, &b, 16);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
    }
  };
  // acc[r] = bias of channel 32 nt + 8 (r >> 2) + 4 half + (r & 3)
  auto bias_init = [&](const float* bvec, f32x16& acc) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(bvec + 32 * nt + 8 * g + 4 * half);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[4 * g + j] = v[j];
    }
  };
  using ID = std::integral_constant<int, DIL>;
  using I1 = std::integral_constant<int, 1>;
  using IPX = std::integral_constant<int, PITCH>;
  using IPH = std::integral_constant<int, PH>;

  // measurement only (p.dbg != NULL): shader-clock ticks per phase, summed over the steps of this wave
  unsigned long long tph[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
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
  int pstep = 0;
  for (Seq tk(g0, g1, nsteps); tk.valid(); ++pstep) {
    const int b = tk.b;
    const bool warm = tk.warm;
    const int t0 = tk.tile() * TT;
    const int par = pstep & 1;                        // raw-chunk buffer of this pseudo-step

    // ---- c1: t = b1 + W1 * lrelu(x) -----------------------------------------------------------------
    f32x16 acc;
    bias_init(bsm, acc);
    mark(0);
    {
      __syncthreads();
      mark(1);
      mma(ID{}, IPX{}, std::integral_constant<int, 0>{}, xs + (it & 1) * BUF + xl_off, wl1, wr1, acc);
      mark(2);
      ++it;
    }
    if constexpr (NCH == 2) {
      __syncthreads();
      mark(1);
      mma(ID{}, IPX{}, std::integral_constant<int, 1>{}, xs + (it & 1) * BUF + xl_off, wl1, wr1, acc);
      mark(2);
      ++it;
    }
    static_assert(NCH <= 2, "chunk loops are written out");
    // t: rounded to bf16 (what the two-launch path stores), activated in fp32, rounded again (what its loader
    // stages); rows outside [0, L) are zero (c2 pads t, not x).  4 consecutive channels per 8-byte LDS store.
    {
      const bool inside = t0 + trow0 + l31 < L;
      unsigned char* hrow = hb + (2 * P2 + trow0 + l31) * PH + (32 * nt + 4 * half) * 2;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] = bf2f(f2bf(acc[4 * g + j]));
          v[j] = v[j] > 0.f ? v[j] : v[j] * p.slope;
        }
        u32x2 o = {pack2(v[0], v[1]), pack2(v[2], v[3])};
        if (!inside) o = u32x2{0u, 0u};
        *reinterpret_cast<u32x2*>(hrow + 16 * g) = o;
      }
    }
    mark(3);
    __syncthreads();                                  // (t in LDS)
    mark(4);

    Seq nx = tk;
    nx.advance();
    const bool next_valid = nx.valid();
    const bool next_fresh = next_valid && !nx.warm && nx.i == 0;

    if (!warm) {
      // ---- c2 out of the t tile: output row o (global t0 - P2 + o) needs t rows [o, o + K - 1] of the tile ----
      bias_init(bsm + C, acc);
      mma(I1{}, IPH{}, std::integral_constant<int, 0>{}, hb + hl_off, wl2, wr2, acc);
      if constexpr (NCH == 2) mma(I1{}, IPH{}, std::integral_constant<int, 1>{}, hb + hl_off + 64, wl2, wr2, acc);
      mark(5);
      // ---- residual: raw x of this wave's channel tile, out of xr ----------------------------------
      identity(xr + (par * NCH + nt) * RBUF + xl_off, acc);
      // ---- MRF running sum: staged rounds -----------------------------------------------------------
      for (int c = 0; c < n_add; ++c, ++it) {
        __syncthreads();
        if (c == nt) identity(xs + (it & 1) * BUF + xl_off, acc);
      }
      mark(6);
      // ---- epilogue: scale, round, store 4 consecutive channels per 8-byte store ----------------------
      const int t = t0 - P2 + trow0 + l31;
      if (t >= 0 && t < L) {
        uint16_t* orow = p.out + ((int64_t)b * L + t) * C + 32 * nt + 4 * half;
        const float scale = p.scale;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          u32x2 o = {pack2(acc[4 * g] * scale, acc[4 * g + 1] * scale), pack2(acc[4 * g + 2] * scale, acc[4 * g + 3] * scale)};
          *reinterpret_cast<u32x2*>(orow + 8 * g) = o;
        }
      }
    }
    mark(7);
    __syncthreads();                                  // (step done with hb / xs / xr)
    if (next_valid) {                                 // last 2 P2 rows of t -> left context of the next step
      for (int e = tid; e < 2 * P2 * (C / 2); e += 64 * NMW) {
        const int row = e / (C / 2), q = e - row * (C / 2);
        uint32_t* d = reinterpret_cast<uint32_t*>(hb + row * PH) + q;
        *d = next_fresh ? 0u : reinterpret_cast<const uint32_t*>(hb + (TT + row) * PH)[q];
      }
    }
    tk = nx;
    mark(8);
  }
  if (dbg && lane == 0) {
#pragma unroll
    for (int q = 0; q < 9; ++q) p.dbg[((size_t)blockIdx.x * (NMW + NLD) + wave) * 9 + q] = tph[q];
  }
}

inline int resident(const void* kernel, int threads, std::atomic<int>* cache) {
  int dev = 0;
  const bool known = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 16;
  if (known) {
    const int v = cache[dev].load(std::memory_order_relaxed);
    if (v > 0) return v;
  }
  int per_cu = 0, slots = 512;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) == hipSuccess &&
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, 0) == hipSuccess && per_cu > 0)
    slots = per_cu * prop.multiProcessorCount;
  if (known) cache[dev].store(slots, std::memory_order_relaxed);
  return slots;
}


template <int K, int DIL, int C, int TT>
int launch(const ov_respair_bf16_params* p, hipStream_t stream) {
  auto kernel = respair_bf16cl_kernel<K, DIL, C, TT>;
  static std::atomic<int> cache[16];
  const int slots = resident(reinterpret_cast<const void*>(kernel), 64 * (NMW + NLD), cache);
  constexpr int P2 = (K - 1) / 2;
  const long S = (long)p->B * ((p->L + P2 + TT - 1) / TT);
  long nwg = p->nwg > 0 ? p->nwg : slots;
  if (nwg > S) nwg = S;
  hipLaunchKernelGGL(kernel, dim3((unsigned)nwg), dim3(64 * (NMW + NLD)), 0, stream, *p);
  return hipGetLastError() == hipSuccess ? OV_OK : OV_E_LAUNCH;
}

// LDS per workgroup = weights (2 C^2 K 2 B) + chunk buffers + raw chunks + t tile: C = 32 -> 116 / 132 / 155 KB at
// K = 3 / 7 / 11; C = 64 -> 130 KB at K = 3, over the 160 KB of a CU beyond that (those shapes stay on two launches).
template <int K, int DIL>
int launch_by_width(const ov_respair_bf16_params* p, hipStream_t stream) {
  if (p->C == 32) return launch<K, DIL, 32, 256>(p, stream);
  if constexpr (K == 3) {
    if (p->C == 64) return launch<K, DIL, 64, 128>(p, stream);
  }
  return OV_E_UNSUPPORTED;
}

}  // namespace ovk16p

using namespace ovk16p;

extern "C" {

int ov_resblock_pair_bf16_supported(int C, int K, int dil) {
  const bool kd = (K == 3 || K == 7 || K == 11) && (dil == 1 || dil == 3 || dil == 5);
  return kd && (C == 32 || (C == 64 && K == 3)) ? 1 : 0;
}

int ov_resblock_pair_bf16cl(const ov_respair_bf16_params* p, ov_stream_t stream) {
  if (!p || !p->x || !p->w1 || !p->b1 || !p->w2 || !p->b2 || !p->out) return OV_E_BADARG;
  if (p->B <= 0 || p->L <= 0 || p->C <= 0 || p->nwg < 0) return OV_E_BADARG;
  if (p->out == p->x) return OV_E_BADARG;
  if (!(p->slope > 0.f && p->slope <= 1.f)) return OV_E_UNSUPPORTED;   // the loaders evaluate lrelu as max(x, slope x)
  if (!ov_resblock_pair_bf16_supported(p->C, p->K, p->dil)) return OV_E_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(p->x) & 15) || (reinterpret_cast<uintptr_t>(p->w1) & 15) ||
      (reinterpret_cast<uintptr_t>(p->w2) & 15) || (p->add && (reinterpret_cast<uintptr_t>(p->add) & 15)))
    return OV_E_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
#define OV16P_CASE(KK, DD) if (p->K == KK && p->dil == DD) return launch_by_width<KK, DD>(p, st);
  OV16P_CASE(3, 1) OV16P_CASE(3, 3) OV16P_CASE(3, 5)
  OV16P_CASE(7, 1) OV16P_CASE(7, 3) OV16P_CASE(7, 5)
  OV16P_CASE(11, 1) OV16P_CASE(11, 3) OV16P_CASE(11, 5)
#undef OV16P_CASE
  return OV_E_UNSUPPORTED;
}

}  // extern "C"
