// Implicit-GEMM Conv1d on the gfx950 fp32 matrix pipe (v_mfma_f32_32x32x2_f32).
//
// GEMM view of y[b][co][t] = sum_{ci,j} W[co][ci][j] * act(x[b][ci][t + j*DIL - PAD]):
//   M = output rows (co), N = time, K_gemm = (ci, tap).
// One 256-thread workgroup (4 wave64, one per SIMD) owns an M_BLK x N_BLK output tile of one
// utterance.  Input channels are walked in chunks of CI_CHUNK: the chunk's rows, with the
// (K-1)*DIL receptive-field halo, are staged ONCE into LDS (leaky-ReLU applied on the way in, so
// the activation costs one VALU op per staged element instead of one per tap), and every tap of
// every MFMA B operand is then a conflict-free ds_read_b32 at a compile-time offset.  The A
// operand (weights) never touches LDS: weights are pre-packed at load time into MFMA fragment
// order, so a wave fetches the fragments of 4 consecutive k-steps with one coalesced 1 KiB
// global_load_dwordx4 (L2-resident), prefetched one group ahead in registers.  The staging loads
// of chunk c+1 are issued before the MFMA loop of chunk c and written to LDS after it
// (issue-early / write-late), so HBM latency hides under the matrix work.
//
// fp32 MFMA is bit-exact fmaf-chain arithmetic (no TF32-style truncation on gfx950), so parity
// with the fp32 reference is limited only by summation order.
#pragma once
#include <hip/hip_runtime.h>

#include "openvoice_amd.h"

namespace ovk {
namespace v1 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int CI_CHUNK = 16;         // input channels staged per LDS fill
constexpr int UNIT = 8;              // input channels per unrolled unit (4 ci-pairs x K taps)
constexpr int UPC = CI_CHUNK / UNIT; // units per chunk
constexpr int REC = 256;             // floats per packed-weight record (64 lanes x 4 k-steps)

// units are padded to a multiple of 4 (the largest units-per-chunk of any kernel variant)
__host__ __device__ inline int packed_units(int cin) { return ((cin + UNIT - 1) / UNIT + 3) / 4 * 4; }

__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }
__device__ __forceinline__ float sigmoidf(float v) { return 1.f / (1.f + __expf(-v)); }

// K taps, dilation DIL; wave tile = (32*WM) x (32*WN); WVM x WVN waves per workgroup;
// VEC: 16-byte staging loads (needs L % 4 == 0 and 16-byte aligned rows).
template <int K, int DIL, int WM, int WN, int WVM, int WVN, bool VEC>
__global__ __launch_bounds__(256, 2) void conv1d_mfma_v1_kernel(const ov_conv1d_params p) {
  static_assert(WVM * WVN == 4, "4 waves per workgroup");
  constexpr int N_BLK = 32 * WN * WVN;
  constexpr int PAD = (K - 1) * DIL / 2;
  constexpr int PADA = (PAD + 3) / 4 * 4;
  constexpr int XS = N_BLK + 2 * PADA;  // LDS row stride (floats), multiple of 4
  constexpr int XS4 = XS / 4;
  constexpr int NITEM = VEC ? CI_CHUNK * XS4 : CI_CHUNK * XS;
  constexpr int NV = (NITEM + 255) / 256;

  __shared__ __attribute__((aligned(16))) float xs[CI_CHUNK * XS];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WVN, wn = wave % WVN;
  const int b = blockIdx.z;
  const int t0 = blockIdx.x * N_BLK;
  const int L = p.L, Cin = p.Cin;
  const float* __restrict__ xb = p.x + (int64_t)b * p.x_bstride;
  const float slope = p.in_slope;

  const int nunits = packed_units(Cin);
  const int nchunks = nunits / UPC;
  const int recs_per_mtile = nunits * K + 1;
  const int mtile0 = (blockIdx.y * WVM + wm) * WM;

  const f32x4* __restrict__ wq[WM];
#pragma unroll
  for (int i = 0; i < WM; ++i)
    wq[i] = reinterpret_cast<const f32x4*>(p.w) + (int64_t)(mtile0 + i) * recs_per_mtile * 64 + lane;

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- staging: global -> registers (early) -> LDS (late) -----------------------------------
  f32x4 stg4[VEC ? NV : 1];
  float stg1[VEC ? 1 : NV];
  auto stage_load = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + 256 * i;
      if constexpr (VEC) {
        const int row = idx / XS4, c4 = idx - row * XS4;
        const int ci = chunk * CI_CHUNK + row;
        const int t = t0 - PADA + 4 * c4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (idx < NITEM && ci < Cin && t >= 0 && t < L)
          v = *reinterpret_cast<const f32x4*>(xb + (int64_t)ci * L + t);
        stg4[i] = v;
      } else {
        const int row = idx / XS, c = idx - row * XS;
        const int ci = chunk * CI_CHUNK + row;
        const int t = t0 - PADA + c;
        float v = 0.f;
        if (idx < NITEM && ci < Cin && t >= 0 && t < L) v = xb[(int64_t)ci * L + t];
        stg1[i] = v;
      }
    }
  };
  auto stage_write = [&]() {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + 256 * i;
      if (idx < NITEM) {
        if constexpr (VEC) {
          f32x4 v = stg4[i];
          v[0] = lrelu(v[0], slope); v[1] = lrelu(v[1], slope);
          v[2] = lrelu(v[2], slope); v[3] = lrelu(v[3], slope);
          *reinterpret_cast<f32x4*>(xs + 4 * idx) = v;  // row*XS + 4*c4 == 4*idx
        } else {
          xs[idx] = lrelu(stg1[i], slope);
        }
      }
    }
  };

  // per-lane LDS base of the B operand: row (lane>>5) of a ci pair, column n of this wave
  const float* xl = xs + (lane >> 5) * XS + wn * (32 * WN) + (lane & 31) + (PADA - PAD);

  f32x4 a_cur[WM], a_nxt[WM];
#pragma unroll
  for (int i = 0; i < WM; ++i) a_cur[i] = wq[i][0];
  int rec = 0;

  stage_load(0);
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    __syncthreads();  // every wave finished reading the previous chunk
    stage_write();
    __syncthreads();
    if (chunk + 1 < nchunks) stage_load(chunk + 1);
#pragma unroll
    for (int uu = 0; uu < UPC; ++uu) {
      const float* xu = xl + uu * UNIT * XS;
#pragma unroll
      for (int g = 0; g < K; ++g) {
        ++rec;  // the record after the last real one is zero padding written by the packer
#pragma unroll
        for (int i = 0; i < WM; ++i) a_nxt[i] = wq[i][(int64_t)rec * 64];
        // Pin the prefetch here: without this hipcc sinks the loads next to their first use and
        // every group starts with an L2-latency stall.
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int s = 4 * g + u;
          const int pp = s / K, tap = s - pp * K;
          float bv[WN];
#pragma unroll
          for (int j = 0; j < WN; ++j) bv[j] = xu[(2 * pp) * XS + 32 * j + tap * DIL];
#pragma unroll
          for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[i][u], bv[j], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < WM; ++i) a_cur[i] = a_nxt[i];
      }
    }
  }

  // ---- epilogue ---------------------------------------------------------------------------
  // C/D fragment: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  const int half = lane >> 5;
  const int epi = p.epi;
  const float scale = p.scale;
  const float* mrow = p.mask ? p.mask + (int64_t)b * L : nullptr;
  const float* bias = p.bias;
  const float* bias_b = p.bias_b ? p.bias_b + (int64_t)b * p.bias_b_bstride : nullptr;

  if (epi == OV_EPI_GATE || epi == OV_EPI_POSTERIOR) {
    if constexpr (WM == 2) {
      const int q = blockIdx.y * WVM + wm;  // pair index: packed tiles 2q (tanh | m), 2q+1 (sigmoid | logs)
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int col = t0 + wn * (32 * WN) + 32 * j + (lane & 31);
        const bool colok = col < L;
        const float mk = (mrow && colok) ? mrow[col] : 1.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rit = (r & 3) + 8 * (r >> 2) + 4 * half;
          const int ch = q * 32 + rit;
          const int row0 = q * 64 + rit, row1 = row0 + 32;
          float v0 = acc[0][j][r], v1 = acc[1][j][r];
          if (bias) { v0 += bias[row0]; v1 += bias[row1]; }
          if (bias_b) { v0 += bias_b[row0]; v1 += bias_b[row1]; }
          if (colok && ch < p.Cout) {
            const int64_t o = (int64_t)b * p.out_bstride + (int64_t)ch * L + col;
            if (epi == OV_EPI_GATE) {
              p.out[o] = tanhf(v0) * (1.f / (1.f + expf(-v1)));
            } else {
              const float nz = p.res[(int64_t)b * p.res_bstride + (int64_t)ch * L + col];
              p.out[o] = (v0 * mk + nz * scale * expf(v1 * mk)) * mk;
            }
          }
        }
      }
    }
    return;
  }

#pragma unroll
  for (int i = 0; i < WM; ++i) {
    const int mt = mtile0 + i;
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int col = t0 + wn * (32 * WN) + 32 * j + (lane & 31);
      const bool colok = col < L;
      const float mk = (mrow && colok) ? mrow[col] : 1.f;
      f32x16 v = acc[i][j];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (bias) v[r] += bias[row];
        if (bias_b) v[r] += bias_b[row];
      }
      if (epi == OV_EPI_CONVT) {
        const int s = p.phase_s;
        float* ob = p.out + (int64_t)b * p.out_bstride;
        const int64_t Lout = (int64_t)L * s;
        if (s == 8) {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int co = mt * 4 + c;
            if (colok && co < p.Cout) {
              f32x4 o = {v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]};
              *reinterpret_cast<f32x4*>(ob + co * Lout + 8 * (int64_t)col + 4 * half) = o;
            }
          }
        } else if (s == 2) {
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) {
            const int r = 2 * jj;
            const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const int co = row >> 1;
            if (colok && co < p.Cout) {
              f32x2 o = {v[r], v[r + 1]};
              *reinterpret_cast<f32x2*>(ob + co * Lout + 2 * (int64_t)col) = o;
            }
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const int co = row / s, ph = row - co * s;
            if (colok && co < p.Cout) ob[co * Lout + (int64_t)s * col + ph] = v[r];
          }
        }
        continue;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (!colok || row >= p.Cout) continue;
        const int64_t o = (int64_t)b * p.out_bstride + (int64_t)row * L + col;
        float val = v[r];
        if (epi == OV_EPI_LINEAR) {
          if (p.flags & OV_F_MASK_V) val *= mk;
          if (p.res) val += p.res[(int64_t)b * p.res_bstride + (int64_t)row * L + col];
          if (p.add) val += p.add[(int64_t)b * p.add_bstride + (int64_t)row * L + col];
          p.out[o] = val * scale;
        } else if (epi == OV_EPI_RESSKIP) {
          if (row < p.split) {
            p.out[o] = (p.out[o] + val) * mk;
          } else {
            const int64_t o2 = (int64_t)b * p.out2_bstride + (int64_t)(row - p.split) * L + col;
            p.out2[o2] = (p.flags & OV_F_OUT2_INIT) ? val : p.out2[o2] + val;
          }
        } else {  // OV_EPI_COUPLE
          const float m = val * mk;
          const float x1 = p.out[o];
          p.out[o] = scale > 0.f ? m + x1 * mk : (x1 - m) * mk;
        }
      }
    }
  }
}

typedef int (*conv_launch_fn)(const ov_conv1d_params*, hipStream_t);

template <int K, int DIL, int WM, int WN, int WVM, int WVN, bool VEC>
int conv1d_v1_launch(const ov_conv1d_params* p, hipStream_t stream) {
  constexpr int M_BLK = 32 * WM * WVM, N_BLK = 32 * WN * WVN;
  dim3 grid((p->L + N_BLK - 1) / N_BLK, (p->M + M_BLK - 1) / M_BLK, p->B);
  hipLaunchKernelGGL((conv1d_mfma_v1_kernel<K, DIL, WM, WN, WVM, WVN, VEC>), grid, dim3(256), 0, stream, *p);
  return hipGetLastError() == hipSuccess ? OV_OK : OV_E_LAUNCH;
}

// Tile ids used by the dispatcher.
enum { TILE_128x128 = 0, TILE_64x256 = 1, TILE_32x512 = 2 };

struct ConvVariant {
  int K, dil, tile, vec;
  conv_launch_fn fn;
};

// Each conv1d_inst_*.hip translation unit exports one table.
extern const ConvVariant kV1VariantsA[];
extern const int kV1NumVariantsA;
extern const ConvVariant kV1VariantsB[];
extern const int kV1NumVariantsB;
extern const ConvVariant kV1VariantsC[];
extern const int kV1NumVariantsC;
extern const ConvVariant kV1VariantsS[];
extern const int kV1NumVariantsS;

}  // namespace v1
}  // namespace ovk
