// Measurement tool: which ingredient of the conv main loop costs matrix-pipe time?  A stripped copy of the
// k-step loop of conv1d_mfma.h (64x64 wave tile, 4 accumulators, A from global in fragment order one group
// ahead, B from LDS one k-step ahead) with each ingredient switchable.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool LDS, bool GLB, bool SCHED, int K>
__global__ __launch_bounds__(256) void loop(const f32x4* __restrict__ w, float* out, int groups_total, int wrecs) {
  __shared__ float xs[32 * 160];
  for (int i = threadIdx.x; i < 32 * 160; i += 256) xs[i] = (float)i * 1e-3f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const float* xl = xs + (lane >> 5) * 160 + (wave & 1) * 64 + (lane & 31);
  uint32_t widx[2] = {(uint32_t)((wave >> 1) * 2) * 64u * 97u + lane, (uint32_t)((wave >> 1) * 2 + 1) * 64u * 97u + lane};
  f32x4 a_cur[2], a_nxt[2];
  a_cur[0] = w[widx[0]]; a_cur[1] = w[widx[1]];
  float bcur[2], bnxt[2];
  bcur[0] = xl[0]; bcur[1] = xl[32];
  int rec = 0;
  constexpr int STEPS = 4 * 4 * K;   // one 32-channel chunk
  for (int g0 = 0; g0 < groups_total; g0 += STEPS / 4) {
#pragma unroll
    for (int sa = 0; sa < STEPS; ++sa) {
      const int u = sa & 3;
      if (u == 0) {
        rec = (rec + 1) % wrecs;
        if (GLB) { a_nxt[0] = (w + (size_t)rec * 64)[widx[0]]; a_nxt[1] = (w + (size_t)rec * 64)[widx[1]]; }
        else { a_nxt[0] = a_cur[0]; a_nxt[1] = a_cur[1]; }
      }
      {
        const int sn1 = (sa + 1) % STEPS;
        const int uu = sn1 / (4 * K), sn = sn1 - uu * (4 * K);
        const int pp = sn / K, tap = sn - pp * K;
        if (LDS) { bnxt[0] = xl[(uu * 8 + 2 * pp) * 160 + tap]; bnxt[1] = xl[(uu * 8 + 2 * pp) * 160 + 32 + tap]; }
        else { bnxt[0] = bcur[0]; bnxt[1] = bcur[1]; }
      }
      if (SCHED) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[i][u], bcur[j], acc[i][j], 0, 0, 0);
      if (SCHED) __builtin_amdgcn_sched_barrier(0);
      bcur[0] = bnxt[0]; bcur[1] = bnxt[1];
      if (u == 3) { a_cur[0] = a_nxt[0]; a_cur[1] = a_nxt[1]; }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  if (s == 12345.678f) out[0] = s;
}

template <bool LDS, bool GLB, bool SCHED, int K>
void run(const f32x4* w, float* d, int wg_per_cu) {
  const int groups = 16 * K * 40;   // 40 chunks
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  loop<LDS, GLB, SCHED, K><<<256 * wg_per_cu, 256>>>(w, d, 16 * K, 96);
  hipEventRecord(e0);
  loop<LDS, GLB, SCHED, K><<<256 * wg_per_cu, 256>>>(w, d, groups, 96);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = 256.0 * wg_per_cu * 4 * (double)groups * 16 * 4096.0;
  printf("K=%d lds=%d glb=%d sched=%d wg/CU=%d  %.3f ms  %.1f %% of peak\n", K, LDS, GLB, SCHED, wg_per_cu, ms,
         flops / ms / 1e9 / 157.3 * 100);
}

int main() {
  f32x4* w; float* d;
  hipMalloc(&w, 4 * 97 * 64 * sizeof(f32x4) + 1024); hipMemset(w, 0, 4 * 97 * 64 * sizeof(f32x4)); hipMalloc(&d, 4);
  for (int wg = 1; wg <= 2; ++wg) {
    run<false, false, false, 3>(w, d, wg);
    run<true, false, false, 3>(w, d, wg);
    run<false, true, false, 3>(w, d, wg);
    run<true, true, false, 3>(w, d, wg);
    run<true, true, true, 3>(w, d, wg);
    run<true, true, true, 11>(w, d, wg);
  }
  return 0;
}
