// Measurement tool: the main loop of mainloop_ceiling.hip wrapped in the per-tile structure of the real kernel
// (accumulator init, NCH chunks, epilogue stores of a 64x64 wave tile), one tile per workgroup, to find what
// the tile boundary costs.  Variants: STORE = write the fragment to global, CHK = chunk loop with LDS buffer
// switch, RES = read a residual fragment at tile start.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool STORE, bool RES, int K, int NCH>
__global__ __launch_bounds__(256) void tile(const f32x4* __restrict__ w, float* __restrict__ out,
                                            const float* __restrict__ res, int L) {
  __shared__ float xs[2 * 32 * 160];
  for (int i = threadIdx.x; i < 2 * 32 * 160; i += 256) xs[i] = (float)i * 1e-3f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int tile_id = blockIdx.x % 431, b = blockIdx.x / 431;
  f32x16 acc[2][2];
  const uint32_t col0 = tile_id * 128 + wn * 64 + (lane & 31);
  float* outb = out + (size_t)b * 128 * L;
  const float* resb = res + (size_t)b * 128 * L;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) {
    const uint32_t voff = ((wm * 2 + i) * 32 + 4 * half) * L + col0 + 32 * j;
    for (int r = 0; r < 16; ++r) acc[i][j][r] = RES ? (resb + (size_t)((r & 3) + 8 * (r >> 2)) * L)[voff] : 0.f;
  }
  uint32_t widx[2] = {(uint32_t)(wm * 2) * 64u * 97u + lane, (uint32_t)(wm * 2 + 1) * 64u * 97u + lane};
  f32x4 a_cur[2], a_nxt[2];
  a_cur[0] = w[widx[0]]; a_cur[1] = w[widx[1]];
  int rec = 0;
  constexpr int STEPS = 4 * 4 * K;
  for (int ch = 0; ch < NCH; ++ch) {
    const float* xl = xs + (ch & 1) * 32 * 160 + half * 160 + wn * 64 + (lane & 31);
    float bcur[2], bnxt[2];
    bcur[0] = xl[0]; bcur[1] = xl[32];
#pragma unroll
    for (int sa = 0; sa < STEPS; ++sa) {
      const int u = sa & 3;
      if (u == 0) { ++rec; a_nxt[0] = (w + (size_t)rec * 64)[widx[0]]; a_nxt[1] = (w + (size_t)rec * 64)[widx[1]]; }
      if (sa + 1 < STEPS) {
        const int uu = (sa + 1) / (4 * K), sn = (sa + 1) - uu * (4 * K);
        const int pp = sn / K, tap = sn - pp * K;
        bnxt[0] = xl[(uu * 8 + 2 * pp) * 160 + tap]; bnxt[1] = xl[(uu * 8 + 2 * pp) * 160 + 32 + tap];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[i][u], bcur[j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (sa + 1 < STEPS) { bcur[0] = bnxt[0]; bcur[1] = bnxt[1]; }
      if (u == 3) { a_cur[0] = a_nxt[0]; a_cur[1] = a_nxt[1]; }
    }
  }
  if (STORE) {
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) {
      const uint32_t voff = ((wm * 2 + i) * 32 + 4 * half) * L + col0 + 32 * j;
      for (int r = 0; r < 16; ++r) (outb + (size_t)((r & 3) + 8 * (r >> 2)) * L)[voff] = acc[i][j][r] * 0.5f;
    }
  } else {
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 12345.678f) out[0] = s;
  }
}

template <bool STORE, bool RES, int K, int NCH>
void run(const f32x4* w, float* out, const float* res, int L) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int wgs = 431 * 32;
  tile<STORE, RES, K, NCH><<<wgs, 256>>>(w, out, res, L);
  hipEventRecord(e0);
  for (int i = 0; i < 3; ++i) tile<STORE, RES, K, NCH><<<wgs, 256>>>(w, out, res, L);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  const double flops = (double)wgs * 4 * NCH * (16.0 * K) * 4 * 4096.0;   // 16K k-steps x 4 MFMAs per wave and chunk
  printf("K=%d chunks=%d store=%d res=%d  %.3f ms  %.1f %% of peak\n", K, NCH, STORE, RES, ms, flops / ms / 1e9 / 157.3 * 100);
}

int main() {
  const int L = 55104;
  f32x4* w; float *out, *res;
  hipMalloc(&w, 4 * 200 * 64 * sizeof(f32x4)); hipMemset(w, 0, 4 * 200 * 64 * sizeof(f32x4));
  hipMalloc(&out, (size_t)32 * 128 * L * 4); hipMalloc(&res, (size_t)32 * 128 * L * 4);
  hipMemset(res, 0, (size_t)32 * 128 * L * 4);
  run<false, false, 3, 4>(w, out, res, L);
  run<true, false, 3, 4>(w, out, res, L);
  run<true, true, 3, 4>(w, out, res, L);
  run<false, false, 11, 4>(w, out, res, L);
  run<true, false, 11, 4>(w, out, res, L);
  run<true, false, 3, 8>(w, out, res, L);
  return 0;
}
