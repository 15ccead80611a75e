// Measurement tool: issue-rate ceiling of v_mfma_f32_32x32x2_f32 as a function of waves per SIMD and of
// independent accumulators per wave.  Built and run on the GPU box:  hipcc --offload-arch=gfx950 -O3 ... && ./a.out
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x, b = b0 - threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16 / NACC; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.678f) out[0] = s;
}

template <int NACC>
void run(int wg_per_cu, float* d) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  mfma_loop<NACC><<<256 * wg_per_cu, 256>>>(d, 100, 1.f, 2.f);
  hipEventRecord(e0);
  mfma_loop<NACC><<<256 * wg_per_cu, 256>>>(d, iters, 1.f, 2.f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = 256.0 * wg_per_cu * 4 * iters * 16 * 4096.0;
  printf("acc=%d waves/SIMD=%d  %.3f ms  %.1f TF/s  %.1f %% of 157.3\n", NACC, wg_per_cu, ms, flops / ms / 1e9,
         flops / ms / 1e9 / 157.3 * 100);
}

int main() {
  float* d; hipMalloc(&d, 4);
  for (int w = 1; w <= 4; ++w) run<4>(w, d);
  for (int w = 1; w <= 4; ++w) run<2>(w, d);
  for (int w = 1; w <= 2; ++w) run<1>(w, d);
  return 0;
}
