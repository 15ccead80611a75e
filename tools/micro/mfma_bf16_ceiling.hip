// Measurement tool: what the bf16 matrix pipe SUSTAINS on MI355X under its power limit -- the denominator the bf16
// generator (BASELINE.json configs[4]) should be priced against.  v_mfma_f32_32x32x16_bf16 in a bare loop, 1 or 2
// waves per SIMD on all 256 CUs, for ~150 ms per configuration (the chip needs tens of ms to settle at its power-limited
// clock), with (a) all-zero operands, (b) random bf16 operands held in registers, (c) random operands where the A
// operand of every MFMA is re-read from LDS (ds_read_b128) as the conv kernels do.  Prints TFLOP/s and the shader
// clock that rate implies (2 516.6 TFLOP/s at 2.4 GHz).   hipcc --offload-arch=gfx950 -O3 -o bin/mfma_bf16_ceiling ...
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <bool LDS>
__global__ __launch_bounds__(512) void mfma_loop(const u32x4* __restrict__ src, float* out, int iters) {
  __shared__ u32x4 tile[4 * 512];
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  u32x4 a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    a[i] = src[(threadIdx.x * 4 + i) & 4095];
    b[i] = src[(threadIdx.x * 4 + i + 2048) & 4095];
    tile[i * 512 + threadIdx.x] = a[i];
  }
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        u32x4 av = a[i];
        if (LDS) av = tile[((i + u) & 3) * 512 + threadIdx.x];
        bf16x8 x, y;
        __builtin_memcpy(&x, &av, 16);
        __builtin_memcpy(&y, &b[(i + u) & 3], 16);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[i], 0, 0, 0);
      }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.678f) out[0] = s;
}

template <bool LDS>
void run(const char* what, const u32x4* src, float* d, int waves_per_simd, double target_ms) {
  const int threads = 256 * waves_per_simd;          // one workgroup per CU
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  int iters = (int)(350000 / waves_per_simd * (target_ms / 150.0));
  mfma_loop<LDS><<<256, threads>>>(src, d, iters / 3);              // ramp to the sustained clock
  hipEventRecord(e0);
  mfma_loop<LDS><<<256, threads>>>(src, d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = 256.0 * 4 * waves_per_simd * (double)iters * 16 * 32768.0;
  const double tf = flops / ms / 1e9;
  printf("%-44s waves/SIMD=%d  %8.2f ms  %7.1f TFLOP/s  = %.3f of 2516.6  -> %.2f GHz sustained\n", what,
         waves_per_simd, ms, tf, tf / 2516.6, tf / 2516.6 * 2.4);
}

int main() {
  u32x4* src;
  float* d;
  hipMalloc(&src, 4096 * sizeof(u32x4));
  hipMalloc(&d, 4);
  uint32_t* h = (uint32_t*)malloc(4096 * 16);
  // zero operands
  hipMemset(src, 0, 4096 * 16);
  for (int w = 1; w <= 2; ++w) run<false>("zero operands, registers", src, d, w, 150);
  // random bf16 operands: sign + exponent around 1.0 (0x3f80) +- 3, random mantissa -- activations / weights of O(1)
  srand(7);
  for (int i = 0; i < 4096 * 4; ++i) {
    uint32_t v = 0;
    for (int half = 0; half < 2; ++half) {
      const uint32_t sign = rand() & 1, exp = 124 + rand() % 6, man = rand() & 0x7f;
      v |= ((sign << 15) | (exp << 7) | man) << (16 * half);
    }
    h[i] = v;
  }
  hipMemcpy(src, h, 4096 * 16, hipMemcpyHostToDevice);
  for (int w = 1; w <= 2; ++w) run<false>("random bf16 operands, registers", src, d, w, 150);
  for (int w = 1; w <= 2; ++w) run<true>("random operands, A re-read from LDS per MFMA", src, d, w, 150);
  for (int w = 1; w <= 2; ++w) run<false>("random bf16 operands, registers, 600 ms", src, d, w, 600);
  return 0;
}
