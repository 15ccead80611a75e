// Measurement tool: what the matrix pipes SUSTAIN under the chip's power limit on operands that look like data --
// the denominators the measured kernels should be read against, next to the instruction-rate peaks of the guide.
//   fp32  v_mfma_f32_32x32x2_f32   (peak 157.3 TFLOP/s at 2.4 GHz)   zero / random operands
//   bf16  v_mfma_f32_32x32x16_bf16 (peak 2516.6)                       random operands  (cf. mfma_bf16_ceiling.hip)
//   bf16  v_mfma_f32_16x16x32_bf16 (peak 2516.6)                       random operands: does the other shape draw less?
// One workgroup per CU, 2 waves per SIMD, ~200 ms per configuration after a ramp launch.
//   hipcc --offload-arch=gfx950 -O3 -o bin/mfma_sustained_ceilings mfma_sustained_ceilings.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void f32_loop(const float* __restrict__ src, float* out, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = src[(threadIdx.x * 4 + i) & 4095]; b[i] = src[(threadIdx.x * 4 + i + 2048) & 4095]; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(i + u) & 3], b[i], acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.678f) out[0] = s;
}

__global__ __launch_bounds__(512) void bf16_32_loop(const u32x4* __restrict__ src, float* out, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  u32x4 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = src[(threadIdx.x * 4 + i) & 4095]; b[i] = src[(threadIdx.x * 4 + i + 2048) & 4095]; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        bf16x8 x, y;
        __builtin_memcpy(&x, &a[(i + u) & 3], 16);
        __builtin_memcpy(&y, &b[i], 16);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[i], 0, 0, 0);
      }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.678f) out[0] = s;
}

__global__ __launch_bounds__(512) void bf16_16_loop(const u32x4* __restrict__ src, float* out, int iters) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i)
    for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
  u32x4 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = src[(threadIdx.x * 4 + i) & 4095]; b[i] = src[(threadIdx.x * 4 + i + 2048) & 4095]; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        bf16x8 x, y;
        __builtin_memcpy(&x, &a[(i + u) & 3], 16);
        __builtin_memcpy(&y, &b[i & 3], 16);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, acc[i], 0, 0, 0);
      }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i)
    for (int r = 0; r < 4; ++r) s += acc[i][r];
  if (s == 12345.678f) out[0] = s;
}

template <typename K, typename P>
void run(const char* what, K kernel, const P* src, float* d, double flop_per_wave_iter, double peak, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  kernel<<<256, 512>>>(src, d, iters / 3);
  (void)hipEventRecord(e0);
  kernel<<<256, 512>>>(src, d, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double tf = 256.0 * 8 * (double)iters * flop_per_wave_iter / ms / 1e9;
  printf("%-52s %8.2f ms  %8.1f TFLOP/s  = %.3f of %.1f  -> %.2f GHz sustained\n", what, ms, tf, tf / peak, peak,
         tf / peak * 2.4);
}

int main() {
  void* src;
  float* d;
  (void)hipMalloc(&src, 4096 * 16);
  (void)hipMalloc(&d, 4);
  uint32_t* h = (uint32_t*)malloc(4096 * 16);
  srand(11);
  (void)hipMemset(src, 0, 4096 * 16);
  run("fp32 32x32x2, zero operands", f32_loop, (const float*)src, d, 16 * 4096.0, 157.3, 110000);
  for (int i = 0; i < 4096 * 4; ++i) {       // random fp32, magnitude ~1, random mantissa
    const uint32_t sign = rand() & 1, exp = 124 + rand() % 6, man = ((uint32_t)rand() << 8 ^ (uint32_t)rand()) & 0x7fffff;
    h[i] = (sign << 31) | (exp << 23) | man;
  }
  (void)hipMemcpy(src, h, 4096 * 16, hipMemcpyHostToDevice);
  run("fp32 32x32x2, random fp32 operands", f32_loop, (const float*)src, d, 16 * 4096.0, 157.3, 110000);
  for (int i = 0; i < 4096 * 4; ++i) {       // random bf16 pairs
    uint32_t v = 0;
    for (int half = 0; half < 2; ++half) {
      const uint32_t sign = rand() & 1, exp = 124 + rand() % 6, man = rand() & 0x7f;
      v |= ((sign << 15) | (exp << 7) | man) << (16 * half);
    }
    h[i] = v;
  }
  (void)hipMemcpy(src, h, 4096 * 16, hipMemcpyHostToDevice);
  run("bf16 32x32x16, random bf16 operands", bf16_32_loop, (const u32x4*)src, d, 16 * 32768.0, 2516.6, 230000);
  run("bf16 16x16x32, random bf16 operands", bf16_16_loop, (const u32x4*)src, d, 32 * 16384.0, 2516.6, 230000);
  (void)hipMemset(src, 0, 4096 * 16);
  run("bf16 16x16x32, zero operands", bf16_16_loop, (const u32x4*)src, d, 32 * 16384.0, 2516.6, 230000);
  return 0;
}
