// Measurement tool: do the fp32 matrix pipe (v_mfma_f32_32x32x2_f32) and the fp32 vector pipe (v_pk_fma_f32) of a
// SIMD add up when one wave of each kind shares the SIMD, and what does the shader clock do under the combined load?
// A workgroup = MW matrix waves + VW vector waves per SIMD (x 4 SIMDs); each kind runs a dependency-free loop of its
// own instruction.  Reported per configuration: matrix TF/s, vector TF/s, sum, shader clock (s_memtime / wall clock).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/coissue_ceiling.hip -o /tmp/coissue && /tmp/coissue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// role: waves [0, 4*MW) matrix, [4*MW, 4*(MW+VW)) vector.  viters / miters sized so both kinds run about equally long.
__global__ __launch_bounds__(1024) void coissue(float* out, long long* clk, int nm_waves, int miters, int viters,
                                                float a0, float b0) {
  const int wave = threadIdx.x >> 6;
  const long long c0 = clock64();
  const long long w0 = wall_clock64();
  float s = 0.f;
  if (wave < nm_waves) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const float a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    for (int it = 0; it < miters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i)
      for (int r = 0; r < 16; ++r) s += acc[i][r];
  } else {
    f32x2 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x2{0.f, 0.f};
    f32x2 w = {a0 * 1e-3f, b0 * 1e-3f};
    const f32x2 x = {1.f + threadIdx.x * 1e-6f, 1.f - threadIdx.x * 1e-6f};
    for (int it = 0; it < viters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_elementwise_fma(w, x, acc[i]);
    }
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1];
  }
  const long long c1 = clock64();
  const long long w1 = wall_clock64();
  if (s == 12345.678f) out[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

static void run(int mw, int vw, float* d, long long* dclk, double ms_target) {
  // one MFMA = 64 cycles for 4096 FLOP; one v_pk_fma_f32 = 256 FLOP (assumed 4 cycles)
  const int nwaves = 4 * (mw + vw);
  const int miters = (int)(ms_target * 2.3e6 / (16 * 64.0) / (mw > 0 ? mw : 1));
  const int viters = (int)(ms_target * 2.3e6 / (64 * 4.0) / (vw > 0 ? vw : 1));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  coissue<<<256, 64 * nwaves>>>(d, dclk, 4 * mw, miters, viters, 1.f, 2.f);   // warm-up + clock ramp
  hipEventRecord(e0);
  coissue<<<256, 64 * nwaves>>>(d, dclk, 4 * mw, miters, viters, 1.f, 2.f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long clk[2]; hipMemcpy(clk, dclk, sizeof(clk), hipMemcpyDeviceToHost);
  const double mfl = 256.0 * 4 * mw * (double)miters * 16 * 4096.0;
  const double vfl = 256.0 * 4 * vw * (double)viters * 64 * 256.0;
  printf("matrix waves/SIMD=%d vector waves/SIMD=%d  %.2f ms  matrix %.1f TF/s  vector %.1f TF/s  sum %.1f  "
         "(wg0: %lld clk / %lld wall ticks @100MHz = %.2f GHz if s_memtime counts shader clocks)\n",
         mw, vw, ms, mfl / ms / 1e9, vfl / ms / 1e9, (mfl + vfl) / ms / 1e9, clk[0], clk[1],
         clk[1] ? clk[0] / (clk[1] * 10.0) : 0.0);
}

int main() {
  float* d; hipMalloc(&d, 4);
  long long* dclk; hipMalloc(&dclk, 16);
  const double ms = 40.0;
  run(1, 0, d, dclk, ms);
  run(2, 0, d, dclk, ms);
  run(0, 1, d, dclk, ms);
  run(0, 2, d, dclk, ms);
  run(1, 1, d, dclk, ms);
  run(2, 1, d, dclk, ms);
  run(1, 2, d, dclk, ms);
  run(2, 2, d, dclk, ms);
  return 0;
}
