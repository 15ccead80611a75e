// Measurement tool: how many bytes per clock can the waves of ONE CU pull through the L1 / TA path as 1 KiB records
// (64 lanes x global_load_dwordx4, scalar base + lane offset -- the weight stream of the bf16 conv kernels), all 256 CUs
// reading the SAME working set at the same time?  Working sets: 16 KiB (L1), 704 KiB (one conv's weights at C = 128,
// K = 11: L2), 8 MiB (beyond one XCD's L2); 4 or 8 streaming waves per CU, requests 8 deep per wave.
//   hipcc --offload-arch=gfx950 -O3 -o bin/l2_stream_ceiling l2_stream_ceiling.hip && bin/l2_stream_ceiling
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// every wave walks the whole working set `passes` times, its start offset staggered by a quarter of the set per wave
__global__ __launch_bounds__(512) void stream(const u32x4* __restrict__ src, uint32_t* out, int records, int passes,
                                              unsigned long long* cycles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  typedef const __attribute__((address_space(1))) u32x4* gp;
  u32x4 acc = {0u, 0u, 0u, 0u};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int p = 0; p < passes; ++p) {
    int r = (records / nw) * wave;
    for (int i = 0; i < records; i += 8) {
      u32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        int rr = r + u;
        if (rr >= records) rr -= records;
        gp base = (gp)(src) + (size_t)__builtin_amdgcn_readfirstlane(rr) * 64;
        v[u] = base[lane];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc ^= v[u];
      r += 8;
      if (r >= records) r -= records;
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0 && wave == 0) cycles[blockIdx.x] = t1 - t0;
  if (acc[0] == 0x12345678u && acc[1] == 1u) out[0] = acc[2] ^ acc[3];
}

int main() {
  const size_t maxb = 16u << 20;
  std::vector<uint32_t> h(maxb / 4);
  uint32_t s = 12345u;
  for (auto& w : h) { s = s * 1664525u + 1013904223u; w = s; }
  u32x4* src; uint32_t* out; unsigned long long* cyc;
  hipMalloc(&src, maxb); hipMalloc(&out, 64); hipMalloc(&cyc, 256 * 8);
  hipMemcpy(src, h.data(), maxb, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  printf("%10s %6s %10s %12s %14s %14s\n", "set KiB", "waves", "ms", "GB/s chip", "B/clk/CU(ev)", "B/tick/CU");
  for (int kib : {16, 704, 8192}) {
    for (int nw : {4, 8}) {
      const int records = kib, passes = (int)((512ll << 10) / kib) + 1;   // ~512 MiB per wave
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(stream, dim3(256), dim3(64 * nw), 0, 0, src, out, records, passes, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep == 0) continue;
        unsigned long long hc[256]; hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost);
        double mean = 0; for (int i = 0; i < 256; ++i) mean += (double)hc[i]; mean /= 256;
        const double bytes_cu = (double)records * 1024.0 * passes * nw;
        printf("%10d %6d %10.3f %12.0f %14.1f %14.1f\n", kib, nw, ms, bytes_cu * 256 / ms / 1e6,
               bytes_cu / (ms * 1e-3 * 2.4e9), bytes_cu / mean);
      }
    }
  }
  return 0;
}
