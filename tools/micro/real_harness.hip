// Measurement tool: the production conv kernel (conv1d_mfma.h, compiled from source here) launched from a bare
// HIP program with hipEvent timing, to compare like for like with tile_ceiling.hip in one process.
//   hipcc --offload-arch=gfx950 -O3 -I include -I openvoice_amd/csrc [-DOV_EXP=1] tools/micro/real_harness.hip
#include "conv1d_mfma.h"
#include <cstdio>
#include <cstdlib>
using namespace ovk;

template <int K, int DIL, int CHUNK, int NLD>
void run(const ov_conv1d_params& base, int tpw) {
  ov_conv1d_params p = base;
  p.K = K; p.dil = DIL; p.tiles_per_wg = tpw;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  conv1d_launch<K, DIL, OV_TILE_128x128, CHUNK, true, OV_EPI_LINEAR, NLD>(&p, 0);
  (void)hipEventRecord(e0);
  for (int i = 0; i < 3; ++i) conv1d_launch<K, DIL, OV_TILE_128x128, CHUNK, true, OV_EPI_LINEAR, NLD>(&p, 0);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  const double flops = 2.0 * p.M * p.Cin * K * (double)p.L * p.B;
  printf("exp=%d real k=%d d=%d chunk=%d nld=%d tpw=%d  %.3f ms  %.1f %% of peak  (%s)\n", OV_EXP, K, DIL, CHUNK, NLD, tpw, ms,
         flops / ms / 1e9 / 157.3 * 100, hipGetErrorString(hipGetLastError()));
}

template <int K, int DIL, int CHUNK, int NLD>
void run_grid(const ov_conv1d_params& base, int nwg) {
  ov_conv1d_params p = base;
  p.K = K; p.dil = DIL;
  auto kernel = conv1d_mfma_kernel<K, DIL, OV_TILE_128x128, CHUNK, true, OV_EPI_LINEAR, NLD>;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(kernel, dim3(nwg), dim3(64 * (4 + NLD)), 0, 0, p);
  (void)hipEventRecord(e0);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kernel, dim3(nwg), dim3(64 * (4 + NLD)), 0, 0, p);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  const double flops = 2.0 * p.M * p.Cin * K * (double)p.L * p.B;
  printf("exp=%d real k=%d d=%d grid=%d  %.3f ms  %.1f %% of peak\n", OV_EXP, K, DIL, nwg, ms, flops / ms / 1e9 / 157.3 * 100);
}

int main(int argc, char** argv) {
  const int B = 32, C = 128, L = 55104;
  const size_t n = (size_t)B * C * L;
  float *x, *out, *w, *bias;
  (void)hipMalloc(&x, n * 4 + 4096); (void)hipMalloc(&out, n * 4 + 4096);
  const size_t wn = (size_t)4 * (packed_units(C) * 11 + 1) * REC;
  (void)hipMalloc(&w, wn * 4); (void)hipMalloc(&bias, 4096);
  float* h = (float*)malloc(n * 4);
  unsigned s = 12345u;
  for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((int)(s >> 8) % 20001 - 10000) * 1e-4f; }
  (void)hipMemcpy(x, h, n * 4, hipMemcpyHostToDevice);
  for (size_t i = 0; i < wn; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((int)(s >> 8) % 20001 - 10000) * 1e-5f; }
  (void)hipMemcpy(w, h, wn * 4, hipMemcpyHostToDevice);
  (void)hipMemset(bias, 0, 4096);
  ov_conv1d_params p{};
  p.x = x; p.w = w; p.bias = bias; p.out = out;
  p.x_bstride = p.out_bstride = p.res_bstride = p.add_bstride = (int64_t)C * L;
  p.B = B; p.Cin = C; p.L = L; p.x_ld = L; p.out_ld = L; p.M = C; p.Cout = C;
  p.epi = OV_EPI_LINEAR; p.in_slope = 0.1f; p.scale = 1.f;
  {
    auto k = conv1d_mfma_kernel<3, 1, OV_TILE_128x128, 32, true, OV_EPI_LINEAR, 2>;
    printf("resident workgroups (occupancy API): %d\n", query_resident_workgroups(reinterpret_cast<const void*>(k), 384));
  }
  for (int g : {512, 511, 510, 509, 508, 504, 500, 496, 480, 448, 384, 256, 255, 1024, 1023, 768, 767}) run_grid<3, 1, 32, 2>(p, g);
  for (int g : {512, 511, 509}) { run_grid<3, 5, 32, 2>(p, g); run_grid<11, 1, 32, 2>(p, g); run_grid<7, 1, 32, 2>(p, g); }
  for (int tpw : {0, 1}) {
    run<3, 1, 32, 2>(p, tpw);
    if (tpw < 2) { run<3, 5, 32, 2>(p, tpw); run<11, 1, 32, 2>(p, tpw); }
  }
  return 0;
}
