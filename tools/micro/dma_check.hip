// Measurement / bring-up tool: the LDS-DMA staging variant of the conv kernel against the register-staged variant
// of the same tile, in a bare HIP program (no Python: runs in ~2 s).  Bitwise comparison on small ragged shapes,
// then interleaved timing at the benchmark shape with the clocks pre-heated.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I openvoice_amd/csrc tools/micro/dma_check.hip -o tools/micro/build/dma_check
#include "conv1d_mfma.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace ovk;

__global__ void fill(float* p, size_t n, unsigned seed, float scale) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    p[i] = ((int)(h >> 8) % 20001 - 10000) * 1e-4f * scale;   // [-1, 1] * scale, both signs (exercises the leaky ReLU)
  }
}
__global__ void diff(const float* a, const float* b, size_t n, unsigned long long* cnt, float* mx) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const unsigned ua = __float_as_uint(a[i]), ub = __float_as_uint(b[i]);
    if (ua != ub) { atomicAdd(cnt, 1ull); atomicMax((int*)mx, __float_as_int(fabsf(a[i] - b[i]))); }
  }
}

// same layout as ov_conv1d_pack_f32 (ov_api.hip): one zero record after the last real one of every 32-row tile
static std::vector<float> pack(const std::vector<float>& w, int Cout, int Cin, int K) {
  const int mtiles = (Cout + 127) / 128 * 128 / 32, nu = packed_units(Cin);
  const size_t recs = (size_t)nu * K + 1;
  std::vector<float> dst((size_t)mtiles * recs * REC, 0.f);
  for (int mt = 0; mt < mtiles; ++mt)
    for (int U = 0; U < nu; ++U)
      for (int g = 0; g < K; ++g) {
        float* rec = dst.data() + ((size_t)mt * recs + (size_t)U * K + g) * REC;
        for (int lane = 0; lane < 64; ++lane)
          for (int u = 0; u < 4; ++u) {
            const int s = 4 * g + u, pp = s / K, tap = s - pp * K;
            const int ci = UNIT * U + 2 * pp + (lane >> 5), co = 32 * mt + (lane & 31);
            if (co < Cout && ci < Cin) rec[lane * 4 + u] = w[((size_t)co * Cin + ci) * K + tap];
          }
      }
  return dst;
}

struct Bufs { float *x, *res, *o1, *o2, *w, *bias; unsigned long long* cnt; float* mx; };

template <int K, int DIL, int WM, int WN, int WVM, int WVN, int CHUNK, int NLD_REG>
void check(const Bufs& b, int C, int B, int L, bool with_res) {
  std::vector<float> w((size_t)C * C * K);
  unsigned s = 777u + K * 31 + DIL;
  for (auto& v : w) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) % 20001 - 10000) * 1e-4f / sqrtf((float)C * K); }
  auto pk = pack(w, C, C, K);
  (void)hipMemcpy(b.w, pk.data(), pk.size() * 4, hipMemcpyHostToDevice);
  const size_t n = (size_t)B * C * L;
  ov_conv1d_params p{};
  p.x = b.x; p.w = b.w; p.bias = b.bias; p.res = with_res ? b.res : nullptr;
  p.x_bstride = p.out_bstride = p.res_bstride = p.add_bstride = (int64_t)C * L;
  p.B = B; p.Cin = C; p.L = L; p.x_ld = L; p.out_ld = L; p.M = C; p.Cout = C; p.K = K; p.dil = DIL;
  p.epi = OV_EPI_LINEAR; p.in_slope = 0.1f; p.scale = 1.f;
  (void)hipMemset(b.o1, 0xff, n * 4); (void)hipMemset(b.o2, 0x7f, n * 4);
  (void)hipMemset(b.cnt, 0, 8); (void)hipMemset(b.mx, 0, 4);
  p.out = b.o1; const int r1 = conv1d_launch<K, DIL, WM, WN, WVM, WVN, CHUNK, STAGE_VEC, OV_EPI_LINEAR, NLD_REG>(&p, 0);
  p.out = b.o2; const int r2 = conv1d_launch<K, DIL, WM, WN, WVM, WVN, CHUNK, STAGE_DMA, OV_EPI_LINEAR, 1>(&p, 0);
  diff<<<512, 256>>>(b.o1, b.o2, n, b.cnt, b.mx);
  unsigned long long cnt = 0; float mx = 0;
  (void)hipMemcpy(&cnt, b.cnt, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&mx, b.mx, 4, hipMemcpyDeviceToHost);
  float probe[2]; (void)hipMemcpy(probe, b.o2 + n / 2, 8, hipMemcpyDeviceToHost);
  printf("check k=%d d=%d tile=%dx%d chunk=%d C=%d B=%d L=%d res=%d: rc %d/%d  differing %llu of %zu  max|diff| %.3e  (out[n/2]=%g)  %s\n",
         K, DIL, 32 * WM * WVM, 32 * WN * WVN, CHUNK, C, B, L, (int)with_res, r1, r2, cnt, n, mx, probe[0],
         hipGetErrorString(hipGetLastError()));
}

template <int K, int DIL>
void timing(const Bufs& b, int C, int B, int L) {
  std::vector<float> w((size_t)C * C * K, 0.01f);
  auto pk = pack(w, C, C, K);
  (void)hipMemcpy(b.w, pk.data(), pk.size() * 4, hipMemcpyHostToDevice);
  ov_conv1d_params p{};
  p.x = b.x; p.w = b.w; p.bias = b.bias; p.out = b.o1;
  p.x_bstride = p.out_bstride = p.res_bstride = p.add_bstride = (int64_t)C * L;
  p.B = B; p.Cin = C; p.L = L; p.x_ld = L; p.out_ld = L; p.M = C; p.Cout = C; p.K = K; p.dil = DIL;
  p.epi = OV_EPI_LINEAR; p.in_slope = 0.1f; p.scale = 1.f;
  auto reg = [&] { conv1d_launch<K, DIL, OV_TILE_128x128, 32, STAGE_VEC, OV_EPI_LINEAR, 2>(&p, 0); };
  auto dma = [&] { conv1d_launch<K, DIL, OV_TILE_128x128, 32, STAGE_DMA, OV_EPI_LINEAR, 1>(&p, 0); };
  hipEvent_t e[5]; for (auto& x : e) (void)hipEventCreate(&x);
  for (int i = 0; i < 30; ++i) reg();          // pre-heat (profiles/r01_s36: the clock needs ~50 ms of matrix work)
  float tr = 0, td = 0;
  for (int round = 0; round < 2; ++round) {    // interleaved: drift hits both alike
    (void)hipEventRecord(e[0]); for (int i = 0; i < 4; ++i) reg();
    (void)hipEventRecord(e[1]); for (int i = 0; i < 4; ++i) dma();
    (void)hipEventRecord(e[2]); (void)hipEventSynchronize(e[2]);
    float a, c; (void)hipEventElapsedTime(&a, e[0], e[1]); (void)hipEventElapsedTime(&c, e[1], e[2]);
    tr += a / 8; td += c / 8;
  }
  const double fl = 2.0 * C * C * K * (double)L * B;
  printf("time  k=%d d=%d C=%d B=%d L=%d: register-staged (2 loader waves) %.3f ms %.1f %%   lds-dma (1 loader wave) %.3f ms %.1f %% of fp32 MFMA peak  %s\n",
         K, DIL, C, B, L, tr, fl / tr / 1e9 / 157.3 * 100, td, fl / td / 1e9 / 157.3 * 100, hipGetErrorString(hipGetLastError()));
}

int main() {
  const size_t big = (size_t)32 * 128 * 55104;
  Bufs b{};
  (void)hipMalloc(&b.x, big * 4 + 4096); (void)hipMalloc(&b.res, (size_t)2 * 256 * 2312 * 4);
  (void)hipMalloc(&b.o1, big * 4 + 4096); (void)hipMalloc(&b.o2, (size_t)2 * 256 * 2312 * 4);
  (void)hipMalloc(&b.w, (size_t)8 << 20); (void)hipMalloc(&b.bias, 4096); (void)hipMalloc(&b.cnt, 8); (void)hipMalloc(&b.mx, 4);
  fill<<<1024, 256>>>(b.x, big, 1u, 1.f); fill<<<64, 256>>>(b.res, (size_t)2 * 256 * 2312, 2u, 1.f);
  fill<<<1, 256>>>(b.bias, 1024, 3u, 0.1f);
  // tiles: 128x128 = <2,2,2,2>, 64x256 = <2,2,1,4>, 32x256 = <1,2,1,4>
  check<3, 1, 2, 2, 2, 2, 32, 2>(b, 128, 2, 2312, false);
  check<3, 1, 2, 2, 2, 2, 32, 2>(b, 128, 2, 2312, true);
  check<11, 5, 2, 2, 2, 2, 32, 2>(b, 256, 2, 1096, true);
  check<7, 3, 2, 2, 1, 4, 16, 4>(b, 64, 2, 2312, true);
  check<3, 1, 1, 2, 1, 4, 16, 4>(b, 32, 2, 2312, false);
  check<11, 5, 1, 2, 1, 4, 16, 4>(b, 32, 2, 2312, true);
  timing<3, 1>(b, 128, 32, 55104);
  timing<11, 1>(b, 128, 32, 55104);
  (void)hipDeviceSynchronize();
  return 0;
}
