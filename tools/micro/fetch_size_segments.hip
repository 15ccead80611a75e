// Measurement tool: what rocprofv3's FETCH_SIZE counts for the split-precision conv's input pattern.  That kernel reads a
// channels-last bf16 tensor [rows][C] as 32-channel chunks: one request = 16 rows x 64 B (64 lanes x 16 B), 64-byte row
// segments at a stride of the row size (128 / 256 / 512 B for C = 64 / 128 / 256), the segments of a row one chunk-time
// apart.  profiles/r05_s20 shows FETCH_SIZE x 2 (the factor measured on wide copies) at 1.02 / 1.18 / 1.22 x the bytes for the
// three widths -- re-fetched lines, or a different count per request shape?  Here: every byte of a 1 GiB tensor read exactly
// once in that pattern (through registers and through the LDS-DMA path), next to a wide contiguous read of the same bytes.
//   hipcc --offload-arch=gfx950 -O3 -o bin/fetch_size_segments fetch_size_segments.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- bin/fetch_size_segments
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) char* gptr;

// workgroup = 128 rows; for each 64-byte segment of the rows in turn, its 4 waves read 2 x (16 rows x 64 B) each
template <int STRIDE, bool DMA>
__global__ __launch_bounds__(256) void seg_read(const char* __restrict__ src, uint32_t* out) {
  __shared__ __attribute__((aligned(16))) char lds[4 * 2 * 1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t row0 = (size_t)blockIdx.x * 128 + wave * 32;
  u32x4 acc = {0u, 0u, 0u, 0u};
#pragma unroll 1
  for (int seg = 0; seg < STRIDE / 64; ++seg) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const char* p = src + (row0 + 16 * h + lane / 4) * STRIDE + seg * 64 + (lane % 4) * 16;
      if constexpr (DMA) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                         (__attribute__((address_space(3))) void*)(lds + (wave * 2 + h) * 1024), 16, 0, 0);
      } else {
        acc ^= *reinterpret_cast<const u32x4*>(p);
      }
    }
    if constexpr (DMA) {
      __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
      acc ^= *reinterpret_cast<const u32x4*>(lds + wave * 2048 + lane * 16);
    }
    __builtin_amdgcn_s_sleep(32);           // the segments of a row are a chunk-time apart in the real kernel
  }
  if (acc[0] == 0x12345678u && acc[1] == 1u) out[0] = acc[2] ^ acc[3];
}

__global__ __launch_bounds__(256) void wide_read(const u32x4* __restrict__ src, uint32_t* out, size_t n) {
  u32x4 acc = {0u, 0u, 0u, 0u};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc ^= src[i];
  if (acc[0] == 0x12345678u && acc[1] == 1u) out[0] = acc[2] ^ acc[3];
}

template <int STRIDE, bool DMA>
static void run(const char* src, uint32_t* out, size_t bytes) {
  const size_t rows = bytes / STRIDE;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((seg_read<STRIDE, DMA>), dim3((unsigned)(rows / 128)), dim3(256), 0, 0, src, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("seg_read<%d, %s>: %.3f ms, %.2f TB/s\n", STRIDE, DMA ? "lds-dma" : "registers", ms, bytes / ms / 1e9);
}

int main() {
  const size_t bytes = (size_t)1 << 30;
  char* src; uint32_t* out;
  hipMalloc(&src, bytes); hipMalloc(&out, 64);
  hipMemset(src, 1, bytes);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(wide_read, dim3(4096), dim3(256), 0, 0, (const u32x4*)src, out, bytes / 16);
    run<128, false>(src, out, bytes); run<256, false>(src, out, bytes); run<512, false>(src, out, bytes);
    run<128, true>(src, out, bytes); run<256, true>(src, out, bytes); run<512, true>(src, out, bytes);
  }
  hipDeviceSynchronize();
  return 0;
}
