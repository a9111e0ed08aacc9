// Diagnostic microbenchmark (not part of the library): per-CU streaming rate of
// "load everything up front" kernels as a function of bytes per block, block
// count, access shape and cache state.  hipcc --offload-arch=gfx950 -O3 membench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// each wave loads NL x 1 KiB; shape 0: contiguous 1 KiB per instruction;
// shape 1: 16 rows x 64 B per instruction (row stride 512 B)
template <int NL, int SHAPE>
__global__ __launch_bounds__(256) void k_read(const u32x4* __restrict__ src, float* out, size_t block_stride_v4, int share) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // share: all blocks with the same (blockIdx & 7) read the same region (L2 sharing on one XCD)
  const size_t blk = share ? (blockIdx.x & 7) : blockIdx.x;
  const u32x4* base = src + blk * block_stride_v4 + (size_t)wave * NL * 64;
  u32x4 v[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    size_t off;
    if (SHAPE == 0) off = (size_t)i * 64 + lane;
    else { const int r = lane & 15, q = lane >> 4; off = (size_t)r * 32 + (size_t)(i % 8) * 4 + q + (size_t)(i / 8) * 512; }
    v[i] = base[off];
  }
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < NL; ++i) s += v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
  if (s == 0x12345678u) out[blockIdx.x] = 1.f;
}
__global__ void k_write(u32x4* dst, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = u32x4{(uint32_t)i, 1u, 2u, 3u};
}
__global__ void k_empty() {}

template <int NL, int SHAPE>
int run(const char* name, u32x4* buf, size_t nbuf, float* out, int grid, int share, int dirty) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t stride = (size_t)4 * NL * 64;
  const int iters = 200;
  // baseline: writer only (or empty)
  float base_ms = 0, ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) {
      if (dirty) hipLaunchKernelGGL(k_write, dim3(256), dim3(256), 0, 0, buf, stride * (share ? 8 : grid));
      else hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0);
      if (rep == 1) hipLaunchKernelGGL((k_read<NL, SHAPE>), dim3(grid), dim3(256), 0, 0, buf, out, stride, share);
    }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(rep ? &ms : &base_ms, e0, e1));
  }
  const double us = (ms - base_ms) * 1e3 / iters;
  const double kb = NL * 4.0;
  printf("%-10s grid %3d  %4.0f KiB/block  shape %d share %d dirty %d : %6.2f us  -> %6.1f GB/s per block, %6.2f TB/s chip\n",
         name, grid, kb, SHAPE, share, dirty, us, kb * 1024 / us / 1e3, kb * 1024 * grid / us / 1e6);
  return 0;
}
int main() {
  const size_t nbuf = (size_t)64 << 20;  // 1 GiB of u32x4? no: 64M x 16 B = 1 GiB
  u32x4* buf; float* out;
  CK(hipMalloc(&buf, nbuf * 16)); CK(hipMalloc(&out, 4096));
  hipLaunchKernelGGL(k_write, dim3(1024), dim3(256), 0, 0, buf, nbuf);
  CK(hipDeviceSynchronize());
  for (int dirty = 0; dirty < 2; ++dirty)
    for (int share = 0; share < 2; ++share)
      for (int grid : {64, 112, 224}) {
        run<8, 0>("8/wave", buf, nbuf, out, grid, share, dirty);
        run<16, 0>("16/wave", buf, nbuf, out, grid, share, dirty);
        run<40, 0>("40/wave", buf, nbuf, out, grid, share, dirty);
        run<40, 1>("40/wave", buf, nbuf, out, grid, share, dirty);
      }
  return 0;
}
