// Diagnostic: how fast can a kernel read data that the PREVIOUS kernel wrote, as a
// function of the producer's store flavour (plain / nontemporal / sc1 write-through)?
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define AS1 __attribute__((address_space(1)))

template <int MODE>
__global__ void k_write(u32x4* dst, size_t n, uint32_t tag) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    u32x4 v = u32x4{(uint32_t)i, tag, 2u, 3u};
    if (MODE == 0) dst[i] = v;
    else if (MODE == 1) __builtin_nontemporal_store(v, dst + i);
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"((u32x4 AS1*)(dst + i)), "v"(v) : "memory");
  }
}
template <int NL>
__global__ __launch_bounds__(256) void k_read(const u32x4* __restrict__ src, float* out, size_t block_stride_v4) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const u32x4* base = src + blockIdx.x * block_stride_v4 + (size_t)wave * NL * 64;
  u32x4 v[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) v[i] = base[(size_t)i * 64 + lane];
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < NL; ++i) s += v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
  if (s == 0x12345678u) out[blockIdx.x] = 1.f;
}
template <int MODE, int NL>
int run(u32x4* buf, float* out, int grid, int wgrid) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t stride = (size_t)4 * NL * 64, n = stride * grid;
  const int iters = 300;
  float ms[2];
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) {
      hipLaunchKernelGGL(k_write<MODE>, dim3(wgrid), dim3(256), 0, 0, buf, n, (uint32_t)i);
      if (rep == 1) hipLaunchKernelGGL((k_read<NL>), dim3(grid), dim3(256), 0, 0, buf, out, stride);
    }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms[rep], e0, e1));
  }
  const double mb = NL * 4.0 * grid / 1024.0;
  printf("store mode %d (0 plain,1 nt,2 sc0sc1)  %5.1f MB by %3d writers, %3d readers x %3d KiB: write %.2f us, +read %.2f us (%.2f TB/s)\n",
         MODE, mb, wgrid, grid, NL * 4, ms[0] * 1e3 / iters, (ms[1] - ms[0]) * 1e3 / iters,
         mb / 1024 / 1024 * 1e6 / ((ms[1] - ms[0]) * 1e3 / iters) / 1e0 * 1.048576);
  return 0;
}
int main() {
  u32x4* buf; float* out;
  CK(hipMalloc(&buf, (size_t)256 << 20)); CK(hipMalloc(&out, 4096));
  for (int wg : {64, 192}) {
    run<0, 16>(buf, out, 192, wg); run<1, 16>(buf, out, 192, wg); run<2, 16>(buf, out, 192, wg);
    run<0, 40>(buf, out, 112, wg); run<1, 40>(buf, out, 112, wg); run<2, 40>(buf, out, 112, wg);
  }
  return 0;
}
