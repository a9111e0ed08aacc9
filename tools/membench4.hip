// Diagnostic: does the SHAPE of k_update's state accesses cost bandwidth?  The read-modify-write
// pass of tools/membench3 over [R][256] fp32 matrices (rows of 1 KB, as the H x H layers), a
// work-group owning a TR x TC tile: 64 x 32 (k_update: 128-byte row pieces, 1 KB apart),
// 32 x 64, 16 x 128, 8 x 256 (= contiguous 8 KB).  Same bytes, same arithmetic, 256 threads x two
// float4 of each of the four arrays.
//   hipcc --offload-arch=gfx950 -O3 -o tools/membench4 tools/membench4.hip && tools/membench4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));
#define AS1 __attribute__((address_space(1)))
__device__ __forceinline__ f4 ld(const float* p) { return *(const f4 AS1*)p; }
__device__ __forceinline__ void st(float* p, f4 v) { *(f4 AS1*)p = v; }
constexpr int W = 256;  // columns

template <int TR, int TC, bool WT>
__global__ __launch_bounds__(256) void k_tile(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                              float* __restrict__ t, uint16_t* __restrict__ c, int R) {
  constexpr int TPR = TC / 4, RPP = 256 / TPR, NP = TR / RPP;  // threads per row, rows per pass, passes
  static_assert(TR * TC == 2048 && NP >= 1, "2048 floats per tile");
  const int tiles_x = W / TC;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x % tiles_x;
  const int tr = threadIdx.x / TPR, tc = (threadIdx.x % TPR) * 4;
  f4 a[NP], b[NP], d[NP], e[NP];
#pragma unroll
  for (int u = 0; u < NP; ++u) {
    const size_t i = (size_t)(ty * TR + tr + RPP * u) * W + tx * TC + tc;
    a[u] = ld(p + i), b[u] = ld(m + i), d[u] = ld(v + i), e[u] = ld(t + i);
  }
#pragma unroll
  for (int u = 0; u < NP; ++u) {
    const size_t i = (size_t)(ty * TR + tr + RPP * u) * W + tx * TC + tc;
    f4 pn, mn, vn, tn;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float g = a[u][k] * 1e-3f;
      mn[k] = b[u][k] + (g - b[u][k]) * 0.1f;
      vn[k] = d[u][k] * 0.999f + (0.001f * g) * g;
      const float den = __builtin_amdgcn_sqrtf(vn[k]) * 31.6f + 1e-8f;
      pn[k] = a[u][k] + -3e-4f * (mn[k] * __builtin_amdgcn_rcpf(den));
      tn[k] = e[u][k] + 0.005f * (pn[k] - e[u][k]);
    }
    if (WT) {
      __builtin_nontemporal_store(pn, (f4 AS1*)(p + i));  // (placeholder: plain stores below are the default)
      st(m + i, mn), st(v + i, vn), st(t + i, tn);
    } else {
      st(p + i, pn), st(m + i, mn), st(v + i, vn), st(t + i, tn);
    }
    u2 w;
    w.x = (__builtin_bit_cast(uint32_t, pn[0]) >> 16) | (__builtin_bit_cast(uint32_t, pn[1]) & 0xffff0000u);
    w.y = (__builtin_bit_cast(uint32_t, pn[2]) >> 16) | (__builtin_bit_cast(uint32_t, pn[3]) & 0xffff0000u);
    *(u2 AS1*)(c + i) = w;
  }
}

template <int TR, int TC>
int run(float* p, float* m, float* v, float* t, uint16_t* c, int R) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = (R / TR) * (W / TC), iters = 400;
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k_tile<TR, TC, false>), dim3(grid), dim3(256), 0, 0, p, m, v, t, c, R);
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((k_tile<TR, TC, false>), dim3(grid), dim3(256), 0, 0, p, m, v, t, c, R);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / iters, bytes = (double)R * W * (4.0 * 8 + 2);
  printf("tile %3d x %3d  grid %5d  %7.2f us / pass (launch to launch)  %.2f TB/s of %5.1f MB\n", TR, TC, grid, us,
         bytes / us / 1e6, bytes / 1e6);
  return 0;
}

int main(int argc, char** argv) {
  const int K = argc > 1 ? atoi(argv[1]) : 8;
  const int R = K * 1024;  // K seeds x (4 nets x 256 rows) of H x H layer state: 8 MB per array at K = 8
  const size_t n = (size_t)R * W;
  float *p, *m, *v, *t; uint16_t* c;
  CK(hipMalloc(&p, n * 4)); CK(hipMalloc(&m, n * 4)); CK(hipMalloc(&v, n * 4)); CK(hipMalloc(&t, n * 4));
  CK(hipMalloc(&c, n * 2));
  CK(hipMemset(p, 0, n * 4)); CK(hipMemset(m, 0, n * 4)); CK(hipMemset(v, 0, n * 4)); CK(hipMemset(t, 0, n * 4));
  printf("K = %d: [%d][256] fp32 per array (%.1f MB)\n", K, R, n * 4 / 1e6);
  for (int rep = 0; rep < 2; ++rep) {
    if (run<64, 32>(p, m, v, t, c, R)) return 1;
    if (run<32, 64>(p, m, v, t, c, R)) return 1;
    if (run<16, 128>(p, m, v, t, c, R)) return 1;
    if (run<8, 256>(p, m, v, t, c, R)) return 1;
  }
  return 0;
}
