// Diagnostic: the ceiling of k_update's state stream.  A read-modify-write pass over the optimiser
// state of K seeds (params, exp_avg, exp_avg_sq [K P] fp32, target [K P_q] fp32; 42 MB at K = 8,
// resident in the 256 MB Infinity Cache between launches) with Adam-like arithmetic, a bf16 copy
// store, and NOTHING else: no GEMM, no LDS, no descriptors.  What this kernel cannot do, k_update
// cannot either.  Variants: float4s in flight per thread, work-groups per CU, grid size.
//   hipcc --offload-arch=gfx950 -O3 -o tools/membench3 tools/membench3.hip && tools/membench3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));
#define AS1 __attribute__((address_space(1)))

__device__ __forceinline__ f4 ld(const float* p) { return *(const f4 AS1*)p; }
__device__ __forceinline__ void st(float* p, f4 v) { *(f4 AS1*)p = v; }

// UN float4s of each array in flight per thread; TGT: also the target array (read + write); MATH:
// Adam arithmetic (IEEE sqrt and divisions, as adam_apply) or a plain add
template <int UN, bool TGT, bool MATH>
__global__ __launch_bounds__(256) void k_rmw(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                             float* __restrict__ t, uint16_t* __restrict__ c, size_t n4) {
  const size_t base = ((size_t)blockIdx.x * UN) * 256 + threadIdx.x;
  f4 a[UN], b[UN], d[UN], e[UN];
#pragma unroll
  for (int u = 0; u < UN; ++u) {
    const size_t i = base + (size_t)u * 256;
    const size_t j = i < n4 ? i : n4 - 1;
    a[u] = ld(p + 4 * j), b[u] = ld(m + 4 * j), d[u] = ld(v + 4 * j);
    if (TGT) e[u] = ld(t + 4 * j);
  }
#pragma unroll
  for (int u = 0; u < UN; ++u) {
    const size_t i = base + (size_t)u * 256;
    if (i >= n4) continue;
    f4 pn, mn, vn, tn;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (MATH) {
        const float g = a[u][k] * 1e-3f;
        mn[k] = b[u][k] + (g - b[u][k]) * 0.1f;
        vn[k] = d[u][k] * 0.999f + (0.001f * g) * g;
        const float den = sqrtf(vn[k]) / 0.0316f + 1e-8f;
        pn[k] = a[u][k] + -3e-4f * (mn[k] / den);
      } else {
        pn[k] = a[u][k] + 1.f, mn[k] = b[u][k] + 1.f, vn[k] = d[u][k] + 1.f;
      }
      if (TGT) tn[k] = e[u][k] + 0.005f * (pn[k] - e[u][k]);
    }
    st(p + 4 * i, pn), st(m + 4 * i, mn), st(v + 4 * i, vn);
    if (TGT) st(t + 4 * i, tn);
    u2 w;
    w.x = (__builtin_bit_cast(uint32_t, pn[0]) >> 16) | (__builtin_bit_cast(uint32_t, pn[1]) & 0xffff0000u);
    w.y = (__builtin_bit_cast(uint32_t, pn[2]) >> 16) | (__builtin_bit_cast(uint32_t, pn[3]) & 0xffff0000u);
    *(u2 AS1*)(c + 4 * i) = w;
  }
}

template <int UN, bool TGT, bool MATH>
int run(float* p, float* m, float* v, float* t, uint16_t* c, size_t n, const char* what) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t n4 = n / 4;
  const int grid = (int)((n4 + (size_t)UN * 256 - 1) / ((size_t)UN * 256));
  const int iters = 400;
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k_rmw<UN, TGT, MATH>), dim3(grid), dim3(256), 0, 0, p, m, v, t, c, n4);
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((k_rmw<UN, TGT, MATH>), dim3(grid), dim3(256), 0, 0, p, m, v, t, c, n4);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / iters;
  const double bytes = (double)n * 4 * (TGT ? 8 : 6) + (double)n * 2;
  printf("%-28s n=%8zu floats  UN=%d grid=%6d  %7.2f us / pass (launch to launch)  %.2f TB/s of %5.1f MB\n", what, n, UN, grid,
         us, bytes / us / 1e6, bytes / 1e6);
  return 0;
}

int main(int argc, char** argv) {
  const int K = argc > 1 ? atoi(argv[1]) : 8;
  const size_t P = 300819, n = ((size_t)K * P + 1023) / 1024 * 1024;
  float *p, *m, *v, *t; uint16_t* c;
  CK(hipMalloc(&p, n * 4)); CK(hipMalloc(&m, n * 4)); CK(hipMalloc(&v, n * 4)); CK(hipMalloc(&t, n * 4));
  CK(hipMalloc(&c, n * 2));
  CK(hipMemset(p, 0, n * 4)); CK(hipMemset(m, 0, n * 4)); CK(hipMemset(v, 0, n * 4)); CK(hipMemset(t, 0, n * 4));
  printf("K = %d seeds: %.1f MB of fp32 state per array\n", K, n * 4 / 1e6);
  // (the trivial-kernel launch-to-launch floor, for scale)
  run<1, false, false>(p, m, v, t, c, 1024, "launch floor (4 KB)");
  run<1, false, false>(p, m, v, t, c, n, "p m v, add");
  run<2, false, false>(p, m, v, t, c, n, "p m v, add");
  run<4, false, false>(p, m, v, t, c, n, "p m v, add");
  run<8, false, false>(p, m, v, t, c, n, "p m v, add");
  run<1, false, true>(p, m, v, t, c, n, "p m v, adam math");
  run<2, false, true>(p, m, v, t, c, n, "p m v, adam math");
  run<4, false, true>(p, m, v, t, c, n, "p m v, adam math");
  run<8, false, true>(p, m, v, t, c, n, "p m v, adam math");
  run<2, true, true>(p, m, v, t, c, n, "p m v t, adam + polyak");
  run<4, true, true>(p, m, v, t, c, n, "p m v t, adam + polyak");
  run<8, true, true>(p, m, v, t, c, n, "p m v t, adam + polyak");
  return 0;
}
