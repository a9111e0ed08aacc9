// CVaR over a reward-model ensemble (ref:1003-1011, ref:1185-1187):
//   out[c] = mean of the n_tail smallest of preds[0..S)[c]
// preds is the [S][N] prediction matrix the reference builds on the host
// (ref:978, 2 GB at S=500, N=1M) -- here it stays in HBM and each wave selects
// the tail of 64 columns from an LDS copy: loads are 256-byte coalesced row
// segments, the column lives in one LDS bank per lane (conflict free), the k-th
// smallest value is found by a 32-step bisection on the order-preserving integer
// image of the floats (exact, no sort), and ties at the threshold are counted so
// that the sum equals the partition-based mean up to fp32 summation order.
#include "../../include/iqlhip.h"
#include "common.h"

namespace iqlhip {

__device__ __forceinline__ uint32_t f2key(float f) {
  const uint32_t u = __builtin_bit_cast(uint32_t, f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k) {
  const uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __builtin_bit_cast(float, u);
}

template <int COLS>
__global__ __launch_bounds__(64) void k_cvar(const float *__restrict__ preds, int S, int64_t N, int n_tail,
                                             float *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint32_t *keys = reinterpret_cast<uint32_t *>(smem);  // [S][COLS]
  const int lane = threadIdx.x;
  const int64_t col0 = (int64_t)blockIdx.x * COLS;
  const bool on = lane < COLS && col0 + lane < N;
  for (int k = 0; k < S; ++k)
    if (lane < COLS) keys[k * COLS + lane] = on ? f2key(preds[(size_t)k * N + col0 + lane]) : 0u;
  __syncthreads();
  if (!on) return;
  // smallest key t such that #(keys <= t) >= n_tail  == the n_tail-th smallest key
  uint32_t lo = 0u, hi = 0xffffffffu;
  while (lo < hi) {
    const uint32_t mid = lo + ((hi - lo) >> 1);
    int cnt = 0;
    for (int k = 0; k < S; ++k) cnt += keys[k * COLS + lane] <= mid ? 1 : 0;
    if (cnt >= n_tail)
      hi = mid;
    else
      lo = mid + 1u;
  }
  const float thr = key2f(lo);
  float sum = 0.f;
  int less = 0;
  for (int k = 0; k < S; ++k) {
    const uint32_t kk = keys[k * COLS + lane];
    if (kk < lo) {
      sum += key2f(kk);
      ++less;
    }
  }
  sum += (float)(n_tail - less) * thr;
  out[col0 + lane] = sum / (float)n_tail;
}

hipError_t launch_cvar(const float *preds, int S, int64_t N, int n_tail, float *out, hipStream_t st) {
  // 64 columns per wave while the LDS copy fits (S <= 600), then 32, then 16
  const int cols = S <= 600 ? 64 : (S <= 1200 ? 32 : 16);
  const size_t sm = (size_t)S * cols * sizeof(uint32_t);
  const int64_t grid = (N + cols - 1) / cols;
  hipError_t e = hipSuccess;
  if (cols == 64) {
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_cvar<64>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_cvar<64>, dim3((unsigned)grid), dim3(64), sm, st, preds, S, N, n_tail, out);
  } else if (cols == 32) {
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_cvar<32>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_cvar<32>, dim3((unsigned)grid), dim3(64), sm, st, preds, S, N, n_tail, out);
  } else {
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_cvar<16>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_cvar<16>, dim3((unsigned)grid), dim3(64), sm, st, preds, S, N, n_tail, out);
  }
  return hipGetLastError();
}

}  // namespace iqlhip
