// CVaR over a reward-model ensemble (ref:1003-1011, ref:1185-1187):
//   out[c] = mean of the n_tail smallest of preds[0..S)[c]
// preds is the [S][N] prediction matrix the reference builds on the host (ref:978, 2 GB at
// S=500, N=1M) -- here it stays in HBM.  One work-group copies COLS >= 32 columns (128-byte row
// segments: whole HBM lines) into LDS as order-preserving integer keys, column-major, and L
// consecutive lanes share a column (L ~ S / 16: 2 lanes at S = 20, 32 lanes = 1024-thread
// work-groups at S = 500): the n_tail-th smallest key is found by a 32-step bisection (exact, no sort) in which
// every lane counts its quarter-rows with 16-byte LDS reads and the L counts meet in log2(L)
// shuffles; ties at the threshold are counted, so the sum equals the partition-based mean up to
// fp32 summation order.  HBM traffic: 4 S bytes per column, read once, in COLS * 4 byte row
// segments.  LDS rows are padded so that 16-byte reads of neighbouring columns hit disjoint banks.
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

// minimum over aligned groups of W consecutive lanes (W a power of two <= 32), result in every lane
template <int W>
__device__ __forceinline__ uint32_t lane_min_u32(uint32_t v) {
  if constexpr (W >= 2) v = min(v, dpp_mov<0xB1>(v));
  if constexpr (W >= 4) v = min(v, dpp_mov<0x4E>(v));
  if constexpr (W >= 8) v = min(v, dpp_mov<0x141>(v));
  if constexpr (W >= 16) v = min(v, dpp_mov<0x140>(v));
  if constexpr (W >= 32) {
    uint32_t a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    v = min(a, b);
  }
  return v;
}

constexpr int SMALL_TAIL = 8;  // up to this many tail elements are taken one distinct minimum at a time

// COLS columns per work-group, L consecutive lanes per column (COLS * L threads)
template <int COLS, int L>
__global__ __launch_bounds__(COLS *L) void k_cvar(const float *__restrict__ preds, int S, int64_t N,
                                                   int n_tail, int SP, float *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint32_t *keys = reinterpret_cast<uint32_t *>(smem);  // [COLS][SP], rows >= S hold 0xffffffff
  const int tid = threadIdx.x;
  const int64_t col0 = (int64_t)blockIdx.x * COLS;
  {
    const int c = tid % COLS;
    const bool on = col0 + c < N;
    for (int k = tid / COLS; k < SP; k += L)  // (COLS * L threads: L rows per pass)
      keys[c * SP + k] = (k < S && on) ? f2key(ldg(preds + (size_t)k * N + col0 + c)) : 0xffffffffu;
  }
  __syncthreads();
  const int c = tid / L, p = tid % L;
  const uint32_t *col = keys + c * SP;
  const int S4 = round_up(S, 4);  // (keys beyond S compare greater than every threshold below 2^32 - 1)
  if (n_tail <= SMALL_TAIL) {
    // ---- short tails (n_tail = 1 is the snapshot ensemble of ref:1152 at S = 20: the minimum):
    // at most n_tail passes, each taking the next distinct minimum with its multiplicity --
    // n_tail x S compares instead of the 32 x S of the bisection below ----
    uint32_t prev = 0u;
    int taken = 0;
    float sum = 0.f;
#pragma unroll 1
    for (int it = 0; it < n_tail; ++it) {
      uint32_t lm = 0xffffffffu;
      int lc = 0;
      for (int k = 4 * p; k < S4; k += 4 * L) {
        const uint4 v = *reinterpret_cast<const uint4 *>(col + k);
        const uint32_t kk[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (it == 0 || kk[e] > prev) {
            lc = kk[e] < lm ? 1 : (kk[e] == lm ? lc + 1 : lc);
            lm = min(lm, kk[e]);
          }
        }
      }
      const uint32_t gm = lane_min_u32<L>(lm);
      const int gc = lane_sum<L>(lm == gm ? lc : 0);
      const int take = min(gc, n_tail - taken);  // 0 once the tail is complete
      if (take > 0) sum += (float)take * key2f(gm);
      taken += take, prev = gm;
    }
    if (p == 0 && col0 + c < N) out[col0 + c] = sum / (float)n_tail;
    return;
  }
  // smallest key t such that #(keys <= t) >= n_tail  == the n_tail-th smallest key
  uint32_t lo = 0u, hi = 0xffffffffu;
#pragma unroll 1
  for (int it = 0; it < 32; ++it) {
    const uint32_t mid = lo + ((hi - lo) >> 1);
    int cnt = 0;
    for (int k = 4 * p; k < S4; k += 4 * L) {
      const uint4 v = *reinterpret_cast<const uint4 *>(col + k);
      cnt += (v.x <= mid ? 1 : 0) + (v.y <= mid ? 1 : 0) + (v.z <= mid ? 1 : 0) + (v.w <= mid ? 1 : 0);
    }
    cnt = lane_sum<L>(cnt);
    if (lo < hi) {
      if (cnt >= n_tail)
        hi = mid;
      else
        lo = mid + 1u;
    }
  }
  const float thr = key2f(lo);
  float sum = 0.f;
  int less = 0;
  for (int k = 4 * p; k < S4; k += 4 * L) {
    const uint4 v = *reinterpret_cast<const uint4 *>(col + k);
    const uint32_t kk[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (kk[e] < lo) sum += key2f(kk[e]), ++less;
  }
  sum = lane_sum<L>(sum), less = lane_sum<L>(less);
  if (p == 0 && col0 + c < N) out[col0 + c] = (sum + (float)(n_tail - less) * thr) / (float)n_tail;
}

// LDS row stride (words): >= S rounded to 4, and = 32 mod 64, so that the 16 lanes one 16-byte LDS
// read serves at a time (two columns at L = 8) touch 64 distinct banks
static int cvar_row_stride(int S) {
  int sp = round_up(S, 4);
  while (sp % 64 != 32) sp += 4;
  return sp;
}

hipError_t launch_cvar(const float *preds, int S, int64_t N, int n_tail, float *out, hipStream_t st) {
  const int SP = cvar_row_stride(S);
  int L = 2;  // lanes per column: about one per 16 rows
  while (L < 32 && L * 16 < S) L *= 2;
#define CVAR_LAUNCH(C, LL)                                                                                   \
  do {                                                                                                       \
    const size_t sm = (size_t)(C) * SP * sizeof(uint32_t);                                                   \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_cvar<C, LL>),                         \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);               \
    if (e != hipSuccess) return e;                                                                           \
    hipLaunchKernelGGL((k_cvar<C, LL>), dim3((unsigned)((N + (C)-1) / (C))), dim3((C) * (LL)), sm, st, preds, S, \
                       N, n_tail, SP, out);                                                                  \
  } while (0)
  if (L == 2)
    CVAR_LAUNCH(128, 2);
  else if (L == 4)
    CVAR_LAUNCH(64, 4);
  else if (L == 8)
    CVAR_LAUNCH(32, 8);
  else if (L == 16)
    CVAR_LAUNCH(32, 16);
  else if ((size_t)32 * SP * 4 <= 160 * 1024)
    CVAR_LAUNCH(32, 32);
  else  // S > ~1200: the LDS image of 32 columns no longer fits, 64-byte row segments
    CVAR_LAUNCH(16, 32);
#undef CVAR_LAUNCH
  return hipGetLastError();
}

}  // namespace iqlhip
