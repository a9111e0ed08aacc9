// CVaR over a reward-model ensemble (ref:1003-1011, ref:1185-1187):
//   out[c] = mean of the n_tail smallest of preds[0..S)[c]
// preds is the [S][N] prediction matrix the reference builds on the host (ref:978, 2 GB at
// S=500, N=1M) -- here it stays in HBM.  One work-group copies COLS >= 32 columns (128-byte row
// segments: whole HBM lines) into LDS as order-preserving integer keys, column-major, and L
// consecutive lanes share a column (L ~ S / 16: 2 lanes at S = 20, 32 lanes = 1024-thread
// work-groups at S = 500): the n_tail-th smallest key is found by counting passes (exact, no sort) in which
// every lane counts its quarter-rows with 16-byte LDS reads and the L counts meet in log2(L)
// shuffles (the probes: see below); ties at the threshold are counted, so the sum equals the
// partition-based mean up to fp32 summation order.  HBM traffic: 4 S bytes per column, read once, in COLS * 4 byte row
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

// Row k of column c lives at LDS word c SP + (k ^ swz(c)): the copy writes one row of 32 columns
// per half-wave, i.e. words SP apart -- with SP = 32 (mod 64) only TWO banks without the swizzle (a
// 16-way conflict on every store of the copy); the mask spreads them over 8 x 2 banks.  It is a
// multiple of 4 below 32: a 16-byte read of four consecutive rows stays one aligned 16-byte read, and
// a row never leaves its 32-word block (SP is a multiple of 32).
__device__ __forceinline__ int swz(int c) { return 4 * ((c >> 1) & 7); }

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
      keys[c * SP + (k ^ swz(c))] = (k < S && on) ? f2key(ldg(preds + (size_t)k * N + col0 + c)) : 0xffffffffu;
  }
  __syncthreads();
  const int c = tid / L, p = tid % L;
  const uint32_t *col = keys + c * SP;
  const int sw = swz(c);  // (k below is a multiple of 4: the four keys of a 16-byte read stay together)
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
        const uint4 v = *reinterpret_cast<const uint4 *>(col + (k ^ sw));
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
  // The n_tail-th smallest key t = the smallest t with #(keys <= t) >= n_tail, found exactly and
  // without sorting.  Invariant: t in [lo, hi], c_lo = #(keys < lo) < n_tail <= c_hi = #(keys <= hi).
  // One pass finds the column's minimum and maximum (the bracket); then counting passes shrink it,
  // the probe taken in turn where a linear interpolation of the counts between the bracket's VALUES
  // puts the rank, at the middle of the value range (ensemble predictions of one transition are a
  // smooth sample: a few passes bring the bracket down to a handful of elements) and at the middle
  // of the key range (the guarantee: at most 96 passes); a
  // bracket of <= BR elements is finished by taking its distinct minima one at a time: ~8 + 3 passes
  // for 25 of 500 normal samples instead of the 32 of a bisection over all 2^32 keys (round 2): 2.7 ->
  // 2.05 ms at S = 500, N = 1M.  What remains is the copy into LDS: 128-byte pieces of 500 rows 4 MB
  // apart (a TLB entry per row and work-group); requesting four rows per thread at once did not move it.
  constexpr int BR = 8;
  uint32_t lo, hi;
  {
    uint32_t mn = 0xffffffffu, mx = 0u;
    for (int k = 4 * p; k < S4; k += 4 * L) {
      const uint4 v = *reinterpret_cast<const uint4 *>(col + (k ^ sw));
      const uint32_t kk[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        mn = min(mn, kk[e]);
        if (k + e < S) mx = max(mx, kk[e]);  // (rows S .. hold the padding key 0xffffffff)
      }
    }
    lo = lane_min_u32<L>(mn), hi = ~lane_min_u32<L>(~mx);
  }
  int c_lo = 0, c_hi = S;
#pragma unroll 1
  for (int it = 0, ph = 0; it < 100 && lo < hi && c_hi - c_lo > BR; ++it, ph = ph == 2 ? 0 : ph + 1) {
    const uint32_t span = hi - lo;  // >= 1; the probe lies in [lo, hi - 1]
    uint32_t mid = lo + (span >> 1);  // ph == 2: the middle of the KEY range (the guarantee)
    if (ph < 2) {
      // ph == 0: where a linear interpolation of the counts between the bracket's VALUES puts the rank;
      // ph == 1: the middle of the VALUE range (one-sided interpolation steps stall on a convex
      // distribution tail; the keys of floats around zero are almost all of the key range and hold
      // almost none of the data)
      const float frac = ph == 0 ? ((float)(n_tail - c_lo) - 0.5f) / (float)(c_hi - c_lo) : 0.5f;
      const float lv = key2f(lo), hv = key2f(hi);
      const float mv = lv + (hv - lv) * frac;
      if (mv >= lv && mv <= hv) {  // (false for NaN / overflowing brackets: the key midpoint then)
        const uint32_t mk = f2key(mv);
        mid = mk < lo ? lo : (mk >= hi ? hi - 1u : mk);
      }
    }
    int cnt = 0;
    for (int k = 4 * p; k < S4; k += 4 * L) {
      const uint4 v = *reinterpret_cast<const uint4 *>(col + (k ^ sw));
      cnt += (v.x <= mid ? 1 : 0) + (v.y <= mid ? 1 : 0) + (v.z <= mid ? 1 : 0) + (v.w <= mid ? 1 : 0);
    }
    cnt = lane_sum<L>(cnt);
    if (cnt >= n_tail)
      hi = mid, c_hi = cnt;
    else
      lo = mid + 1u, c_lo = cnt;
  }
  if (lo < hi) {
    // a handful of elements in [lo, hi]: the (n_tail - c_lo)-th smallest of them, one distinct minimum
    // (with its multiplicity) per pass -- at most BR passes
    int need = n_tail - c_lo;  // >= 1
#pragma unroll 1
    for (int it = 0; it <= BR && need > 0; ++it) {
      uint32_t lm = 0xffffffffu;
      int lc = 0;
      for (int k = 4 * p; k < S4; k += 4 * L) {
        const uint4 v = *reinterpret_cast<const uint4 *>(col + (k ^ sw));
        const uint32_t kk[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (kk[e] >= lo) {
            lc = kk[e] < lm ? 1 : (kk[e] == lm ? lc + 1 : lc);
            lm = min(lm, kk[e]);
          }
        }
      }
      const uint32_t gm = lane_min_u32<L>(lm);
      const int gc = lane_sum<L>(lm == gm ? lc : 0);
      need -= gc;
      if (need > 0) lo = gm + 1u;  // (gm < hi here: the bracket still holds the rank)
      else lo = gm;
    }
  }
  const float thr = key2f(lo);
  float sum = 0.f;
  int less = 0;
  for (int k = 4 * p; k < S4; k += 4 * L) {
    const uint4 v = *reinterpret_cast<const uint4 *>(col + (k ^ sw));
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
