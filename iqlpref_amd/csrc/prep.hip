// Device-side dataset preparation: what the reference does in Python loops over
// the N transitions before training starts --
//   keep mask + episode step       ref:701-716 (= ref:1236-1253, 1127-1141, 938-951)
//   per-episode return range       ref:344-360  return_reward_range
//   reward normalisation           ref:363-401  modify_reward
//   state mean / std, z-scoring    ref:132-139, 1438-1448
// -- as HBM-bound integer / float kernels.  Everything here is one pass over N elements
// (N ~ 1e6): the kernels are simple and coalesced, none is worth more than that.
//
// Building block: an exclusive "index of the last flagged element" scan,
//   last[i] = max{ j < i : flag[j] }  (or -1),
// in two launches (per-chunk last index; then look-back over the chunk table + an in-chunk
// scan).  The episode-step counter and the episode segmentation are closed forms of it.
#include <cstring>

#include "../../include/iqlhip.h"
#include "common.h"

namespace iqlhip {

constexpr int CHUNK = 1024;  // elements per work-group of the scan (256 threads x 4)

// ---- flags -----------------------------------------------------------------
// Counter reset AFTER transition i (ref:708-716 with a timeouts array):
//   final and not terminate_on_end -> dropped, counter := 0          (kind 1)
//   else terminal or final          -> counter := 0 then += 1 -> 1   (kind 2)
//   else                            -> counter += 1                  (kind 0)
__device__ __forceinline__ int reset_kind(const uint8_t *term, const uint8_t *tmo, int64_t i, int toe) {
  const bool fin = tmo[i] != 0, done = term[i] != 0;
  if (fin && !toe) return 1;
  return (done || fin) ? 2 : 0;
}

template <int MODE>  // 0: reset flags of (terminals, timeouts); 1: terminals alone
__device__ __forceinline__ bool flag_at(const uint8_t *term, const uint8_t *tmo, int64_t i, int toe) {
  if constexpr (MODE == 0)
    return reset_kind(term, tmo, i, toe) != 0;
  else
    return term[i] != 0;
}

template <int MODE>
__global__ __launch_bounds__(256) void k_chunk_last(const uint8_t *__restrict__ term,
                                                    const uint8_t *__restrict__ tmo, int64_t n, int toe,
                                                    int64_t *__restrict__ chunk_last) {
  __shared__ long long red[4];
  const int64_t base = (int64_t)blockIdx.x * CHUNK + threadIdx.x * 4;
  long long best = -1;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (base + k < n && flag_at<MODE>(term, tmo, base + k, toe)) best = base + k;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const long long o = __shfl_xor(best, off);
    best = o > best ? o : best;
  }
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    long long b = red[0];
    for (int w = 1; w < 4; ++w) b = red[w] > b ? red[w] : b;
    chunk_last[blockIdx.x] = b;
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void k_last_scan(const uint8_t *__restrict__ term,
                                                   const uint8_t *__restrict__ tmo, int64_t n, int toe,
                                                   const int64_t *__restrict__ chunk_last,
                                                   int64_t *__restrict__ last) {
  __shared__ long long carry_s;
  __shared__ long long wave_last[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) {  // look back over the chunk table (resets are at most an episode apart)
    long long c = -1;
    for (int64_t b = (int64_t)blockIdx.x - 1; b >= 0 && c < 0; --b) c = chunk_last[b];
    carry_s = c;
  }
  const int64_t base = (int64_t)blockIdx.x * CHUNK + tid * 4;
  long long own[4], run = -1;  // own[k]: last flagged index among this thread's elements < k
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    own[k] = run;
    if (base + k < n && flag_at<MODE>(term, tmo, base + k, toe)) run = base + k;
  }
  // inclusive max-scan of `run` over the wave, then over the 4 waves
  long long inc = run;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const long long o = __shfl_up(inc, off);
    if (lane >= off) inc = o > inc ? o : inc;
  }
  if (lane == 63) wave_last[wave] = inc;
  __syncthreads();
  long long before = carry_s;  // everything before this thread: carry, earlier waves, earlier lanes
  for (int w = 0; w < wave; ++w) before = wave_last[w] > before ? wave_last[w] : before;
  const long long prev_lane = __shfl_up(inc, 1);
  if (lane > 0) before = prev_lane > before ? prev_lane : before;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (base + k < n) last[base + k] = own[k] > before ? own[k] : before;
}

// ---- keep mask / episode step, timeouts present ---------------------------------
// ep_steps[i] = value of the counter when transition i is visited; `last` = index of the last
// reset before i.  (tests: bit-exact against the reference loop)
__global__ void k_keep_steps(const uint8_t *__restrict__ term, const uint8_t *__restrict__ tmo, int64_t n1,
                             int toe, const int64_t *__restrict__ last, uint8_t *__restrict__ keep,
                             int64_t *__restrict__ ep_steps) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n1) return;
  const int64_t l = last[i];
  int64_t ep;
  if (l < 0)
    ep = i;  // never reset: one increment per earlier transition
  else
    ep = (i - 1 - l) + (reset_kind(term, tmo, l, toe) == 2 ? 1 : 0);
  ep_steps[i] = ep;
  keep[i] = (tmo[i] != 0 && !toe) ? 0 : 1;
}

// ---- keep mask / episode step, no timeouts: final = (counter == M - 1) -------------
// The counter feeds back into the reset condition, so the scan above does not apply.  One wave
// walks the array 64 transitions at a time: inside a terminal-free run that starts at index a
// with counter c the value at a + t is a closed form (ramp, then a sawtooth of period M - base);
// terminals (found with a ballot) start a new run.  This path is the fallback for datasets
// without a timeouts array (every D4RL v2 file has one).
__global__ __launch_bounds__(64) void k_keep_steps_counter(const uint8_t *__restrict__ term, int64_t n1,
                                                          int64_t M, int toe, uint8_t *__restrict__ keep,
                                                          int64_t *__restrict__ ep_steps) {
  const int lane = threadIdx.x;
  const int64_t base = toe ? 1 : 0;  // counter after a `final` transition
  int64_t a = 0, c = 0;              // current run: starts at index a with counter c
  auto value = [&](int64_t i) -> int64_t {
    const int64_t t = i - a;
    if (c > M - 1) return c + t;               // already past M - 1: never final again in this run
    const int64_t f0 = M - 1 - c;              // offset of the run's first final transition
    if (t <= f0) return c + t;
    if (base <= M - 1) return (t - f0 - 1) % (M - base) + base;
    return base + (t - f0 - 1);                // M == 1 with terminate_on_end
  };
  for (int64_t p = 0; p < n1; p += 64) {
    const int64_t i = p + lane;
    const bool in = i < n1;
    const bool tm = in && term[i] != 0;
    unsigned long long pending = __ballot(tm);
    int64_t v = in ? value(i) : 0;
    while (pending) {  // terminals of this group, in order: each one ends the current run
      const int L = __ffsll((long long)pending) - 1;
      pending &= pending - 1;
      const int64_t vt = __shfl(v, L);  // counter at the terminal (wave-uniform)
      a = p + L + 1;
      c = (vt == M - 1) ? base : 1;     // final there -> base, else a kept terminal -> 1
      if (lane > L && in) v = value(i);
    }
    if (in) {
      ep_steps[i] = v;
      keep[i] = (v == M - 1 && !toe) ? 0 : 1;
    }
  }
}

// ---- per-episode returns (ref:344-360) ------------------------------------------
// An episode ends at a terminal or after M transitions (counted from the previous episode's
// end).  `last` = index of the last terminal before i, so i starts an episode iff
// (i - last - 1) % M == 0.  The thread of every episode start walks its episode in order and
// adds float(r) in double exactly as the reference loop does -> the sums are bit-identical.
// trj_lens[i] = length of i's episode (the trailing partial episode: its length so far).
__device__ __forceinline__ unsigned long long dbl_key(double x) {  // order-preserving integer image
  const unsigned long long u = (unsigned long long)__double_as_longlong(x);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__global__ void k_episode_walk(const float *__restrict__ rew, const uint8_t *__restrict__ term, int64_t n,
                               int64_t M, const int64_t *__restrict__ last, double *__restrict__ trj_lens,
                               unsigned long long *__restrict__ minmax_key, int64_t *__restrict__ n_complete) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if ((i - last[i] - 1) % M != 0) return;
  double ret = 0.0;
  int64_t j = i, len = 0;
  bool complete = false;
  while (j < n) {
    ret += (double)rew[j];
    ++len;
    const bool d = term[j] != 0;
    ++j;
    if (d || len == M) {
      complete = true;
      break;
    }
  }
  for (int64_t k = i; k < j; ++k) trj_lens[k] = (double)len;
  if (complete) {
    const unsigned long long key = dbl_key(ret);
    atomicMin(minmax_key, key);  // min / max are order independent: deterministic
    atomicMax(minmax_key + 1, key);
    atomicAdd((unsigned long long *)n_complete, 1ull);
  }
}

// ---- reward normalisation (ref:363-401), numpy's in-place float32 semantics --------
//   rewards op= python scalar  -> the scalar is rounded to float32, the op runs in float32
//   rewards -= float64 array   -> the op runs in float64, the result is rounded to float32
struct RewardOps {
  int32_t sub_first;    // 0 none, 1 `-= min_ret` (float32), 2 `-= min_ret / trj_lens` (float64)
  int32_t scale;        // `/= (max_ret - min_ret)` then `*= max_episode_steps`
  int32_t sub_one;      // `-= 1.0`
  double min_ret, range;
  float steps;
};
__global__ void k_modify_reward(float *__restrict__ rew, int64_t n, const double *__restrict__ trj_lens,
                                RewardOps op) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float r = rew[i];
  if (op.sub_first == 1) r = r - (float)op.min_ret;
  if (op.sub_first == 2) r = (float)((double)r - op.min_ret / trj_lens[i]);
  if (op.scale) {
    r = r / (float)op.range;
    r = r * op.steps;
  }
  if (op.sub_one) r = r - 1.0f;
  rew[i] = r;
}

// ---- state statistics (ref:132-135): column mean and std in double, fixed order ------
// pass 1: per-block column sums; pass 2 (host-launched again with the mean): squared deviations.
// Partial sums of a column are added in block order by one thread: run-to-run deterministic.
constexpr int STAT_ROWS = 2048;  // rows per block
__global__ __launch_bounds__(256) void k_col_partial(const float *__restrict__ x, int64_t n, int S,
                                                     const double *__restrict__ mean,  // null: plain sums
                                                     double *__restrict__ partial) {
  // thread t handles column t % S for rows r0 + t / S, stepping by 256 / S rows (S <= 256)
  const int per = 256 / S;
  const int col = threadIdx.x % S, sub = threadIdx.x / S;
  __shared__ double acc[256];
  double s = 0.0;
  if (sub < per) {
    const int64_t r0 = (int64_t)blockIdx.x * STAT_ROWS;
    const int64_t r1 = r0 + STAT_ROWS < n ? r0 + STAT_ROWS : n;
    const double m = mean ? mean[col] : 0.0;
    for (int64_t r = r0 + sub; r < r1; r += per) {
      const double v = (double)x[r * S + col] - m;
      s += mean ? v * v : v;
    }
  }
  acc[threadIdx.x] = s;
  __syncthreads();
  if (sub == 0) {
    double t = 0.0;
    for (int k = 0; k < per; ++k) t += acc[k * S + col];
    partial[(size_t)blockIdx.x * S + col] = t;
  }
}
__global__ void k_col_final(const double *__restrict__ partial, int nblk, int S, int64_t n, int pass,
                            double eps, double *__restrict__ mean, float *__restrict__ mean_f,
                            float *__restrict__ std_f) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= S) return;
  double t = 0.0;
  for (int b = 0; b < nblk; ++b) t += partial[(size_t)b * S + col];
  if (pass == 0) {
    mean[col] = t / (double)n;
    mean_f[col] = (float)(t / (double)n);
  } else {
    std_f[col] = (float)(sqrt(t / (double)n) + eps);  // states.std(0) + eps
  }
}

// ---- pack with fused z-scoring (ref:1438-1456): rows = [ (s-m)/sd | a | r | d | (s'-m)/sd ] ---
__global__ void k_pack_norm(float *__restrict__ rows, int stride, int S, int A, int64_t first, int64_t n,
                            const float *__restrict__ obs, const float *__restrict__ act,
                            const float *__restrict__ rew, const float *__restrict__ nxt,
                            const float *__restrict__ done, const float *__restrict__ mean,
                            const float *__restrict__ sd) {
  const int NO = round_up(S + A + 2, 4), W = NO + S;  // s' starts on a 16-byte boundary
  const int64_t total = n * (int64_t)stride;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = e / stride;
    const int c = (int)(e - row * stride);
    float v = 0.f;
    if (c < S)
      v = (obs[row * S + c] - mean[c]) / sd[c];
    else if (c < S + A)
      v = act[row * A + (c - S)];
    else if (c == S + A)
      v = rew[row];
    else if (c == S + A + 1)
      v = done[row];
    else if (c >= NO && c < W)
      v = (nxt[row * S + (c - NO)] - mean[c - NO]) / sd[c - NO];
    rows[(first + row) * stride + c] = v;
  }
}

// ------------------------------------------------------------------ launchers --
template <int MODE>
static hipError_t last_flag_scan(const uint8_t *term, const uint8_t *tmo, int64_t n, int toe, int64_t *last,
                                 hipStream_t st) {
  const int nchunk = (int)((n + CHUNK - 1) / CHUNK);
  int64_t *chunk_last = nullptr;
  hipError_t e = hipMallocAsync((void **)&chunk_last, sizeof(int64_t) * nchunk, st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_chunk_last<MODE>, dim3(nchunk), dim3(256), 0, st, term, tmo, n, toe, chunk_last);
  hipLaunchKernelGGL(k_last_scan<MODE>, dim3(nchunk), dim3(256), 0, st, term, tmo, n, toe, chunk_last, last);
  e = hipGetLastError();
  (void)hipFreeAsync(chunk_last, st);
  return e;
}

hipError_t launch_keep_steps(const uint8_t *term, const uint8_t *tmo, int64_t n, int64_t M, int toe,
                             uint8_t *keep, int64_t *ep_steps, hipStream_t st) {
  const int64_t n1 = n - 1;
  if (n1 <= 0) return hipSuccess;
  if (!tmo) {
    hipLaunchKernelGGL(k_keep_steps_counter, dim3(1), dim3(64), 0, st, term, n1, M, toe, keep, ep_steps);
    return hipGetLastError();
  }
  int64_t *last = nullptr;
  hipError_t e = hipMallocAsync((void **)&last, sizeof(int64_t) * n1, st);
  if (e != hipSuccess) return e;
  e = last_flag_scan<0>(term, tmo, n1, toe, last, st);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_keep_steps, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, st, term, tmo, n1, toe,
                       last, keep, ep_steps);
    e = hipGetLastError();
  }
  (void)hipFreeAsync(last, st);
  return e;
}

// out[0] = min return, out[1] = max return (doubles), out[2] = number of complete episodes
hipError_t launch_reward_range(const float *rew, const uint8_t *term, int64_t n, int64_t M, double *trj_lens,
                               double *out3, hipStream_t st) {
  int64_t *last = nullptr;
  unsigned long long *keys = nullptr;  // [0] min key, [1] max key, [2] count
  hipError_t e = hipMallocAsync((void **)&last, sizeof(int64_t) * n, st);
  if (e != hipSuccess) return e;
  e = hipMallocAsync((void **)&keys, 3 * sizeof(unsigned long long), st);
  if (e != hipSuccess) return e;
  const unsigned long long init[3] = {~0ull, 0ull, 0ull};
  e = hipMemcpyAsync(keys, init, sizeof(init), hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = last_flag_scan<1>(term, nullptr, n, 0, last, st);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_episode_walk, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, rew, term, n, M,
                       last, trj_lens, keys, reinterpret_cast<int64_t *>(keys + 2));
    e = hipGetLastError();
  }
  unsigned long long h[3] = {0, 0, 0};
  if (e == hipSuccess) e = hipMemcpyAsync(h, keys, sizeof(h), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)hipFreeAsync(last, st);
  (void)hipFreeAsync(keys, st);
  if (e != hipSuccess) return e;
  auto unkey = [](unsigned long long k) {
    const unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    double d;
    memcpy(&d, &u, 8);
    return d;
  };
  out3[0] = unkey(h[0]), out3[1] = unkey(h[1]), out3[2] = (double)h[2];
  return hipSuccess;
}

hipError_t launch_modify_reward(float *rew, int64_t n, const double *trj_lens, int sub_first, int scale,
                                int sub_one, double min_ret, double range, float steps, hipStream_t st) {
  RewardOps op;
  op.sub_first = sub_first, op.scale = scale, op.sub_one = sub_one;
  op.min_ret = min_ret, op.range = range, op.steps = steps;
  hipLaunchKernelGGL(k_modify_reward, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, rew, n, trj_lens, op);
  return hipGetLastError();
}

hipError_t launch_state_stats(const float *x, int64_t n, int S, double eps, float *mean_f, float *std_f,
                              hipStream_t st) {
  const int nblk = (int)((n + STAT_ROWS - 1) / STAT_ROWS);
  double *partial = nullptr, *mean = nullptr;
  hipError_t e = hipMallocAsync((void **)&partial, sizeof(double) * (size_t)nblk * S, st);
  if (e != hipSuccess) return e;
  e = hipMallocAsync((void **)&mean, sizeof(double) * S, st);
  if (e != hipSuccess) return e;
  for (int pass = 0; pass < 2; ++pass) {
    hipLaunchKernelGGL(k_col_partial, dim3(nblk), dim3(256), 0, st, x, n, S, pass ? mean : nullptr, partial);
    hipLaunchKernelGGL(k_col_final, dim3((S + 63) / 64), dim3(64), 0, st, partial, nblk, S, n, pass, eps, mean,
                       mean_f, std_f);
  }
  e = hipGetLastError();
  (void)hipFreeAsync(partial, st);
  (void)hipFreeAsync(mean, st);
  return e;
}

hipError_t launch_pack_norm(float *rows, int stride, int S, int A, int64_t first, int64_t n, const float *obs,
                            const float *act, const float *rew, const float *nxt, const float *done,
                            const float *mean, const float *sd, hipStream_t st) {
  const int64_t total = n * (int64_t)stride;
  int grid = (int)((total + 255) / 256);
  if (grid > 256 * 8) grid = 256 * 8;
  hipLaunchKernelGGL(k_pack_norm, dim3(grid), dim3(256), 0, st, rows, stride, S, A, first, n, obs, act, rew, nxt,
                     done, mean, sd);
  return hipGetLastError();
}

}  // namespace iqlhip
