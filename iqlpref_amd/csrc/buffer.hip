// Replay buffer kernels: interleave the five dataset arrays into packed rows
// (ref:193-209 load_d4rl_dataset) and gather a batch (ref:211-221 sample).
// Pure HBM work: a sampled transition is ONE contiguous 16-byte aligned row
// [s|a|r|d|s'] read by consecutive lanes instead of five scattered reads.
#include "../../include/iqlhip.h"
#include "common.h"

namespace iqlhip {

__global__ void k_pack(float *__restrict__ rows, int stride, int S, int A, int64_t first, int64_t n,
                       const float *__restrict__ obs, const float *__restrict__ act,
                       const float *__restrict__ rew, const float *__restrict__ nxt,
                       const float *__restrict__ done) {
  const int NO = round_up(S + A + 2, 4), W = NO + S;  // s' starts on a 16-byte boundary
  const int64_t total = n * (int64_t)stride;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = e / stride;
    const int c = (int)(e - row * stride);
    float v = 0.f;
    if (c < S)
      v = obs[row * S + c];
    else if (c < S + A)
      v = act[row * A + (c - S)];
    else if (c == S + A)
      v = rew[row];
    else if (c == S + A + 1)
      v = done[row];
    else if (c >= NO && c < W)
      v = nxt[row * S + (c - NO)];
    rows[(first + row) * stride + c] = v;
  }
}

// One wave per sampled row: the row is read with coalesced dword loads and
// scattered to the five dense outputs the reference API returns.
__global__ __launch_bounds__(256) void k_sample(const float *__restrict__ rows, int64_t n_rows, int stride,
                                                int S, int A, int batch, const int64_t *__restrict__ idx,
                                                uint64_t seed, uint64_t step, float *__restrict__ s,
                                                float *__restrict__ a, float *__restrict__ r,
                                                float *__restrict__ s2, float *__restrict__ d,
                                                int64_t *__restrict__ idx_out) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= batch) return;
  int64_t ix = idx ? idx[b] : philox_index(seed, step, (uint32_t)b, (uint64_t)n_rows);
  ix = ix < 0 ? 0 : (ix >= n_rows ? n_rows - 1 : ix);
  if (idx_out && lane == 0) idx_out[b] = ix;
  const float *src = rows + ix * stride;
  const int NO = round_up(S + A + 2, 4), W = NO + S;  // s' starts on a 16-byte boundary
  for (int c = lane; c < W; c += 64) {
    const float v = src[c];
    if (c < S)
      s[(size_t)b * S + c] = v;
    else if (c < S + A)
      a[(size_t)b * A + (c - S)] = v;
    else if (c == S + A)
      r[b] = v;
    else if (c == S + A + 1)
      d[b] = v;
    else if (c >= NO)
      s2[(size_t)b * S + (c - NO)] = v;
  }
}

hipError_t launch_pack(float *rows, int stride, int S, int A, int64_t first, int64_t n, const float *obs,
                       const float *act, const float *rew, const float *nxt, const float *done,
                       hipStream_t st) {
  const int64_t total = n * (int64_t)stride;
  int grid = (int)((total + 255) / 256);
  if (grid > 256 * 8) grid = 256 * 8;
  hipLaunchKernelGGL(k_pack, dim3(grid), dim3(256), 0, st, rows, stride, S, A, first, n, obs, act, rew, nxt,
                     done);
  return hipGetLastError();
}

hipError_t launch_sample(const iqlhip_replay_view &v, int batch, const int64_t *idx, uint64_t seed,
                         uint64_t step, float *s, float *a, float *r, float *s2, float *d,
                         int64_t *idx_out, hipStream_t st) {
  hipLaunchKernelGGL(k_sample, dim3((batch + 3) / 4), dim3(256), 0, st, v.rows, v.n_rows, v.row_stride,
                     v.state_dim, v.action_dim, batch, idx, seed, step, s, a, r, s2, d, idx_out);
  return hipGetLastError();
}

}  // namespace iqlhip
