// Generic fp32 MLP forward over many rows on the exact-f32 matrix path
// (v_mfma_f32_16x16x4_f32).  Serves nn.Module.forward() of the network
// containers outside the autocast region (ref:452-543, eval_actor ref:306-319)
// and the Markovian reward relabel (ref:719-724, ref:986-991, ref:1176-1178).
//
// Two launches per call:
//   k_mlp_repack  every layer's weights (torch [out][in] or x@W [in][out]) are rewritten once
//                 into the fragment-major image of common.h (zero padded to 16 x 16): a B
//                 fragment is then ONE contiguous 1 KiB read per wave instruction instead of
//                 64 scalar loads with stride K between lanes.
//   k_mlp_f32     one work-group = 64 rows.  The activations live in ONE LDS buffer
//                 [64][lda] (67 KB at width 256: two work-groups per CU); a layer is computed
//                 into registers (wave w owns n-tiles w, w+4, w+8, w+12; four 16-row M tiles
//                 share every B fragment; A fragments are shared by the wave's four n-tiles),
//                 then written back in place behind a barrier.  Per 16-deep k-step a wave issues
//                 4 LDS reads, 4 fragment loads (the next step's, prefetched) and 64 MFMAs.
#include "../../include/iqlhip.h"
#include "common.h"

namespace iqlhip {

constexpr int MROWS = 64;
constexpr int MAXT = 4;  // n-tiles per wave: widths <= 256

struct MlpArgs {
  int32_t n_layers;
  int32_t dims[IQLHIP_MLP_MAX_LAYERS + 1];
  const float *Wf[IQLHIP_MLP_MAX_LAYERS];  // fragment-major [round16(out)][round16(in)]
  const float *b[IQLHIP_MLP_MAX_LAYERS];
  int32_t hidden_act;  // 0 relu, 1 tanh
  int32_t out_act;     // 0 none, 1 tanh
  int32_t lda;         // LDS row stride (floats)
  // nn.Dropout(p) behind every hidden activation (ref:436-437) in train mode: keep iff the
  // element's Philox word >= drop_thr, kept values times 1 / (1 - p); drop_thr = 0: no dropout
  uint32_t drop_thr;
  float drop_scale;
  uint64_t drop_seed;
  uint32_t drop_call;
};

struct RepackArgs {
  int32_t n_layers;
  int32_t dims[IQLHIP_MLP_MAX_LAYERS + 1];
  const float *W[IQLHIP_MLP_MAX_LAYERS];
  float *Wf[IQLHIP_MLP_MAX_LAYERS];
  int32_t w_in_out;  // 1: W is [in][out]
};

__global__ void k_mlp_repack(const RepackArgs R) {
  using P = Prec<false>;
  const int l = blockIdx.y;
  if (l >= R.n_layers) return;
  const int K = R.dims[l], N = R.dims[l + 1];
  const int Kp = round_up(K, 16), Np = round_up(N, 16);
  const float *W = R.W[l];
  float *Wf = R.Wf[l];
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < Np * Kp; e += gridDim.x * blockDim.x) {
    // consecutive threads read consecutive source elements of the layout at hand
    const int f = R.w_in_out ? e % Np : e / Kp, k = R.w_in_out ? e / Np : e % Kp;
    float v = 0.f;
    if (f < N && k < K) v = R.w_in_out ? W[(size_t)k * N + f] : W[(size_t)f * K + k];
    Wf[fidx<P>(f, k, Kp / 16)] = v;
  }
}

// activation codes of iqlhip_mlp_desc: 0 / 1 are the torch containers' (hidden: relu / tanh; output:
// none / tanh); 8 + i is entry i of reward_models/q_mlp.py:121-130
//   cos, tanh, relu, softplus, sin, leaky_relu, swish, none
__device__ __forceinline__ float act_flax(float v, int i) {
  switch (i) {
    case 0: return cosf(v);
    case 1: return tanhf(v);
    case 2: return fmaxf(v, 0.f);
    case 3: return fmaxf(v, 0.f) + log1pf(expf(-fabsf(v)));  // jax.nn.softplus = logaddexp(x, 0)
    case 4: return sinf(v);
    case 5: return v >= 0.f ? v : 0.01f * v;                  // jax.nn.leaky_relu, slope 0.01
    case 6: return v / (1.f + expf(-v));                      // swish = x * sigmoid(x)
    default: return v;
  }
}
// The kernel is instantiated per activation family (ACT: 0 relu hidden layers, 1 tanh hidden layers,
// 2 anything of the table above).  The epilogues are unrolled 64-fold over register tiles: with the
// whole table inlined at every site the kernel was 0.5 MB of code and every site an instruction-cache
// miss (the relu path spent more time there than in its MFMAs).  The table variant therefore writes
// raw sums to LDS and applies the activation in ONE rolled pass.
template <int ACT>
__device__ __forceinline__ float act_apply(float v, int kind) {  // hidden layers
  if constexpr (ACT == 0) return fmaxf(v, 0.f);
  if constexpr (ACT == 1) return tanhf(v);
  return kind >= 8 ? act_flax(v, kind - 8) : (kind == 0 ? fmaxf(v, 0.f) : tanhf(v));
}
template <int ACT>
__device__ __forceinline__ float out_apply(float v, int kind) {  // output layer
  if constexpr (ACT != 2) return kind == 1 ? tanhf(v) : v;
  return kind >= 8 ? act_flax(v, kind - 8) : (kind == 1 ? tanhf(v) : v);
}

template <int ACT>
__global__ __launch_bounds__(256, 2) void k_mlp_f32(const MlpArgs M, const float *__restrict__ x, int64_t n,
                                                    int x_stride, float *__restrict__ out, int out_stride) {
  using P = Prec<false>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *buf = reinterpret_cast<float *>(smem);  // [MROWS][lda]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * MROWS;
  const int lda = M.lda;

  // input rows, zero padded to a multiple of 16 columns
  {
    const int K0 = M.dims[0], K0p = round_up(K0, 16);
    for (int e = tid; e < MROWS * K0p; e += 256) {
      const int rr = e / K0p, c = e - rr * K0p;
      float v = 0.f;
      if (row0 + rr < n && c < K0) v = ldg(x + (size_t)(row0 + rr) * x_stride + c);
      buf[rr * lda + c] = v;
    }
  }
  __syncthreads();

  for (int l = 0; l < M.n_layers; ++l) {
    const int K = M.dims[l], N = M.dims[l + 1];
    const int nk = round_up(K, 16) / 16, ntile = round_up(N, 16) / 16;
    const float *Wf = M.Wf[l];
    const float *bias = M.b[l];
    const bool last = l == M.n_layers - 1;
    if (last && ntile < 4) {
      // ---- narrow output layer (N < 64, e.g. the reward head N = 1): with n-tiles over waves
      // only `ntile` waves would work through all of K; instead the four waves split the k-steps,
      // every wave accumulates all n-tiles for its share, and the partial tiles meet in LDS ----
      f32x4 pacc[3][4];
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int m = 0; m < 4; ++m) pacc[t][m] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int ks = wave; ks < nk; ks += 4) {
        uint4 a[4], bq[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) bq[t] = ldg16(Wf + frag_off<P>(t < ntile ? t : 0, ks, nk, lane));
#pragma unroll
        for (int m = 0; m < 4; ++m)
          a[m] = *reinterpret_cast<const uint4 *>(buf + (16 * m + r) * lda + 16 * ks + 4 * q);
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          if (t < ntile) {
#pragma unroll
            for (int m = 0; m < 4; ++m) P::mma(a[m], bq[t], pacc[t][m]);
          }
        }
      }
      __syncthreads();  // the activations have been read: the buffer holds the partial tiles now
      f32x4 *red = reinterpret_cast<f32x4 *>(buf);  // [4 waves][ntile][4 m][64 lanes]
#pragma unroll
      for (int t = 0; t < 3; ++t)
        if (t < ntile)
#pragma unroll
          for (int m = 0; m < 4; ++m) red[((wave * ntile + t) * 4 + m) * 64 + lane] = pacc[t][m];
      __syncthreads();
      for (int e = tid; e < ntile * 4 * 64; e += 256) {
        const int ln = e & 63, m = (e >> 6) & 3, t = e >> 8;
        f32x4 sum = red[((0 * ntile + t) * 4 + m) * 64 + ln];
#pragma unroll
        for (int w = 1; w < 4; ++w) sum += red[((w * ntile + t) * 4 + m) * 64 + ln];
        const int ncol = 16 * t + (ln & 15);
        if (ncol < N) {
          const float bvv = ldg(bias + ncol);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int rr = 16 * m + 4 * (ln >> 4) + i;
            if (row0 + rr < n) stg(out + (size_t)(row0 + rr) * out_stride + ncol, out_apply<ACT>(sum[i] + bvv, M.out_act));
          }
        }
      }
      __syncthreads();
      continue;
    }
    // this wave's n-tiles: wave, wave + 4, ... (clamped copies beyond ntile are computed on a
    // valid fragment and dropped: no branches in the load stream)
    int tile[MAXT];
    bool live[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
      live[t] = wave + 4 * t < ntile;
      tile[t] = live[t] ? wave + 4 * t : 0;
    }
    f32x4 acc[MAXT][4];
#pragma unroll
    for (int t = 0; t < MAXT; ++t)
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[t][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bv[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; ++t) bv[t] = ldg(bias + (16 * tile[t] + r < N ? 16 * tile[t] + r : N - 1));
    if (live[0]) {  // wave-uniform: a wave without tiles (narrow layers) skips the k loop
      uint4 bq[MAXT];
#pragma unroll
      for (int t = 0; t < MAXT; ++t) bq[t] = ldg16(Wf + frag_off<P>(tile[t], 0, nk, lane));
      for (int ks = 0; ks < nk; ++ks) {
        uint4 bn[MAXT];  // next k-step's fragments, in flight during this step's MFMAs
        const int kn = ks + 1 < nk ? ks + 1 : ks;
#pragma unroll
        for (int t = 0; t < MAXT; ++t) bn[t] = ldg16(Wf + frag_off<P>(tile[t], kn, nk, lane));
        uint4 a[4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
          a[m] = *reinterpret_cast<const uint4 *>(buf + (16 * m + r) * lda + 16 * ks + 4 * q);
        // component-major issue order: the four MFMAs of one (n-tile, M-tile) pair accumulate into
        // the same registers (40-cycle dependent latency vs 32-cycle issue); between two of them the
        // wave issues the other 15 pairs
        float4 af[4], bf[MAXT];
#pragma unroll
        for (int m = 0; m < 4; ++m) af[m] = __builtin_bit_cast(float4, a[m]);
#pragma unroll
        for (int t = 0; t < MAXT; ++t) bf[t] = __builtin_bit_cast(float4, bq[t]);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
          for (int t = 0; t < MAXT; ++t) {
            if (live[t]) {
#pragma unroll
              for (int m = 0; m < 4; ++m)
                acc[t][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m][c], bf[t][c], acc[t][m], 0, 0, 0);
            }
          }
        }
#pragma unroll
        for (int t = 0; t < MAXT; ++t) bq[t] = bn[t];
      }
    }
    __syncthreads();  // every wave has read its last A fragment: the buffer may be overwritten
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
      if (!live[t]) continue;
      const int ncol = 16 * tile[t] + r;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int rr = 16 * m + 4 * q + i;
          float v = acc[t][m][i] + bv[t];
          if constexpr (ACT == 2) {
            buf[rr * lda + ncol] = v;  // raw: the table is applied in the rolled pass below
          } else if (!last) {
            buf[rr * lda + ncol] = ncol < N ? act_apply<ACT>(v, M.hidden_act) : 0.f;  // zero K padding
          } else if (ncol < N && row0 + rr < n) {
            stg(out + (size_t)(row0 + rr) * out_stride + ncol, out_apply<ACT>(v, M.out_act));
          }
        }
      }
    }
    __syncthreads();
    if constexpr (ACT == 2) {
      const int Np = 16 * ntile;
#pragma unroll 1
      for (int e = tid; e < MROWS * Np; e += 256) {
        const int rr = e / Np, c = e - rr * Np;
        const float v = buf[rr * lda + c];
        if (!last) {
          buf[rr * lda + c] = c < N ? act_apply<2>(v, M.hidden_act) : 0.f;
        } else if (c < N && row0 + rr < n) {
          stg(out + (size_t)(row0 + rr) * out_stride + c, out_apply<2>(v, M.out_act));
        }
      }
      __syncthreads();
    }
    if (M.drop_thr && !last) {
      // Dropout of this hidden layer (oracle/philox.py mlp_dropout_keep): one Philox4x32 block per
      // (row, 4 consecutive units); a rolled pass over LDS, the 64-fold unrolled epilogues stay lean
      const int Nq = 4 * ntile;
#pragma unroll 1
      for (int e = tid; e < MROWS * Nq; e += 256) {
        const int rr = e / Nq, cq = e - rr * Nq;
        const uint64_t R = (uint64_t)(row0 + rr);
        const Philox4 ph = philox4x32_10((uint32_t)R, M.drop_call, (uint32_t)cq | ((uint32_t)l << 16),
                                         STREAM_MLP_DROPOUT, (uint32_t)M.drop_seed, (uint32_t)(M.drop_seed >> 32));
        float4 v = *reinterpret_cast<float4 *>(buf + rr * lda + 4 * cq);
        v.x = ph.x >= M.drop_thr ? v.x * M.drop_scale : 0.f;
        v.y = ph.y >= M.drop_thr ? v.y * M.drop_scale : 0.f;
        v.z = ph.z >= M.drop_thr ? v.z * M.drop_scale : 0.f;
        v.w = ph.w >= M.drop_thr ? v.w * M.drop_scale : 0.f;
        *reinterpret_cast<float4 *>(buf + rr * lda + 4 * cq) = v;
      }
      __syncthreads();
    }
  }
}

// Widths beyond 256 (up to 1024; ref:417-449 accepts any hidden_dim): one wave per 16 rows, the
// activations ping-pong between two LDS images [16][lda], every B fragment one contiguous read of the
// fragment-major image.  No register tiling across rows or n-tiles: this variant exists for coverage.
__global__ __launch_bounds__(64) void k_mlp_wide(const MlpArgs M, const float *__restrict__ x, int64_t n, int x_stride,
                                                 float *__restrict__ out, int out_stride) {
  using P = Prec<false>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x, r = lane & 15, q = lane >> 4, lda = M.lda;
  float *in = reinterpret_cast<float *>(smem), *nxt = in + 16 * lda;
  const int64_t row0 = (int64_t)blockIdx.x * 16;
  {
    const int K0 = M.dims[0], K0p = round_up(K0, 16);
    for (int e = lane; e < 16 * K0p; e += 64) {
      const int rr = e / K0p, c = e - rr * K0p;
      float v = 0.f;
      if (row0 + rr < n && c < K0) v = ldg(x + (size_t)(row0 + rr) * x_stride + c);
      in[rr * lda + c] = v;
    }
  }
  __syncthreads();
  for (int l = 0; l < M.n_layers; ++l) {
    const int K = M.dims[l], N = M.dims[l + 1];
    const int nk = round_up(K, 16) / 16, ntile = round_up(N, 16) / 16;
    const float *Wf = M.Wf[l], *bias = M.b[l];
    const bool last = l == M.n_layers - 1;
    for (int nt = 0; nt < ntile; ++nt) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int ks = 0; ks < nk; ++ks) {
        const uint4 a = *reinterpret_cast<const uint4 *>(in + r * lda + ks * 16 + q * 4);
        const uint4 b = ldg16(Wf + frag_off<P>(nt, ks, nk, lane));
        P::mma(a, b, acc);
      }
      const int col = nt * 16 + r;
      const float bv = col < N ? ldg(bias + col) : 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t R = row0 + 4 * q + i;
        float v = acc[i] + bv;
        if (!last) {
          v = col < N ? act_apply<2>(v, M.hidden_act) : 0.f;
          if (M.drop_thr) {  // oracle/philox.py mlp_dropout_keep: block (row, unit / 4 | layer << 16), word unit % 4
            const Philox4 ph = philox4x32_10((uint32_t)R, M.drop_call, (uint32_t)(col >> 2) | ((uint32_t)l << 16),
                                             STREAM_MLP_DROPOUT, (uint32_t)M.drop_seed, (uint32_t)(M.drop_seed >> 32));
            const uint32_t w = (col & 3) == 0 ? ph.x : (col & 3) == 1 ? ph.y : (col & 3) == 2 ? ph.z : ph.w;
            v = w >= M.drop_thr ? v * M.drop_scale : 0.f;
          }
          nxt[(4 * q + i) * lda + col] = v;
        } else if (col < N && R < n) {
          stg(out + (size_t)R * out_stride + col, out_apply<2>(v, M.out_act));
        }
      }
    }
    __syncthreads();
    float *t_ = in;
    in = nxt, nxt = t_;
  }
}

hipError_t launch_mlp_f32(const iqlhip_mlp_desc &d, const float *x, int64_t n, int x_stride, float *out,
                          int out_stride, hipStream_t st) {
  MlpArgs M;
  RepackArgs R;
  M.n_layers = R.n_layers = d.n_layers;
  int maxd = 0;
  size_t total = 0, off[IQLHIP_MLP_MAX_LAYERS];
  for (int i = 0; i <= d.n_layers; ++i) {
    M.dims[i] = R.dims[i] = d.dims[i];
    if (i < d.n_layers && d.dims[i] > maxd) maxd = d.dims[i];  // widths that pass through the LDS buffer
  }
  for (int i = 0; i < d.n_layers; ++i) {
    off[i] = total;
    total += (size_t)round_up(d.dims[i], 16) * round_up(d.dims[i + 1], 16);
  }
  float *wf = nullptr;  // stream-ordered scratch for the fragment-major images (<= 8 x 256 KB)
  hipError_t e = hipMallocAsync((void **)&wf, total * sizeof(float), st);
  if (e != hipSuccess) return e;
  for (int i = 0; i < d.n_layers; ++i) {
    R.W[i] = d.weights[i], R.Wf[i] = wf + off[i];
    M.Wf[i] = wf + off[i], M.b[i] = d.biases[i];
  }
  R.w_in_out = d.w_in_out;
  M.hidden_act = d.hidden_act, M.out_act = d.out_act;
  M.lda = round_up(maxd, 16) + 4;
  M.drop_thr = 0, M.drop_scale = 1.f, M.drop_seed = d.dropout_seed, M.drop_call = d.dropout_call;
  if (d.dropout_p > 0.f) {  // the trainer's fp32 convention (api.hip): thr = p 2^32, scale = 1 / (1 - p)
    const double thr = (double)d.dropout_p * 4294967296.0;
    M.drop_thr = thr >= 4294967295.0 ? 0xffffffffu : (thr < 1.0 ? 1u : (uint32_t)thr);
    M.drop_scale = 1.0f / (float)(1.0 - (double)d.dropout_p);
  }
  hipLaunchKernelGGL(k_mlp_repack, dim3(64, d.n_layers), dim3(256), 0, st, R);
  // the activations [64][lda], or the partial tiles of a k-split output layer (4 waves x <= 3 n-tiles)
  size_t sm = (size_t)MROWS * M.lda * sizeof(float);
  const int nt_last = round_up(d.dims[d.n_layers], 16) / 16;
  if (nt_last < 4 && sm < (size_t)4 * nt_last * 4 * 64 * 16) sm = (size_t)4 * nt_last * 4 * 64 * 16;
  bool wide = false;
  for (int i = 0; i <= d.n_layers; ++i) wide |= d.dims[i] > 256;
  if (wide) {
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_mlp_wide), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024);
    if (e == hipSuccess) {
      int maxw = maxd;
      if (d.dims[d.n_layers] > maxw) maxw = d.dims[d.n_layers];
      M.lda = round_up(maxw, 16) + 4;
      hipLaunchKernelGGL(k_mlp_wide, dim3((unsigned)((n + 15) / 16)), dim3(64), (size_t)2 * 16 * M.lda * sizeof(float),
                         st, M, x, n, x_stride, out, out_stride);
      e = hipGetLastError();
    }
    (void)hipFreeAsync(wf, st);
    return e;
  }
  // (set on every call: the attribute is per device, and a process may drive several)
  const int act = (d.hidden_act >= 8 || d.out_act >= 8) ? 2 : d.hidden_act;  // instantiation, see act_apply
  auto kern = act == 0 ? k_mlp_f32<0> : act == 1 ? k_mlp_f32<1> : k_mlp_f32<2>;
  e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                          160 * 1024);
  if (e == hipSuccess) {
    const int64_t grid = (n + MROWS - 1) / MROWS;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), sm, st, M, x, n, x_stride, out, out_stride);
    e = hipGetLastError();
  }
  (void)hipFreeAsync(wf, st);
  return e;
}

}  // namespace iqlhip
