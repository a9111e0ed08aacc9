// Generic fp32 MLP forward over many rows on the exact-f32 matrix path
// (v_mfma_f32_16x16x4_f32).  Serves nn.Module.forward() of the network
// containers outside the autocast region (ref:452-543, eval_actor ref:306-319)
// and the Markovian reward relabel (ref:719-724, ref:986-991, ref:1176-1178).
//
// One work-group = 64 rows: four 16-row M tiles share every weight fragment, so
// the weights (L2 resident) are read once per 64 rows; activations ping-pong
// between two LDS buffers and never touch HBM.  Weights are read from the fp32
// masters in either layout ([out][in] torch Linear, or [in][out] x@W).
#include "../../include/iqlhip.h"
#include "common.h"

namespace iqlhip {

constexpr int MROWS = 64;

struct MlpArgs {
  int32_t n_layers;
  int32_t dims[IQLHIP_MLP_MAX_LAYERS + 1];
  const float *W[IQLHIP_MLP_MAX_LAYERS];
  const float *b[IQLHIP_MLP_MAX_LAYERS];
  int32_t w_in_out;    // 1: W is [in][out]
  int32_t hidden_act;  // 0 relu, 1 tanh
  int32_t out_act;     // 0 none, 1 tanh
  int32_t lda;         // LDS row stride (floats)
};

__device__ __forceinline__ float act_apply(float v, int kind) {
  return kind == 0 ? fmaxf(v, 0.f) : tanhf(v);
}

__global__ __launch_bounds__(256) void k_mlp_f32(const MlpArgs M, const float *__restrict__ x, int64_t n,
                                                 int x_stride, float *__restrict__ out, int out_stride) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *buf0 = reinterpret_cast<float *>(smem);
  float *buf1 = buf0 + MROWS * M.lda;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * MROWS;
  const int lda = M.lda;

  // input rows, zero padded to a multiple of 16 columns
  {
    const int K0 = M.dims[0], K0p = round_up(K0, 16);
    for (int e = tid; e < MROWS * K0p; e += 256) {
      const int rr = e / K0p, c = e - rr * K0p;
      float v = 0.f;
      if (row0 + rr < n && c < K0) v = x[(size_t)(row0 + rr) * x_stride + c];
      buf0[rr * lda + c] = v;
    }
  }
  __syncthreads();

  float *in = buf0, *ob = buf1;
  for (int l = 0; l < M.n_layers; ++l) {
    const int K = M.dims[l], N = M.dims[l + 1];
    const int Kp = round_up(K, 16), ntile = (N + 15) / 16;
    const float *W = M.W[l];
    const float *bias = M.b[l];
    const bool last = l == M.n_layers - 1;
    for (int jt = wave; jt < ntile; jt += 4) {
      f32x4 acc[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int ncol = 16 * jt + r;
      for (int kb = 0; kb < Kp; kb += 16) {
        float bw[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int k = kb + 4 * q + c;
          bw[c] = (k < K && ncol < N)
                      ? (M.w_in_out ? W[(size_t)k * N + ncol] : W[(size_t)ncol * K + k])
                      : 0.f;
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const float4 a = *reinterpret_cast<const float4 *>(in + (16 * m + r) * lda + kb + 4 * q);
          acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bw[0], acc[m], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bw[1], acc[m], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bw[2], acc[m], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bw[3], acc[m], 0, 0, 0);
        }
      }
      const float bv = ncol < N ? bias[ncol] : 0.f;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int rr = 16 * m + 4 * q + i;
          float v = acc[m][i] + bv;
          if (!last) {
            v = ncol < N ? act_apply(v, M.hidden_act) : 0.f;
            ob[rr * lda + ncol] = v;
          } else if (ncol < N && row0 + rr < n) {
            if (M.out_act == 1) v = tanhf(v);
            out[(size_t)(row0 + rr) * out_stride + ncol] = v;
          }
        }
      }
    }
    __syncthreads();
    float *t = in;
    in = ob, ob = t;
  }
}

hipError_t launch_mlp_f32(const iqlhip_mlp_desc &d, const float *x, int64_t n, int x_stride, float *out,
                          int out_stride, hipStream_t st) {
  MlpArgs M;
  M.n_layers = d.n_layers;
  int maxd = 0;
  for (int i = 0; i <= d.n_layers; ++i) {
    M.dims[i] = d.dims[i];
    if (i < d.n_layers && d.dims[i] > maxd) maxd = d.dims[i];
    if (i > 0 && i < d.n_layers && d.dims[i] > maxd) maxd = d.dims[i];
  }
  for (int i = 0; i < d.n_layers; ++i) M.W[i] = d.weights[i], M.b[i] = d.biases[i];
  M.w_in_out = d.w_in_out, M.hidden_act = d.hidden_act, M.out_act = d.out_act;
  M.lda = round_up(maxd, 16) + 4;
  const size_t sm = (size_t)2 * MROWS * M.lda * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_mlp_f32),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const int64_t grid = (n + MROWS - 1) / MROWS;
  hipLaunchKernelGGL(k_mlp_f32, dim3((unsigned)grid), dim3(256), sm, st, M, x, n, x_stride, out, out_stride);
  return hipGetLastError();
}

}  // namespace iqlhip
