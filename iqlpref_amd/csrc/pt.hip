// Preference-transformer reward relabel (ref:1223-1309 qlearning_dataset_pt;
// architecture reward_models/pref_transformer.py:170-277, reward_models/ops.py:6-117).
//
// One work-group walks a queue of windows.  A window is `len` consecutive
// transitions (len <= query_length; left padding of the reference's batch is
// never materialised: padded keys get -1e4 and vanish from the fp32 softmax).
// The reference only reads value[:, 0, -1, 0], so for the (single) GPT-2 block
// only the LAST action token needs a query, attention output and MLP; every
// token still needs its key and value.
//
// embd_dim == 64 == wave size: lane e of a wave owns embedding element e of the
// token that wave is processing, LayerNorm statistics are wave reductions, a
// lane keeps ITS row of Wk and Wv (128 VGPRs) for the whole kernel, keys (bf16,
// as ops.py:74-76 casts them) and values live in LDS.  fp32 arithmetic except
// the q.k products, which follow the reference's bf16 cast.
#include "../../include/iqlhip.h"
#include "common.h"

namespace iqlhip {

constexpr int E = 64;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}
__device__ __forceinline__ float seg_sum(float v, int width) {  // lanes grouped by `width` (pow2)
  for (int m = width >> 1; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}
// LayerNorm over the 64 lanes (flax/torch: biased variance, eps inside the sqrt)
__device__ __forceinline__ float layer_norm(float x, float w, float b, float eps) {
  const float mu = wave_sum(x) * (1.0f / E);
  const float d = x - mu;
  const float var = wave_sum(d * d) * (1.0f / E);
  return d / sqrtf(var + eps) * w + b;
}

__global__ __launch_bounds__(256) void k_pt_relabel(const iqlhip_pt_weights W, const float *__restrict__ obs,
                                                    const float *__restrict__ act, int64_t n_rows,
                                                    const int64_t *__restrict__ win_start,
                                                    const int32_t *__restrict__ win_len, int64_t n_win,
                                                    int ql, float *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int S = W.state_dim, A = W.action_dim, I = W.inter_dim, NH = W.num_heads;
  const int HD = E / NH;
  const int Tmax = 2 * ql;
  // ---- LDS carve ----
  float *Vs = reinterpret_cast<float *>(smem);               // [Tmax][64]
  uint16_t *Kb = reinterpret_cast<uint16_t *>(Vs + (size_t)Tmax * E);  // [Tmax][64] bf16
  float *wsT = reinterpret_cast<float *>(Kb + (size_t)Tmax * E);       // [S][64]
  float *waT = wsT + S * E;                                   // [A][64]
  float *wqT = waT + A * E;                                   // [64][64]
  float *woT = wqT + E * E;                                   // [64][64]
  float *hs = woT + E * E;                                    // [4][64] per-wave token scratch
  float *xlast = hs + 4 * E;                                  // [64]
  float *hlast = xlast + E;                                   // [64]
  float *ovec = hlast + E;                                    // [64] attention output / LN1 output
  float *part = ovec + E;                                     // [4][64] cross-wave partials
  float *stat = part + 4 * E;                                 // [4 waves][16 heads] x 2
  float *hid = stat + 2 * 4 * 16;                             // [I]
  float *lg = hid + I;                                        // [Tmax][NH] logits

  // ---- weights that stay on chip for the whole queue ----
  for (int e = tid; e < S * E; e += 256) wsT[e] = W.state_wT[e];
  for (int e = tid; e < A * E; e += 256) waT[e] = W.action_wT[e];
  for (int e = tid; e < E * E; e += 256) wqT[e] = W.q_wT[e], woT[e] = W.attn_out_wT[e];
  float wk[E], wv[E];
#pragma unroll
  for (int j = 0; j < E; ++j) {
    wk[j] = W.qkv_w[(size_t)(E + lane) * E + j];
    wv[j] = W.qkv_w[(size_t)(2 * E + lane) * E + j];
  }
  const float bs = W.state_b[lane], ba = W.action_b[lane];
  const float slw = W.sln_w[lane], slb = W.sln_b[lane];
  const float l0w = W.ln0_w[lane], l0b = W.ln0_b[lane];
  const float l1w = W.ln1_w[lane], l1b = W.ln1_b[lane];
  const float lfw = W.lnf_w[lane], lfb = W.lnf_b[lane];
  const float bq = W.qkv_b[lane], bk = W.qkv_b[E + lane], bv = W.qkv_b[2 * E + lane];
  const float bo = W.attn_out_b[lane], bmo = W.mlp_out_b[lane];
  const float pw = W.pref_w_last[lane];
  const float eps = W.eps;
  const float inv_sqrt_hd = 1.0f / sqrtf((float)HD);
  __syncthreads();

  for (int64_t win = blockIdx.x; win < n_win; win += gridDim.x) {
    const int64_t start = win_start[win];
    const int len = win_len[win];
    const int T = 2 * len;
    // ================= every token: embedding, LN, key / value =================
    for (int t = wave; t < T; t += 4) {
      const int k = t >> 1;
      const int64_t row = start + k;
      float x;
      if ((t & 1) == 0) {
        x = bs;
        const float *src = obs + (size_t)row * S;
        for (int j = 0; j < S; ++j) x += src[j] * wsT[j * E + lane];
      } else {
        x = ba;
        const float *src = act + (size_t)row * A;
        for (int j = 0; j < A; ++j) x += src[j] * waT[j * E + lane];
      }
      x += W.temb[(size_t)k * E + lane];  // timestep k of the window (ref:1281,1291)
      x = layer_norm(x, slw, slb, eps);   // stacked_layer_norm
      const float h = layer_norm(x, l0w, l0b, eps);  // block pre-LN
      hs[wave * E + lane] = h;
      if (t == T - 1) xlast[lane] = x, hlast[lane] = h;
      float kk = bk, vv = bv;
#pragma unroll
      for (int j4 = 0; j4 < E; j4 += 4) {
        const float4 hv = *reinterpret_cast<const float4 *>(&hs[wave * E + j4]);
        kk += hv.x * wk[j4] + hv.y * wk[j4 + 1] + hv.z * wk[j4 + 2] + hv.w * wk[j4 + 3];
        vv += hv.x * wv[j4] + hv.y * wv[j4 + 1] + hv.z * wv[j4 + 2] + hv.w * wv[j4 + 3];
      }
      Kb[(size_t)t * E + lane] = f2bf(kk);
      Vs[(size_t)t * E + lane] = vv;
    }
    __syncthreads();
    // ================= last token: query, attention over all keys =================
    float qv = bq;
#pragma unroll 8
    for (int j = 0; j < E; ++j) qv += hlast[j] * wqT[j * E + lane];
    const float qb = rbf(qv);  // ops.py:74
    const int head = lane / HD;
    float lmax = -3.0e38f;
    for (int t = wave; t < T; t += 4) {
      float s = seg_sum(qb * bf2f(Kb[(size_t)t * E + lane]), HD);
      s = rbf(rbf(s) * inv_sqrt_hd);  // bf16 product tensor, bf16 scale (ops.py:76-79)
      if ((lane % HD) == 0) lg[t * NH + head] = s;
      lmax = fmaxf(lmax, s);
    }
    if ((lane % HD) == 0) stat[wave * 16 + head] = lmax;
    __syncthreads();
    float gmax = stat[head];
#pragma unroll
    for (int w = 1; w < 4; ++w) gmax = fmaxf(gmax, stat[w * 16 + head]);
    float lsum = 0.f, oacc = 0.f;
    for (int t = wave; t < T; t += 4) {
      const float p = expf(lg[t * NH + head] - gmax);
      lsum += p;
      oacc += p * Vs[(size_t)t * E + lane];
    }
    part[wave * E + lane] = oacc;
    if ((lane % HD) == 0) stat[64 + wave * 16 + head] = lsum;
    __syncthreads();
    {
      float o = part[lane] + part[E + lane] + part[2 * E + lane] + part[3 * E + lane];
      const float den = stat[64 + head] + stat[64 + 16 + head] + stat[64 + 32 + head] + stat[64 + 48 + head];
      if (wave == 0) ovec[lane] = o / den;
    }
    __syncthreads();
    // ---- out projection + residual, LN1 (all waves redundantly: 64 FMAs) ----
    float x1 = bo + xlast[lane];
#pragma unroll 8
    for (int j = 0; j < E; ++j) x1 += ovec[j] * woT[j * E + lane];
    const float h1 = layer_norm(x1, l1w, l1b, eps);
    __syncthreads();  // everyone has read ovec
    if (wave == 0) ovec[lane] = h1;
    __syncthreads();
    // ---- MLP: hidden units j = lane + 64 m, m split over the waves ----
    for (int m = wave; m < I / E; m += 4) {
      const int j = lane + E * m;
      float a = W.mlp_in_b[j];
#pragma unroll 8
      for (int e = 0; e < E; ++e) a += ovec[e] * W.mlp_in_wT[(size_t)e * I + j];
      hid[j] = fmaxf(a, 0.f);
    }
    __syncthreads();
    {
      const int j0 = wave * (I / 4), j1 = j0 + I / 4;
      float a = 0.f;
      for (int j = j0; j < j1; ++j) a += hid[j] * W.mlp_out_wT[(size_t)j * E + lane];
      part[wave * E + lane] = a;
    }
    __syncthreads();
    if (wave == 0) {
      const float x2 = bmo + part[lane] + part[E + lane] + part[2 * E + lane] + part[3 * E + lane] + x1;
      const float y = layer_norm(x2, lfw, lfb, eps);  // gpt.layer_norm
      const float v = wave_sum(y * pw) + W.pref_b_last;
      if (lane == 0) out[win] = v;
    }
    __syncthreads();  // LDS is reused by the next window
  }
}

size_t pt_smem_bytes(const iqlhip_pt_weights &W, int ql) {
  const size_t Tmax = 2 * (size_t)ql;
  return Tmax * E * 4 + Tmax * E * 2 + (size_t)(W.state_dim + W.action_dim) * E * 4 + 2 * E * E * 4 +
         (4 * E + 3 * E + 4 * E + 2 * 4 * 16 + W.inter_dim + Tmax * W.num_heads) * 4 + 64;
}

hipError_t launch_pt(const iqlhip_pt_weights &W, const float *obs, const float *act, int64_t n_rows,
                     const int64_t *win_start, const int32_t *win_len, int64_t n_win, int ql, float *out,
                     hipStream_t st) {
  const size_t sm = pt_smem_bytes(W, ql);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_pt_relabel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) return e;
  int64_t grid = n_win < 1024 ? n_win : 1024;  // 4 work-groups per CU queue the windows
  hipLaunchKernelGGL(k_pt_relabel, dim3((unsigned)grid), dim3(256), sm, st, W, obs, act, n_rows, win_start,
                     win_len, n_win, ql, out);
  return hipGetLastError();
}

}  // namespace iqlhip
