// Preference-transformer reward relabel (ref:1223-1309 qlearning_dataset_pt;
// architecture reward_models/pref_transformer.py:170-277, reward_models/ops.py:6-117).
//
// One work-group (8 waves, two per CU) walks a queue of windows.  A window is `len` consecutive
// transitions (len <= query_length; left padding of the reference's batch is never materialised:
// padded keys get -1e4 and vanish from the fp32 softmax).  The reference only reads
// value[:, 0, -1, 0], so for the (single) GPT-2 block only the LAST action token needs a query,
// an attention output and an MLP; every token still contributes its key and value.
//
// Because there is ONE query per window, attention is a streaming reduction over the tokens:
// keys and values are never stored.  A wave job = 16 tokens of one kind (state / action):
//   embedding      H^T = W . X^T on the exact-fp32 matrix cores (v_mfma_f32_16x16x4_f32), computed
//                  TRANSPOSED: lane (token r, quarter q) holds features 16 mt + 4 q + i of ITS
//                  token.  The weights are the A operand (fragment-major LDS image), the dataset
//                  rows the B operand (global).  Accumulators start from bias + timestep embedding.
//   LayerNorms     (stacked, block pre-LN) on that layout: 16 local adds + two cross-quarter
//                  swaps (v_permlane16/32_swap) per sum.
//   K^T | V^T      = Wkv . H^T: the LayerNorm output registers ARE the B fragments (no LDS round
//                  trip); A = the K | V rows of attention.in_linear from LDS.
//   attention      q . k per head from the lane's 16 key features (bf16 products and scale as
//                  ops.py:74-79) + cross-quarter sums; a lane-local online softmax (running max m,
//                  sum l, weighted values o of the lane's tokens); at the end of the window one
//                  reduction over the 16 token lanes per wave and one over the waves in LDS.
// The query of a window (the block input of its last action token through rows 0..63 of
// attention.in_linear, rounded to bf16) is prepared one window ahead by the last wave, which
// is idle in the second job round of a full window (14 jobs on 8 waves).
// Everything behind the attention -- out projection, residual, LayerNorm 1, GPT2MLP, residual,
// final LayerNorm, value head -- is parked per window and runs once per PT_SLOTS windows on the
// matrix cores (tail_batch), so those weights are fetched once per PT_SLOTS windows.
// Cross-lane sums are DPP / v_permlane*_swap (common.h lane_sum), never ds_bpermute.
#include "../../include/iqlhip.h"
#include "common.h"

namespace iqlhip {

constexpr int E = 64;
constexpr int PT_WAVES = 8;
constexpr int PT_SLOTS = 8;  // windows whose tail is batched (<= PT_WAVES: one final LayerNorm per wave)
constexpr int VLD = E + 4;   // row stride of the parked rows read as MFMA A fragments (conflict-free)
constexpr int CB = 96;       // floats a wave leaves for the cross-wave combine: o[64], m[16], l[16]
constexpr int KCH = 3;       // 16-deep k-steps of the embedding GEMM per register chunk (S, A <= 48: one chunk)

__device__ __forceinline__ float wave_sum(float v) { return lane_sum<64>(v); }
__device__ __forceinline__ float sum16(float v) { return lane_sum<16>(v); }  // the 16 lanes sharing lane >> 4
__device__ __forceinline__ float max16(float v) {
  v = fmaxf(v, dpp_mov<0xB1>(v));
  v = fmaxf(v, dpp_mov<0x4E>(v));
  v = fmaxf(v, dpp_mov<0x141>(v));
  return fmaxf(v, dpp_mov<0x140>(v));
}
__device__ __forceinline__ float quarters_sum(float v) { return xor32_sum(xor16_sum(v)); }  // over lane >> 4
// LayerNorm over the 64 lanes (flax/torch: biased variance, eps inside the sqrt)
__device__ __forceinline__ float layer_norm(float x, float w, float b, float eps) {
  const float mu = wave_sum(x) * (1.0f / E);
  const float d = x - mu;
  const float var = wave_sum(d * d) * (1.0f / E);
  return d / sqrtf(var + eps) * w + b;
}
// LayerNorm of a token in the transposed layout: x[mt][i] = feature 16 mt + 4 q + i of the lane's
// token, the other three quarters of the token sit in the lanes r + 16 q'; w / b: LDS vectors [64]
__device__ __forceinline__ void layer_norm_t(f32x4 (&x)[4], const float *w, const float *b, int q, float eps) {
  float s = 0.f;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) s += (x[mt][0] + x[mt][1]) + (x[mt][2] + x[mt][3]);
  const float mu = quarters_sum(s) * (1.0f / E);
  float v = 0.f;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) x[mt][i] -= mu, v += x[mt][i] * x[mt][i];
  }
  const float inv = 1.0f / sqrtf(quarters_sum(v) * (1.0f / E) + eps);  // (flax multiplies by rsqrt as well)
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const float4 w4 = *reinterpret_cast<const float4 *>(w + 16 * mt + 4 * q);
    const float4 b4 = *reinterpret_cast<const float4 *>(b + 16 * mt + 4 * q);
    x[mt][0] = x[mt][0] * inv * w4.x + b4.x, x[mt][1] = x[mt][1] * inv * w4.y + b4.y;
    x[mt][2] = x[mt][2] * inv * w4.z + b4.z, x[mt][3] = x[mt][3] * inv * w4.w + b4.w;
  }
}

__global__ __launch_bounds__(64 * PT_WAVES, 4) void k_pt_relabel(const iqlhip_pt_weights W,
                                                                  const float *__restrict__ obs,
                                                                  const float *__restrict__ act, int64_t n_rows,
                                                                  const int64_t *__restrict__ win_start,
                                                                  const int32_t *__restrict__ win_len,
                                                                  const int32_t *__restrict__ win_t0,
                                                                  int64_t n_win, int ql, float *__restrict__ out) {
  using P = Prec<false>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int S = W.state_dim, A = W.action_dim, I = W.inter_dim, NH = W.num_heads;
  const int HD = E / NH;  // a power of two >= 4
  const int nks_s = round_up(S, 16) / 16, nks_a = round_up(A, 16) / 16;
  // ---- LDS carve ----
  float *wsF = reinterpret_cast<float *>(smem);  // fragment-major [64][16 nks_s] state_linear
  float *waF = wsF + nks_s * 16 * E;             // fragment-major [64][16 nks_a] action_linear
  float *wkvF = waF + nks_a * 16 * E;            // [8 mt][4 ks][64 lanes][4] K | V rows of attention.in_linear
  float *fvec = wkvF + 8 * 4 * 64 * 4;           // [8][64] per-feature vectors of the token jobs
  // (three buffers: the one a window uses is rewritten two windows later, a barrier after its last read)
  float *qbuf = fvec + 8 * E;                    // [3][64] query of the previous / this / the next window (bf16 values)
  float *xnext = qbuf + 3 * E;                   // [3][64] residual stream of the last token, likewise
  float *cbuf = xnext + 3 * E;                   // [2][PT_WAVES][CB] per-wave attention partials
  float *pend_x = cbuf + 2 * PT_WAVES * CB;      // [PT_SLOTS][64] residual stream of parked windows
  float *pend_o = pend_x + PT_SLOTS * E;         // [PT_SLOTS][VLD] attention output
  float *pend_h = pend_o + PT_SLOTS * VLD;       // [PT_SLOTS][VLD] LN1(x1)
  float *hidb = pend_h + PT_SLOTS * VLD;         // [PT_SLOTS][I + 4] MLP hidden
  float *ksplit = cbuf;                          // [2 K halves][PT_SLOTS][64]: no window's partials are live in tail_batch

  // ---- weights that stay on chip for the whole queue ----
  // embedding weights as MFMA fragments (common.h fidx): element (feature f, input k), zero padded
  for (int e = tid; e < nks_s * 16 * E; e += 64 * PT_WAVES) {
    const int k = e / E, f = e - k * E;
    wsF[fidx<P>(f, k, nks_s)] = k < S ? W.state_wT[(size_t)k * E + f] : 0.f;
  }
  for (int e = tid; e < nks_a * 16 * E; e += 64 * PT_WAVES) {
    const int k = e / E, f = e - k * E;
    waF[fidx<P>(f, k, nks_a)] = k < A ? W.action_wT[(size_t)k * E + f] : 0.f;
  }
  // K | V projection (rows 64..191 of attention.in_linear.weight [192][64]): one 16-byte fragment
  // per (m-tile, k-step, lane); m-tile mt < 4 -> key features 16 mt.., mt >= 4 -> value features
  for (int f = wave; f < 32; f += PT_WAVES)  // f = 4 mt + ks
    *reinterpret_cast<uint4 *>(wkvF + (f * 64 + lane) * 4) =
        ldg16(W.qkv_w + (size_t)(E + 16 * (f >> 2) + r) * E + 16 * (f & 3) + 4 * q);
  if (tid < E) {
    fvec[tid] = W.state_b[tid], fvec[E + tid] = W.action_b[tid];
    fvec[2 * E + tid] = W.sln_w[tid], fvec[3 * E + tid] = W.sln_b[tid];
    fvec[4 * E + tid] = W.ln0_w[tid], fvec[5 * E + tid] = W.ln0_b[tid];
    fvec[6 * E + tid] = W.qkv_b[E + tid], fvec[7 * E + tid] = W.qkv_b[2 * E + tid];  // key / value bias
  }
  const float eps = W.eps;
  const float inv_sqrt_hd = 1.0f / sqrtf((float)HD);
  __syncthreads();

  // ---- one token tile through embedding and both LayerNorms (transposed layout) ----
  // x: after stacked_layer_norm (the residual stream); h: after the block's pre-LayerNorm
  auto embed_ln = [&](int kind, int mt_, int64_t start, int len, int t0, f32x4 (&x)[4], f32x4 (&h)[4]) {
    const float *src = kind ? act : obs;
    const int D = kind ? A : S, nks = kind ? nks_a : nks_s;
    const float *wF = kind ? waF : wsF;
    // the lane's token (clamped past len: those columns are dropped by the consumers)
    const int tok = 16 * mt_ + r < len ? 16 * mt_ + r : len - 1;
    const float *rowp = src + (size_t)(start + tok) * D;
    // accumulators start from bias + timestep embedding (timestep = t0 + position in the window;
    // t0 = 0 in ref:1281,1291, the true step in custom_offline:209)
    const float *tp = W.temb + (size_t)(t0 + tok) * E + 4 * q;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const float4 tb = __builtin_bit_cast(float4, ldg16(tp + 16 * mt));
      const float4 bb = *reinterpret_cast<const float4 *>(fvec + kind * E + 16 * mt + 4 * q);
      x[mt] = f32x4{tb.x + bb.x, tb.y + bb.y, tb.z + bb.z, tb.w + bb.w};
    }
    for (int ks0 = 0; ks0 < nks; ks0 += KCH) {
      float4 xb[KCH];  // B fragments: inputs 16 ks + 4 q .. + 3 of the lane's token (zero past D)
#pragma unroll
      for (int kk = 0; kk < KCH; ++kk) {
        float v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int k = 16 * (ks0 + kk) + 4 * q + c;
          const float xv = ldg(rowp + (k < D ? k : D - 1));
          v[c] = k < D ? xv : 0.f;
        }
        xb[kk] = make_float4(v[0], v[1], v[2], v[3]);
      }
#pragma unroll
      for (int kk = 0; kk < KCH; ++kk) {
        if (ks0 + kk < nks) {
          float4 wa[4];
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
            wa[mt] = *reinterpret_cast<const float4 *>(wF + frag_off<P>(mt, ks0 + kk, nks, lane));
          const float xc[4] = {xb[kk].x, xb[kk].y, xb[kk].z, xb[kk].w};
#pragma unroll
          for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
              const float wc[4] = {wa[mt].x, wa[mt].y, wa[mt].z, wa[mt].w};
              x[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[c], xc[c], x[mt], 0, 0, 0);
            }
          }
        }
      }
    }
    layer_norm_t(x, fvec + 2 * E, fvec + 3 * E, q, eps);  // stacked_layer_norm
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) h[mt] = x[mt];
    layer_norm_t(h, fvec + 4 * E, fvec + 5 * E, q, eps);  // block pre-LN
  };

  // A window (start, len, t0) as the kernel uses it: clamped so that no row of obs / act and no
  // entry of the timestep table outside the arrays is ever addressed, whatever a caller hands in
  // (valid windows -- 1 <= len <= ql, start + len <= n_rows, t0 + len <= n_temb, the precondition
  // stated in iqlhip.h -- pass through unchanged; a few scalar instructions per window).
  auto load_window = [&](int64_t win, int64_t &start, int &len, int &t0) {
    int l = win_len[win];
    const int64_t cap = n_rows < (int64_t)W.n_temb ? n_rows : (int64_t)W.n_temb;
    l = l > ql ? ql : l;
    l = (int64_t)l > cap ? (int)cap : l;
    l = l < 1 ? 1 : l;
    int64_t s0 = win_start[win];
    s0 = s0 < 0 ? 0 : (s0 > n_rows - l ? n_rows - l : s0);
    int t = win_t0 ? win_t0[win] : 0;
    t = t < 0 ? 0 : (t > W.n_temb - l ? W.n_temb - l : t);
    start = s0, len = l, t0 = t;
  };

  // ---- the query and the residual stream of a window's last action token, into buffer `pb` ----
  auto prepare_query = [&](int64_t win, int pb) {
    int64_t start;
    int len, t0;
    load_window(win, start, len, t0);
    const int mt_ = (len - 1) >> 4;
    f32x4 x[4], h[4];
    embed_ln(1, mt_, start, len, t0, x, h);
    // Q^T = Wq . H^T (rows 0..63 of attention.in_linear), A fragments from global / L2
    f32x4 qa[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const float4 bb = __builtin_bit_cast(float4, ldg16(W.qkv_b + 16 * mt + 4 * q));
      qa[mt] = f32x4{bb.x, bb.y, bb.z, bb.w};
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      float4 wq[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
        wq[mt] = __builtin_bit_cast(float4, ldg16(W.qkv_w + (size_t)(16 * mt + r) * E + 16 * ks + 4 * q));
#pragma unroll
      for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const float wc[4] = {wq[mt].x, wq[mt].y, wq[mt].z, wq[mt].w};
          qa[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[c], h[ks][c], qa[mt], 0, 0, 0);
        }
      }
    }
    if (16 * mt_ + r == len - 1) {  // the four lanes (one per quarter) of the last token
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        *reinterpret_cast<float4 *>(qbuf + pb * E + 16 * mt + 4 * q) =
            make_float4(rbf(qa[mt][0]), rbf(qa[mt][1]), rbf(qa[mt][2]), rbf(qa[mt][3]));  // ops.py:74
        *reinterpret_cast<float4 *>(xnext + pb * E + 16 * mt + 4 * q) =
            make_float4(x[mt][0], x[mt][1], x[mt][2], x[mt][3]);
      }
    }
  };

  // ---- everything behind the attention of the parked tokens (slots < n) ----
  // Slots are the M rows of 16-row MFMA tiles (rows >= PT_SLOTS alias rows 0..7: their results are
  // dropped).  B fragments come straight from the torch-layout weights ([out][in]: 16 B per lane).
  // Every stage is spread over the waves so that no wave holds more than a few fragments:
  //   S1  x1 = x + o . Wo^T + b                     [16 x 64] . [64 x 64], one n-tile per wave 0..3
  //   S2  h1 = LN1(x1)                              one slot per wave
  //   S3  hidden = relu(h1 . Win^T + b)             [16 x 64] . [64 x I], n-tiles over the waves
  //   S4  hidden . Wout^T                           [16 x I] . [I x 64], (n-tile, K half) per wave
  //   S5  out = value head(LNf(x1 + S4 + b))        one slot per wave
  const int ldh = I + 4;
  auto tail_batch = [&](int n, int64_t first) {
    __syncthreads();  // pend_x / pend_o of every slot written
    if (wave < 4) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      uint4 bw[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) bw[ks] = ldg16(W.attn_out_w + (size_t)(16 * wave + r) * E + 16 * ks + 4 * q);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const f32x4 oa = *reinterpret_cast<const f32x4 *>(pend_o + (r & (PT_SLOTS - 1)) * VLD + 16 * ks + 4 * q);
#pragma unroll
        for (int c = 0; c < 4; ++c)
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(oa[c], __builtin_bit_cast(float4, bw[ks])[c], acc, 0, 0, 0);
      }
      if (q < PT_SLOTS / 4) {
        const float bias = W.attn_out_b[16 * wave + r];
#pragma unroll
        for (int i = 0; i < 4; ++i) pend_x[(4 * q + i) * E + 16 * wave + r] += acc[i] + bias;
      }
    }
    __syncthreads();
    if (wave < PT_SLOTS)
      pend_h[wave * VLD + lane] = layer_norm(pend_x[wave * E + lane], W.ln1_w[lane], W.ln1_b[lane], eps);
    __syncthreads();
    {
      f32x4 ha[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        ha[ks] = *reinterpret_cast<const f32x4 *>(pend_h + (r & (PT_SLOTS - 1)) * VLD + 16 * ks + 4 * q);
      for (int nt = wave; nt < I / 16; nt += PT_WAVES) {
        uint4 bw[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) bw[ks] = ldg16(W.mlp_in_w + (size_t)(16 * nt + r) * E + 16 * ks + 4 * q);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int c = 0; c < 4; ++c)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ha[ks][c], __builtin_bit_cast(float4, bw[ks])[c], acc, 0, 0, 0);
        if (q < PT_SLOTS / 4) {
          const float bi = W.mlp_in_b[16 * nt + r];
#pragma unroll
          for (int i = 0; i < 4; ++i) hidb[(4 * q + i) * ldh + 16 * nt + r] = fmaxf(acc[i] + bi, 0.f);
        }
      }
    }
    __syncthreads();
    {
      // wave -> (n-tile wave & 3, K half wave >> 2); a half is I / 32 k-steps of 16
      const int nt = wave & 3, kq = wave >> 2, nkq = I / 32;
      const float *wrow = W.mlp_out_w + (size_t)(16 * nt + r) * I + 16 * kq * nkq + 4 * q;
      const float *arow = hidb + (r & (PT_SLOTS - 1)) * ldh + 16 * kq * nkq + 4 * q;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int ks0 = 0; ks0 < nkq; ks0 += 4) {
        uint4 bw[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) bw[kk] = ldg16(wrow + 16 * (ks0 + kk));
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const f32x4 av = *reinterpret_cast<const f32x4 *>(arow + 16 * (ks0 + kk));
#pragma unroll
          for (int c = 0; c < 4; ++c)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c], __builtin_bit_cast(float4, bw[kk])[c], acc, 0, 0, 0);
        }
      }
      if (q < PT_SLOTS / 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) ksplit[(kq * PT_SLOTS + 4 * q + i) * E + 16 * nt + r] = acc[i];
      }
    }
    __syncthreads();
    if (wave < n) {  // slot = wave, feature = lane
      const float x2 = pend_x[wave * E + lane] + W.mlp_out_b[lane] + ksplit[wave * E + lane] +
                       ksplit[(PT_SLOTS + wave) * E + lane];
      const float y = layer_norm(x2, W.lnf_w[lane], W.lnf_b[lane], eps);  // gpt.layer_norm
      const float v = wave_sum(y * W.pref_w_last[lane]) + W.pref_b_last;
      if (lane == 0) out[first + (int64_t)wave * gridDim.x] = v;
    }
    __syncthreads();  // the slots are refilled by the next windows
  };

  int nslot = 0, par = 0, qb = 0;  // par: cbuf buffer of this window; qb: its query / residual buffer
  int64_t batch_first = 0;
  if (wave == PT_WAVES - 1 && (int64_t)blockIdx.x < n_win) prepare_query(blockIdx.x, 0);
  __syncthreads();

  for (int64_t win = blockIdx.x; win < n_win; win += gridDim.x, par ^= 1, qb = qb == 2 ? 0 : qb + 1) {
    int64_t start;
    int len, t0;  // t0: timestep of the window's first transition
    load_window(win, start, len, t0);
    const int nmt = (len + 15) >> 4;           // 16-token tiles per kind
    // (the query -- per lane the components that meet its key features 16 mt + 4 q + i -- is read from
    // LDS where a job needs it: 16 registers that are not live across the two GEMMs of a job)
    // lane-local online softmax over the lane's tokens: running max, sum and weighted values per
    // m-tile.  The head of features 16 mt + 4 q + i is (16 mt) / HD for HD >= 16, 2 mt + (q >> 1) for
    // HD = 8, 4 mt + q for HD = 4: after the reductions below s[mt] is that head's logit.
    float m_run[4], l_run[4];
    f32x4 o_run[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) m_run[mt] = -3.0e38f, l_run[mt] = 0.f, o_run[mt] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int job = wave; job < 2 * nmt; job += PT_WAVES) {
      const int kind = job >= nmt ? 1 : 0;  // 0: state tokens, 1: action tokens
      const int mt_ = kind ? job - nmt : job;
      // K^T = Wk . H^T + bias (h IS the B operand), the logits, THEN V^T = Wv . H^T + bias and the
      // softmax update: keys and values are never live together (16 registers less at the peak: the
      // kernel runs four waves per SIMD on 128 registers).  200k antmaze windows 11.69 -> 11.35 ms.
      // (Tried on top: the next job's dataset rows requested one job ahead, also across the window
      // barrier -- 12 more registers in flight: 83 spilled registers, 12.34 ms.)
      f32x4 h[4];
      {
        f32x4 x[4];
        embed_ln(kind, mt_, start, len, t0, x, h);
      }
      auto project = [&](const int half, f32x4 (&out)[4]) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const float4 bb = *reinterpret_cast<const float4 *>(fvec + 6 * E + 16 * (4 * half + mt) + 4 * q);
          out[mt] = f32x4{bb.x, bb.y, bb.z, bb.w};
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          float4 wa[4];  // four m-tiles at a time bound the live A fragments; component-major issue order
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
            wa[mt] = *reinterpret_cast<const float4 *>(wkvF + ((4 * (4 * half + mt) + ks) * 64 + lane) * 4);
#pragma unroll
          for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
              const float wc[4] = {wa[mt].x, wa[mt].y, wa[mt].z, wa[mt].w};
              out[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[c], h[ks][c], out[mt], 0, 0, 0);
            }
          }
        }
      };
      // ---- this token's logits: bf16 q . bf16 k per head (exact products, fp32 sums), rounded to
      // bf16, scaled, rounded (ops.py:74-79) ----
      float s[4];
      {
        f32x4 kt[4];
        project(0, kt);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const f32x4 qv = *reinterpret_cast<const f32x4 *>(qbuf + qb * E + 16 * mt + 4 * q);
          float p = qv[0] * rbf(kt[mt][0]);
          p = fmaf(qv[1], rbf(kt[mt][1]), p), p = fmaf(qv[2], rbf(kt[mt][2]), p);
          s[mt] = fmaf(qv[3], rbf(kt[mt][3]), p);
        }
      }
      if (HD >= 16) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) s[mt] = quarters_sum(s[mt]);
        if (HD == 32) {
          const float a0 = s[0] + s[1], a1 = s[2] + s[3];
          s[0] = s[1] = a0, s[2] = s[3] = a1;
        } else if (HD == 64) {
          const float a0 = (s[0] + s[1]) + (s[2] + s[3]);
          s[0] = s[1] = s[2] = s[3] = a0;
        }
      } else if (HD == 8) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) s[mt] = xor16_sum(s[mt]);
      }
      f32x4 vt[4];
      project(1, vt);
      if (16 * mt_ + r < len) {  // (tokens past the window do not exist)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const float sc = rbf(rbf(s[mt]) * inv_sqrt_hd);
          const float mn = fmaxf(m_run[mt], sc);
          const float c_old = __expf(m_run[mt] - mn), pw = __expf(sc - mn);
          l_run[mt] = l_run[mt] * c_old + pw;
#pragma unroll
          for (int i = 0; i < 4; ++i) o_run[mt][i] = o_run[mt][i] * c_old + pw * vt[mt][i];
          m_run[mt] = mn;
        }
      }
    }
    // ---- the wave's tokens: one reduction over the 16 token lanes, then the waves meet in LDS ----
    {
      float *cb = cbuf + (par * PT_WAVES + wave) * CB;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const float M = max16(m_run[mt]);
        const float c = __expf(m_run[mt] - M);
        const float L = sum16(l_run[mt] * c);
        f32x4 O;
#pragma unroll
        for (int i = 0; i < 4; ++i) O[i] = sum16(o_run[mt][i] * c);
        if (r == 0) {
          *reinterpret_cast<f32x4 *>(cb + 16 * mt + 4 * q) = O;
          cb[E + 4 * q + mt] = M, cb[E + 16 + 4 * q + mt] = L;
        }
      }
    }
    // the last wave prepares the next window while the others finish their second job round
    if (wave == PT_WAVES - 1 && win + (int64_t)gridDim.x < n_win) prepare_query(win + gridDim.x, qb == 2 ? 0 : qb + 1);
    __syncthreads();
    if (wave == 0) {  // park the attention output and the residual stream: the rest runs in tail_batch
      const int f = lane, qq = (f >> 2) & 3, mt = f >> 4;
      const float *cb = cbuf + par * PT_WAVES * CB;
      float M = -3.0e38f;
#pragma unroll
      for (int w = 0; w < PT_WAVES; ++w) M = fmaxf(M, cb[w * CB + E + 4 * qq + mt]);
      float num = 0.f, den = 0.f;
#pragma unroll
      for (int w = 0; w < PT_WAVES; ++w) {
        const float c = __expf(cb[w * CB + E + 4 * qq + mt] - M);
        num += cb[w * CB + f] * c, den += cb[w * CB + E + 16 + 4 * qq + mt] * c;
      }
      pend_o[nslot * VLD + f] = num / den;
      pend_x[nslot * E + f] = xnext[qb * E + f];
    }
    if (nslot == 0) batch_first = win;
    ++nslot;
    if (nslot == PT_SLOTS || win + (int64_t)gridDim.x >= n_win) {
      tail_batch(nslot, batch_first);
      nslot = 0;
    }
  }
}

size_t pt_smem_bytes(const iqlhip_pt_weights &W, int ql) {
  (void)ql;  // keys and values are never stored: the footprint does not depend on the window length
  const size_t ks = (size_t)round_up(W.state_dim, 16) + round_up(W.action_dim, 16);
  return (ks * E + 8 * 4 * 64 * 4 + 8 * E + 6 * E + 2 * PT_WAVES * CB + PT_SLOTS * E + 2 * PT_SLOTS * VLD +
          PT_SLOTS * (W.inter_dim + 4)) * 4 + 64;
}

hipError_t launch_pt(const iqlhip_pt_weights &W, const float *obs, const float *act, int64_t n_rows,
                     const int64_t *win_start, const int32_t *win_len, const int32_t *win_t0, int64_t n_win,
                     int ql, float *out, hipStream_t st) {
  const size_t sm = pt_smem_bytes(W, ql);
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  // persistent work-groups: as many as stay resident (8 waves of <= 128 VGPRs: two per CU when
  // their LDS fits twice)
  const int per_cu = sm <= 80 * 1024 ? 2 : 1;
  int64_t grid = (int64_t)cus * per_cu;
  if (n_win < grid) grid = n_win;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_pt_relabel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_pt_relabel, dim3((unsigned)grid), dim3(64 * PT_WAVES), sm, st, W, obs, act, n_rows,
                     win_start, win_len, win_t0, n_win, ql, out);
  return hipGetLastError();
}

}  // namespace iqlhip
