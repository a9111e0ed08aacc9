// Preference-transformer reward relabel (ref:1223-1309 qlearning_dataset_pt;
// architecture reward_models/pref_transformer.py:170-277, reward_models/ops.py:6-117).
//
// One work-group (8 waves) walks a queue of windows.  A window is `len` consecutive
// transitions (len <= query_length; left padding of the reference's batch is never
// materialised: padded keys get -1e4 and vanish from the fp32 softmax).  The reference only
// reads value[:, 0, -1, 0], so for the (single) GPT-2 block only the LAST action token needs a
// query, attention output and MLP; every token still needs its key and value.
//
// The per-token work is the two batched Linear layers of the model and runs on the exact-fp32
// matrix cores (v_mfma_f32_16x16x4_f32), 16 tokens of one kind (state / action) per wave job:
//   embedding   [16 x S|A] . W^T   A fragments straight from the dataset rows (global), B
//               fragments from a fragment-major LDS image of state_linear / action_linear
//   + bias + timestep embedding, stacked LayerNorm, block pre-LayerNorm: in the MFMA C layout
//               (a token's 64 features sit in 4 accumulators x 16 lanes: sums are 4 adds +
//               4 xor-shuffles)
//   K | V       [16 x 64] . Wkv^T (64 -> 128): A fragments through LDS (the job's own, not yet
//               written V rows serve as the transposition scratch), B fragments (the K and V
//               rows of attention.in_linear) live in 128 VGPRs for the whole kernel
// Keys are stored as bf16 (ops.py:74-76 casts them), values as fp32.  The job that holds the
// window's last action token also projects its tile's queries (one more [16 x 64] . [64 x 64] on
// the matrix cores) and leaves the last token's query (bf16-rounded, ops.py:74) and residual
// stream in LDS.  Per window only the attention of that one query stays on the vector units:
// logits over all keys (bf16 q.k products and scale as the reference), softmax, P.V.  Everything
// behind it -- attention out projection, residual, LayerNorm 1, GPT2MLP, residual, final
// LayerNorm, value head -- is parked per window and runs once per PT_SLOTS windows on the matrix
// cores (tail_batch), so those weights are fetched once per PT_SLOTS windows.
// Cross-lane sums are DPP / v_permlane*_swap (common.h lane_sum), not ds_bpermute.
#include "../../include/iqlhip.h"
#include "common.h"
#include <cstdlib>

namespace iqlhip {

constexpr int E = 64;
constexpr int PT_WAVES = 16;  // 1024 threads: four waves per SIMD, <= 128 VGPRs each
constexpr int VLD = E + 4;  // row stride of the V rows (floats): conflict-free A-fragment reads; column 64 holds 1
constexpr int KLD = E + 8;  // row stride of the bf16 K rows (144 B): conflict-free 16-byte B-fragment reads
constexpr int PT_SLOTS = 8;   // windows whose last-token tail is batched (<= PT_WAVES: one final LayerNorm per wave)

__device__ __forceinline__ float wave_sum(float v) { return lane_sum<64>(v); }
__device__ __forceinline__ float seg_sum(float v, int width) { return lane_sum_rt(v, width); }  // pow2 groups
__device__ __forceinline__ float sum16(float v) { return lane_sum<16>(v); }  // the 16 lanes sharing lane >> 4
__device__ __forceinline__ float max16(float v) {
  v = fmaxf(v, dpp_mov<0xB1>(v));
  v = fmaxf(v, dpp_mov<0x4E>(v));
  v = fmaxf(v, dpp_mov<0x141>(v));
  return fmaxf(v, dpp_mov<0x140>(v));
}
// LayerNorm over the 64 lanes (flax/torch: biased variance, eps inside the sqrt)
__device__ __forceinline__ float layer_norm(float x, float w, float b, float eps) {
  const float mu = wave_sum(x) * (1.0f / E);
  const float d = x - mu;
  const float var = wave_sum(d * d) * (1.0f / E);
  return d / sqrtf(var + eps) * w + b;
}
// The same on the MFMA C layout: x[nt][i] = feature 16 nt + (lane & 15) of token row 4 (lane >> 4) + i
__device__ __forceinline__ void layer_norm_tile(f32x4 (&x)[4], const float (&w)[4], const float (&b)[4],
                                                float eps) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float mu = sum16((x[0][i] + x[1][i]) + (x[2][i] + x[3][i])) * (1.0f / E);
    float d[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) d[nt] = x[nt][i] - mu;
    const float var = sum16((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3])) * (1.0f / E);
    const float inv = 1.0f / sqrtf(var + eps);  // (flax multiplies by rsqrt as well)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) x[nt][i] = d[nt] * inv * w[nt] + b[nt];
  }
}

constexpr int KCH = 3;  // 16-deep k-steps of the embedding GEMM per register chunk (S, A <= 48: one chunk)
__global__ __launch_bounds__(64 * PT_WAVES) void k_pt_relabel(const iqlhip_pt_weights W,
                                                               const float *__restrict__ obs,
                                                               const float *__restrict__ act, int64_t n_rows,
                                                               const int64_t *__restrict__ win_start,
                                                               const int32_t *__restrict__ win_len,
                                                               const int32_t *__restrict__ win_t0,
                                                               int64_t n_win, int ql, float *__restrict__ out,
                                                               int skip_arg) {
  using P = Prec<false>;
#ifdef IQL_STAMPS
  const int skip = skip_arg;  // diagnostic builds: bit mask of phases left out (results are wrong)
#else
  constexpr int skip = 0;
#endif
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int S = W.state_dim, A = W.action_dim, I = W.inter_dim, NH = W.num_heads;
  const int HD = E / NH;
  const int hds = __builtin_ctz(HD);  // HD is a power of two
  const int Tmax = 2 * ql;
  const int nks_s = round_up(S, 16) / 16, nks_a = round_up(A, 16) / 16;
  // ---- LDS carve ----
  float *Vs = reinterpret_cast<float *>(smem);                          // [Tmax][VLD]
  const size_t vs_floats = (size_t)Tmax * VLD > (size_t)4 * PT_SLOTS * E ? (size_t)Tmax * VLD : (size_t)4 * PT_SLOTS * E;
  uint16_t *Kb = reinterpret_cast<uint16_t *>(Vs + vs_floats);           // [Tmax][KLD] bf16
  const size_t kb_floats = (size_t)Tmax * KLD / 2 > (size_t)PT_SLOTS * (I + 4) ? (size_t)Tmax * KLD / 2 : (size_t)PT_SLOTS * (I + 4);
  float *wsF = reinterpret_cast<float *>(Kb) + kb_floats;         // fragment-major [64][16 nks_s]
  float *waF = wsF + nks_s * 16 * E;                                     // fragment-major [64][16 nks_a]
  float *wkvF = waF + nks_a * 16 * E;                                    // [8 nt][4 ks][64 lanes][4] K | V projection
  float *qlast = wkvF + 8 * 4 * 64 * 4;                                  // [64] last token's query (bf16 values)
  float *hlast = qlast + E;                                              // [64] last token's block input LN0(x)
  int *qflag = reinterpret_cast<int *>(hlast + E);                       // [4] hand-over flag of hlast (word 0)
  float *part = hlast + E + 4;                                              // [PT_WAVES][64] cross-wave partials
  float *stat = part + PT_WAVES * E;                                     // [PT_WAVES][16 heads] x 2
  float *lg = stat + 2 * PT_WAVES * 16;                                  // [Tmax][NH] logits
  float *fvec = lg + round_up(Tmax * NH, 4);                             // [8][64] per-feature vectors
  float *pend_x = fvec + 8 * E;                                          // [PT_SLOTS][64] residual stream of parked windows
  float *pend_o = pend_x + PT_SLOTS * E;                                 // [PT_SLOTS][VLD] attention output
  float *pend_h = pend_o + PT_SLOTS * VLD;                               // [PT_SLOTS][VLD] LN1(x1)
  // [PT_SLOTS][I + 4] MLP hidden, on the K rows: tail_batch runs between a window's attention and
  // the next window's token phase, when no key is live (the region is the larger of the two)
  float *hidb = reinterpret_cast<float *>(Kb);

  // ---- weights that stay on chip for the whole queue ----
  // embedding weights as MFMA B fragments (common.h fidx): element (feature f, input k), zero padded
  for (int e = tid; e < nks_s * 16 * E; e += 64 * PT_WAVES) {
    const int k = e / E, f = e - k * E;
    wsF[fidx<P>(f, k, nks_s)] = k < S ? W.state_wT[(size_t)k * E + f] : 0.f;
  }
  for (int e = tid; e < nks_a * 16 * E; e += 64 * PT_WAVES) {
    const int k = e / E, f = e - k * E;
    waF[fidx<P>(f, k, nks_a)] = k < A ? W.action_wT[(size_t)k * E + f] : 0.f;
  }
  // K | V projection (rows 64..191 of attention.in_linear.weight [192][64]) as B fragments in
  // LDS, one 16-byte fragment per (n-tile, k-step, lane): n-tile nt < 4 -> key features 16 nt..,
  // nt >= 4 -> value features; 4 k-steps of 16
  for (int f = wave; f < 32; f += PT_WAVES)  // f = 4 nt + ks
    *reinterpret_cast<uint4 *>(wkvF + (f * 64 + lane) * 4) =
        ldg16(W.qkv_w + (size_t)(E + 16 * (f >> 2) + r) * E + 16 * (f & 3) + 4 * q);
  // V rows start finite (rows past a window's length are multiplied by zero weights), column 64 = 1
  // (the softmax denominator falls out of the P.V product), columns 65..67 = 0: never written again
  for (int e = tid; e < Tmax * VLD; e += 64 * PT_WAVES) Vs[e] = (e % VLD) == E ? 1.f : 0.f;
  for (int e = tid; e < Tmax * KLD / 2; e += 64 * PT_WAVES) reinterpret_cast<uint32_t *>(Kb)[e] = 0u;
  if (tid < 4) qflag[tid] = 0;
  // per-feature vectors of the token jobs (read from LDS in the C layout: feature 16 nt + r)
  if (tid < E) {
    fvec[tid] = W.state_b[tid], fvec[E + tid] = W.action_b[tid];
    fvec[2 * E + tid] = W.sln_w[tid], fvec[3 * E + tid] = W.sln_b[tid];
    fvec[4 * E + tid] = W.ln0_w[tid], fvec[5 * E + tid] = W.ln0_b[tid];
    fvec[6 * E + tid] = W.qkv_b[E + tid], fvec[7 * E + tid] = W.qkv_b[2 * E + tid];  // key / value bias
  }
  // one value per lane (feature = lane) for the last-token phase
  const float eps = W.eps;
  const float inv_sqrt_hd = 1.0f / sqrtf((float)HD);
  __syncthreads();

  // ---- everything behind the attention of the parked tokens (slots < n) ----
  // Slots are the M rows of 16-row MFMA tiles (rows >= PT_SLOTS alias rows 0..7: their results are
  // dropped).  B fragments come straight from the torch-layout weights ([out][in]: 16 B per lane).
  // Every stage is spread over the waves so that no wave holds more than a few fragments:
  //   S1  x1 = x + o . Wo^T + b                     [16 x 64] . [64 x 64], one n-tile per wave 0..3
  //   S2  h1 = LN1(x1)                              one slot per wave
  //   S3  hidden = relu(h1 . Win^T + b)             [16 x 64] . [64 x I], n-tiles over the waves
  //   S4  hidden . Wout^T                           [16 x I] . [I x 64], (n-tile, K quarter) per wave
  //   S5  out = value head(LNf(x1 + S4 + b))        one slot per wave
  // The key / value rows are dead here: K holds the hidden tile, V the K-split partial sums.
  const int ldh = I + 4;
  float *ksplit = Vs;  // [4 K quarters][PT_SLOTS][64]
  auto tail_batch = [&](int n, int64_t first) {
    __syncthreads();  // pend_x / pend_o of every slot written; keys and values no longer read
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
      // wave -> (n-tile wave & 3, K quarter wave >> 2); a quarter is I / 64 k-steps of 16
      const int nt = wave & 3, kq = wave >> 2, nkq = I / 64;
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
      float x2 = pend_x[wave * E + lane] + W.mlp_out_b[lane];
#pragma unroll
      for (int kq = 0; kq < 4; ++kq) x2 += ksplit[(kq * PT_SLOTS + wave) * E + lane];
      const float y = layer_norm(x2, W.lnf_w[lane], W.lnf_b[lane], eps);  // gpt.layer_norm
      const float v = wave_sum(y * W.pref_w_last[lane]) + W.pref_b_last;
      if (lane == 0) out[first + (int64_t)wave * gridDim.x] = v;
    }
    __syncthreads();  // LDS is reused by the next window
    // the first V rows served as scratch: restore what the attention relies on and the token phase
    // never writes (column 64 = 1, 65..67 = 0); the other columns are rewritten before they are read
    for (int e = tid; e < 4 * ((4 * PT_SLOTS * E + VLD - 1) / VLD); e += 64 * PT_WAVES)
      if ((e >> 2) < Tmax) Vs[(e >> 2) * VLD + E + (e & 3)] = (e & 3) == 0 ? 1.f : 0.f;
  };
  int nslot = 0;
  int64_t batch_first = 0;
  int seq = 0;  // window counter of this work-group: the value the hand-over flag takes

  for (int64_t win = blockIdx.x; win < n_win; win += gridDim.x) {
    const int64_t start = win_start[win];
    const int len = win_len[win];
    const int t0 = win_t0 ? win_t0[win] : 0;  // timestep of the window's first transition
    const int T = 2 * len;
    const int nmt = (len + 15) >> 4;  // 16-token tiles per kind
    // ================= every token: embedding, LayerNorms, key / value =================
    // One job = 16 tokens of one kind; with <= 16 jobs (query_length <= 128) every wave has at
    // most one.  The job of the last action tile hands the last token's block input to the
    // query wave (below) through LDS.
    ++seq;
    const int njobs = 2 * nmt, qjob = njobs - 1;
    for (int job = wave; job < njobs; job += PT_WAVES) {
      const int kind = job >= nmt ? 1 : 0;  // 0: state tokens, 1: action tokens
      const int mt = kind ? job - nmt : job;
      const float *src = kind ? act : obs;
      const int D = kind ? A : S, nks = kind ? nks_a : nks_s;
      const float *wF = kind ? waF : wsF;
      // A fragments from the dataset rows: token 16 mt + r of the window (clamped past len: the
      // results of those rows are dropped), inputs 16 ks + 4 q .. + 3 (clamped past D: zeroed)
      const int kr = 16 * mt + r < len ? 16 * mt + r : len - 1;
      const float *rowp = src + (size_t)(start + kr) * D;
      // accumulators start from bias + timestep embedding of tokens 4 q + i (timestep = t0 +
      // position in the window; t0 = 0 in ref:1281,1291, the true step in custom_offline:209)
      f32x4 x[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = 16 * mt + 4 * q + i < len ? 16 * mt + 4 * q + i : len - 1;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) x[nt][i] = (skip & 4) ? 0.5f : ldg(W.temb + (size_t)(t0 + k) * E + 16 * nt + r);
      }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const float bias = fvec[kind * E + 16 * nt + r];
#pragma unroll
        for (int i = 0; i < 4; ++i) x[nt][i] += bias;
      }
      for (int ks0 = 0; ks0 < nks; ks0 += KCH) {
        uint4 a[KCH];
#pragma unroll
        for (int kk = 0; kk < KCH; ++kk) {
          float v[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int k = 16 * (ks0 + kk) + 4 * q + c;
            const float xv = (skip & 4) ? 0.25f : ldg(rowp + (k < D ? k : D - 1));
            v[c] = k < D ? xv : 0.f;
          }
          a[kk] = __builtin_bit_cast(uint4, make_float4(v[0], v[1], v[2], v[3]));
        }
#pragma unroll
        for (int kk = 0; kk < KCH; ++kk) {
          if (ks0 + kk < nks) {
            float4 bf[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
              bf[nt] = *reinterpret_cast<const float4 *>(wF + frag_off<P>(nt, ks0 + kk, nks, lane));
            const float4 af = __builtin_bit_cast(float4, a[kk]);
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
              for (int nt = 0; nt < 4; ++nt)
                x[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[c], bf[nt][c], x[nt], 0, 0, 0);
          }
        }
      }
      float lw[4], lb[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) lw[nt] = fvec[2 * E + 16 * nt + r], lb[nt] = fvec[3 * E + 16 * nt + r];
      if (!(skip & 2)) layer_norm_tile(x, lw, lb, eps);  // stacked_layer_norm
      f32x4 h[4] = {x[0], x[1], x[2], x[3]};
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) lw[nt] = fvec[4 * E + 16 * nt + r], lb[nt] = fvec[5 * E + 16 * nt + r];
      if (!(skip & 2)) layer_norm_tile(h, lw, lb, eps);  // block pre-LN
      // the window's last token (action token len - 1): its residual stream is parked for tail_batch
      if (job == qjob) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (16 * mt + 4 * q + i == len - 1) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) pend_x[nslot * E + 16 * nt + r] = x[nt][i], hlast[16 * nt + r] = h[nt][i];
          }
        }
        // hand the block input of the last token to the query wave: data first, then the flag
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_store(qflag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      // h -> A fragments through the job's own V rows (token j of the tile -> row 2 (16 mt + j) + kind)
      float *scr = Vs + (size_t)(2 * 16 * mt + kind) * VLD;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (16 * mt + 4 * q + i < ql) scr[(size_t)(4 * q + i) * 2 * VLD + 16 * nt + r] = h[nt][i];
      // same wave, LDS operations execute in order: only the compiler must keep the order
      asm volatile("" ::: "memory");
      uint4 ha[4];
      const int rr = 16 * mt + r < ql ? r : 0;  // rows past the LDS image (never stored) re-read row 0
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        ha[ks] = *reinterpret_cast<const uint4 *>(scr + (size_t)rr * 2 * VLD + 16 * ks + 4 * q);
      asm volatile("" ::: "memory");  // the V rows written below are the scratch read above
      // keys (half 0), then values (half 1): four n-tiles at a time bound the live B fragments;
      // component-major: the four K = 4 MFMAs of one fragment pair accumulate into the same
      // registers, the other three n-tiles issue between them (128 cycles vs the 40 of latency)
      f32x4 kv[8];
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) kv[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int ks = 0; ks < ((skip & 16) ? 0 : 4); ++ks) {
          const float4 af = __builtin_bit_cast(float4, ha[ks]);
          float4 bw[4];
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
            bw[nt] = *reinterpret_cast<const float4 *>(wkvF + ((4 * (4 * half + nt) + ks) * 64 + lane) * 4);
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
              kv[4 * half + nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[c], bw[nt][c], kv[4 * half + nt], 0, 0, 0);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = 16 * mt + 4 * q + i;
        if (k < len && !(skip & 8)) {
          const int t = 2 * k + kind;
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            Kb[(size_t)t * KLD + 16 * nt + r] = f2bf(kv[nt][i] + fvec[6 * E + 16 * nt + r]);
            Vs[(size_t)t * VLD + 16 * nt + r] = kv[4 + nt][i] + fvec[7 * E + 16 * nt + r];
          }
        }
      }
    }
    if (wave == PT_WAVES - 1) {
      // ---- the last token's query (rows 0..63 of attention.in_linear): lane = feature, this
      // lane's weight row requested before the wait (64 registers nothing else needs here), the
      // dot product in input order; rounded to bf16 (ops.py:74) ----
      uint4 wq[16];
#pragma unroll
      for (int g = 0; g < 16; ++g) wq[g] = ldg16(W.qkv_w + (size_t)lane * E + 4 * g);
      float acc = W.qkv_b[lane];
      while (__hip_atomic_load(qflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != seq)
        __builtin_amdgcn_s_sleep(1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const float4 hv = *reinterpret_cast<const float4 *>(hlast + 4 * g);
        const float4 wv = __builtin_bit_cast(float4, wq[g]);
        acc = fmaf(hv.x, wv.x, acc), acc = fmaf(hv.y, wv.y, acc);
        acc = fmaf(hv.z, wv.z, acc), acc = fmaf(hv.w, wv.w, acc);
      }
      qlast[lane] = rbf(acc);
    }
    __syncthreads();
    // ================= last token: query, attention over all keys =================
    if (skip & 1) {
      if (tid == 0) out[win] = 0.f;
      continue;
    }
    // ---- logits of the one query over all keys: [16 heads x 64] . [64 x T] on the bf16 matrix
    // cores.  Row m of the A operand is the query masked to head m's features, so C[m][t] is head
    // m's bf16 q.k product (exact products, fp32 accumulation, rounded to bf16 as ops.py:76) ----
    {
      uint4 qa[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int f0 = 32 * ks + 8 * q;
        const float4 v0 = *reinterpret_cast<const float4 *>(qlast + f0);
        const float4 v1 = *reinterpret_cast<const float4 *>(qlast + f0 + 4);
        const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        uint32_t w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const uint32_t lo = ((f0 + 2 * e) >> hds) == r ? f2bf(v[2 * e]) : 0u;
          const uint32_t hi = ((f0 + 2 * e + 1) >> hds) == r ? f2bf(v[2 * e + 1]) : 0u;
          w[e] = lo | (hi << 16);
        }
        qa[ks] = make_uint4(w[0], w[1], w[2], w[3]);
      }
      float lmax[4] = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
      for (int j = wave; 16 * j < T; j += PT_WAVES) {
        const int t = 16 * j + r;
        const uint16_t *krow = Kb + (size_t)(t < T ? t : T - 1) * KLD + 8 * q;
        const uint4 k0 = *reinterpret_cast<const uint4 *>(krow), k1 = *reinterpret_cast<const uint4 *>(krow + 32);
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        Prec<true>::mma(qa[0], k0, c);
        Prec<true>::mma(qa[1], k1, c);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (4 * q + i < NH && t < T) {
            const float sc = rbf(rbf(c[i]) * inv_sqrt_hd);  // bf16 product tensor, bf16 scale (ops.py:76-79)
            lg[t * NH + 4 * q + i] = sc;
            lmax[i] = fmaxf(lmax[i], sc);
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float m = max16(lmax[i]);
        if (r == 0 && 4 * q + i < NH) stat[wave * 16 + 4 * q + i] = m;
      }
    }
    __syncthreads();
    // ---- softmax numerators and P.V: [16 heads x T] . [T x 64 (+ the ones column)] on the exact
    // fp32 matrix cores, the keys split over the waves; C[m][f] is wanted for m = head of f ----
    const int head = lane >> hds;
    {
      float gmax = -3.0e38f;
      if (r < NH) {
#pragma unroll
        for (int w = 0; w < PT_WAVES; ++w) gmax = fmaxf(gmax, stat[w * 16 + r]);
      }
      f32x4 oc[5];
#pragma unroll
      for (int nt = 0; nt < 5; ++nt) oc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int ks = wave; 4 * ks < T; ks += PT_WAVES) {
        const int t = 4 * ks + q;
        float a = 0.f;
        if (r < NH && t < T) a = expf(lg[t * NH + r] - gmax);
        const float *vrow = Vs + (size_t)(t < T ? t : T - 1) * VLD + r;
        float bv[5];
#pragma unroll
        for (int nt = 0; nt < 5; ++nt) bv[nt] = vrow[16 * nt];
#pragma unroll
        for (int nt = 0; nt < 5; ++nt) oc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[nt], oc[nt], 0, 0, 0);
      }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int h = (16 * nt + r) >> hds;
        if ((h >> 2) == q) {
          const int i = h & 3;
          part[wave * E + 16 * nt + r] = i == 0 ? oc[nt][0] : i == 1 ? oc[nt][1] : i == 2 ? oc[nt][2] : oc[nt][3];
        }
      }
      if (r == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (4 * q + i < NH) stat[PT_WAVES * 16 + wave * 16 + 4 * q + i] = oc[4][i];
      }
    }
    __syncthreads();
    // Every LDS word the next window's token phase writes (keys, values, the query) has been read
    // by now; what is read below (part, stat) is next written behind two more barriers.
    if (wave == 0) {  // park the attention output: the rest of the block runs in tail_batch
      float o = 0.f, den = 0.f;
#pragma unroll
      for (int w = 0; w < PT_WAVES; ++w) o += part[w * E + lane], den += stat[PT_WAVES * 16 + w * 16 + head];
      pend_o[nslot * VLD + lane] = o / den;
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
  const size_t Tmax = 2 * (size_t)ql;
  const size_t ks = (size_t)round_up(W.state_dim, 16) + round_up(W.action_dim, 16);
  const size_t kb = Tmax * KLD * 2 > (size_t)PT_SLOTS * (W.inter_dim + 4) * 4 ? Tmax * KLD * 2
                                                                           : (size_t)PT_SLOTS * (W.inter_dim + 4) * 4;
  const size_t vs = Tmax * VLD * 4 > (size_t)4 * PT_SLOTS * E * 4 ? Tmax * VLD * 4 : (size_t)4 * PT_SLOTS * E * 4;
  return vs + kb + ks * E * 4 + 8 * 4 * 64 * 16 +
         (2 * E + 4 + PT_WAVES * E + 2 * PT_WAVES * 16 + round_up((int)Tmax * W.num_heads, 4) + 8 * E + PT_SLOTS * E +
          2 * PT_SLOTS * VLD) * 4 + 64;
}

hipError_t launch_pt(const iqlhip_pt_weights &W, const float *obs, const float *act, int64_t n_rows,
                     const int64_t *win_start, const int32_t *win_len, const int32_t *win_t0, int64_t n_win,
                     int ql, float *out, hipStream_t st) {
  const size_t sm = pt_smem_bytes(W, ql);
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  // persistent work-groups: as many as stay resident
  const int per_cu = 1;  // 16 waves, ~150 KB of LDS: one work-group per CU
  int64_t grid = (int64_t)cus * per_cu;
  if (n_win < grid) grid = n_win;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_pt_relabel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) return e;
  int skip = 0;
#ifdef IQL_STAMPS
  if (const char *sk = getenv("IQLHIP_PT_SKIP")) skip = atoi(sk);
#endif
  hipLaunchKernelGGL(k_pt_relabel, dim3((unsigned)grid), dim3(64 * PT_WAVES), sm, st, W, obs, act, n_rows,
                     win_start, win_len, win_t0, n_win, ql, out, skip);
  return hipGetLastError();
}

}  // namespace iqlhip
