// General layer-wise IQL step for MI355X (gfx950): the shapes the tuned three-Linear step
// (iql_step.hip) is not built for -- n_hidden = 1..6 hidden layers of any width 1..1024
// (ref:417-449 MLP, :452-543 the networks; ref:581-662 the step).  Same arithmetic (step_math.h),
// same arenas, same index / dropout streams; three plain launches per step:
//
//   kd_forward   (2E+3 evaluations) x B/16 slabs, four or eight waves each: the slab's 16 transitions gathered
//                from the packed replay rows, then every Linear on MFMA -- A fragments from a
//                row-major LDS image of the previous layer's output, B fragments straight from the
//                row-major compute copy W[n][k] in L2 -- with the hidden activations of the trained
//                nets stored feature-major for the other two kernels.
//   kd_backward  (E+2 trained nets) x B/16 slabs, four or eight waves each: loss terms, d(out), then
//                dZ_{l-1} = (dZ_l W_l) * relu' layer by layer (B fragments from the transposed copies
//                Wt[k][n]), deltas stored feature-major; per-slab loss partial sums.
//   kd_update    64 x 64 tiles of every weight matrix: dW = dZ^T X (K = batch, both operands
//                feature-major planes), Adam, the compute / transposed / target copies; biases by the
//                tiles of the first column block; one block for the losses and log_std.
//
// This path is about coverage, not speed: a work-group is one latency chain and re-reads its
// network's weights from L2.  Every shipped configuration (n_hidden = 2, hidden_dim 64 / 128 / 256)
// runs on the tuned step instead.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/iqlhip.h"
#include "common.h"
#include "iql_deep.h"
#include "step_math.h"

namespace iqlhip {

namespace {

template <bool BF16>
__device__ __forceinline__ void put4T(typename Prec<BF16>::T *dst, const float v[4]) {
  if constexpr (BF16) {
    uint2 u;
    u.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
    u.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
    stg8(dst, u);
  } else {
    stg16(dst, make_float4(v[0], v[1], v[2], v[3]));
  }
}
template <bool BF16>
__device__ __forceinline__ void get4T(const typename Prec<BF16>::T *src, float v[4]) {
  if constexpr (BF16) {
    const uint2 u = ldg8(src);
    v[0] = bf2f((uint16_t)(u.x & 0xffff)), v[1] = bf2f((uint16_t)(u.x >> 16));
    v[2] = bf2f((uint16_t)(u.y & 0xffff)), v[3] = bf2f((uint16_t)(u.y >> 16));
  } else {
    const float4 f = __builtin_bit_cast(float4, ldg16(src));
    v[0] = f.x, v[1] = f.y, v[2] = f.z, v[3] = f.w;
  }
}
template <bool BF16>
__device__ __forceinline__ float frag_sum(const uint4 &f) {
  if constexpr (BF16) {
    const uint32_t w[4] = {f.x, f.y, f.z, f.w};
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += bf2f((uint16_t)(w[i] & 0xffff)) + bf2f((uint16_t)(w[i] >> 16));
    return s;
  } else {
    const float4 v = __builtin_bit_cast(float4, f);
    return (v.x + v.y) + (v.z + v.w);
  }
}

// Philox stream of the Dropout behind hidden layer l (0-based): layers 0 / 1 use the tuned step's
// streams 1 / 2, deeper ones 8 + l (3 is the stand-alone MLP's): oracle/philox.py
__device__ __forceinline__ uint32_t drop_stream(int l) { return l < 2 ? STREAM_DROPOUT1 + (uint32_t)l : 8u + (uint32_t)l; }

// keep mask of the 4 rows a lane owns (rows 4 rowblk .. +3) of hidden unit col of hidden layer l
__device__ __forceinline__ void deep_keep4(const DeepDesc &D, const DeepStep &A, int l, int rowblk, int col,
                                           bool keep[4]) {
  if (col >= D.H) {
    keep[0] = keep[1] = keep[2] = keep[3] = false;
    return;
  }
  if (A.drop_keep) {
    const uint8_t *m = A.drop_keep + ((size_t)l * D.B + (size_t)rowblk * 4) * D.H + col;
#pragma unroll
    for (int i = 0; i < 4; ++i) keep[i] = ldg(m + (size_t)i * D.H) != 0;
  } else {
    const Philox4 ph = philox4x32_10((uint32_t)(rowblk * D.H + col), (uint32_t)A.step,
                                     (uint32_t)((uint64_t)A.step >> 32), drop_stream(l), (uint32_t)D.seed,
                                     (uint32_t)(D.seed >> 32));
    keep[0] = ph.x >= D.drop_thr, keep[1] = ph.y >= D.drop_thr;
    keep[2] = ph.z >= D.drop_thr, keep[3] = ph.w >= D.drop_thr;
  }
}

// Two 16 x 16 output tiles that share their A operand: acc_j = A[16][K] (LDS, row-major) x B_j[16][K]^T
// (global, fragment-major: brow_j = this lane's fragment of k-step 0, the next k-step one fragment on --
// a wave-wide fragment load is one contiguous 1 KiB read; a row-major image made it 16 half-used lines
// and the weight stream, not latency, bounds these kernels); K = nk MFMA steps.  Eight steps of operands (16 global + 8 LDS fragments)
// are requested before the first MFMA of a chunk.
template <bool BF16>
__device__ __forceinline__ void tile_mma2(const typename Prec<BF16>::T *arow, const typename Prec<BF16>::T *brow0,
                                          const typename Prec<BF16>::T *brow1, int nk, f32x4 &acc0, f32x4 &acc1) {
  using P = Prec<BF16>;
  acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
  int ks = 0;
  for (; ks + 8 <= nk; ks += 8) {  // (a 256-wide layer in bf16 is ONE such chunk: one memory round trip per tile pair)
    uint4 b0[8], b1[8], a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      b0[u] = ldg16(brow0 + (size_t)(ks + u) * 64 * P::EPV);
      b1[u] = ldg16(brow1 + (size_t)(ks + u) * 64 * P::EPV);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] = *reinterpret_cast<const uint4 *>(arow + (size_t)(ks + u) * P::KM);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      P::mma(a[u], b0[u], acc0);
      P::mma(a[u], b1[u], acc1);
    }
  }
  for (; ks + 4 <= nk; ks += 4) {
    uint4 b0[4], b1[4], a[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      b0[u] = ldg16(brow0 + (size_t)(ks + u) * 64 * P::EPV);
      b1[u] = ldg16(brow1 + (size_t)(ks + u) * 64 * P::EPV);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) a[u] = *reinterpret_cast<const uint4 *>(arow + (size_t)(ks + u) * P::KM);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      P::mma(a[u], b0[u], acc0);
      P::mma(a[u], b1[u], acc1);
    }
  }
  for (; ks < nk; ++ks) {
    const uint4 b0 = ldg16(brow0 + (size_t)ks * 64 * P::EPV), b1 = ldg16(brow1 + (size_t)ks * 64 * P::EPV);
    const uint4 a = *reinterpret_cast<const uint4 *>(arow + (size_t)ks * P::KM);
    P::mma(a, b0, acc0);
    P::mma(a, b1, acc1);
  }
}

// The waves of a slab share its LDS images; wave w owns the tile pairs w, w + W, ...  W = 4 or 8 (blockDim):
// eight when a layer has that many pairs and the launch leaves CUs idle anyway (one seed at batch 256: 24.8k
// against 20.8k steps/s at three hidden layers of 256 units, 18.1k against 14.6k at 512 units); four when
// the launch fills the chip (E = 4 at batch 1024: 14.7k against 13.4k) or the layers are narrow (96 units:
// 32.2k against 29.0k) -- deep_threads() below.
// (two instantiations, THREADS = 256 / 512: one binary bounded for 512 threads costs the 256-thread launches
// 10-20 %)

// All Linear layers of one evaluation for the 16 rows whose inputs sit in `in` (LDS, row-major,
// zero padded to the first layer's Kpad).  hidden(l, col, a[4]) receives the activations behind
// hidden layer l (after ReLU and Dropout), final(col, z[4]) the outputs (after tanh).
template <bool BF16, class Hidden, class Final>
__device__ __forceinline__ void deep_layers(const DeepDesc &D, const DeepEval &N, const DeepStep *A, bool dropout,
                                            int rowblk, typename Prec<BF16>::T *in, typename Prec<BF16>::T *out,
                                            const DeepNet *TN, Hidden hidden, Final final) {
  using P = Prec<BF16>;
  using T = typename P::T;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, q = lane >> 4, ldw = D.lds_w, NL = D.NL;
  for (int l = 0; l < NL; ++l) {
    const DeepLin &Lk = N.lin[l];
    const int Kpad = Lk.Kpad, ntile = Lk.Npad / 16, Nn = Lk.N, nk = Kpad / P::KM;
    const T *W = reinterpret_cast<const T *>(Lk.w);
    const float *bias = Lk.b;
    const bool last = l == NL - 1;
    // (every descriptor field of this layer is read HERE, into registers: behind a global store the compiler
    // must assume the descriptor changed and would fetch the field again -- a scalar-load round trip per tile)
    T *const plane = (TN && !last) ? reinterpret_cast<T *>(TN->hT[l + 1]) : nullptr;
    const bool tanh_out = N.tanh_out != 0;
    const T *arow = in + (size_t)r16 * ldw + q * P::EPV;
    for (int nt = 2 * wave; nt < ntile; nt += 2 * ((int)blockDim.x >> 6)) {
      const bool two = nt + 1 < ntile;  // (the output layer may have a single tile: its twin is computed and dropped)
      const int col0 = nt * 16 + r16, col1 = two ? col0 + 16 : col0;
      f32x4 acc[2];
      // (requested ahead of the weight fragments: a load behind the MFMAs is a memory round trip per tile pair)
      const float braw[2] = {ldg(bias + (col0 < Nn ? col0 : 0)), ldg(bias + (col1 < Nn ? col1 : 0))};
      tile_mma2<BF16>(arow, W + frag_off<P>(nt, 0, nk, lane), W + frag_off<P>(two ? nt + 1 : nt, 0, nk, lane), nk, acc[0],
                      acc[1]);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (j == 1 && !two) break;
        const int col = j ? col1 : col0;
        // nn.Linear under autocast: bf16 inputs, weights AND bias, fp32 accumulation, bf16 result
        const float bv = col < Nn ? P::round(braw[j]) : 0.f;
        float z[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) z[i] = P::round(acc[j][i] + bv);
        if (!last) {
#pragma unroll
          for (int i = 0; i < 4; ++i) z[i] = fmaxf(z[i], 0.f);
          if (dropout) {
            bool keep[4];
            deep_keep4(D, *A, l, rowblk + q, col, keep);
#pragma unroll
            for (int i = 0; i < 4; ++i) z[i] = P::round(z[i] * (keep[i] ? D.drop_scale : 0.f));
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) out[(size_t)(4 * q + i) * ldw + col] = P::from_f32(z[i]);
          hidden(plane, col, z);
        } else {
          if (tanh_out) {
#pragma unroll
            for (int i = 0; i < 4; ++i) z[i] = P::round(tanhf(z[i]));
          }
          final(col, z);
        }
      }
    }
    __syncthreads();
    T *t_ = in;
    in = out, out = t_;
  }
}

__device__ __forceinline__ int64_t deep_row_index(const DeepDesc &D, const DeepStep &A, int row) {
  int64_t ix;
  if (A.idx_mode == 1)
    ix = ldg(A.idx + row);
  else if (A.idx_mode == 2)
    ix = row;
  else
    ix = philox_index(D.seed, (uint64_t)A.step, (uint32_t)row, (uint64_t)A.n_rows);
  return ix < 0 ? 0 : (ix >= A.n_rows ? A.n_rows - 1 : ix);
}

// ------------------------------------------------------------------------
// kd_forward: grid (B/16, 2E+3), four waves.
// ------------------------------------------------------------------------
template <bool BF16, int THREADS>
__global__ __launch_bounds__(THREADS) void kd_forward(const DeepDesc *__restrict__ Dp, const DeepStep A) {
  using P = Prec<BF16>;
  using T = typename P::T;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const DeepDesc &D = *Dp;
  const DeepEval &N = D.ev[blockIdx.y];
  const int row0 = blockIdx.x * 16, q = (threadIdx.x & 63) >> 4, ldw = D.lds_w, BP = D.BP;
  T *bufA = reinterpret_cast<T *>(smem), *bufB = bufA + (size_t)16 * ldw;
  // ---- the slab's inputs: 16 threads per row (the first 256 threads) ----
  if (threadIdx.x < 256) {
    const int r = threadIdx.x >> 4, c0 = threadIdx.x & 15, row = row0 + r;
    const int64_t ix = deep_row_index(D, A, row);
    const float *src = A.rows + (size_t)ix * A.row_stride;
    const int K0 = N.lin[0].Kpad, in_dim = N.in_dim, in_off = N.in_off;
    T *xT = reinterpret_cast<T *>(D.net[0].hT[0]);
    for (int c = c0; c < K0; c += 16) {
      const float v = c < in_dim ? ldg(src + in_off + c) : 0.f;
      const T tv = P::from_f32(v);
      bufA[(size_t)r * ldw + c] = tv;
      if (N.stage && c < in_dim) stg(xT + (size_t)c * BP + row, tv);  // (s | a) feature-major: layer-0 X operand
    }
    if (N.stage) {  // this evaluation's input is (s | a): reward, done and the fp32 actions ride along
      const int S = D.S, SA = D.S + D.A;
      for (int c = c0; c < D.A; c += 16) stg(D.actf + (size_t)row * D.A + c, ldg(src + S + c));
      if (c0 < 2) stg(D.rd + (size_t)row * 2 + c0, ldg(src + SA + c0));
    }
  }
  __syncthreads();
  const int slot = N.train_slot;
  const DeepNet *TN = slot >= 0 ? &D.net[slot] : nullptr;
  float *outs = D.outs;
  const int out_dim = N.out_dim, out_col = N.out_col;
  deep_layers<BF16>(
      D, N, &A, N.dropout != 0, row0 / 4, bufA, bufB, TN,
      [&](T *plane, int col, const float a[4]) {
        if (plane) put4T<BF16>(plane + (size_t)col * BP + row0 + 4 * q, a);
      },
      [&](int col, const float z[4]) {
        if (col < out_dim) stg16(outs + (size_t)(out_col + col) * BP + row0 + 4 * q, make_float4(z[0], z[1], z[2], z[3]));
      });
}

// Forward on dense inputs (iqlhip_forward): n rows of s [n][S] (and a [n][A]); out[row][col0 + j],
// row stride out_ld.  Eval mode: no Dropout.
template <bool BF16>
__global__ __launch_bounds__(256) void kd_infer(const DeepDesc *__restrict__ Dp, int ev, const float *__restrict__ s,
                                               const float *__restrict__ a, int64_t n, float *__restrict__ out,
                                               int out_ld, int col0) {
  using P = Prec<BF16>;
  using T = typename P::T;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const DeepDesc &D = *Dp;
  const DeepEval &N = D.ev[ev];
  const int q = (threadIdx.x & 63) >> 4, ldw = D.lds_w;
  const int64_t row0 = (int64_t)blockIdx.x * 16;
  T *bufA = reinterpret_cast<T *>(smem), *bufB = bufA + (size_t)16 * ldw;
  if (threadIdx.x < 256) {
    const int r = threadIdx.x >> 4, c0 = threadIdx.x & 15;
    const int64_t row = row0 + r < n ? row0 + r : n - 1;
    const int K0 = N.lin[0].Kpad, in_dim = N.in_dim, S = D.S;
    for (int c = c0; c < K0; c += 16) {
      float v = 0.f;
      if (c < in_dim) v = c < S ? ldg(s + row * S + c) : ldg(a + row * D.A + (c - S));
      bufA[(size_t)r * ldw + c] = P::from_f32(v);
    }
  }
  __syncthreads();
  const int out_dim = N.out_dim;
  deep_layers<BF16>(
      D, N, nullptr, false, 0, bufA, bufB, nullptr, [&](T *, int, const float *) {},
      [&](int col, const float z[4]) {
        if (col < out_dim) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int64_t row = row0 + 4 * q + i;
            if (row < n) stg(out + row * out_ld + col0 + col, z[i]);
          }
        }
      });
}

// ------------------------------------------------------------------------
// kd_backward: grid (B/16, E+2), four waves (the loss terms by the first, the layer walk by all).
// ------------------------------------------------------------------------
template <bool BF16, int THREADS>
__global__ __launch_bounds__(THREADS) void kd_backward(const DeepDesc *__restrict__ Dp, const DeepStep A) {
  using P = Prec<BF16>;
  using T = typename P::T;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const DeepDesc &D = *Dp;
  const int net = blockIdx.y, slab = blockIdx.x, row0 = slab * 16;
  const DeepNet &N = D.net[net];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, q = lane >> 4;
  const int ldw = D.lds_w, BP = D.BP, L = D.NL - 1;
  T *dzin = reinterpret_cast<T *>(smem), *dzout = dzin + (size_t)16 * ldw;
  // the output layer's delta tile is narrower than one MFMA K step: zero what the GEMM reads beyond it
  for (int e = threadIdx.x; e < 16 * ldw; e += (int)blockDim.x) dzin[e] = P::from_f32(0.f);
  __syncthreads();
  // ---- loss terms and d(loss)/d(out) (ref:581-637) ----
  if (wave == 0) {
    const int E = D.E, Aq = D.A, odim = N.N[L];
    const float *outs = D.outs;
    const float fB = (float)D.B;
    float lsum = 0.f;
    for (int nt = 0; nt < N.Npad[L] / 16; ++nt) {
      const int j = nt * 16 + r16, jc = j < Aq ? j : Aq - 1;
      const bool valid = j < odim;
      const float ls = D.deterministic ? 0.f : ldg(D.params + D.off_log_std + jc);
      float dz[4], gs = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int b = row0 + 4 * q + i;
        LossIn x;
#pragma unroll
        for (int e = 0; e < MAX_CRITICS; ++e) x.qt[e] = ldg(outs + (size_t)(D.out_qt + (e < E ? e : 0)) * BP + b);
        x.vv = ldg(outs + (size_t)D.out_v * BP + b);
        x.nv = ldg(outs + (size_t)D.out_nv * BP + b);
        x.qv = ldg(outs + (size_t)(net < E ? net : 0) * BP + b);
        x.mean = ldg(outs + (size_t)(D.out_mean + jc) * BP + b);
        x.act = ldg(D.actf + (size_t)b * Aq + jc);
        x.ls = ls;
        x.rew = ldg(D.rd + (size_t)b * 2), x.done = ldg(D.rd + (size_t)b * 2 + 1);
        float d3, lt, g;
        loss_terms<BF16>(D, net, x, fB, d3, lt, g);
        dz[i] = valid ? d3 : 0.f;
        lsum += valid ? lt : 0.f;
        gs += valid ? g : 0.f;
        dzin[(size_t)(4 * q + i) * ldw + j] = P::from_f32(dz[i]);
      }
      put4T<BF16>(reinterpret_cast<T *>(N.dzT[L]) + (size_t)j * BP + row0 + 4 * q, dz);
      if (net == D.net_a && !D.deterministic) {
        gs = xor32_sum(xor16_sum(gs));  // over the four row groups: this slab's 16 rows
        if (q == 0 && valid) stg(D.lsp + (size_t)slab * Aq + j, gs);
      }
    }
    lsum = lane_sum<64>(lsum);
    if (lane == 0) stg(D.lossp + (size_t)net * D.nslab + slab, lsum);
  }
  __syncthreads();
  // ---- dZ_{l-1} = (dZ_l W_l) [* dropout] * relu'(h_{l-1}) ----
  const bool dropout = net == D.net_a && D.has_dropout;
  for (int l = L; l >= 1; --l) {
    const T *Wt = reinterpret_cast<const T *>(N.wt[l]);
    const T *hp = reinterpret_cast<const T *>(N.hT[l]);
    T *zp = reinterpret_cast<T *>(N.dzT[l - 1]);
    const int NK = N.NKpad[l], nk = NK / P::KM, Kp = N.Kpad[l];
    const T *arow = dzin + (size_t)r16 * ldw + q * P::EPV;
    for (int kt = 2 * wave; kt < Kp / 16; kt += 2 * ((int)blockDim.x >> 6)) {  // (Kp = Hp: an even number of tiles)
      float h[2][4];
      f32x4 acc[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) get4T<BF16>(hp + (size_t)((kt + j) * 16 + r16) * BP + row0 + 4 * q, h[j]);
      tile_mma2<BF16>(arow, Wt + frag_off<P>(kt, 0, nk, lane), Wt + frag_off<P>(kt + 1, 0, nk, lane), nk, acc[0], acc[1]);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = (kt + j) * 16 + r16;
        float g[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) g[i] = P::round(acc[j][i]);
        if (dropout) {
          bool keep[4];
          deep_keep4(D, A, l - 1, row0 / 4 + q, col, keep);
#pragma unroll
          for (int i = 0; i < 4; ++i) g[i] = P::round(g[i] * (keep[i] ? D.drop_scale : 0.f));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          g[i] = h[j][i] > 0.f ? g[i] : 0.f;
          dzout[(size_t)(4 * q + i) * ldw + col] = P::from_f32(g[i]);
        }
        put4T<BF16>(zp + (size_t)col * BP + row0 + 4 * q, g);
      }
    }
    __syncthreads();
    T *t_ = dzin;
    dzin = dzout, dzout = t_;
  }
}

// ------------------------------------------------------------------------
// kd_update: one work-group (4 waves) per 64 x 64 tile of a weight matrix.
// ------------------------------------------------------------------------
// sum of n floats `stride` apart, added in index order; eight loads are in flight at a time (a rolled
// load-add loop is one memory round trip per element: 16 slabs were 11 us of the update launch)
__device__ __forceinline__ float ordered_sum(const float *p, int stride, int n) {
  float s = 0.f;
  for (int k0 = 0; k0 < n; k0 += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = ldg(p + (size_t)(k0 + u < n ? k0 + u : n - 1) * stride);
#pragma unroll
    for (int u = 0; u < 8; ++u) s += k0 + u < n ? v[u] : 0.f;
  }
  return s;
}

template <bool BF16>
__device__ __forceinline__ void deep_misc(const DeepDesc &D, const DeepStep &A) {
  __shared__ float lm[MAX_TRAIN];
  const int tid = threadIdx.x, nslab = D.nslab;
  if (tid < D.ntrain) {
    lm[tid] = ordered_sum(D.lossp + (size_t)tid * nslab, 1, nslab) / (float)D.B;
  }
  __syncthreads();
  if (tid == 0 && A.losses_out) {
    float ql = 0.f;
    for (int e = 0; e < D.E; ++e) ql += lm[e];  // q_loss = sum(mse) / E (ref:606)
    stg(A.losses_out + 0, lm[D.net_v]);
    stg(A.losses_out + 1, ql / (float)D.E);
    stg(A.losses_out + 2, lm[D.net_a]);
  }
  if (!D.deterministic && tid < D.A) {  // log_std (ref:452-474): std = exp(clamp(log_std))
    const float gs = ordered_sum(D.lsp + tid, D.A, nslab);
    const int64_t o = D.off_log_std + tid;
    float p = ldg(D.params + o), m = ldg(D.exp_avg + o), v = ldg(D.exp_avg_sq + o);
    const bool inside = p >= -20.f && p <= 2.f;
    const float sd = expf(fminf(fmaxf(p, -20.f), 2.f));
    const float g = inside ? gs * sd : 0.f;
    if (D.grads) stg(D.grads + o, g);
    adam_apply<(BF16 && IQL_ADAM_FAST)>(p, m, v, g, A.coef, A.coef.neg_step[2]);
    stg(D.params + o, p), stg(D.exp_avg + o, m), stg(D.exp_avg_sq + o, v);
  }
}

// Columns of a tile: DEEP_TQ 16-wide groups of the Q operand per wave (the P operand gives every wave 16 rows).
// Two instantiations: 64 x 16 tiles -- four times the work-groups, a quarter of the chain per lane: the launch
// is a latency chain for one seed at batch 256 (update 13.1 -> 10.4 us at three hidden layers of 256 units,
// 10.2 -> 5.4 us at one) -- and 64 x 64 tiles from batch 1024 on, where the operand panels (batch x 2 bytes per
// row) are the traffic (E = 4 at batch 1024: 26.7 us against 32.3), and from 768 units on (H = 1024: 43.8 us
// against 46.6), provided such tiles still make ~150 work-groups (deep_create).
template <bool BF16, int DEEP_TQ>
__global__ __launch_bounds__(256) void kd_update(const DeepDesc *__restrict__ Dp, const DeepItem *__restrict__ items,
                                                 const DeepStep A) {
  using P = Prec<BF16>;
  using T = typename P::T;
  constexpr bool AF = BF16 && IQL_ADAM_FAST;
  const DeepDesc &D = *Dp;
  const DeepItem it = items[blockIdx.x];
  if (it.net < 0) {
    deep_misc<BF16>(D, A);
    return;
  }
  // Two orientations of the same tile.  Rows of the [out][in] tensors that start on 16-byte boundaries
  // (in-features a multiple of 4: every hidden layer of such a width) are computed TRANSPOSED,
  // dW^T = X^T dZ: in the MFMA C layout a lane then holds 4 consecutive in-features k of ONE out-feature n,
  // i.e. 16 contiguous bytes -- masters, moments, target and compute copies move as 16-byte (8-byte bf16)
  // accesses.  Other shapes (the first layer: in-features = state_dim [+ action_dim]) keep dW = dZ^T X, where
  // the 16 lanes of a row group hold 16 consecutive k of one n: 64-byte runs of scalar accesses (the
  // transposed form there is 64 scattered words per instruction: the first layer's tiles took 6 of the
  // launch's 12.8 us).  The wave's 16 rows come from operand P, the four 16-wide column groups from Q.
  const DeepNet &N = D.net[it.net];
  const int l = it.layer, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r16 = lane & 15, q = lane >> 4;
  const int BP = D.BP, nk = BP / P::KM;
  const int Nn = N.N[l], Kn = N.K[l];
  const int nkw = N.Kpad[l] / P::KM, nkt = N.NKpad[l] / P::KM;  // k-steps per row of the copies W[n][k] / Wt[k][n]
  const bool vec = (Kn & 3) == 0;  // (scalar)
  const T *Xp = reinterpret_cast<const T *>(N.hT[l]) + (size_t)(it.i0 + r16) * BP + q * P::EPV;
  const T *Zp = reinterpret_cast<const T *>(N.dzT[l]) + (size_t)(it.o0 + r16) * BP + q * P::EPV;
  const T *Pop = (vec ? Xp : Zp) + (size_t)16 * wave * BP, *Qop = vec ? Zp : Xp;
  // bias = sum of the deltas over the batch: from the dZ fragments, by the tiles of the first column block
  const bool bias_q = vec && it.i0 == 0 && wave == 0, bias_p = !vec && it.i0 == 0;  // (scalar)
  f32x4 acc[DEEP_TQ];
  float bsum[DEEP_TQ];
#pragma unroll
  for (int t = 0; t < DEEP_TQ; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f}, bsum[t] = 0.f;
  int ks = 0;
  for (; ks + 4 <= nk; ks += 4) {  // four k-steps (20 fragments) requested before the first MFMA
    uint4 a[4], b[4][DEEP_TQ];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a[u] = ldg16(Pop + (size_t)(ks + u) * P::KM);
#pragma unroll
      for (int t = 0; t < DEEP_TQ; ++t) b[u][t] = ldg16(Qop + (size_t)t * 16 * BP + (size_t)(ks + u) * P::KM);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (bias_p) bsum[0] += frag_sum<BF16>(a[u]);
#pragma unroll
      for (int t = 0; t < DEEP_TQ; ++t) {
        if (bias_q) bsum[t] += frag_sum<BF16>(b[u][t]);
        P::mma(a[u], b[u][t], acc[t]);
      }
    }
  }
  for (; ks < nk; ++ks) {
    const uint4 a = ldg16(Pop + (size_t)ks * P::KM);
    if (bias_p) bsum[0] += frag_sum<BF16>(a);
#pragma unroll
    for (int t = 0; t < DEEP_TQ; ++t) {
      const uint4 b = ldg16(Qop + (size_t)t * 16 * BP + (size_t)ks * P::KM);
      if (bias_q) bsum[t] += frag_sum<BF16>(b);
      P::mma(a, b, acc[t]);
    }
  }
  // (descriptor fields the epilogue needs, read once into registers: behind its first global store the compiler
  // must assume the descriptors changed and fetched D.params, N.has_target, ... again before EVERY access -- a
  // chain of scalar-load round trips that was half of this launch)
  const float neg_step = A.coef.neg_step[N.group];
  T *const wc = reinterpret_cast<T *>(N.wc[l]), *const wt = reinterpret_cast<T *>(N.wt[l]), *const tc = reinterpret_cast<T *>(N.tc[l]);
  const int64_t off_w = N.off_w[l], toff_w = N.toff_w[l], off_b = N.off_b[l], toff_b = N.toff_b[l];
  float *const Pp = D.params, *const Pm = D.exp_avg, *const Pv = D.exp_avg_sq, *const Pt = D.target, *const Pg = D.grads;
  const bool has_target = N.has_target != 0;
  const struct {
    float tau, one_m_tau;
    int polyak_convex;
  } PK = {D.tau, D.one_m_tau, D.polyak_convex};
  const AdamCoef coef = A.coef;
  if (vec) {
    const int k0 = it.i0 + 16 * wave + 4 * q;  // (k0 + 3 < Kn whenever k0 < Kn: both are multiples of 4)
#pragma unroll
    for (int t = 0; t < DEEP_TQ; ++t) {
      const int n = it.o0 + 16 * t + r16;
      if (n >= Nn || k0 >= Kn) continue;
      const int64_t o = off_w + (int64_t)n * Kn + k0, to = toff_w + (int64_t)n * Kn + k0;
      const float4 pp = __builtin_bit_cast(float4, ldg16(Pp + o)), mm = __builtin_bit_cast(float4, ldg16(Pm + o));
      const float4 vv = __builtin_bit_cast(float4, ldg16(Pv + o));
      float p[4] = {pp.x, pp.y, pp.z, pp.w}, m[4] = {mm.x, mm.y, mm.z, mm.w}, v[4] = {vv.x, vv.y, vv.z, vv.w};
      float tg[4] = {0.f, 0.f, 0.f, 0.f}, g[4];
      if (has_target) {
        const float4 tt = __builtin_bit_cast(float4, ldg16(Pt + to));
        tg[0] = tt.x, tg[1] = tt.y, tg[2] = tt.z, tg[3] = tt.w;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        g[i] = P::round(acc[t][i]);  // parameter gradients are the bf16 results widened (autocast)
        adam_apply<AF>(p[i], m[i], v[i], g[i], coef, neg_step);
        if (has_target) tg[i] = polyak(PK, tg[i], p[i]);
      }
      if (Pg) stg16(Pg + o, make_float4(g[0], g[1], g[2], g[3]));
      stg16(Pp + o, make_float4(p[0], p[1], p[2], p[3]));
      stg16(Pm + o, make_float4(m[0], m[1], m[2], m[3]));
      stg16(Pv + o, make_float4(v[0], v[1], v[2], v[3]));
      put4T<BF16>(wc + fidx<P>(n, k0, nkw), p);  // (4 consecutive, 4-aligned k of one row stay contiguous)
      if (has_target) {
        stg16(Pt + to, make_float4(tg[0], tg[1], tg[2], tg[3]));
        put4T<BF16>(tc + fidx<P>(n, k0, nkw), tg);
      }
      if (wt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) stg(wt + fidx<P>(k0 + i, n, nkt), P::from_f32(p[i]));
      }
    }
  } else {
    const int n0 = it.o0 + 16 * wave + 4 * q;
#pragma unroll
    for (int t = 0; t < DEEP_TQ; ++t) {
      const int k = it.i0 + 16 * t + r16;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int n = n0 + i;
        if (n < Nn && k < Kn) {
          const float g = P::round(acc[t][i]);
          const int64_t o = off_w + (int64_t)n * Kn + k;
          float p = ldg(Pp + o), m = ldg(Pm + o), v = ldg(Pv + o);
          if (Pg) stg(Pg + o, g);
          adam_apply<AF>(p, m, v, g, coef, neg_step);
          stg(Pp + o, p), stg(Pm + o, m), stg(Pv + o, v);
          stg(wc + fidx<P>(n, k, nkw), P::from_f32(p));
          if (wt) stg(wt + fidx<P>(k, n, nkt), P::from_f32(p));
          if (has_target) {
            const int64_t to = toff_w + (int64_t)n * Kn + k;
            const float tn = polyak(PK, ldg(Pt + to), p);
            stg(Pt + to, tn);
            stg(tc + fidx<P>(n, k, nkw), P::from_f32(tn));
          }
        }
      }
    }
  }
  if (bias_q || bias_p) {  // bf16(g.sum(0)) under autocast
#pragma unroll
    for (int t = 0; t < DEEP_TQ; ++t) {
      if (bias_p && t > 0) break;
      const float gs = P::round(xor32_sum(xor16_sum(bsum[t])));
      const int n = bias_q ? it.o0 + 16 * t + r16 : it.o0 + 16 * wave + r16;
      if (q == 0 && n < Nn) {
        const int64_t o = off_b + n;
        float p = ldg(Pp + o), m = ldg(Pm + o), v = ldg(Pv + o);
        if (Pg) stg(Pg + o, gs);
        adam_apply<AF>(p, m, v, gs, coef, neg_step);
        stg(Pp + o, p), stg(Pm + o, m), stg(Pv + o, v);
        if (has_target) {
          const int64_t to = toff_b + n;
          stg(Pt + to, polyak(PK, ldg(Pt + to), p));
        }
      }
    }
  }
}

// Compute / transposed / target copies from the fp32 masters: grid (blocks, ntrain * NL).
template <bool BF16>
__global__ __launch_bounds__(256) void kd_sync(const DeepDesc *__restrict__ Dp) {
  using P = Prec<BF16>;
  using T = typename P::T;
  const DeepDesc &D = *Dp;
  const int net = blockIdx.y / D.NL, l = blockIdx.y % D.NL;
  const DeepNet &N = D.net[net];
  const int Nn = N.N[l], Kn = N.K[l], nkw = N.Kpad[l] / P::KM, nkt = N.NKpad[l] / P::KM;
  T *wc = reinterpret_cast<T *>(N.wc[l]), *wt = reinterpret_cast<T *>(N.wt[l]), *tc = reinterpret_cast<T *>(N.tc[l]);
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < (int64_t)Nn * Kn; e += (int64_t)gridDim.x * 256) {
    const int n = (int)(e / Kn), k = (int)(e % Kn);
    const T pv = P::from_f32(ldg(D.params + N.off_w[l] + e));
    stg(wc + fidx<P>(n, k, nkw), pv);
    if (wt) stg(wt + fidx<P>(k, n, nkt), pv);
    if (N.has_target) stg(tc + fidx<P>(n, k, nkw), P::from_f32(ldg(D.target + N.toff_w[l] + e)));
  }
}

}  // namespace

// ========================================================================
// host side
// ========================================================================
struct DeepTrainer {
  int tq = 4;  // kd_update's tile width in 16-column groups (1 or 4)
  DeepDesc D;
  DeepDesc *dD = nullptr;
  DeepItem *ditems = nullptr;
  int n_items = 0;
  void *ws = nullptr;
  bool bf16 = true;
  size_t lds_bytes = 0;
};

template <typename T>
static T *carve_(char *&p, size_t n) {
  T *r = reinterpret_cast<T *>(p);
  p += (n * sizeof(T) + 255) / 256 * 256;
  return r;
}

bool deep_shape_ok(const iqlhip_trainer_config &c, int n_hidden, const char **why) {
  if (n_hidden < 1 || n_hidden > DEEP_MAX_LIN - 1) return *why = "n_hidden must be in 1..6", false;
  if (c.hidden_dim < 1 || c.hidden_dim > DEEP_MAX_H) return *why = "hidden_dim must be in 1..1024", false;
  return true;
}

// off: the arena layout's offsets, 2 (n_hidden + 1) per net (W0 b0 W1 b1 ...), then log_std
hipError_t deep_create(DeepTrainer **out, const iqlhip_trainer_config &cfg, int n_hidden, const iqlhip_arenas &ar,
                       const int64_t *off) {
  DeepTrainer *t = new (std::nothrow) DeepTrainer();
  if (!t) return hipErrorOutOfMemory;
  t->bf16 = cfg.precision == IQLHIP_PREC_BF16;
  const int es = t->bf16 ? 2 : 4, KM = t->bf16 ? 32 : 16, EPV = t->bf16 ? 8 : 4;
  DeepDesc &D = t->D;
  memset(&D, 0, sizeof(D));
  const int S = cfg.state_dim, A = cfg.action_dim, H = cfg.hidden_dim, B = cfg.batch_size;
  const int E = cfg.n_critics > 0 ? cfg.n_critics : 2, NT = E + 2, NF = 2 * E + 3, NL = n_hidden + 1;
  D.S = S, D.A = A, D.H = H, D.Hp = round_up(H, 32), D.B = B, D.BP = round_up(B, 32), D.NL = NL;
  D.E = E, D.ntrain = NT, D.nfwd = NF, D.net_v = E, D.net_a = E + 1;
  D.out_v = E, D.out_qt = E + 1, D.out_nv = 2 * E + 1, D.out_mean = 2 * E + 2;
  D.OUTW = round_up(D.out_mean + A, 4);
  D.next_off = round_up(S + A + 2, 4);
  D.opad = round_up(A, 16);
  D.nslab = B / 16;
  const char *tq_env = getenv("IQLHIP_GENERAL_TQ");  // (A/B knob: 1 or 4; the rule sits with the item table below)
  D.deterministic = cfg.deterministic, D.has_dropout = cfg.dropout_p > 0.f, D.polyak_convex = cfg.polyak_form == 1;
  D.two_over_B = 2.0f / (float)B, D.inv_E = 1.0f / (float)E;
  D.discount = cfg.discount, D.tau = cfg.tau, D.beta = cfg.beta, D.iql_tau = cfg.iql_tau;
  D.one_m_tau = (float)(1.0 - (double)cfg.tau);
  if (D.has_dropout) {
    const float scale = 1.0f / (float)(1.0 - (double)cfg.dropout_p);
    if (t->bf16) {  // noise.div_(1-p) happens in bf16 under autocast (ATen _dropout_impl)
      uint32_t u;
      memcpy(&u, &scale, 4);
      u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
      memcpy(&D.drop_scale, &u, 4);
    } else {
      D.drop_scale = scale;
    }
    const double thr = (double)cfg.dropout_p * 4294967296.0;
    D.drop_thr = thr >= 4294967295.0 ? 0xffffffffu : (uint32_t)thr;
  }
  D.seed = cfg.seed;
  D.params = ar.params, D.exp_avg = ar.exp_avg, D.exp_avg_sq = ar.exp_avg_sq, D.target = ar.target, D.grads = ar.grads;
  D.off_log_std = off[NT * 2 * NL];
  const int k0max = round_up(S + A, 32);
  D.lds_w = std::max(std::max(k0max, D.Hp), 32) + EPV;
  t->lds_bytes = (size_t)2 * 16 * D.lds_w * es;
  // ---- per-net geometry ----
  auto r64 = [](int x) { return (size_t)round_up(x, 64); };
  size_t total = 0;
  auto add = [&](size_t bytes) { total += (bytes + 255) / 256 * 256; };
  for (int n = 0; n < NT; ++n) {
    DeepNet &N = D.net[n];
    const int in = n < E ? S + A : S, outd = n == E + 1 ? A : 1;
    for (int l = 0; l < NL; ++l) {
      N.K[l] = l == 0 ? in : H, N.Kpad[l] = l == 0 ? round_up(in, 32) : D.Hp;
      N.N[l] = l == NL - 1 ? outd : H, N.Npad[l] = l == NL - 1 ? round_up(outd, 16) : D.Hp;
      N.NKpad[l] = round_up(N.Npad[l], KM);
      N.off_w[l] = off[(n * NL + l) * 2], N.off_b[l] = off[(n * NL + l) * 2 + 1];
      N.toff_w[l] = n < E ? N.off_w[l] : -1, N.toff_b[l] = n < E ? N.off_b[l] : -1;
      add((size_t)N.Npad[l] * N.Kpad[l] * es);                 // wc
      if (l >= 1) add((size_t)N.Kpad[l] * N.NKpad[l] * es);    // wt
      if (n < E) add((size_t)N.Npad[l] * N.Kpad[l] * es);      // tc
      if (l >= 1) add(r64(N.Kpad[l]) * D.BP * es);             // hT[l]
      add(r64(N.Npad[l]) * D.BP * es);                         // dzT[l]
    }
    N.has_target = n < E, N.group = n < E ? 0 : (n == E ? 1 : 2);
  }
  add(r64(k0max) * D.BP * es);             // xT
  add((size_t)B * 2 * 4), add((size_t)B * A * 4);
  add((size_t)D.OUTW * D.BP * 4);
  add((size_t)NT * D.nslab * 4), add((size_t)D.nslab * A * 4);
  add(sizeof(DeepDesc));
  // update items.  Tile width (kd_update<.., TQ>): 64 x 64 tiles where the operand panels are the traffic (batch
  // >= 1024) or the matrices are large (width >= 768) AND such tiles still make ~150 work-groups; 64 x 16 otherwise
  // (measured with IQLHIP_GENERAL_TQ, tools/general_run.py: E = 4 / batch 1024 with three hidden layers, 216 wide
  // tiles: 26.4 us against 32.1; with two, 144: 25.7 against 18.6; width 1024: 43.9 against 46.5)
  {
    int wide = 0;
    for (int n = 0; n < NT; ++n)
      for (int l = 0; l < NL; ++l) wide += ((D.net[n].N[l] + 63) / 64) * ((D.net[n].K[l] + 63) / 64);
    t->tq = ((D.BP >= 1024 || D.Hp >= 768) && wide >= 150) ? 4 : 1;
    if (tq_env) t->tq = atoi(tq_env) == 1 ? 1 : 4;
  }
  std::vector<DeepItem> items;
  for (int n = 0; n < NT; ++n)
    for (int l = 0; l < NL; ++l)
    {
      // (kd_update: in-features a multiple of 4 -> transposed tile, 64 in-features x 16 tq out-features;
      // else 64 out-features x 16 tq in-features)
      const bool vec = (D.net[n].K[l] & 3) == 0;
      const int so = vec ? 16 * t->tq : 64, si = vec ? 64 : 16 * t->tq;
      for (int o0 = 0; o0 < D.net[n].N[l]; o0 += so)
        for (int i0 = 0; i0 < D.net[n].K[l]; i0 += si) items.push_back(DeepItem{n, l, o0, i0});
    }
  items.push_back(DeepItem{-1, 0, 0, 0});
  t->n_items = (int)items.size();
  add(items.size() * sizeof(DeepItem));
  hipError_t e = hipMalloc(&t->ws, total);
  if (e != hipSuccess) {
    delete t;
    return e;
  }
  if ((e = hipMemset(t->ws, 0, total)) != hipSuccess) {
    (void)hipFree(t->ws);
    delete t;
    return e;
  }
  char *p = reinterpret_cast<char *>(t->ws);
  void *xT = nullptr;
  for (int n = 0; n < NT; ++n) {
    DeepNet &N = D.net[n];
    for (int l = 0; l < NL; ++l) {
      N.wc[l] = carve_<char>(p, (size_t)N.Npad[l] * N.Kpad[l] * es);
      N.wt[l] = l >= 1 ? carve_<char>(p, (size_t)N.Kpad[l] * N.NKpad[l] * es) : nullptr;
      N.tc[l] = n < E ? carve_<char>(p, (size_t)N.Npad[l] * N.Kpad[l] * es) : nullptr;
      N.hT[l] = l >= 1 ? carve_<char>(p, r64(N.Kpad[l]) * D.BP * es) : nullptr;
      N.dzT[l] = carve_<char>(p, r64(N.Npad[l]) * D.BP * es);
    }
  }
  xT = carve_<char>(p, r64(k0max) * D.BP * es);
  for (int n = 0; n < NT; ++n) D.net[n].hT[0] = xT;
  D.rd = carve_<float>(p, (size_t)B * 2);
  D.actf = carve_<float>(p, (size_t)B * A);
  D.outs = carve_<float>(p, (size_t)D.OUTW * D.BP);
  D.lossp = carve_<float>(p, (size_t)NT * D.nslab);
  D.lsp = carve_<float>(p, (size_t)D.nslab * A);
  t->dD = carve_<DeepDesc>(p, 1);
  t->ditems = carve_<DeepItem>(p, items.size());
  // ---- evaluations: q_e, v, actor, target q_e, next_v ----
  auto mk = [&](int f, int net, bool target, int in_off, int out_col, int slot) {
    DeepEval &F = D.ev[f];
    const DeepNet &N = D.net[net];
    const float *base = target ? D.target : D.params;
    for (int l = 0; l < NL; ++l) {
      F.lin[l].w = target ? N.tc[l] : N.wc[l];
      F.lin[l].b = base + (target ? N.toff_b[l] : N.off_b[l]);
      F.lin[l].K = N.K[l], F.lin[l].Kpad = N.Kpad[l], F.lin[l].N = N.N[l], F.lin[l].Npad = N.Npad[l];
    }
    F.in_off = in_off, F.in_dim = N.K[0];
    F.out_dim = N.N[NL - 1], F.out_col = out_col, F.train_slot = slot;
    F.tanh_out = net == D.net_a, F.dropout = net == D.net_a && D.has_dropout, F.stage = f == 0;
  };
  for (int e2 = 0; e2 < E; ++e2) mk(e2, e2, false, 0, e2, e2);
  mk(E, D.net_v, false, 0, D.out_v, D.net_v);
  mk(E + 1, D.net_a, false, 0, D.out_mean, D.net_a);
  for (int e2 = 0; e2 < E; ++e2) mk(E + 2 + e2, e2, true, 0, D.out_qt + e2, -1);
  mk(2 * E + 2, D.net_v, false, D.next_off, D.out_nv, -1);
  if ((e = hipMemcpy(t->dD, &D, sizeof(D), hipMemcpyHostToDevice)) == hipSuccess)
    e = hipMemcpy(t->ditems, items.data(), items.size() * sizeof(DeepItem), hipMemcpyHostToDevice);
  if (e == hipSuccess && t->lds_bytes > 48 * 1024) {
    const int lb = (int)t->lds_bytes;
    const void *fns[] = {(const void *)kd_forward<true, 256>,   (const void *)kd_forward<false, 256>,
                         (const void *)kd_forward<true, 512>,   (const void *)kd_forward<false, 512>,
                         (const void *)kd_backward<true, 256>,  (const void *)kd_backward<false, 256>,
                         (const void *)kd_backward<true, 512>,  (const void *)kd_backward<false, 512>,
                         (const void *)kd_infer<true>,          (const void *)kd_infer<false>};
    for (const void *f : fns)
      if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, lb);
  }
  if (e != hipSuccess) {
    (void)hipFree(t->ws);
    delete t;
    return e;
  }
  *out = t;
  return hipSuccess;
}

void deep_destroy(DeepTrainer *t) {
  if (!t) return;
  if (t->ws) (void)hipFree(t->ws);
  delete t;
}

hipError_t deep_sync_weights(DeepTrainer *t, hipStream_t st) {
  const dim3 grid(64, t->D.ntrain * t->D.NL);
  if (t->bf16)
    hipLaunchKernelGGL(kd_sync<true>, grid, dim3(256), 0, st, t->dD);
  else
    hipLaunchKernelGGL(kd_sync<false>, grid, dim3(256), 0, st, t->dD);
  return hipGetLastError();
}

// One step: three launches; ev (optional, 4 events) brackets them for the per-kernel timing.
static int deep_threads(const DeepDesc &D, int n_wgs) { return D.Hp >= 256 && n_wgs <= 256 ? 512 : 256; }

hipError_t deep_step(DeepTrainer *t, const DeepStep &a, hipStream_t st, hipEvent_t *ev) {
  const DeepDesc &D = t->D;
  const dim3 gf(D.nslab, D.nfwd), gb(D.nslab, D.ntrain), gu(t->n_items);
  const int tf = deep_threads(D, D.nslab * D.nfwd), tb = deep_threads(D, D.nslab * D.ntrain);
  hipError_t e;
#define DEEP_EV(k) \
  if (ev && (e = hipEventRecord(ev[k], st)) != hipSuccess) return e;
  DEEP_EV(0);
  if (t->bf16 && tf == 512)
    hipLaunchKernelGGL((kd_forward<true, 512>), gf, dim3(512), t->lds_bytes, st, t->dD, a);
  else if (t->bf16)
    hipLaunchKernelGGL((kd_forward<true, 256>), gf, dim3(256), t->lds_bytes, st, t->dD, a);
  else if (tf == 512)
    hipLaunchKernelGGL((kd_forward<false, 512>), gf, dim3(512), t->lds_bytes, st, t->dD, a);
  else
    hipLaunchKernelGGL((kd_forward<false, 256>), gf, dim3(256), t->lds_bytes, st, t->dD, a);
  DEEP_EV(1);
  if (t->bf16 && tb == 512)
    hipLaunchKernelGGL((kd_backward<true, 512>), gb, dim3(512), t->lds_bytes, st, t->dD, a);
  else if (t->bf16)
    hipLaunchKernelGGL((kd_backward<true, 256>), gb, dim3(256), t->lds_bytes, st, t->dD, a);
  else if (tb == 512)
    hipLaunchKernelGGL((kd_backward<false, 512>), gb, dim3(512), t->lds_bytes, st, t->dD, a);
  else
    hipLaunchKernelGGL((kd_backward<false, 256>), gb, dim3(256), t->lds_bytes, st, t->dD, a);
  DEEP_EV(2);
  if (t->bf16 && t->tq == 1)
    hipLaunchKernelGGL((kd_update<true, 1>), gu, dim3(256), 0, st, t->dD, t->ditems, a);
  else if (t->bf16)
    hipLaunchKernelGGL((kd_update<true, 4>), gu, dim3(256), 0, st, t->dD, t->ditems, a);
  else if (t->tq == 1)
    hipLaunchKernelGGL((kd_update<false, 1>), gu, dim3(256), 0, st, t->dD, t->ditems, a);
  else
    hipLaunchKernelGGL((kd_update<false, 4>), gu, dim3(256), 0, st, t->dD, t->ditems, a);
  DEEP_EV(3);
#undef DEEP_EV
  return hipGetLastError();
}

// which: 0 critics, 1 V, 2 actor mean, 3 target critics (iqlhip_forward)
hipError_t deep_infer(DeepTrainer *t, int which, const float *s, const float *a, int64_t n, float *out,
                      hipStream_t st) {
  const DeepDesc &D = t->D;
  const dim3 grid((unsigned)((n + 15) / 16));
  auto go = [&](int evn, int out_ld, int col0) {
    if (t->bf16)
      hipLaunchKernelGGL(kd_infer<true>, grid, dim3(256), t->lds_bytes, st, t->dD, evn, s, a, n, out, out_ld, col0);
    else
      hipLaunchKernelGGL(kd_infer<false>, grid, dim3(256), t->lds_bytes, st, t->dD, evn, s, a, n, out, out_ld, col0);
  };
  if (which == 0 || which == 3)
    for (int e = 0; e < D.E; ++e) go((which == 3 ? D.E + 2 : 0) + e, D.E, e);
  else if (which == 1)
    go(D.E, 1, 0);
  else
    go(D.E + 1, D.A, 0);
  return hipGetLastError();
}

}  // namespace iqlhip
