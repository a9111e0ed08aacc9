// IQL optimisation step for MI355X (gfx950): three kernels per step.
//
//   k_forward   7 MLP evaluations x B/16 row slabs, one work-group each:
//               gather the slab's transitions from the packed replay rows,
//               3 Linear layers on MFMA (activations staged in LDS, weights
//               streamed from L2 straight into B fragments), hidden activations
//               of the trained nets stored feature-major for the backward GEMMs.
//   k_backward  4 trained nets x B/16 slabs: loss terms (expectile, TD, AWR),
//               d(out), dZ2 (VALU outer product), dZ1 (MFMA), all stored
//               feature-major; per-slab loss partial sums.
//   k_update    weight-gradient GEMMs (K = batch) fused with Adam, the
//               compute-precision weight copies, and the Polyak target update;
//               the gradient tile never leaves registers.
//
// Reference: /root/reference/algorithms/offline/iql.py:581-662 (order of
// operations), :404-405 (expectile), :127-129 (Polyak), torch.optim.Adam
// (_single_tensor_adam) and CosineAnnealingLR closed form.
//
// Data layout: activations / deltas are [feature][batch] ("T" suffix) so that
// both operands of dW = dZ^T X are K(batch)-contiguous 16-byte fragments, and
// the MFMA C/D layout (4 consecutive rows per lane) stores them with 8/16-byte
// writes.
#include "common.h"
#include "iql_step.h"

namespace iqlhip {

constexpr int SLAB = 16;  // batch rows per work-group (one MFMA M tile)
constexpr int MAXT = 4;   // n-tiles per wave at H = 256

// ------------------------------------------------------------------------
// acc[jj] += X[16][K] (LDS, row stride xs) * W^T, W = [N][K] global, for the
// wave's n-tiles jt = wave*tpw + jj.
// ------------------------------------------------------------------------
template <bool BF16>
__device__ __forceinline__ void slab_gemm(const typename Prec<BF16>::T *x, int xs, int K,
                                          const typename Prec<BF16>::T *W, int tpw, int wave,
                                          int lane, f32x4 acc[MAXT]) {
  using P = Prec<BF16>;
  const int r = lane & 15, q = lane >> 4;
  const typename P::T *xrow = x + r * xs + P::EPV * q;
  const typename P::T *wrow = W + (size_t)(16 * wave * tpw + r) * K + P::EPV * q;
  for (int kb = 0; kb < K; kb += P::KM) {
    const uint4 a = *reinterpret_cast<const uint4 *>(xrow + kb);
#pragma unroll
    for (int jj = 0; jj < MAXT; ++jj) {
      if (jj < tpw) {
        const uint4 b = *reinterpret_cast<const uint4 *>(wrow + (size_t)jj * 16 * K + kb);
        P::mma(a, b, acc[jj]);
      }
    }
  }
}

template <bool BF16>
__device__ __forceinline__ void store4T(typename Prec<BF16>::T *dst, const float v[4]) {
  using P = Prec<BF16>;
  if constexpr (BF16) {
    uint2 u;
    u.x = (uint32_t)P::from_f32(v[0]) | ((uint32_t)P::from_f32(v[1]) << 16);
    u.y = (uint32_t)P::from_f32(v[2]) | ((uint32_t)P::from_f32(v[3]) << 16);
    *reinterpret_cast<uint2 *>(dst) = u;
  } else {
    *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

template <bool BF16>
__device__ __forceinline__ void load4T(const typename Prec<BF16>::T *src, float v[4]) {
  if constexpr (BF16) {
    const uint2 u = *reinterpret_cast<const uint2 *>(src);
    v[0] = bf2f((uint16_t)(u.x & 0xffff));
    v[1] = bf2f((uint16_t)(u.x >> 16));
    v[2] = bf2f((uint16_t)(u.y & 0xffff));
    v[3] = bf2f((uint16_t)(u.y >> 16));
  } else {
    const float4 f = *reinterpret_cast<const float4 *>(src);
    v[0] = f.x, v[1] = f.y, v[2] = f.z, v[3] = f.w;
  }
}

// keep mask of the 4 rows a lane owns (rows 4*rowblk .. +3) of hidden unit col
__device__ __forceinline__ void dropout_keep4(const TrainerDesc &D, const DevArgs &A, int64_t step,
                                              int layer, int rowblk, int col, bool keep[4]) {
  if (A.drop_keep) {
    const uint8_t *m = A.drop_keep +
                       (((size_t)(step - A.base_step) * 2 + layer) * D.B + (size_t)rowblk * 4) * D.H + col;
#pragma unroll
    for (int i = 0; i < 4; ++i) keep[i] = m[(size_t)i * D.H] != 0;
  } else {
    const Philox4 ph = philox4x32_10((uint32_t)(rowblk * D.H + col), (uint32_t)step,
                                     (uint32_t)((uint64_t)step >> 32),
                                     layer == 0 ? STREAM_DROPOUT1 : STREAM_DROPOUT2,
                                     (uint32_t)D.seed, (uint32_t)(D.seed >> 32));
    keep[0] = ph.x >= D.drop_thr;
    keep[1] = ph.y >= D.drop_thr;
    keep[2] = ph.z >= D.drop_thr;
    keep[3] = ph.w >= D.drop_thr;
  }
}

// ------------------------------------------------------------------------
// Three Linear layers for one 16-row slab whose (padded) input is already in
// LDS at xs.  Used by the training forward kernel and by iqlhip_forward.
// ------------------------------------------------------------------------
template <bool BF16>
__device__ __forceinline__ void mlp_slab(const FwdNet &N, const TrainerDesc &D, const DevArgs *Ap,
                                         int64_t step, int slab, typename Prec<BF16>::T *xs,
                                         typename Prec<BF16>::T *h1, typename Prec<BF16>::T *h2,
                                         float *red, float *out, int out_stride, int64_t row0,
                                         int64_t n_valid) {
  using P = Prec<BF16>;
  using T = typename P::T;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int H = D.H, HP = H + P::EPV, K1P = D.k1max + P::EPV;
  const int tpw = H / 64;
  const int k1 = N.k1pad;
  // ---- hidden layers ----
  T *hin = xs;
  int hin_stride = K1P, K = k1;
  const T *Wl = reinterpret_cast<const T *>(N.w1c);
  const float *bl = N.b1;
  T *hout = h1;
#pragma unroll 1
  for (int layer = 0; layer < 2; ++layer) {
    f32x4 acc[MAXT];
#pragma unroll
    for (int jj = 0; jj < MAXT; ++jj) acc[jj] = f32x4{0.f, 0.f, 0.f, 0.f};
    slab_gemm<BF16>(hin, hin_stride, K, Wl, tpw, wave, lane, acc);
#pragma unroll
    for (int jj = 0; jj < MAXT; ++jj) {
      if (jj < tpw) {
        const int col = 16 * (wave * tpw + jj) + r;
        const float bias = P::round(bl[col]);
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = fmaxf(P::round(acc[jj][i] + bias), 0.f);
        if (N.dropout) {
          bool keep[4];
          dropout_keep4(D, *Ap, step, layer, slab * 4 + q, col, keep);
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = keep[i] ? P::round(v[i] * D.drop_scale) : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) hout[(4 * q + i) * HP + col] = P::from_f32(v[i]);
        if (N.train_slot >= 0) {
          T *dst = reinterpret_cast<T *>(D.hT) +
                   ((size_t)(N.train_slot * 2 + layer) * H + col) * D.BP + slab * SLAB + 4 * q;
          store4T<BF16>(dst, v);
        }
      }
    }
    __syncthreads();
    hin = hout, hin_stride = HP, K = H;
    Wl = reinterpret_cast<const T *>(N.w2c);
    bl = N.b2;
    hout = h2;
  }

  // ---- output layer: K split over the 4 waves, reduced through LDS ----
  const int nt3 = N.out_pad / 16;  // 1 or 2
  {
    f32x4 acc3[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    const T *W3 = reinterpret_cast<const T *>(N.w3c);
    for (int kb = wave * P::KM; kb < H; kb += 4 * P::KM) {
      const uint4 a = *reinterpret_cast<const uint4 *>(h2 + r * HP + kb + P::EPV * q);
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) {
        if (jt < nt3) {
          const uint4 b = *reinterpret_cast<const uint4 *>(W3 + (size_t)(16 * jt + r) * H + kb + P::EPV * q);
          P::mma(a, b, acc3[jt]);
        }
      }
    }
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
      *reinterpret_cast<f32x4 *>(red + ((wave * 2 + jt) * 64 + lane) * 4) = acc3[jt];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
      if (jt < nt3) {
        f32x4 s = *reinterpret_cast<f32x4 *>(red + ((0 * 2 + jt) * 64 + lane) * 4);
#pragma unroll
        for (int w = 1; w < 4; ++w) {
          const f32x4 p = *reinterpret_cast<f32x4 *>(red + ((w * 2 + jt) * 64 + lane) * 4);
          s += p;
        }
        const int col = 16 * jt + r;
        if (col < N.out_dim) {
          const float bias = P::round(N.b3[col]);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float v = P::round(s[i] + bias);
            if (N.tanh_out) v = P::round(tanhf(v));
            if (row0 + 4 * q + i < n_valid) out[(size_t)(row0 + 4 * q + i) * out_stride + N.out_col + col] = v;
          }
        }
      }
    }
  }
}

// ========================================================================
// k_forward
// ========================================================================
template <bool BF16>
__global__ __launch_bounds__(256) void k_forward(const TrainerDesc D, const DevArgs *__restrict__ Ap,
                                                 const DevCtr *__restrict__ Cp) {
  using P = Prec<BF16>;
  using T = typename P::T;
  const DevArgs &A = *Ap;
  const int nslab = D.B / SLAB;
  const int fnet = blockIdx.x / nslab, slab = blockIdx.x % nslab;
  const FwdNet &N = D.fwd[fnet];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int H = D.H, HP = H + P::EPV, K1P = D.k1max + P::EPV;
  const int tpw = H / 64;
  const int64_t step = Cp->ctr[0];

  extern __shared__ __attribute__((aligned(16))) char smem[];
  T *xs = reinterpret_cast<T *>(smem);             // [16][K1P]
  T *h1 = xs + SLAB * K1P;                          // [16][HP]
  T *h2 = h1 + SLAB * HP;                           // [16][HP]
  float *red = reinterpret_cast<float *>(h2 + SLAB * HP);  // [4][2][64][4]
  int64_t *sidx = reinterpret_cast<int64_t *>(red + 4 * 2 * 64 * 4);  // [16]

  // ---- batch indices of this slab (ref:211-214) ----
  if (tid < SLAB) {
    const int row = slab * SLAB + tid;
    int64_t ix;
    if (A.idx_mode == 1)
      ix = A.idx[(size_t)(step - A.base_step) * D.B + row];
    else if (A.idx_mode == 2)
      ix = row;
    else
      ix = philox_index(D.seed, (uint64_t)step, (uint32_t)row, (uint64_t)A.n_rows);
    ix = ix < 0 ? 0 : (ix >= A.n_rows ? A.n_rows - 1 : ix);
    sidx[tid] = ix;
  }
  __syncthreads();

  // ---- gather the slab's input rows into LDS (zero padded to k1pad) ----
  const int k1 = N.k1pad;
  for (int e = tid; e < SLAB * k1; e += 256) {
    const int rr = e / k1, c = e - rr * k1;
    float v = 0.f;
    if (c < N.in_dim) v = A.rows[(size_t)sidx[rr] * A.row_stride + N.in_off + c];
    xs[rr * K1P + c] = P::from_f32(v);
    if (N.stage && c < N.in_dim)
      reinterpret_cast<T *>(D.xT)[(size_t)c * D.BP + slab * SLAB + rr] = P::from_f32(v);
  }
  if (N.stage) {
    const int sa = D.S + D.A;
    if (tid < SLAB * 2) {
      const int rr = tid >> 1, w = tid & 1;
      D.rd[(size_t)(slab * SLAB + rr) * 2 + w] = A.rows[(size_t)sidx[rr] * A.row_stride + sa + w];
    }
    for (int e = tid; e < SLAB * D.A; e += 256) {
      const int rr = e / D.A, c = e - rr * D.A;
      D.actf[(size_t)(slab * SLAB + rr) * D.A + c] = A.rows[(size_t)sidx[rr] * A.row_stride + D.S + c];
    }
  }
  __syncthreads();

  mlp_slab<BF16>(N, D, Ap, step, slab, xs, h1, h2, red, D.outs, D.OUTW, (int64_t)slab * SLAB, D.B);
}

// ========================================================================
// k_infer: forward pass of one network on dense inputs (iqlhip_forward;
// ref:452-543 forward()).  Rows beyond n are computed on zeros and not stored.
// ========================================================================
template <bool BF16>
__global__ __launch_bounds__(256) void k_infer(const TrainerDesc D, const FwdNet N, const float *__restrict__ s,
                                               const float *__restrict__ a, int64_t n, float *out,
                                               int out_stride) {
  using P = Prec<BF16>;
  using T = typename P::T;
  const int tid = threadIdx.x;
  const int H = D.H, HP = H + P::EPV, K1P = D.k1max + P::EPV;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T *xs = reinterpret_cast<T *>(smem);
  T *h1 = xs + SLAB * K1P;
  T *h2 = h1 + SLAB * HP;
  float *red = reinterpret_cast<float *>(h2 + SLAB * HP);
  const int64_t row0 = (int64_t)blockIdx.x * SLAB;
  const int k1 = N.k1pad;
  for (int e = tid; e < SLAB * k1; e += 256) {
    const int rr = e / k1, c = e - rr * k1;
    float v = 0.f;
    if (row0 + rr < n && c < N.in_dim)
      v = (c < D.S) ? s[(size_t)(row0 + rr) * D.S + c] : a[(size_t)(row0 + rr) * D.A + (c - D.S)];
    xs[rr * K1P + c] = P::from_f32(v);
  }
  __syncthreads();
  mlp_slab<BF16>(N, D, nullptr, 0, 0, xs, h1, h2, red, out, out_stride, row0, n);
}

// ========================================================================
// k_backward
// ========================================================================
template <bool BF16>
__global__ __launch_bounds__(256) void k_backward(const TrainerDesc D, const DevArgs *__restrict__ Ap,
                                                  DevCtr *__restrict__ Cp) {
  using P = Prec<BF16>;
  using T = typename P::T;
  const DevArgs &A = *Ap;
  const int nslab = D.B / SLAB;
  const int net = blockIdx.x / nslab, slab = blockIdx.x % nslab;
  const TrainNet &N = D.net[net];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int H = D.H, HP = H + P::EPV, B = D.B, BP = D.BP;
  const int tpw = H / 64;
  const float fB = (float)B;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  T *dz2s = reinterpret_cast<T *>(smem);                       // [16][HP]
  float *dz3 = reinterpret_cast<float *>(dz2s + SLAB * HP);    // [16][32]
  float *lterm = dz3 + SLAB * 32;                              // [16][32] loss terms
  float *gstd = lterm + SLAB * 32;                             // [16][32]
  float *eadv = gstd + SLAB * 32;                              // [16]

  if (blockIdx.x == 0 && tid == 0) Cp->ctr[1] = Cp->ctr[0] + 1;

  // ---- per-row loss terms and d(loss)/d(out)  (ref:581-637) ----
  for (int e = tid; e < SLAB * 32; e += 256) dz3[e] = 0.f, lterm[e] = 0.f, gstd[e] = 0.f;
  __syncthreads();
  if (tid < SLAB) {
    const int b = slab * SLAB + tid;
    const float *o = D.outs + (size_t)b * D.OUTW;
    const float adv = P::round(fminf(o[OUT_QT1], o[OUT_QT2]) - o[OUT_V]);  // ref:583-587
    if (net == NET_V) {
      const float w = fabsf(D.iql_tau - (adv < 0.f ? 1.f : 0.f));  // ref:404-405
      lterm[tid * 32] = w * P::round(adv * adv);
      float g;
      if constexpr (BF16)
        g = rbf(rbf(w / fB) * (2.f * adv));
      else
        g = (w / fB) * (2.f * adv);
      dz3[tid * 32] = -g;  // adv = target_q - v
    } else if (net == NET_A) {
      eadv[tid] = fminf(P::round(expf(P::round(D.beta * adv))), 100.f);  // ref:622
    } else {
      const float rew = D.rd[(size_t)b * 2], done = D.rd[(size_t)b * 2 + 1];
      const float target = rew + (1.f - done) * D.discount * o[OUT_NV];  // ref:604
      const float diff = o[net == NET_Q1 ? OUT_Q1 : OUT_Q2] - target;
      lterm[tid * 32] = diff * diff;
      dz3[tid * 32] = P::round(diff / fB);  // 0.5 * 2 (q - t) / B
    }
  }
  __syncthreads();
  if (net == NET_A) {
    for (int e = tid; e < SLAB * D.A; e += 256) {
      const int rr = e / D.A, j = e - rr * D.A;
      const int b = slab * SLAB + rr;
      const float mean = D.outs[(size_t)b * D.OUTW + OUT_MEAN + j];
      const float act = D.actf[(size_t)b * D.A + j];
      const float gbc = eadv[rr] / fB;
      float gm, bc;
      if (!D.deterministic) {
        float ls = D.params[D.off_log_std + j];
        ls = fminf(fmaxf(ls, -20.f), 2.f);
        const float sd = expf(ls), var = sd * sd, z = act - mean;
        // -log_prob (torch.distributions.Normal.log_prob)
        bc = (z * z) / (2.f * var) + logf(sd) + 0.9189385332046727f;
        gm = P::round(-gbc * (z / var));
        gstd[rr * 32 + j] = gbc * (-(z * z) / (var * sd) + 1.f / sd);
      } else {
        const float z = mean - act;  // ref:629
        bc = z * z;
        gm = P::round(gbc * 2.f * z);
      }
      lterm[rr * 32 + j] = eadv[rr] * bc;
      dz3[rr * 32 + j] = P::round(gm * (1.f - mean * mean));  // tanh backward
    }
    __syncthreads();
  }
  // per-slab partial sums (fixed order -> deterministic)
  if (tid == 0) {
    float s = 0.f;
    for (int rr = 0; rr < SLAB; ++rr)
      for (int j = 0; j < N.out_dim; ++j) s += lterm[rr * 32 + j];
    D.lossp[net * nslab + slab] = s;
  }
  if (net == NET_A && !D.deterministic && tid >= 64 && tid < 64 + D.A) {
    const int j = tid - 64;
    float s = 0.f;
    for (int rr = 0; rr < SLAB; ++rr) s += gstd[rr * 32 + j];
    D.lsp[(size_t)slab * D.A + j] = s;
  }
  // d(out), feature-major, for the layer-3 weight gradient
  for (int e = tid; e < N.out_dim * SLAB; e += 256) {
    const int j = e / SLAB, rr = e - j * SLAB;
    reinterpret_cast<T *>(D.dz3T)[((size_t)net * D.opmax + j) * BP + slab * SLAB + rr] =
        P::from_f32(dz3[rr * 32 + j]);
  }

  // ---- dZ2 = (dZ3 W3) * relu'(h2)   (VALU: K = out_dim <= 32) ----
  if (tid < H) {
    const int c = tid;
    const T *W3 = reinterpret_cast<const T *>(N.wc[2]);
    const T *h2T = reinterpret_cast<const T *>(D.hT) + ((size_t)(net * 2 + 1) * H + c) * BP + slab * SLAB;
    T *dst = reinterpret_cast<T *>(D.dz2T) + ((size_t)net * H + c) * BP + slab * SLAB;
    float s[SLAB];
#pragma unroll
    for (int rr = 0; rr < SLAB; ++rr) s[rr] = 0.f;
    for (int j = 0; j < N.out_dim; ++j) {
      const float w = P::to_f32(W3[(size_t)j * H + c]);
#pragma unroll
      for (int rr = 0; rr < SLAB; ++rr) s[rr] += dz3[rr * 32 + j] * w;
    }
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      float hv[4], outv[4];
      load4T<BF16>(h2T + 4 * g4, hv);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int rr = 4 * g4 + i;
        float sv = P::round(s[rr]);
        if (D.has_dropout && net == NET_A) sv = P::round(sv * D.drop_scale);
        outv[i] = hv[i] > 0.f ? sv : 0.f;
        dz2s[rr * HP + c] = P::from_f32(outv[i]);
      }
      store4T<BF16>(dst + 4 * g4, outv);
    }
  }
  __syncthreads();

  // ---- dZ1 = (dZ2 W2) * relu'(h1)   (MFMA, B operand = transposed copy) ----
  {
    f32x4 acc[MAXT];
#pragma unroll
    for (int jj = 0; jj < MAXT; ++jj) acc[jj] = f32x4{0.f, 0.f, 0.f, 0.f};
    slab_gemm<BF16>(dz2s, HP, H, reinterpret_cast<const T *>(N.w2ct), tpw, wave, lane, acc);
#pragma unroll
    for (int jj = 0; jj < MAXT; ++jj) {
      if (jj < tpw) {
        const int col = 16 * (wave * tpw + jj) + r;
        const T *h1T = reinterpret_cast<const T *>(D.hT) + ((size_t)(net * 2 + 0) * H + col) * BP +
                       slab * SLAB + 4 * q;
        float hv[4], outv[4];
        load4T<BF16>(h1T, hv);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float s = P::round(acc[jj][i]);
          if (D.has_dropout && net == NET_A) s = P::round(s * D.drop_scale);
          outv[i] = hv[i] > 0.f ? s : 0.f;
        }
        store4T<BF16>(reinterpret_cast<T *>(D.dz1T) + ((size_t)net * H + col) * BP + slab * SLAB + 4 * q,
                      outv);
      }
    }
  }
}

// ========================================================================
// k_update
// ========================================================================
struct AdamCoef {
  float one_m_b1, b2, one_m_b2, neg_step[3], bc2_sqrt, eps;  // neg_step per group q / v / actor
};

__device__ __forceinline__ float adam_apply(float &p, float &m, float &v, float g, const AdamCoef &c,
                                            float neg_step) {
  m = m + (g - m) * c.one_m_b1;                 // exp_avg.lerp_(grad, 1 - beta1)
  v = v * c.b2 + (c.one_m_b2 * g) * g;          // mul_(beta2).addcmul_(g, g, 1 - beta2)
  const float denom = sqrtf(v) / c.bc2_sqrt + c.eps;
  p = p + neg_step * (m / denom);               // addcdiv_(exp_avg, denom, -step_size)
  return p;
}

template <bool BF16>
__global__ __launch_bounds__(256) void k_update(const TrainerDesc D, const DevArgs *__restrict__ Ap,
                                                DevCtr *__restrict__ Cp,
                                                const UpdItem *__restrict__ items, int n_items) {
  using P = Prec<BF16>;
  using T = typename P::T;
  const DevArgs &A = *Ap;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int H = D.H, B = D.B, BP = D.BP;
  const int nslab = B / SLAB;
  const int64_t t1 = Cp->ctr[1];  // 1-based Adam step of this update

  __shared__ AdamCoef coef;
  if (tid == 0) {
    const double bc1 = 1.0 - pow(D.beta1, (double)t1);
    const double bc2 = 1.0 - pow(D.beta2, (double)t1);
    // CosineAnnealingLR closed form; t1-1 scheduler steps have been taken (ref:636-637)
    const double lr_a = A.lr_a_base * (1.0 + cos(M_PI * (double)(t1 - 1) / (double)D.t_max)) * 0.5;
    coef.one_m_b1 = (float)(1.0 - D.beta1);
    coef.b2 = (float)D.beta2;
    coef.one_m_b2 = (float)(1.0 - D.beta2);
    coef.neg_step[0] = (float)(-(A.lr_q / bc1));
    coef.neg_step[1] = (float)(-(A.lr_v / bc1));
    coef.neg_step[2] = (float)(-(lr_a / bc1));
    coef.bc2_sqrt = (float)sqrt(bc2);
    coef.eps = (float)D.eps;
  }
  __syncthreads();

  if ((int)blockIdx.x >= n_items) {
    // ---------------- misc block: log_std, logged losses, step counter ------------
    if (!D.deterministic && tid < D.A) {
      float g = 0.f;
      for (int s = 0; s < nslab; ++s) g += D.lsp[(size_t)s * D.A + tid];
      const int64_t o = D.off_log_std + tid;
      const float ls = D.params[o];
      const float lsc = fminf(fmaxf(ls, -20.f), 2.f);
      g = g * expf(lsc) * ((ls >= -20.f && ls <= 2.f) ? 1.f : 0.f);
      float p = ls, m = D.exp_avg[o], v = D.exp_avg_sq[o];
      adam_apply(p, m, v, g, coef, coef.neg_step[2]);
      D.params[o] = p, D.exp_avg[o] = m, D.exp_avg_sq[o] = v;
      if (D.grads) D.grads[o] = g;
    }
    if (tid == 64) {
      float s[4] = {0.f, 0.f, 0.f, 0.f};
      for (int n = 0; n < 4; ++n)
        for (int k = 0; k < nslab; ++k) s[n] += D.lossp[n * nslab + k];
      const float fB = (float)B;
      const float vl = s[NET_V] / fB;
      const float ql = (s[NET_Q1] / fB + s[NET_Q2] / fB) / 2.f;  // ref:606
      const float al = s[NET_A] / fB;
      Cp->last_losses[0] = vl, Cp->last_losses[1] = ql, Cp->last_losses[2] = al;
      Cp->loss_sum[0] += vl, Cp->loss_sum[1] += ql, Cp->loss_sum[2] += al;
      if (A.losses_out) {
        float *lo = A.losses_out + (size_t)(t1 - 1 - A.base_step) * 3;
        lo[0] = vl, lo[1] = ql, lo[2] = al;
      }
      Cp->ctr[0] = t1;
    }
    return;
  }

  const UpdItem it = items[blockIdx.x];
  const TrainNet &N = D.net[it.net];
  const int L = it.layer;
  const int Odim = (L == 2) ? N.out_dim : H;
  const int Idim = (L == 0) ? N.in_dim : H;
  const int Kw = (L == 0) ? N.k1pad : H;  // row stride of the compute copy
  const int Ipad = round_up(Idim, 16);
  const int Opad = (L == 2) ? N.out_pad : H;
  const int o_base = it.o0 + 16 * wave * it.wo;
  const int i_base = it.i0 + 64 * wave * it.wi;
  if (o_base >= Opad || i_base >= Ipad) return;
  const float neg_step = coef.neg_step[it.net == NET_V ? 1 : (it.net == NET_A ? 2 : 0)];

  const T *Asrc = (L == 0) ? reinterpret_cast<const T *>(D.dz1T) + (size_t)it.net * H * BP
                : (L == 1) ? reinterpret_cast<const T *>(D.dz2T) + (size_t)it.net * H * BP
                           : reinterpret_cast<const T *>(D.dz3T) + (size_t)it.net * D.opmax * BP;
  const T *Bsrc = (L == 0) ? reinterpret_cast<const T *>(D.xT)
                           : reinterpret_cast<const T *>(D.hT) + (size_t)(it.net * 2 + (L - 1)) * H * BP;
  const bool do_bias = (it.i0 == 0) && (it.wi == 0 || wave == 0);

  f32x4 acc[4];
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) acc[jj] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  const T *arow = Asrc + (size_t)(o_base + r) * BP + P::EPV * q;
  const T *brow = Bsrc + (size_t)(i_base + r) * BP + P::EPV * q;
  for (int kb = 0; kb < BP; kb += P::KM) {
    const uint4 a = *reinterpret_cast<const uint4 *>(arow + kb);
    if (do_bias) {
      if constexpr (BF16) {
        const uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) bsum += bf2f((uint16_t)(w[i] & 0xffff)) + bf2f((uint16_t)(w[i] >> 16));
      } else {
        const float4 f = __builtin_bit_cast(float4, a);
        bsum += (f.x + f.y) + (f.z + f.w);
      }
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      if (i_base + 16 * jj < Ipad) {
        const uint4 b = *reinterpret_cast<const uint4 *>(brow + (size_t)jj * 16 * BP + kb);
        P::mma(a, b, acc[jj]);
      }
    }
  }

  // ---- Adam on the tile (gradient stays in registers) ----
  float *Pm = D.params, *Mm = D.exp_avg, *Vm = D.exp_avg_sq;
  T *wc = reinterpret_cast<T *>(N.wc[L]);
  T *tc = N.has_target ? reinterpret_cast<T *>(N.tc[L]) : nullptr;
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    const int i = i_base + 16 * jj + r;
    if (i_base + 16 * jj < Ipad && i < Idim) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int o = o_base + 4 * q + k;
        if (o < Odim) {
          const int64_t e = N.off_w[L] + (int64_t)o * Idim + i;
          const float g = P::round(acc[jj][k]);
          float p = Pm[e], m = Mm[e], v = Vm[e];
          adam_apply(p, m, v, g, coef, neg_step);
          Pm[e] = p, Mm[e] = m, Vm[e] = v;
          if (D.grads) D.grads[e] = g;
          wc[(size_t)o * Kw + i] = P::from_f32(p);
          if (L == 1) reinterpret_cast<T *>(N.w2ct)[(size_t)i * H + o] = P::from_f32(p);
          if (N.has_target) {
            const int64_t te = N.toff_w[L] + (int64_t)o * Idim + i;
            float tv = D.target[te];
            tv = tv + D.tau * (p - tv);  // lerp_ (ref:127-129)
            D.target[te] = tv;
            tc[(size_t)o * Kw + i] = P::from_f32(tv);
          }
        }
      }
    }
  }
  // ---- bias gradient = row sums of dZ^T ----
  if (do_bias) {
    bsum += __shfl_xor(bsum, 16);
    bsum += __shfl_xor(bsum, 32);
    const int o = o_base + r;
    if (q == 0 && o < Odim) {
      const int64_t e = N.off_b[L] + o;
      const float g = P::round(bsum);
      float p = Pm[e], m = Mm[e], v = Vm[e];
      adam_apply(p, m, v, g, coef, neg_step);
      Pm[e] = p, Mm[e] = m, Vm[e] = v;
      if (D.grads) D.grads[e] = g;
      if (N.has_target) {
        const int64_t te = N.toff_b[L] + o;
        float tv = D.target[te];
        tv = tv + D.tau * (p - tv);
        D.target[te] = tv;
      }
    }
  }
}

// ========================================================================
// k_sync_weights: rebuild every compute-precision copy from the fp32 masters
// ========================================================================
template <bool BF16>
__global__ void k_sync_weights(const TrainerDesc D) {
  using P = Prec<BF16>;
  using T = typename P::T;
  const int net = blockIdx.y;
  const TrainNet &N = D.net[net];
  const int H = D.H;
  for (int L = 0; L < 3; ++L) {
    const int Odim = (L == 2) ? N.out_dim : H;
    const int Idim = (L == 0) ? N.in_dim : H;
    const int Kw = (L == 0) ? N.k1pad : H;
    const int total = Odim * Idim;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
      const int o = e / Idim, i = e - o * Idim;
      const float p = D.params[N.off_w[L] + e];
      reinterpret_cast<T *>(N.wc[L])[(size_t)o * Kw + i] = P::from_f32(p);
      if (L == 1) reinterpret_cast<T *>(N.w2ct)[(size_t)i * H + o] = P::from_f32(p);
      if (N.has_target)
        reinterpret_cast<T *>(N.tc[L])[(size_t)o * Kw + i] = P::from_f32(D.target[N.toff_w[L] + e]);
    }
  }
}

// explicit instantiations + launchers used by the host driver (api.hip)
size_t fwd_smem_bytes(bool bf16, int H, int k1max) {
  const int es = bf16 ? 2 : 4, epv = bf16 ? 8 : 4;
  return (size_t)SLAB * (k1max + epv) * es + 2 * (size_t)SLAB * (H + epv) * es + 4 * 2 * 64 * 4 * 4 +
         SLAB * 8;
}
size_t bwd_smem_bytes(bool bf16, int H) {
  const int es = bf16 ? 2 : 4, epv = bf16 ? 8 : 4;
  return (size_t)SLAB * (H + epv) * es + 3 * SLAB * 32 * 4 + SLAB * 4;
}

hipError_t launch_forward(bool bf16, const TrainerDesc &D, const DevArgs *a, const DevCtr *c,
                          hipStream_t st) {
  const int grid = N_FWD * (D.B / SLAB);
  const size_t sm = fwd_smem_bytes(bf16, D.H, D.k1max);
  if (bf16)
    hipLaunchKernelGGL(k_forward<true>, dim3(grid), dim3(256), sm, st, D, a, c);
  else
    hipLaunchKernelGGL(k_forward<false>, dim3(grid), dim3(256), sm, st, D, a, c);
  return hipGetLastError();
}
hipError_t launch_backward(bool bf16, const TrainerDesc &D, const DevArgs *a, DevCtr *c,
                           hipStream_t st) {
  const int grid = N_TRAIN * (D.B / SLAB);
  const size_t sm = bwd_smem_bytes(bf16, D.H);
  if (bf16)
    hipLaunchKernelGGL(k_backward<true>, dim3(grid), dim3(256), sm, st, D, a, c);
  else
    hipLaunchKernelGGL(k_backward<false>, dim3(grid), dim3(256), sm, st, D, a, c);
  return hipGetLastError();
}
hipError_t launch_update(bool bf16, const TrainerDesc &D, const DevArgs *a, DevCtr *c,
                         const UpdItem *items, int n_items, hipStream_t st) {
  if (bf16)
    hipLaunchKernelGGL(k_update<true>, dim3(n_items + 1), dim3(256), 0, st, D, a, c, items, n_items);
  else
    hipLaunchKernelGGL(k_update<false>, dim3(n_items + 1), dim3(256), 0, st, D, a, c, items, n_items);
  return hipGetLastError();
}
hipError_t launch_infer(bool bf16, const TrainerDesc &D, const FwdNet &N, const float *s, const float *a,
                        int64_t n, float *out, int out_stride, hipStream_t st) {
  const int grid = (int)((n + SLAB - 1) / SLAB);
  const size_t sm = fwd_smem_bytes(bf16, D.H, D.k1max);
  if (bf16)
    hipLaunchKernelGGL(k_infer<true>, dim3(grid), dim3(256), sm, st, D, N, s, a, n, out, out_stride);
  else
    hipLaunchKernelGGL(k_infer<false>, dim3(grid), dim3(256), sm, st, D, N, s, a, n, out, out_stride);
  return hipGetLastError();
}
hipError_t launch_sync_weights(bool bf16, const TrainerDesc &D, hipStream_t st) {
  if (bf16)
    hipLaunchKernelGGL(k_sync_weights<true>, dim3(32, N_TRAIN), dim3(256), 0, st, D);
  else
    hipLaunchKernelGGL(k_sync_weights<false>, dim3(32, N_TRAIN), dim3(256), 0, st, D);
  return hipGetLastError();
}

}  // namespace iqlhip
