// IQL optimisation step for MI355X (gfx950): three kernels per step.
//
//   k_forward   (2E+3) MLP evaluations x B/16 row slabs, one work-group each (E critics; E = 2
//               is the reference's TwinQ): the slab's transitions (prefetched by the previous
//               k_update, else gathered from the packed replay rows), 3 Linear layers on
//               MFMA (activations staged in LDS, weights streamed from L2 straight into B
//               fragments), hidden activations of the trained nets stored feature-major for
//               the backward GEMMs.  A spare job writes the step's Adam coefficients.
//   k_backward  (E+2) trained nets x B/16 slabs x 2 halves of W2^T: loss terms (expectile,
//               TD, AWR), d(out), dZ2 (VALU outer product), dZ1 (MFMA), all stored
//               feature-major; per-slab loss partial sums.
//   k_update    weight-gradient GEMMs (K = batch) fused with Adam, the compute-precision
//               weight copies and the Polyak target update: 64x32 tiles for the H-wide
//               layers, 16-row strips with a flat state stream for layer 1; idle slots of
//               the XCD-major item table prefetch the next step's batch.
//
// Every kernel is a latency chain (one 16-row slab / one tile per work-group, 100-260
// work-groups on 256 CUs).  Rules the code follows, each measured:
//   * every global load a work-group will need is requested before its first dependent
//     instruction, in the order the values are consumed (loads return in order);
//   * load phases are branch-free: clamped always-valid addresses, the guard moves to the use
//     (a guarded load is a branch followed by s_waitcnt vmcnt(0)); wave-uniform conditions are
//     scalar; the first k-chunk is straight-line (a loop pre-header drains pending loads);
//   * descriptor scalars used later are pinned at entry (pin_s) so that no scalar load misses
//     in the middle of a kernel;
//   * no runtime-indexed register arrays, no flat (generic) accesses, descriptors in device
//     memory rather than the kernarg segment.
//
// Reference: /root/reference/algorithms/offline/iql.py:581-662 (order of
// operations), :404-405 (expectile), :127-129 (Polyak), torch.optim.Adam
// (_single_tensor_adam) and CosineAnnealingLR closed form.
//
// Data layout: activations / deltas are [feature][batch] ("T" suffix) so that
// both operands of dW = dZ^T X are K(batch)-contiguous 16-byte fragments, and
// the MFMA C/D layout (4 consecutive rows per lane) stores them with 8/16-byte
// writes.
#include <cstdlib>

#include "common.h"
#include "iql_step.h"
#include "step_math.h"

namespace iqlhip {

constexpr int SLAB = 16;  // batch rows per work-group (one MFMA M tile)

// Diagnostic build only (-DIQL_STAMPS): per work-group wall-clock (100 MHz) and
// shader-cycle stamps into D.dbg[kernel][block][8][2]; never compiled into the
// shipped library.
#ifdef IQL_STAMPS
#define STAMP(kern, slot)                                                                   \
  do {                                                                                      \
    if (D.dbg && threadIdx.x == 0 && blockIdx.x < 512) {                                                      \
      unsigned long long *d_ = D.dbg + (((size_t)(kern) * 512 + blockIdx.x) * 8 + (slot)) * 2; \
      d_[0] = wall_clock64();                                                               \
      d_[1] = clock64();                                                                    \
    }                                                                                       \
  } while (0)
#else
#define STAMP(kern, slot) \
  do {                    \
  } while (0)
#endif

// Descriptor fields that a kernel first touches in its middle are fetched there, by a scalar
// load that misses the scalar cache (~0.5 us each, serial).  pin_s() is a use at the top of the
// kernel: the loads join the first batch and the values stay in SGPRs.
template <class T>
__device__ __forceinline__ void pin_s(T v) {
  if constexpr (sizeof(T) == 8) {
    asm volatile("" ::"s"(__builtin_bit_cast(uint64_t, v)));
  } else {
    static_assert(sizeof(T) == 4, "pin_s: 4- or 8-byte values");
    asm volatile("" ::"s"(__builtin_bit_cast(uint32_t, v)));
  }
}

template <bool BF16, int H>
struct KCfg {
  using P = Prec<BF16>;
  static constexpr int TPW = H / 64;                 // n-tiles (16 columns) per wave
  static constexpr int NK2 = H / P::KM;              // k-steps of an H-deep GEMM
  static constexpr int NKC = NK2 < 8 ? NK2 : 8;      // k-steps held in registers at once
  static constexpr int NCH = NK2 / NKC;              // register chunks per H-deep GEMM
  static constexpr int NK1 = BF16 ? 4 : 8;           // layer-1 k-steps in registers (k1pad <= 128)
  static constexpr int NK3 = NK2 >= 4 ? NK2 / 4 : 1; // layer-3 k-steps per wave (K split over waves)
  static constexpr int HP = H + P::EPV;              // LDS row stride: +16 B breaks bank conflicts
};

// w[ks][jj] = B fragment (n-tile tile0+jj, k-step ks0+ks) of a fragment-major weight image
// with nkw k-steps per tile: one contiguous 1 KiB read per wave instruction
template <class P, int NK, int TPW>
__device__ __forceinline__ void load_w(uint4 (&w)[NK][TPW], const typename P::T *W, int nkw, int ks0,
                                       int nk, int tile0, int lane) {
#pragma unroll
  for (int ks = 0; ks < NK; ++ks) {
    if (ks < nk) {
#pragma unroll
      for (int jj = 0; jj < TPW; ++jj)
        w[ks][jj] = ldg16(W + frag_off<P>(tile0 + jj, ks0 + ks, nkw, lane));
    }
  }
}

// acc[jj] += x[16][k0 .. k0 + nk*KM) (LDS, row stride xs) * w
template <class P, int NK, int TPW>
__device__ __forceinline__ void mma_w(const typename P::T *x, int xs, int k0, int nk,
                                      const uint4 (&w)[NK][TPW], f32x4 (&acc)[TPW], int lane) {
  const int r = lane & 15, q = lane >> 4;
  const typename P::T *xrow = x + r * xs + k0 + P::EPV * q;
#pragma unroll
  for (int ks = 0; ks < NK; ++ks) {
    if (ks < nk) {
      const uint4 a = *reinterpret_cast<const uint4 *>(xrow + ks * P::KM);
#pragma unroll
      for (int jj = 0; jj < TPW; ++jj) P::mma(a, w[ks][jj], acc[jj]);
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
    stg8(dst, u);
  } else {
    stg16(dst, make_float4(v[0], v[1], v[2], v[3]));
  }
}

// IQL_WT_ACT (A/B build): the bf16 activations / deltas one kernel hands to the next leave
// write-through (8-byte stores, lean bf16 paths only).
#ifndef IQL_WT_ACT
#define IQL_WT_ACT 0
#endif
__device__ __forceinline__ void act_store8(uint16_t *base, size_t elem, uint2 u) {
#if IQL_WT_ACT
  stg8_wt(base, (uint32_t)elem * 2u, u);
#else
  stg8(base + elem, u);
#endif
}
// IQL_WT_FWD8: the forward's 8-byte hidden-activation stores (16-row work-groups) write-through -- the
// forward's, not the backward's (IQL_WT_ACT covers both: no gain).  A/B on one box
// (round 4, f5, three rounds): one seed 66.1k -> 67.0k steps/s, two seeds equal.
#ifndef IQL_WT_FWD8
#define IQL_WT_FWD8 1
#endif
__device__ __forceinline__ void act_store8_fwd(uint16_t *base, size_t elem, uint2 u) {
#if IQL_WT_FWD8
  stg8_wt(base, (uint32_t)elem * 2u, u);
#else
  act_store8(base, elem, u);
#endif
}

// The packed bf16 C-layout registers of two row tiles m (even) and m + 1 of one feature column, stored
// to a feature-major plane as ONE 16-byte store per lane (permlane16_swap): the lane of row group q
// ends up with rows 16 (m + (q & 1)) + 4 (q & ~1) .. + 7 of its column.  Same bytes at the same
// addresses as two act_store8 calls.
// IQL_WT_ACT16: these 16-byte stores leave write-through (sc0 sc1).  A wave's store instruction covers one
// whole 1 KiB fragment (8 full lines), so nothing is merged in L2 anyway, and what a kernel leaves dirty
// is written back at its END, by the L2s of the few XCDs that hold it: in-kernel stamps (round 4, four
// critics at batch 1024, plain stores) show the last forward work-group done 7.0 us after the first
// started and the first backward work-group stamping at +12.3 us -- 7 MB of activations leaving six L2s;
// with these stores the two are 1.6 us apart.
// A/B on one box (round 4, d2): four critics at batch 1024 31.2k -> 35.3k steps/s, 8 seeds per launch
// 208.7k -> 220.3k (throughput kernels), 4 seeds 156.5k -> 168.0k (k_forward<.., 2, 2>), one seed
// unchanged (66.1k: its 16-row work-groups have no tile pairs).  -DIQL_WT_ACT16=0 builds plain stores.
#ifndef IQL_WT_ACT16
#define IQL_WT_ACT16 1
#endif
__device__ __forceinline__ void act_store16(uint16_t *plane, size_t elem, uint4 v) {
#if IQL_WT_ACT16
  stg16_wt(plane, (uint32_t)elem * 2u, __builtin_bit_cast(float4, v));
#else
  stg16(plane + elem, __builtin_bit_cast(float4, v));
#endif
}
__device__ __forceinline__ void act_store16_pair(uint16_t *plane, int col, int row_m, int q, int nkb, uint2 um,
                                                 uint2 um1) {
  permlane16_swap(um.x, um1.x);
  permlane16_swap(um.y, um1.y);
  const int k0 = row_m + 16 * (q & 1) + 4 * (q & ~1);
  act_store16(plane, fidx<Prec<true>>(col, k0, nkb), make_uint4(um.x, um.y, um1.x, um1.y));
}

// The forward's partial output planes [part][output column][batch row]: a lane's four rows of a column
// as one 16-byte store.  Written through by the throughput kernel only (A/B with the row-major planes
// of 4-byte stores, round 4, f4: 8 seeds per launch 225.4k -> 229.1k steps/s, ONE seed 66.2k -> 64.4k).
template <bool WT>
__device__ __forceinline__ void outs_store4(float *plane, size_t elem, float4 v) {
  if constexpr (WT)
    stg16_wt(plane, (uint32_t)elem * 4u, v);
  else
    stg16(plane + elem, v);
}

// relu(round(acc + bias)) of the four batch rows a lane holds for one hidden unit (MFMA C layout),
// in the compute type.  bf16: relu before the rounding (the same value: rounding is monotone and
// keeps zero), two values per v_cvt_pk_bf16_f32, no round trip through f32 -- a third of the
// vector instructions of round -> relu -> convert, and these epilogues are what bounds a forward
// work-group (one or two waves per SIMD).  u.x = rows 0,1, u.y = rows 2,3.
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ uint32_t pk_bf16(float lo, float hi) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){lo, hi}, bf16x2));
}
__device__ __forceinline__ uint2 relu_bias_bf16x4(const f32x4 &acc, float bias) {
  const float t0 = fmaxf(acc[0] + bias, 0.f), t1 = fmaxf(acc[1] + bias, 0.f);
  const float t2 = fmaxf(acc[2] + bias, 0.f), t3 = fmaxf(acc[3] + bias, 0.f);
  return make_uint2(pk_bf16(t0, t1), pk_bf16(t2, t3));
}

template <bool BF16>
__device__ __forceinline__ void load4T(const typename Prec<BF16>::T *src, float v[4]) {
  if constexpr (BF16) {
    const uint2 u = ldg8(src);
    v[0] = bf2f((uint16_t)(u.x & 0xffff));
    v[1] = bf2f((uint16_t)(u.x >> 16));
    v[2] = bf2f((uint16_t)(u.y & 0xffff));
    v[3] = bf2f((uint16_t)(u.y >> 16));
  } else {
    const float4 f = __builtin_bit_cast(float4, ldg16(src));
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
    for (int i = 0; i < 4; ++i) keep[i] = ldg(m + (size_t)i * D.H) != 0;
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
// Three Linear layers for one 16-row slab.  `fill` writes the (zero padded)
// input rows into LDS; it runs AFTER the weight fragments of all three layers
// have been requested, so the replay gather overlaps the weight fetch.
// ------------------------------------------------------------------------
template <bool BF16, int H, class Issue, class Fill>
__device__ __forceinline__ void mlp_slab(const FwdNet &N, const TrainerDesc &D, const DevArgs *Ap,
                                         int64_t step, int slab, char *smem, float *out, int out_stride,
                                         int64_t row0, int64_t n_valid, Issue issue, Fill fill) {
  using K = KCfg<BF16, H>;
  using P = Prec<BF16>;
  using T = typename P::T;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: wave-uniform branches stay scalar
  const int r = lane & 15, q = lane >> 4;
  constexpr int HP = K::HP, TPW = K::TPW;
  const int K1P = D.k1max + P::EPV;
  T *xs = reinterpret_cast<T *>(smem);                      // [16][K1P]
  T *h1 = xs + SLAB * K1P;                                  // [16][HP]
  T *h2 = h1 + SLAB * HP;                                   // [16][HP]
  float *red = reinterpret_cast<float *>(h2 + SLAB * HP);   // [4][2][64][4]

  STAMP(0, 0);
  // everything the epilogues need from the descriptors joins the first batch of scalar loads
  pin_s(D.hT), pin_s(D.BP), pin_s(N.train_slot), pin_s(N.dropout), pin_s(N.b3), pin_s(N.out_dim);
  pin_s(N.out_col), pin_s(N.tanh_out), pin_s(N.out_pad), pin_s(out), pin_s(out_stride), pin_s(n_valid);
  // the input rows first: loads return in order, so the gather must not queue behind
  // the 160 KB of weight fragments requested next
  issue();
  // ---- request every weight fragment this wave will need ----
  const int nk1 = N.k1pad / P::KM;
  const int nt3 = N.out_pad / 16;  // 1 or 2
  // biases first: they are needed at the END of layer 1 / 2, and loads return in order -- behind
  // the weight stream they would hold layer 1's epilogue until all 160 KB have arrived
  float bias1[TPW], bias2[TPW];
#pragma unroll
  for (int jj = 0; jj < TPW; ++jj) {
    bias1[jj] = ldg(N.b1 + 16 * (wave * TPW + jj) + r);
    bias2[jj] = ldg(N.b2 + 16 * (wave * TPW + jj) + r);
  }
  float bias3[2];  // output-layer bias of this lane's column(s), clamped (used for col < out_dim)
#pragma unroll
  for (int jt = 0; jt < 2; ++jt) bias3[jt] = ldg(N.b3 + (16 * jt + r < N.out_dim ? 16 * jt + r : N.out_dim - 1));
  uint4 w1[K::NK1][TPW], w2[K::NKC][TPW], w3[K::NK3][2];
  const T *W1 = reinterpret_cast<const T *>(N.w1c);
  const T *W2 = reinterpret_cast<const T *>(N.w2c);
  const T *W3 = reinterpret_cast<const T *>(N.w3c);
  load_w<P, K::NK1, TPW>(w1, W1, nk1, 0, nk1, wave * TPW, lane);
  load_w<P, K::NKC, TPW>(w2, W2, K::NK2, 0, K::NKC, wave * TPW, lane);
#pragma unroll
  for (int i = 0; i < K::NK3; ++i) {
    const int ks = wave + 4 * i;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)  // unconditional (clamped): no branch, no drain of the queue
      w3[i][jt] = ldg16(W3 + frag_off<P>(jt < nt3 ? jt : 0, ks < K::NK2 ? ks : 0, K::NK2, lane));
  }

  STAMP(0, 1);
  fill(xs, K1P);
  __syncthreads();
  STAMP(0, 2);

  // ---- hidden layer 1 ----
  {
    f32x4 acc[TPW];
#pragma unroll
    for (int jj = 0; jj < TPW; ++jj) acc[jj] = f32x4{0.f, 0.f, 0.f, 0.f};
    mma_w<P, K::NK1, TPW>(xs, K1P, 0, nk1, w1, acc, lane);
#pragma unroll
    for (int jj = 0; jj < TPW; ++jj) {
      const int col = 16 * (wave * TPW + jj) + r;
      const float bias = P::round(bias1[jj]);
      float v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = fmaxf(P::round(acc[jj][i] + bias), 0.f);
      if (N.dropout) {
        bool keep[4];
        dropout_keep4(D, *Ap, step, 0, slab * 4 + q, col, keep);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = keep[i] ? P::round(v[i] * D.drop_scale) : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) h1[(4 * q + i) * HP + col] = P::from_f32(v[i]);
      if (N.train_slot >= 0)
        store4T<BF16>(reinterpret_cast<T *>(D.hT) + (size_t)(N.train_slot * 2 + 0) * H * D.BP +
                          fidx<P>(col, slab * SLAB + 4 * q, D.BP / P::KM), v);
    }
  }
  __syncthreads();
  STAMP(0, 3);

  // ---- hidden layer 2 ----
  {
    f32x4 acc[TPW];
#pragma unroll
    for (int jj = 0; jj < TPW; ++jj) acc[jj] = f32x4{0.f, 0.f, 0.f, 0.f};
    mma_w<P, K::NKC, TPW>(h1, HP, 0, K::NKC, w2, acc, lane);
#pragma unroll 1
    for (int ch = 1; ch < K::NCH; ++ch) {
      load_w<P, K::NKC, TPW>(w2, W2, K::NK2, ch * K::NKC, K::NKC, wave * TPW, lane);
      mma_w<P, K::NKC, TPW>(h1, HP, ch * K::NKC * P::KM, K::NKC, w2, acc, lane);
    }
#pragma unroll
    for (int jj = 0; jj < TPW; ++jj) {
      const int col = 16 * (wave * TPW + jj) + r;
      const float bias = P::round(bias2[jj]);
      float v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = fmaxf(P::round(acc[jj][i] + bias), 0.f);
      if (N.dropout) {
        bool keep[4];
        dropout_keep4(D, *Ap, step, 1, slab * 4 + q, col, keep);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = keep[i] ? P::round(v[i] * D.drop_scale) : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) h2[(4 * q + i) * HP + col] = P::from_f32(v[i]);
      if (N.train_slot >= 0)
        store4T<BF16>(reinterpret_cast<T *>(D.hT) + (size_t)(N.train_slot * 2 + 1) * H * D.BP +
                          fidx<P>(col, slab * SLAB + 4 * q, D.BP / P::KM), v);
    }
  }
  __syncthreads();
  STAMP(0, 4);

  // ---- output layer: K split over the 4 waves, reduced through LDS ----
  {
    f32x4 acc3[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int i = 0; i < K::NK3; ++i) {
      const int ks = wave + 4 * i;
      if (ks < K::NK2) {
        const uint4 a = *reinterpret_cast<const uint4 *>(h2 + r * HP + ks * P::KM + P::EPV * q);
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
          if (jt < nt3) P::mma(a, w3[i][jt], acc3[jt]);
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
        for (int w = 1; w < 4; ++w) s += *reinterpret_cast<f32x4 *>(red + ((w * 2 + jt) * 64 + lane) * 4);
        const int col = 16 * jt + r;
        if (col < N.out_dim) {
          const float bias = P::round(bias3[jt]);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float v = P::round(s[i] + bias);
            if (N.tanh_out) v = P::round(tanhf(v));
            if (row0 + 4 * q + i < n_valid)
              stg(out + (size_t)(row0 + 4 * q + i) * out_stride + N.out_col + col, v);
          }
        }
      }
    }
  }
  STAMP(0, 5);
}

// ------------------------------------------------------------------------
// Batch staging: the B replay rows of step `step` (ref:211-221 sample: indices from the on-device
// Philox stream, an injected index array, or the identity for an explicit batch) are copied
// into D.stage_rows [B][stride] -- contiguous, so k_forward's gather is a coalesced read with
// no index arithmetic in front of it.  Done by the idle work-groups of k_update for the NEXT
// step while the update tiles work (random HBM rows + TLB misses off the critical path), and
// by k_stage for the first step of a call (or every step when prefetching is off).
// 16 lanes per row, 16-byte accesses (row strides are multiples of 4 floats).
// ------------------------------------------------------------------------
// The same pass leaves what k_update needs of the batch besides the rows: the layer-1 input
// (s | a) in compute precision, feature-major fragment layout (xT, the X operand of the layer-1
// weight-gradient GEMM), reward / done (rd) and the actions in fp32 (actf).
// (tid >> 4 rows per pass: 16 for a 256-thread work-group, 32 for k_update's 512)
template <bool BF16>
__device__ __forceinline__ void stage_rows16(const TrainerDesc &D, const DevArgs &A, int64_t step, int row0,
                                             int row_step, int tid) {
  using P = Prec<BF16>;
  using T = typename P::T;
  const int rr = tid >> 4, l16 = tid & 15;
  const int B = D.B, nq = D.stage_stride >> 2, S = D.S, SA = D.S + D.A, nkb = D.BP / P::KM;
  // (two planes, by step parity: the update kernel of step t reads plane t & 1 while its idle
  // work-groups already fill plane (t + 1) & 1)
  T *const xT = reinterpret_cast<T *>(D.xT) + (size_t)(step & 1) * D.xrows * D.BP;
  for (int row = row0 + rr; row < B; row += row_step) {
    int64_t ix;
    if (A.idx_mode == 1)
      ix = ldg(A.idx + (size_t)(step - A.base_step) * B + row);
    else if (A.idx_mode == 2)
      ix = row;
    else
      ix = philox_index(D.seed, (uint64_t)step, (uint32_t)row, (uint64_t)A.n_rows);
    ix = ix < 0 ? 0 : (ix >= A.n_rows ? A.n_rows - 1 : ix);
    const float *src = A.rows + (size_t)ix * A.row_stride;
    float *dst = D.stage_rows + (size_t)row * D.stage_stride;
    for (int c = l16; c < nq; c += 16) {
      const float4 v4 = __builtin_bit_cast(float4, ldg16(src + 4 * c));
      stg16(dst + 4 * c, v4);
      const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int col = 4 * c + e;  // [0, S + A): s | a, then r, d
        if (col < SA) stg(xT + fidx<P>(col, row, nkb), P::from_f32(v[e]));
        if (col >= S && col < SA) stg(D.actf + (size_t)row * D.A + (col - S), v[e]);
        if (col >= SA && col < SA + 2) stg(D.rd + (size_t)row * 2 + (col - SA), v[e]);
      }
    }
  }
}

template <bool BF16>
__global__ __launch_bounds__(256) void k_stage(const TrainerDesc *__restrict__ Dp,
                                               const DevArgs *__restrict__ Ap,
                                               const DevCtr *__restrict__ Cp) {
  Dp += blockIdx.y, Ap += blockIdx.y, Cp += blockIdx.y;
  stage_rows16<BF16>(*Dp, *Ap, Cp->ctr[0], blockIdx.x * 16, gridDim.x * 16, threadIdx.x);
}

__device__ __forceinline__ void write_adam_coef(const TrainerDesc &D, const DevArgs &A, int64_t t1,
                                                AdamCoef *out);

// ========================================================================
// k_forward
//
// One work-group = (evaluation, slab of 16*MT batch rows, part of layer 2).  Layer 2's output
// features are split into SPL parts of 64 (one 16-column tile per wave): a work-group streams
// all of W1 (every part recomputes the cheap first layer), ONE part of W2 (H x 64) and the
// matching K-slice of W3, i.e. 32 + 32 + 2 KB at H = 256 instead of the 160 KB of an unsplit
// slab -- the per-CU fetch rate from L2 (~70 GB/s) is what bounds this kernel.  The output
// layer therefore produces PARTIAL dot products per part (outs[part][row][col], the bias in
// part 0); k_backward adds the parts in a fixed order (fin_outputs) -- deterministic.
// Grid: (2E+3 evaluations + the spare job) x B/(16 MT) slabs x SPL parts; all work-groups
// of an evaluation sit on one XCD (its weights are fetched into that L2 once).
// ========================================================================
template <bool BF16, int H>
struct FCfg {
  using P = Prec<BF16>;
  static constexpr int TPW = H / 64;                  // layer-1 n-tiles per wave (4 waves cover H)
  static constexpr int SPL = TPW >= 4 ? 4 : TPW;      // parts of layer 2 (H = 256: 4, 128: 2, 64: 1)
  static constexpr int HQ = H / SPL;                  // layer-2 features per part (= 64: one tile per wave)
  static constexpr int NK2 = H / P::KM;               // k-steps of layer 2 (K = H)
  static constexpr int NK3 = HQ / P::KM;              // layer-3 k-steps of one part
  static constexpr int NK1 = BF16 ? 4 : 8;            // layer-1 k-steps in registers (k1pad <= 128)
  static constexpr int HP = H + P::EPV;               // LDS row strides: +16 B breaks bank conflicts
  static constexpr int HQP = HQ + P::EPV;
  static_assert(HQ == 64, "one 16-column tile per wave and part");
};

template <bool BF16, int H, int MT, int PW, bool PRE>
__global__ __launch_bounds__(256, PRE ? 1 : (MT >= 4 ? 2 : 3)) void k_forward(const TrainerDesc *__restrict__ Dp,
                                                 const DevArgs *__restrict__ Ap,
                                                 const DevCtr *__restrict__ Cp, const int nsl_,
                                                 const int nfwd_) {
  using C = FCfg<BF16, H>;
  using P = Prec<BF16>;
  using T = typename P::T;
  // PW = parts of hidden layer 2 (64 units each) per work-group: 1 for a lone seed (most
  // work-groups, shortest chains), 2 for seed groups (layer 1 -- whose epilogue is the vector work
  // that bounds a busy CU -- is recomputed SPL / PW times instead of SPL times).  Every part
  // still produces its own partial layer-3 plane: the sums the consumer forms are the same.
  constexpr int ROWS = 16 * MT, TPW = C::TPW, SPL = C::SPL, HP = C::HP, HQP = C::HQP;
  constexpr int NPG = SPL / PW;  // part groups = work-groups per (evaluation, slab)
  static_assert(SPL % PW == 0 && PW <= 2, "parts per work-group: 1 or 2, dividing the parts");
  // blockIdx.y = seed of a group launch (iqlhip_group_*): the descriptors of the seeds of a
  // group are contiguous arrays; a solo launch has gridDim.y = 1
  Dp += blockIdx.y, Ap += blockIdx.y, Cp += blockIdx.y;
  // blocks are dealt round-robin over the 8 XCDs: job j (evaluation j < nfwd, or the spare job
  // nfwd) lives on XCD j & 7, round j >> 3, with all its slabs and parts (speed only).
  // (nsl_ = slabs of 16 MT rows and nfwd_ arrive as preloaded kernel arguments: the job index,
  // and with it the address of this work-group's FwdNet, must not wait for a descriptor load)
  const TrainerDesc &D = *Dp;
  const int idx_ = blockIdx.x >> 3;
  const int per_job = nsl_ * NPG;
  const int fnet = (idx_ / per_job) * 8 + (blockIdx.x & 7);
  const int rest = idx_ % per_job;
  const int slab = rest / NPG;
  const int part0 = __builtin_amdgcn_readfirstlane((rest % NPG) * PW);  // first of this work-group's parts
  if (fnet > nfwd_) return;
  if (fnet == nfwd_) {
    // spare XCD slot: one thread prepares this step's Adam coefficients for k_update
    if (rest == 0 && threadIdx.x == 0) {
      STAMP(0, 0);
      write_adam_coef(D, *Ap, Cp->ctr[0] + 1, const_cast<AdamCoef *>(&Cp->coef));
      const_cast<DevCtr *>(Cp)->coef_step = Cp->ctr[0] + 1;
      STAMP(0, 5);
    }
    if (rest == 0 && !D.deterministic && (int)threadIdx.x < D.A)
      stg(D.ls_snap + threadIdx.x, ldg(D.params + D.off_log_std + threadIdx.x));
    return;
  }
  // ---- ONE batch of scalar loads: everything the load phase needs, before any branch ----
  const FwdNet N = D.fwd[fnet];
  const int B = D.B, BP = D.BP, OUTW = D.OUTW, k1max = D.k1max;
  const float *const stage = D.stage_rows;
  const unsigned sstride = (unsigned)D.stage_stride;
  float *const g_outs = D.outs;
  void *const g_hT = D.hT;
  const int64_t step = Cp->ctr[0];  // dropout masks only
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: wave-uniform branches stay scalar
  const int r = lane & 15, q = lane >> 4;
  const int K1P = k1max + P::EPV;
  T *xs = reinterpret_cast<T *>(smem);   // [ROWS][K1P]  layer-1 input
  T *h1 = xs + ROWS * K1P;               // [ROWS][HP]   all of hidden layer 1
  T *h2 = h1 + ROWS * HP;                // [PW][ROWS][HQP]  this work-group's parts of hidden layer 2
  STAMP(0, 0);

  // ---- the input rows first (loads return in order): the staged batch, 16 lanes per row,
  // lane l16 takes columns 4 l16 .. +3 and 64 + 4 l16 .. +3 of its row's input segment as two
  // 16-byte loads (segments start on 16-byte boundaries: in_off is 0 or next_off).  Clamped,
  // never branching: columns beyond the segment re-read its last 16 bytes and are zeroed at use.
  const int rr = tid >> 4, l16 = tid & 15;
  const int lim4 = round_up(N.in_dim, 4) - 4;  // last valid 16-byte column group
  float4 xq[MT][2];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int row_ = slab * ROWS + 16 * m + rr;
    const unsigned row = (unsigned)(row_ < B ? row_ : B - 1);
    const float *src = stage + N.in_off;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = 4 * l16 + 64 * j;
      xq[m][j] = __builtin_bit_cast(float4, ldg16(src + (row * sstride + (unsigned)(c < lim4 ? c : lim4))));
    }
  }
  // ---- request every weight fragment this wave will need, biases first: they are needed at
  // the END of a layer and loads return in order.  Uniform (SGPR) bases + one per-lane offset. ----
  const int nk1 = N.k1pad / P::KM;
  const int nt3 = N.out_pad / 16;  // 1 or 2
  const int tile2 = part0 * 4 + wave;  // this wave's n-tile of layer 2 in part j: tile2 + 4 j
  float bias1[TPW], bias3[2];
  {
    const float *b1w = N.b1 + 16 * (wave * TPW);
#pragma unroll
    for (int jj = 0; jj < TPW; ++jj) bias1[jj] = ldg(b1w + 16 * jj + r);
  }
  float bias2[PW];
#pragma unroll
  for (int j = 0; j < PW; ++j) bias2[j] = ldg(N.b2 + 16 * (tile2 + 4 * j) + r);
#pragma unroll
  for (int jt = 0; jt < 2; ++jt)  // clamped (used for col < out_dim)
    bias3[jt] = ldg(N.b3 + (16 * jt + r < N.out_dim ? 16 * jt + r : N.out_dim - 1));
  uint4 w1[C::NK1][TPW], w2[PW][C::NK2], w3[PW][C::NK3][2];
  // PRE requests the fragments of all three layers up front (the shortest chain on paper); the
  // default requests layer 2 behind the layer-1 product and layer 3 behind the layer-2 product:
  // a third fewer live registers, three work-groups per CU instead of two, and the other
  // work-groups cover the wait.  Measured faster everywhere: 63.7k vs 62.5k steps/s for one seed,
  // 164.6k vs 157.3k for a group of 8; only the just-in-time order is instantiated.
  auto load_w2 = [&]() {
#pragma unroll
    for (int j = 0; j < PW; ++j) {
      const T *W2w = reinterpret_cast<const T *>(N.w2c) + (size_t)(tile2 + 4 * j) * C::NK2 * 64 * P::EPV;
#pragma unroll
      for (int ks = 0; ks < C::NK2; ++ks) w2[j][ks] = ldg16(W2w + ks * 64 * P::EPV + lane * P::EPV);
    }
  };
  auto load_w3 = [&]() {
#pragma unroll
    for (int j = 0; j < PW; ++j) {
      const T *W3w = reinterpret_cast<const T *>(N.w3c) + (size_t)((part0 + j) * C::NK3) * 64 * P::EPV;
#pragma unroll
      for (int ks = 0; ks < C::NK3; ++ks)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
          if (jt < nt3) w3[j][ks][jt] = ldg16(W3w + (size_t)(jt * C::NK2 + ks) * 64 * P::EPV + lane * P::EPV);
    }
  };
  {
    // (scalar guard: a CU's L1 port moves 64 B / clk, a redundant 1 KiB fragment load costs the
    // work-group 16 cycles of it -- the load phase of this kernel is bound by exactly that)
    const T *W1w = reinterpret_cast<const T *>(N.w1c) + (size_t)(wave * TPW) * nk1 * 64 * P::EPV;
#pragma unroll
    for (int ks = 0; ks < C::NK1; ++ks) {
      if (ks < nk1) {
#pragma unroll
        for (int jj = 0; jj < TPW; ++jj)
          w1[ks][jj] = ldg16(W1w + (size_t)(jj * nk1 + ks) * 64 * P::EPV + lane * P::EPV);
      }
    }
    if constexpr (PRE) load_w2(), load_w3();
  }
  STAMP(0, 1);

  // ---- layer-1 input into LDS ----
#pragma unroll
  for (int m = 0; m < MT; ++m) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = 4 * l16 + 64 * j;
      const float v[4] = {xq[m][j].x, xq[m][j].y, xq[m][j].z, xq[m][j].w};
      if (c < N.k1pad) {
        T tv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) tv[e] = P::from_f32(c + e < N.in_dim ? v[e] : 0.f);
        if constexpr (BF16) {
          uint2 u;
          u.x = (uint32_t)tv[0] | ((uint32_t)tv[1] << 16), u.y = (uint32_t)tv[2] | ((uint32_t)tv[3] << 16);
          *reinterpret_cast<uint2 *>(xs + (16 * m + rr) * K1P + c) = u;
        } else {
          *reinterpret_cast<float4 *>(xs + (16 * m + rr) * K1P + c) = make_float4(tv[0], tv[1], tv[2], tv[3]);
        }
      }
    }
  }
  __syncthreads();
  STAMP(0, 2);

  // ---- hidden layer 1: all H features (wave w: n-tiles w TPW .. +TPW), MT row tiles ----
  {
    f32x4 acc[MT][TPW];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int jj = 0; jj < TPW; ++jj) acc[m][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < C::NK1; ++ks) {
      if (ks < nk1) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const uint4 a = *reinterpret_cast<const uint4 *>(xs + (16 * m + r) * K1P + ks * P::KM + P::EPV * q);
#pragma unroll
          for (int jj = 0; jj < TPW; ++jj) P::mma(a, w1[ks][jj], acc[m][jj]);
        }
      }
    }
    if constexpr (!PRE) {
      __builtin_amdgcn_sched_barrier(0);
      load_w2();
      __builtin_amdgcn_sched_barrier(0);
    }
    bool lean = false;
    if constexpr (BF16) lean = !N.dropout;  // (wave-uniform) see relu_bias_bf16x4
#pragma unroll
    for (int jj = 0; jj < TPW; ++jj) {
      const int tile = wave * TPW + jj;
      const int col = 16 * tile + r;
      const float bias = P::round(bias1[jj]);
      // the part that owns this tile's features stores them for the backward pass
      const bool mine = N.train_slot >= 0 && ((tile * SPL) / (4 * TPW)) / PW * PW == part0;
      if (lean) {
        if constexpr (BF16) {
          uint2 u[MT];
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            u[m] = relu_bias_bf16x4(acc[m][jj], bias);
            T *hrow = h1 + (16 * m + 4 * q) * HP + col;
            hrow[0] = (T)(u[m].x & 0xffff), hrow[HP] = (T)(u[m].x >> 16);
            hrow[2 * HP] = (T)(u[m].y & 0xffff), hrow[3 * HP] = (T)(u[m].y >> 16);
          }
          T *plane = reinterpret_cast<T *>(g_hT) + (size_t)(N.train_slot * 2 + 0) * H * BP;
          if constexpr (MT >= 2 && !IQL_WT_ACT) {  // (two row tiles = 16 contiguous bytes per lane: one store)
            if (mine && slab * ROWS + ROWS <= B) {
#pragma unroll
              for (int m = 0; m < MT; m += 2) act_store16_pair(plane, col, slab * ROWS + 16 * m, q, BP / P::KM, u[m], u[m + 1]);
            }
          } else {
#pragma unroll
            for (int m = 0; m < MT; ++m)
              if (mine && slab * ROWS + 16 * m < B) act_store8_fwd(plane, fidx<P>(col, slab * ROWS + 16 * m + 4 * q, BP / P::KM), u[m]);
          }
        }
        continue;
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int row0 = slab * ROWS + 16 * m;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = fmaxf(P::round(acc[m][jj][i] + bias), 0.f);
        if (N.dropout) {
          bool keep[4];
          dropout_keep4(D, *Ap, step, 0, (row0 >> 2) + q, col, keep);
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = keep[i] ? P::round(v[i] * D.drop_scale) : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) h1[(16 * m + 4 * q + i) * HP + col] = P::from_f32(v[i]);
        if (mine && row0 < B)
          store4T<BF16>(reinterpret_cast<T *>(g_hT) + (size_t)(N.train_slot * 2 + 0) * H * BP +
                            fidx<P>(col, row0 + 4 * q, BP / P::KM), v);
      }
    }
  }
  __syncthreads();
  STAMP(0, 3);

  // ---- hidden layer 2: 64 features per part, one n-tile per wave and part ----
  {
    f32x4 acc[PW][MT];
#pragma unroll
    for (int j = 0; j < PW; ++j)
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[j][m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < C::NK2; ++ks) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const uint4 a = *reinterpret_cast<const uint4 *>(h1 + (16 * m + r) * HP + ks * P::KM + P::EPV * q);
#pragma unroll
        for (int j = 0; j < PW; ++j) P::mma(a, w2[j][ks], acc[j][m]);
      }
    }
    if constexpr (!PRE) {
      __builtin_amdgcn_sched_barrier(0);
      load_w3();
      __builtin_amdgcn_sched_barrier(0);
    }
    bool lean = false;
    if constexpr (BF16) lean = !N.dropout;
#pragma unroll
    for (int j = 0; j < PW; ++j) {
      const int col = 16 * (tile2 + 4 * j) + r;
      const float bias = P::round(bias2[j]);
      T *h2j = h2 + j * ROWS * HQP;
      if (lean) {
        if constexpr (BF16) {
          uint2 u[MT];
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            u[m] = relu_bias_bf16x4(acc[j][m], bias);
            T *hrow = h2j + (16 * m + 4 * q) * HQP + 16 * wave + r;
            hrow[0] = (T)(u[m].x & 0xffff), hrow[HQP] = (T)(u[m].x >> 16);
            hrow[2 * HQP] = (T)(u[m].y & 0xffff), hrow[3 * HQP] = (T)(u[m].y >> 16);
          }
          T *plane = reinterpret_cast<T *>(g_hT) + (size_t)(N.train_slot * 2 + 1) * H * BP;
          if constexpr (MT >= 2 && !IQL_WT_ACT) {
            if (N.train_slot >= 0 && slab * ROWS + ROWS <= B) {
#pragma unroll
              for (int m = 0; m < MT; m += 2) act_store16_pair(plane, col, slab * ROWS + 16 * m, q, BP / P::KM, u[m], u[m + 1]);
            }
          } else {
#pragma unroll
            for (int m = 0; m < MT; ++m)
              if (N.train_slot >= 0 && slab * ROWS + 16 * m < B)
                act_store8_fwd(plane, fidx<P>(col, slab * ROWS + 16 * m + 4 * q, BP / P::KM), u[m]);
          }
        }
        continue;
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int row0 = slab * ROWS + 16 * m;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = fmaxf(P::round(acc[j][m][i] + bias), 0.f);
        if (N.dropout) {
          bool keep[4];
          dropout_keep4(D, *Ap, step, 1, (row0 >> 2) + q, col, keep);
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = keep[i] ? P::round(v[i] * D.drop_scale) : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) h2j[(16 * m + 4 * q + i) * HQP + 16 * wave + r] = P::from_f32(v[i]);
        if (N.train_slot >= 0 && row0 < B)
          store4T<BF16>(reinterpret_cast<T *>(g_hT) + (size_t)(N.train_slot * 2 + 1) * H * BP +
                            fidx<P>(col, row0 + 4 * q, BP / P::KM), v);
      }
    }
  }
  __syncthreads();
  STAMP(0, 4);

  // ---- output layer, one partial plane per part (64 hidden units each): the MT x PW (row tile,
  // part) pairs are dealt over the waves.  No rounding here: the parts are summed in fp32 by
  // the consumer, then rounded once.
  for (int pr = wave; pr < MT * PW; pr += 4) {
    const int m = pr % MT, j = pr / MT;
    const T *h2j = h2 + j * ROWS * HQP;
    f32x4 acc3[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int ks = 0; ks < C::NK3; ++ks) {
      const uint4 a = *reinterpret_cast<const uint4 *>(h2j + (16 * m + r) * HQP + ks * P::KM + P::EPV * q);
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) {
        if (jt < nt3) {
          if constexpr (PW == 1) {
            P::mma(a, w3[0][ks][jt], acc3[jt]);
          } else {  // (j is wave-uniform: a scalar branch, not a register index)
            if (j == 0) P::mma(a, w3[0][ks][jt], acc3[jt]);
            else P::mma(a, w3[PW - 1][ks][jt], acc3[jt]);
          }
        }
      }
    }
    // planes are [part][output column][batch row]: the four rows a lane holds are ONE 16-byte store, a
    // tile's column 64 contiguous bytes (row-major planes took a 4-byte store per element into 64- to
    // 80-byte rows: 0.9 us of the forward's end at four critics / batch 1024, stamps of round 4, g1)
    float *outp = g_outs + (size_t)(part0 + j) * B * OUTW;
    const int row0 = slab * ROWS + 16 * m + 4 * q;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
      const int col = 16 * jt + r;
      if (jt < nt3 && col < N.out_dim && row0 < B) {
        const float bias = part0 + j == 0 ? P::round(bias3[jt]) : 0.f;
        outs_store4<false>(outp, (size_t)(N.out_col + col) * B + row0,
                           make_float4(acc3[jt][0] + bias, acc3[jt][1] + bias, acc3[jt][2] + bias, acc3[jt][3] + bias));
      }
    }
  }
  STAMP(0, 5);
}

// ========================================================================
// k_forward_tp: the forward pass of launches with MANY rows (seed groups, batch-1024 critic
// ensembles; bf16, H = 256).  Same arithmetic as k_forward, element for element -- every
// accumulator sums its k-steps in ascending order from zero, the output layer leaves the same
// partial plane per 64-unit part of hidden layer 2 -- so a seed stepped by this kernel is
// bit-identical to the seed stepped by k_forward.  What differs is the shape of a work-group.
//
// k_forward is cut for ONE seed at batch 256: 16 or 32 rows x one or two 64-unit parts per
// work-group, 450+ short chains on an idle chip.  With 11,000-14,000 row-evaluations per launch
// the stamps (round 4, tools/stamps.py STAMP_E=4 STAMP_B=1024) show those work-groups waiting on
// each other's weight fragments: a 32-row x 128-unit work-group pulls 112 KB of fragments through
// its CU's vector-memory path (~46 B/clk: a 1 KiB fragment per ~22 cycles, MI355X_MICROARCH
// "vmcnt drain"), three of them share a CU, layer 1 is recomputed per part group, and 12 jobs on
// 8 XCDs run as two rounds.  Here a work-group is 64 rows x ALL of a network (8 waves, two n-tiles
// of every layer each): 168 KB of fragments per 64 rows instead of 224, layer 1 once per slab, every
// fragment requested before the first dependent instruction, one work-group per CU (176 of them
// for 4 critics at batch 1024, 224 for 8 seeds at batch 256: the whole launch is resident at once).
// ========================================================================
template <int H, bool DROP>
__global__ __launch_bounds__(512, 1) void k_forward_tp(const TrainerDesc *__restrict__ Dp,
                                                        const DevArgs *__restrict__ Ap,
                                                        const DevCtr *__restrict__ Cp, const int nsl_,
                                                        const int nfwd_) {
  using C = FCfg<true, H>;
  using P = Prec<true>;
  using T = typename P::T;
  constexpr int ROWS = 64, MT = ROWS / 16, NW = 8;
  constexpr int TPW = H / (16 * NW);  // n-tiles per wave, layers 1 and 2
  constexpr int SPL = C::SPL, HP = C::HP, NK2 = C::NK2, NK3 = C::NK3;
  static_assert(H == 256 && TPW == 2 && SPL == 4, "k_forward_tp is built for H = 256");
  Dp += blockIdx.y, Ap += blockIdx.y, Cp += blockIdx.y;
  const TrainerDesc &D = *Dp;
  // job j (evaluation j < nfwd, or the spare job nfwd) lives on XCD j & 7, round j >> 3
  const int idx_ = blockIdx.x >> 3;
  const int fnet = (idx_ / nsl_) * 8 + (blockIdx.x & 7);
  const int slab = idx_ % nsl_;
  if (fnet > nfwd_) return;
  if (fnet == nfwd_) {
    if (slab == 0 && threadIdx.x == 0) {
      STAMP(0, 0);
      write_adam_coef(D, *Ap, Cp->ctr[0] + 1, const_cast<AdamCoef *>(&Cp->coef));
      const_cast<DevCtr *>(Cp)->coef_step = Cp->ctr[0] + 1;
      STAMP(0, 5);
    }
    if (slab == 0 && !D.deterministic && (int)threadIdx.x < D.A)
      stg(D.ls_snap + threadIdx.x, ldg(D.params + D.off_log_std + threadIdx.x));
    return;
  }
  // ---- ONE batch of scalar loads ----
  const FwdNet N = D.fwd[fnet];
  const int B = D.B, BP = D.BP, OUTW = D.OUTW, k1max = D.k1max;
  const float *const stage = D.stage_rows;
  const unsigned sstride = (unsigned)D.stage_stride;
  float *const g_outs = D.outs;
  void *const g_hT = D.hT;
  const int64_t step = Cp->ctr[0];  // dropout masks only
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int K1P = k1max + P::EPV;
  T *xs = reinterpret_cast<T *>(smem);  // [ROWS][K1P]
  T *h1 = xs + ROWS * K1P;              // [ROWS][HP]
  T *h2 = h1 + ROWS * HP;               // [ROWS][HP]
  STAMP(0, 0);

  // ---- the input rows first: 8 lanes per row, four 16-byte loads each (columns 4 l8 + 32 j) ----
  const int rr = tid >> 3, l8 = tid & 7;
  const int lim4 = round_up(N.in_dim, 4) - 4;
  float4 xq[4];
  {
    const int row_ = slab * ROWS + rr;
    const unsigned row = (unsigned)(row_ < B ? row_ : B - 1);
    const float *src = stage + N.in_off;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = 4 * l8 + 32 * j;
      xq[j] = __builtin_bit_cast(float4, ldg16(src + (row * sstride + (unsigned)(c < lim4 ? c : lim4))));
    }
  }
  // ---- every weight fragment this wave will need, in the order of use ----
  const int nk1 = N.k1pad / P::KM;
  const int nt3 = N.out_pad / 16;
  const int tile0 = wave * TPW;       // this wave's first n-tile of layers 1 and 2
  const int part = tile0 / 4;         // the 64-unit part of layer 2 its tiles belong to (layer-3 K-slice)
  float bias1[TPW], bias2[TPW], bias3[2];
#pragma unroll
  for (int jj = 0; jj < TPW; ++jj) {
    bias1[jj] = ldg(N.b1 + 16 * (tile0 + jj) + r);
    bias2[jj] = ldg(N.b2 + 16 * (tile0 + jj) + r);
  }
#pragma unroll
  for (int jt = 0; jt < 2; ++jt)
    bias3[jt] = ldg(N.b3 + (16 * jt + r < N.out_dim ? 16 * jt + r : N.out_dim - 1));
  uint4 w1[C::NK1][TPW], w2[TPW][NK2], w3[NK3][2];
  {
    const T *W1w = reinterpret_cast<const T *>(N.w1c) + (size_t)tile0 * nk1 * 64 * P::EPV;
#pragma unroll
    for (int ks = 0; ks < C::NK1; ++ks) {
      if (ks < nk1) {
#pragma unroll
        for (int jj = 0; jj < TPW; ++jj)
          w1[ks][jj] = ldg16(W1w + (size_t)(jj * nk1 + ks) * 64 * P::EPV + lane * P::EPV);
      }
    }
  }
  // IQL_FTP_LATE: where the 136 KB of W2 / W3 fragments enter the vector-memory queue -- 0: up front (the
  // waves stall issuing them and reach the input barrier late); 1: behind the input barrier, ahead of the
  // layer-1 product; 2: behind the layer-1 product (k_forward's order)
#ifndef IQL_FTP_LATE
#define IQL_FTP_LATE 0
#endif
  auto load_w23 = [&]() {
#pragma unroll
    for (int jj = 0; jj < TPW; ++jj) {
      const T *W2w = reinterpret_cast<const T *>(N.w2c) + (size_t)(tile0 + jj) * NK2 * 64 * P::EPV;
#pragma unroll
      for (int ks = 0; ks < NK2; ++ks) w2[jj][ks] = ldg16(W2w + ks * 64 * P::EPV + lane * P::EPV);
    }
    const T *W3w = reinterpret_cast<const T *>(N.w3c) + (size_t)(part * NK3) * 64 * P::EPV;
#pragma unroll
    for (int ks = 0; ks < NK3; ++ks)
#pragma unroll
      for (int jt = 0; jt < 2; ++jt)
        if (jt < nt3) w3[ks][jt] = ldg16(W3w + (size_t)(jt * NK2 + ks) * 64 * P::EPV + lane * P::EPV);
  };
  if (IQL_FTP_LATE == 0) load_w23();
  STAMP(0, 1);

  // ---- layer-1 input into LDS ----
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = 4 * l8 + 32 * j;
    const float v[4] = {xq[j].x, xq[j].y, xq[j].z, xq[j].w};
    if (c < N.k1pad) {
      T tv[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) tv[e] = P::from_f32(c + e < N.in_dim ? v[e] : 0.f);
      uint2 u;
      u.x = (uint32_t)tv[0] | ((uint32_t)tv[1] << 16), u.y = (uint32_t)tv[2] | ((uint32_t)tv[3] << 16);
      *reinterpret_cast<uint2 *>(xs + rr * K1P + c) = u;
    }
  }
  __syncthreads();
  STAMP(0, 2);
  if (IQL_FTP_LATE == 1) {
    __builtin_amdgcn_sched_barrier(0);
    load_w23();
    __builtin_amdgcn_sched_barrier(0);
  }

  // DROP = false: the lean bf16 epilogues only (relu_bias_bf16x4); the instantiation with dropout
  // carries the Philox masks.  The host picks by the trainer's has_dropout; a network without dropout
  // inside a DROP launch (every net but the actor) takes the general path with an all-ones mask.
  const bool store_h = N.train_slot >= 0;  // (wave-uniform) B is a multiple of 64: no row guards
  // ---- hidden layer 1 ----
  {
    f32x4 acc[MT][TPW];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int jj = 0; jj < TPW; ++jj) acc[m][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < C::NK1; ++ks) {
      if (ks < nk1) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const uint4 a = *reinterpret_cast<const uint4 *>(xs + (16 * m + r) * K1P + ks * P::KM + P::EPV * q);
#pragma unroll
          for (int jj = 0; jj < TPW; ++jj) P::mma(a, w1[ks][jj], acc[m][jj]);
        }
      }
    }
    STAMP(0, 6);
    if (IQL_FTP_LATE == 2) {
      __builtin_amdgcn_sched_barrier(0);
      load_w23();
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int jj = 0; jj < TPW; ++jj) {
      const int col = 16 * (tile0 + jj) + r;
      const float bias = P::round(bias1[jj]);
      if constexpr (!DROP) {
        uint2 u[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          u[m] = relu_bias_bf16x4(acc[m][jj], bias);
          T *hrow = h1 + (16 * m + 4 * q) * HP + col;
          hrow[0] = (T)(u[m].x & 0xffff), hrow[HP] = (T)(u[m].x >> 16);
          hrow[2 * HP] = (T)(u[m].y & 0xffff), hrow[3 * HP] = (T)(u[m].y >> 16);
        }
        if (store_h) {
          T *plane = reinterpret_cast<T *>(g_hT) + (size_t)(N.train_slot * 2 + 0) * H * BP;
#pragma unroll
          for (int m = 0; m < MT; m += 2)
            act_store16_pair(plane, col, slab * ROWS + 16 * m, q, BP / P::KM, u[m], u[m + 1]);
        }
        continue;
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int row0 = slab * ROWS + 16 * m;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = fmaxf(P::round(acc[m][jj][i] + bias), 0.f);
        if (N.dropout) {
          bool keep[4];
          dropout_keep4(D, *Ap, step, 0, (row0 >> 2) + q, col, keep);
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = keep[i] ? P::round(v[i] * D.drop_scale) : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) h1[(16 * m + 4 * q + i) * HP + col] = P::from_f32(v[i]);
        if (store_h)
          store4T<true>(reinterpret_cast<T *>(g_hT) + (size_t)(N.train_slot * 2 + 0) * H * BP +
                            fidx<P>(col, row0 + 4 * q, BP / P::KM), v);
      }
    }
  }
  __syncthreads();
  STAMP(0, 3);

  // ---- hidden layer 2 ----
  {
    f32x4 acc[TPW][MT];
#pragma unroll
    for (int jj = 0; jj < TPW; ++jj)
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[jj][m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < NK2; ++ks) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const uint4 a = *reinterpret_cast<const uint4 *>(h1 + (16 * m + r) * HP + ks * P::KM + P::EPV * q);
#pragma unroll
        for (int jj = 0; jj < TPW; ++jj) P::mma(a, w2[jj][ks], acc[jj][m]);
      }
    }
    STAMP(0, 7);
#pragma unroll
    for (int jj = 0; jj < TPW; ++jj) {
      const int col = 16 * (tile0 + jj) + r;
      const float bias = P::round(bias2[jj]);
      if constexpr (!DROP) {
        uint2 u[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          u[m] = relu_bias_bf16x4(acc[jj][m], bias);
          T *hrow = h2 + (16 * m + 4 * q) * HP + col;
          hrow[0] = (T)(u[m].x & 0xffff), hrow[HP] = (T)(u[m].x >> 16);
          hrow[2 * HP] = (T)(u[m].y & 0xffff), hrow[3 * HP] = (T)(u[m].y >> 16);
        }
        if (store_h) {
          T *plane = reinterpret_cast<T *>(g_hT) + (size_t)(N.train_slot * 2 + 1) * H * BP;
#pragma unroll
          for (int m = 0; m < MT; m += 2)
            act_store16_pair(plane, col, slab * ROWS + 16 * m, q, BP / P::KM, u[m], u[m + 1]);
        }
        continue;
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int row0 = slab * ROWS + 16 * m;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = fmaxf(P::round(acc[jj][m][i] + bias), 0.f);
        if (N.dropout) {
          bool keep[4];
          dropout_keep4(D, *Ap, step, 1, (row0 >> 2) + q, col, keep);
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = keep[i] ? P::round(v[i] * D.drop_scale) : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) h2[(16 * m + 4 * q + i) * HP + col] = P::from_f32(v[i]);
        if (store_h)
          store4T<true>(reinterpret_cast<T *>(g_hT) + (size_t)(N.train_slot * 2 + 1) * H * BP +
                            fidx<P>(col, row0 + 4 * q, BP / P::KM), v);
      }
    }
  }
  __syncthreads();
  STAMP(0, 4);

  // ---- output layer: one partial plane per 64-unit part, as k_forward leaves them (no rounding:
  // the consumer adds the parts in fp32, then rounds once).  The (part, row tile) pairs are dealt
  // over the waves: wave w holds the layer-3 K-slice of part w / 2 and takes two of its row tiles.
  {
    float *outp = g_outs + (size_t)part * B * OUTW;
#pragma unroll
    for (int mm = 0; mm < MT / 2; ++mm) {
      const int m = (wave & 1) * (MT / 2) + mm;
      f32x4 acc3[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int ks = 0; ks < NK3; ++ks) {
        const uint4 a = *reinterpret_cast<const uint4 *>(h2 + (16 * m + r) * HP + 64 * part + ks * P::KM + P::EPV * q);
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
          if (jt < nt3) P::mma(a, w3[ks][jt], acc3[jt]);
      }
      const int row0 = slab * ROWS + 16 * m + 4 * q;
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) {
        const int col = 16 * jt + r;
        if (jt < nt3 && col < N.out_dim) {
          const float bias = part == 0 ? P::round(bias3[jt]) : 0.f;
          outs_store4<true>(outp, (size_t)(N.out_col + col) * B + row0,
                            make_float4(acc3[jt][0] + bias, acc3[jt][1] + bias, acc3[jt][2] + bias, acc3[jt][3] + bias));
        }
      }
    }
  }
  STAMP(0, 5);
}

// ========================================================================
// k_infer: forward pass of one network on dense inputs (iqlhip_forward;
// ref:452-543 forward()).  Rows beyond n are computed on zeros and not stored.
// ========================================================================
template <bool BF16, int H>
__global__ __launch_bounds__(256) void k_infer(const TrainerDesc *__restrict__ Dp, const FwdNet N,
                                               const float *__restrict__ s,
                                               const float *__restrict__ a, int64_t n, float *out,
                                               int out_stride) {
  using P = Prec<BF16>;
  using T = typename P::T;
  const TrainerDesc &D = *Dp;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int64_t row0 = (int64_t)blockIdx.x * SLAB;
  auto fill = [&](T *xs, int K1P) {
    const int k1 = N.k1pad;
    for (int e = threadIdx.x; e < SLAB * k1; e += 256) {
      const int rr = e / k1, c = e - rr * k1;
      float v = 0.f;
      if (row0 + rr < n && c < N.in_dim)
        v = (c < D.S) ? ldg(s + (size_t)(row0 + rr) * D.S + c) : ldg(a + (size_t)(row0 + rr) * D.A + (c - D.S));
      xs[rr * K1P + c] = P::from_f32(v);
    }
  };
  mlp_slab<BF16, H>(N, D, nullptr, 0, 0, smem, out, out_stride, row0, n, [] {}, fill);
}

// (LossIn, loss_terms: step_math.h)

// k_forward leaves every network output as SPL partial dot products (one per part of hidden
// layer 2, the bias in part 0).  fin_loads requests the partials of one slab -- thread
// (row tid / 16, lane16 = tid % 16) takes columns lane16 + 16 c of its row -- and fin_value adds
// them in a fixed order, rounds the sum to the compute precision (the Linear's output dtype
// under autocast) and applies the actor's tanh (ref:462-470).
constexpr int FIN_NC = 4;   // column groups of 16: OUTW = 2E + 2 + A (rounded to 4) <= 52
constexpr int FIN_LD = 64;  // LDS row stride of the finished outputs

// ========================================================================
// k_backward
//
// (Tried: 512 threads, the dZ2 phase split over two groups of 8 rows.  The phase got 0.12 us
// shorter, the load phase 0.7 us longer, the kernel 45 % slower at 8 seeds per launch: kept at 256.)
//
// One work-group = (trained net, 16-row slab, part of the dZ1 columns).  The W2^T stream is what
// bounds this kernel (per-CU fetch rate from L2), so the H input features of layer 2 are split
// into SPL parts of 64 (one 16-column tile per wave): every part redoes the cheap loss / dZ2
// phase for all H hidden units (dZ2 is the GEMM's K operand) and streams H x 64 of W2^T.
// The parts share the stores: part p writes its 64 columns of dZ2, part 0 writes dZ3 and the
// loss partial sums.
// PW parts per work-group (1, 2 or 4; PW divides SPL): with the chip full anyway (seed groups) the
// redundant loss / dZ2 phases are pure cost -- a work-group then owns PW x 64 columns of dZ1, every
// wave PW n-tiles, and there are SPL / PW work-groups per (net, slab).  The same values either way.
// ========================================================================
template <bool BF16, int H, bool PRE, int PW = 1>
__device__ __forceinline__ void backward_body(const TrainerDesc *__restrict__ Dp, DevCtr *__restrict__ Cp,
                                              const int blk, char *smem, const int nslab, const int ntrain) {
  using K = KCfg<BF16, H>;
  using C = FCfg<BF16, H>;
  using P = Prec<BF16>;
  using T = typename P::T;
  constexpr int SPL = C::SPL;
  static_assert(SPL % PW == 0, "parts per work-group divide the parts");
  constexpr int NPG = SPL / PW;  // part groups = work-groups per (net, slab)
  // Job j = NPG net + part group lives on XCD j & 7, round j >> 3 (each L2 fetches only the part of
  // W2^T its work-groups stream).  (nslab, ntrain: preloaded kernel arguments, see k_forward)
  const TrainerDesc &D = *Dp;
  const int idx_ = blk >> 3;
  const int job = (idx_ / nslab) * 8 + (blk & 7), slab = idx_ % nslab;
  const int net = job / NPG;
  const int part = __builtin_amdgcn_readfirstlane(job % NPG);  // (the part GROUP: parts PW part .. + PW)
  if (net >= ntrain) return;
  // ---- ONE batch of scalar loads: everything the load phase needs, before any branch ----
  const int out_dim = D.net[net].out_dim, out_pad = D.net[net].out_pad;
  const void *const p_w2ct = D.net[net].w2ct, *const p_w3t = D.net[net].w3t;
  const int B = D.B, BP = D.BP, OUTW = D.OUTW, n_act = D.A, opmax = D.opmax;
  const float *const g_outs = D.outs, *const g_rd = D.rd, *const g_actf = D.actf;
  const float *const g_ls = D.deterministic ? D.actf : D.ls_snap;  // unused when deterministic
  const T *const g_hT = reinterpret_cast<const T *>(D.hT);
  T *const g_dz1T = reinterpret_cast<T *>(D.dz1T), *const g_dz2T = reinterpret_cast<T *>(D.dz2T);
  T *const g_dz3T = reinterpret_cast<T *>(D.dz3T);
  float *const g_lsp = D.lsp, *const g_lossp = D.lossp;
  const bool drop_on = D.has_dropout && net == D.net_a;
  const float drop_scale = D.drop_scale;
  const bool is_gauss_actor = net == D.net_a && !D.deterministic;
  const int out_qt = D.out_qt, out_v = D.out_v, out_nv = D.out_nv, out_mean = D.out_mean, n_crit = D.E;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: wave-uniform branches stay scalar
  const int tile0 = part * PW * 4 + wave;  // this wave's first n-tile of dZ1 (of H / 16); the others: + 4 t
  const int r = lane & 15, q = lane >> 4;
  constexpr int HP = K::HP;
  const int nkb = BP / P::KM;
  const float fB = (float)B;

  T *dz2s = reinterpret_cast<T *>(smem);                       // [16][HP]
  float *dz3 = reinterpret_cast<float *>(dz2s + SLAB * HP);    // [32][16] d(loss)/d(out), [j][row]
  float *lterm = dz3 + SLAB * 32;                              // [32][16] loss terms
  float *gstd = lterm + SLAB * 32;                             // [32][16] d(loss)/d(std) terms
  float *rowsum = gstd + SLAB * 32;                            // [16]
  float *fin = rowsum + SLAB;                                  // [16][FIN_LD] finished forward outputs

  if (blk == 0 && tid == 0) Cp->ctr[1] = Cp->ctr[0] + 1;
  STAMP(1, 0);

  // ---- the loss inputs first: loads return in order, these must not queue behind the
  // weight stream requested next.  Thread (row tid / 16, lane16 = tid % 16). ----
  const int lrow = tid >> 4, lj = tid & 15;
  const int brow = slab * SLAB + lrow;
  // (the partial planes are [part][column][row]: for THIS phase thread -> (row tid % 16, column tid / 16 + 16 c),
  // 16 consecutive rows of a column = 64 contiguous bytes per 16 lanes)
  const int frow = tid & 15, fcq = tid >> 4;
  float pv[FIN_NC][SPL];
  {
    const float *o = g_outs + (size_t)(slab * SLAB + frow);
#pragma unroll
    for (int c = 0; c < FIN_NC; ++c) {
      const int col = fcq + 16 * c;
      const unsigned cc = (unsigned)(col < OUTW ? col : OUTW - 1);  // clamped, unused beyond OUTW
#pragma unroll
      for (int p = 0; p < SPL; ++p) pv[c][p] = ldg(o + ((unsigned)p * (unsigned)OUTW + cc) * (unsigned)B);
    }
  }
  const float rew = ldg(g_rd + (size_t)brow * 2), done = ldg(g_rd + (size_t)brow * 2 + 1);
  float actv[2], lsv[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {  // outputs lj and lj + 16 (A <= 32); clamped, unused beyond A
    const int jc = lj + 16 * h < n_act ? lj + 16 * h : n_act - 1;
    actv[h] = ldg(g_actf + (size_t)brow * n_act + jc);
    lsv[h] = ldg(g_ls + jc);
  }
  // keep these loads AHEAD of the weight stream (the scheduler otherwise moves some of them
  // behind it, and loads return in order)
  __builtin_amdgcn_sched_barrier(0);
  // ---- request everything else that does not depend on the loss, in the order it is
  // consumed: W3 and h2 for the dZ2 phase, then this part of W2^T for the GEMM, h1 for its
  // epilogue.  All unconditional (clamped). ----
  const int c2 = tid < H ? tid : H - 1;  // hidden unit this thread owns in the dZ2 phase
  // W3^T [H][out_pad]: this unit's weights to every output, 8 (bf16) / 4 (fp32) per 16-byte load
  constexpr int W3G = 32 / P::EPV;  // 16-byte groups covering 32 outputs
  uint4 w3q[W3G];
  {
    const T *w3row = reinterpret_cast<const T *>(p_w3t) + (size_t)c2 * out_pad;
#pragma unroll
    for (int g = 0; g < W3G; ++g)
      if (g * P::EPV < out_pad) w3q[g] = ldg16(w3row + g * P::EPV);  // (scalar guard: no redundant loads)
  }
  float h2v[16];
  {
    const T *h2T = g_hT + (size_t)(net * 2 + 1) * H * BP;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4)
      load4T<BF16>(h2T + fidx<P>(c2, slab * SLAB + 4 * g4, nkb), &h2v[4 * g4]);
  }
  __builtin_amdgcn_sched_barrier(0);
  // PRE: the operands of the closing GEMM are requested here, ahead of everything; otherwise
  // behind the dZ2 phase (fewer live registers, more work-groups per CU; A/B: IQLHIP_BWD_PRE)
  uint4 w2t[PW][K::NK2];
  float h1v[PW][4];
  auto load_gemm_operands = [&]() {
#pragma unroll
    for (int t = 0; t < PW; ++t) {
      const T *W2Tw = reinterpret_cast<const T *>(p_w2ct) + (size_t)(tile0 + 4 * t) * K::NK2 * 64 * P::EPV;
#pragma unroll
      for (int ks = 0; ks < K::NK2; ++ks) w2t[t][ks] = ldg16(W2Tw + ks * 64 * P::EPV + lane * P::EPV);
      load4T<BF16>(g_hT + (size_t)(net * 2 + 0) * H * BP + fidx<P>(16 * (tile0 + 4 * t) + r, slab * SLAB + 4 * q, nkb),
                   h1v[t]);
    }
  };
  if constexpr (PRE) load_gemm_operands();
  STAMP(1, 1);

  // ---- finish the forward outputs of this slab (sum of the parts, rounding, tanh) ----
#pragma unroll
  for (int c = 0; c < FIN_NC; ++c) {
    const int col = fcq + 16 * c;
    if (col < OUTW) {
      float sum = pv[c][0];
#pragma unroll
      for (int p = 1; p < SPL; ++p) sum += pv[c][p];
      const float v = P::round(sum);
      // the actor's mean columns carry its tanh (ref:462-470), rounded like every autocast output
      fin[frow * FIN_LD + col] = (col >= out_mean && col < out_mean + n_act) ? P::round(tanhf(v)) : v;
    }
  }
  __syncthreads();

  // ---- per-row loss terms and d(loss)/d(out)  (ref:581-637), LDS layout [j][16 rows] ----
  {
    const float *f = fin + lrow * FIN_LD;
    LossIn lin;
    // TwinQ.forward = min(q1, q2) of the target critics (ref:531-533, 583-584); min over all E
#pragma unroll
    for (int e = 0; e < MAX_CRITICS; ++e) lin.qt[e] = f[out_qt + (e < n_crit ? e : 0)];
    lin.vv = f[out_v], lin.nv = f[out_nv], lin.qv = f[net < n_crit ? net : 0];
    lin.rew = rew, lin.done = done;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int j = lj + 16 * h;
      if (h == 0 || out_dim > 16) {
        lin.mean = f[out_mean + (j < n_act ? j : n_act - 1)];
        lin.act = actv[h], lin.ls = lsv[h];
        float d3, lt, gs;
        loss_terms<BF16>(D, net, lin, fB, d3, lt, gs);
        if (j < out_dim) dz3[j * SLAB + lrow] = d3, lterm[j * SLAB + lrow] = lt, gstd[j * SLAB + lrow] = gs;
      }
    }
  }
  __syncthreads();
  STAMP(1, 2);

  // per-slab partial sums in a fixed order (deterministic): rows first, then the 16 row sums
  if (tid < SLAB) {
    float s = 0.f;
    for (int j = 0; j < out_dim; ++j) s += lterm[j * SLAB + tid];
    rowsum[tid] = s;
  }
  if (part == 0 && is_gauss_actor && tid >= 64 && tid < 64 + n_act) {
    const int j = tid - 64;
    float s = 0.f;
#pragma unroll
    for (int rr = 0; rr < SLAB; ++rr) s += gstd[j * SLAB + rr];
    stg(g_lsp + (size_t)slab * n_act + j, s);
  }
  // d(out), feature-major, for the layer-3 weight gradient
  for (int e = tid; part == 0 && e < out_dim * SLAB; e += 256) {
    const int j = e / SLAB, rr = e - j * SLAB;
    stg(g_dz3T + (size_t)net * opmax * BP + fidx<P>(j, slab * SLAB + rr, nkb),
        P::from_f32(dz3[j * SLAB + rr]));
  }

  // ---- dZ2 = (dZ3 W3) * relu'(h2)   (VALU: K = out_dim <= 32) ----
  if (tid < H) {
    T *dst = g_dz2T + (size_t)net * H * BP;
    float s[SLAB];
#pragma unroll
    for (int rr = 0; rr < SLAB; ++rr) s[rr] = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      if (j < out_dim) {
        const uint4 &g = w3q[j / P::EPV];
        float w3j;
        if constexpr (BF16) {
          const uint32_t wds[4] = {g.x, g.y, g.z, g.w};
          const uint32_t wd = wds[(j % 8) >> 1];
          w3j = bf2f((uint16_t)((j & 1) ? (wd >> 16) : (wd & 0xffff)));
        } else {
          const float4 gf = __builtin_bit_cast(float4, g);
          const float fs[4] = {gf.x, gf.y, gf.z, gf.w};
          w3j = fs[j % 4];
        }
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const float4 dv = *reinterpret_cast<const float4 *>(&dz3[j * SLAB + 4 * g4]);
          s[4 * g4] += dv.x * w3j, s[4 * g4 + 1] += dv.y * w3j;
          s[4 * g4 + 2] += dv.z * w3j, s[4 * g4 + 3] += dv.w * w3j;
        }
      }
    }
    const bool mine = (c2 / C::HQ) / PW == part;  // the part (group) that owns this hidden unit stores its dZ2
    bool lean = false;
    if constexpr (BF16) lean = !drop_on;  // mask before the rounding (the same value), packed conversion
    if constexpr (BF16 && !IQL_WT_ACT) {
      if (lean) {
        // rows 8 h .. 8 h + 7 of a hidden unit are 16 contiguous bytes of the feature-major plane: two
        // 16-byte stores per thread instead of four 8-byte ones (store instructions, not bytes, bound
        // these epilogues)
#pragma unroll
        for (int h8 = 0; h8 < 2; ++h8) {
          uint2 u2[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int g4 = 2 * h8 + e;
            float t[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) t[i] = h2v[4 * g4 + i] > 0.f ? s[4 * g4 + i] : 0.f;
            u2[e] = make_uint2(pk_bf16(t[0], t[1]), pk_bf16(t[2], t[3]));
            T *drow = dz2s + (4 * g4) * HP + c2;
            drow[0] = (T)(u2[e].x & 0xffff), drow[HP] = (T)(u2[e].x >> 16);
            drow[2 * HP] = (T)(u2[e].y & 0xffff), drow[3 * HP] = (T)(u2[e].y >> 16);
          }
          if (mine)
            act_store16(dst, fidx<P>(c2, slab * SLAB + 8 * h8, nkb), make_uint4(u2[0].x, u2[0].y, u2[1].x, u2[1].y));
        }
      }
    }
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      if (lean) {
        if constexpr (BF16 && IQL_WT_ACT) {
          float t[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) t[i] = h2v[4 * g4 + i] > 0.f ? s[4 * g4 + i] : 0.f;
          const uint2 u = make_uint2(pk_bf16(t[0], t[1]), pk_bf16(t[2], t[3]));
          T *drow = dz2s + (4 * g4) * HP + c2;
          drow[0] = (T)(u.x & 0xffff), drow[HP] = (T)(u.x >> 16);
          drow[2 * HP] = (T)(u.y & 0xffff), drow[3 * HP] = (T)(u.y >> 16);
          if (mine) act_store8(dst, fidx<P>(c2, slab * SLAB + 4 * g4, nkb), u);
        }
        continue;
      }
      float outv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int rr = 4 * g4 + i;
        float sv = P::round(s[rr]);
        if (drop_on) sv = P::round(sv * drop_scale);
        outv[i] = h2v[rr] > 0.f ? sv : 0.f;
        dz2s[rr * HP + c2] = P::from_f32(outv[i]);
      }
      if (mine) store4T<BF16>(dst + fidx<P>(c2, slab * SLAB + 4 * g4, nkb), outv);
    }
  }
  if constexpr (!PRE) {
    __builtin_amdgcn_sched_barrier(0);
    load_gemm_operands();
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();
  STAMP(1, 3);
  if (tid == 0 && part == 0) {
    float s = 0.f;
#pragma unroll
    for (int rr = 0; rr < SLAB; ++rr) s += rowsum[rr];
    stg(g_lossp + net * nslab + slab, s);
  }

  // ---- dZ1 = (dZ2 W2) * relu'(h1)   (MFMA, B operand = transposed copy; PW n-tiles per wave) ----
#pragma unroll
  for (int t = 0; t < PW; ++t) {
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    const T *xrow = dz2s + r * HP + P::EPV * q;
#pragma unroll
    for (int ks = 0; ks < K::NK2; ++ks) {
      const uint4 a = *reinterpret_cast<const uint4 *>(xrow + ks * P::KM);
      P::mma(a, w2t[t][ks], acc);
    }
    const int col = 16 * (tile0 + 4 * t) + r;
    bool lean = false;
    if constexpr (BF16) lean = !drop_on;
    if (lean) {
      if constexpr (BF16) {
        float tv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) tv[i] = h1v[t][i] > 0.f ? acc[i] : 0.f;
        act_store8(g_dz1T + (size_t)net * H * BP, fidx<P>(col, slab * SLAB + 4 * q, nkb),
                   make_uint2(pk_bf16(tv[0], tv[1]), pk_bf16(tv[2], tv[3])));
      }
    } else {
      float outv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float s = P::round(acc[i]);
        if (drop_on) s = P::round(s * drop_scale);
        outv[i] = h1v[t][i] > 0.f ? s : 0.f;
      }
      store4T<BF16>(g_dz1T + (size_t)net * H * BP + fidx<P>(col, slab * SLAB + 4 * q, nkb), outv);
    }
  }
  STAMP(1, 4);
}

// ========================================================================
// k_backward_tp: the backward pass of launches with many rows (bf16, H = 256; see k_forward_tp).
// One work-group = (trained net, 32 batch rows, ALL dZ1 columns): 512 threads, the two halves of
// the work-group run the loss / dZ3 / dZ2 phases of k_backward for one 16-row slab each (the
// same thread -> element map, the same order of every sum), then all eight waves share the
// closing GEMM dZ1 = dZ2 W2 -- two n-tiles x both slabs per wave.  k_backward cuts the dZ1
// columns into parts so that ONE seed fills the chip: every part group repeats the loss / dZ2
// phase and streams its share of W2^T per 16 rows (64 KB per work-group, 768 work-groups for four
// critics at batch 1024).  Here nothing is repeated and W2^T is streamed once per 32 rows
// (192-256 work-groups, one per CU, every operand requested up front).  The same bits as k_backward.
// ========================================================================
template <int H>
__global__ __launch_bounds__(512, 1) void k_backward_tp(const TrainerDesc *__restrict__ Dp,
                                                         const DevArgs *__restrict__ Ap, DevCtr *__restrict__ Cp,
                                                         const int nslab32, const int ntrain, const int nxn) {
  constexpr bool BF16 = true;
  using K = KCfg<true, H>;
  using C = FCfg<true, H>;
  using P = Prec<true>;
  using T = typename P::T;
  constexpr int SPL = C::SPL, NW = 8, TPW = H / (16 * NW), HP = K::HP;
  static_assert(H == 256 && TPW == 2, "k_backward_tp is built for H = 256");
  Dp += blockIdx.y, Cp += blockIdx.y;
  const TrainerDesc &D = *Dp;
  // Virtual job vj = nxn net + (slab32 % nxn) lives on XCD vj & 7, round vj >> 3: with few trained
  // nets (TwinQ: 4) each takes nxn = 2 XCDs, so that a group launch uses all eight.
  const int blk = (int)blockIdx.x;
  const int idx_ = blk >> 3;
  const int per_vj = nslab32 / nxn;
  const int vj = (idx_ / per_vj) * 8 + (blk & 7), s_in = idx_ % per_vj;
  const int net = vj / nxn, slab32 = s_in * nxn + vj % nxn;
  if (net >= ntrain) return;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // ---- ONE batch of scalar loads ----
  const int out_dim = D.net[net].out_dim, out_pad = D.net[net].out_pad;
  const void *const p_w2ct = D.net[net].w2ct, *const p_w3t = D.net[net].w3t;
  const int B = D.B, BP = D.BP, OUTW = D.OUTW, n_act = D.A, opmax = D.opmax;
  const float *const g_outs = D.outs, *const g_rd = D.rd, *const g_actf = D.actf;
  const float *const g_ls = D.deterministic ? D.actf : D.ls_snap;
  const T *const g_hT = reinterpret_cast<const T *>(D.hT);
  T *const g_dz1T = reinterpret_cast<T *>(D.dz1T), *const g_dz2T = reinterpret_cast<T *>(D.dz2T);
  T *const g_dz3T = reinterpret_cast<T *>(D.dz3T);
  float *const g_lsp = D.lsp, *const g_lossp = D.lossp;
  const bool drop_on = D.has_dropout && net == D.net_a;
  const float drop_scale = D.drop_scale;
  const bool is_gauss_actor = net == D.net_a && !D.deterministic;
  const int out_qt = D.out_qt, out_v = D.out_v, out_nv = D.out_nv, out_mean = D.out_mean, n_crit = D.E;
  const int tid_ = threadIdx.x, lane = tid_ & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);
  const int sub = wave >> 2;         // which 16-row slab of the pair this half of the work-group owns
  const int tid = tid_ & 255;        // thread index inside the half: k_backward's tid
  const int slab = slab32 * 2 + sub; // 16-row slab index, as in k_backward
  const int nslab = B / SLAB;
  const int r = lane & 15, q = lane >> 4;
  const int nkb = BP / P::KM;
  const float fB = (float)B;

  constexpr int SUB_BYTES = SLAB * HP * 2 + 3 * SLAB * 32 * 4 + SLAB * 4 + SLAB * FIN_LD * 4;
  static_assert(SUB_BYTES % 16 == 0, "LDS regions stay 16-byte aligned");
  char *const sm = smem + sub * SUB_BYTES;
  T *dz2s = reinterpret_cast<T *>(sm);                         // [16][HP]
  float *dz3 = reinterpret_cast<float *>(dz2s + SLAB * HP);    // [32][16]
  float *lterm = dz3 + SLAB * 32;
  float *gstd = lterm + SLAB * 32;
  float *rowsum = gstd + SLAB * 32;                            // [16]
  float *fin = rowsum + SLAB;                                  // [16][FIN_LD]

  if (blk == 0 && tid_ == 0) Cp->ctr[1] = Cp->ctr[0] + 1;
  STAMP(1, 0);

  // ---- the loss inputs first (thread (row tid / 16, lane16 = tid % 16) of its half) ----
  const int lrow = tid >> 4, lj = tid & 15;
  const int brow = slab * SLAB + lrow;
  const int frow = tid & 15, fcq = tid >> 4;  // (the partial planes are [part][column][row]: see k_backward)
  float pv[FIN_NC][SPL];
  {
    const float *o = g_outs + (size_t)(slab * SLAB + frow);
#pragma unroll
    for (int c = 0; c < FIN_NC; ++c) {
      const int col = fcq + 16 * c;
      const unsigned cc = (unsigned)(col < OUTW ? col : OUTW - 1);
#pragma unroll
      for (int p = 0; p < SPL; ++p) pv[c][p] = ldg(o + ((unsigned)p * (unsigned)OUTW + cc) * (unsigned)B);
    }
  }
  const float rew = ldg(g_rd + (size_t)brow * 2), done = ldg(g_rd + (size_t)brow * 2 + 1);
  float actv[2], lsv[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int jc = lj + 16 * h < n_act ? lj + 16 * h : n_act - 1;
    actv[h] = ldg(g_actf + (size_t)brow * n_act + jc);
    lsv[h] = ldg(g_ls + jc);
  }
  __builtin_amdgcn_sched_barrier(0);
  const int c2 = tid;  // hidden unit this thread owns in the dZ2 phase (H = 256 = threads per half)
  constexpr int W3G = 32 / P::EPV;
  uint4 w3q[W3G];
  {
    const T *w3row = reinterpret_cast<const T *>(p_w3t) + (size_t)c2 * out_pad;
#pragma unroll
    for (int g = 0; g < W3G; ++g)
      if (g * P::EPV < out_pad) w3q[g] = ldg16(w3row + g * P::EPV);
  }
  float h2v[16];
  {
    const T *h2T = g_hT + (size_t)(net * 2 + 1) * H * BP;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4)
      load4T<BF16>(h2T + fidx<P>(c2, slab * SLAB + 4 * g4, nkb), &h2v[4 * g4]);
  }
  __builtin_amdgcn_sched_barrier(0);
  // the operands of the closing GEMM: wave w -> n-tiles 2w, 2w + 1 of dZ1, both slabs of the pair
  const int tile0 = wave * TPW;
  uint4 w2t[TPW][K::NK2];
  float h1v[TPW][2][4];
  // IQL_BTP_LATE: where the 128 KB of W2^T enter the vector-memory queue -- 0: up front, behind the
  // loss inputs (the waves stall issuing them and reach the loss phase's barriers late); 1: behind the
  // finished forward outputs; 2: behind the loss terms.  A/B on one box (round 4, e5; steps/s for four
  // critics at batch 1024 / 8 seeds per launch): 37.6k / 224.4k, 37.8k / 226.4k, 37.7k / 226.4k.
#ifndef IQL_BTP_LATE
#define IQL_BTP_LATE 1
#endif
  auto load_gemm_operands = [&]() {
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const T *W2Tw = reinterpret_cast<const T *>(p_w2ct) + (size_t)(tile0 + t) * K::NK2 * 64 * P::EPV;
#pragma unroll
      for (int ks = 0; ks < K::NK2; ++ks) w2t[t][ks] = ldg16(W2Tw + ks * 64 * P::EPV + lane * P::EPV);
#pragma unroll
      for (int m = 0; m < 2; ++m)
        load4T<BF16>(g_hT + (size_t)(net * 2 + 0) * H * BP +
                         fidx<P>(16 * (tile0 + t) + r, (slab32 * 2 + m) * SLAB + 4 * q, nkb), h1v[t][m]);
    }
  };
  if (IQL_BTP_LATE == 0) load_gemm_operands();
  STAMP(1, 1);

  // ---- finish the forward outputs of this slab (sum of the parts, rounding, tanh) ----
#pragma unroll
  for (int c = 0; c < FIN_NC; ++c) {
    const int col = fcq + 16 * c;
    if (col < OUTW) {
      float sum = pv[c][0];
#pragma unroll
      for (int p = 1; p < SPL; ++p) sum += pv[c][p];
      const float v = P::round(sum);
      fin[frow * FIN_LD + col] = (col >= out_mean && col < out_mean + n_act) ? P::round(tanhf(v)) : v;
    }
  }
  if (IQL_BTP_LATE == 1) {
    __builtin_amdgcn_sched_barrier(0);
    load_gemm_operands();
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();

  // ---- per-row loss terms and d(loss)/d(out)  (ref:581-637) ----
  {
    const float *f = fin + lrow * FIN_LD;
    LossIn lin;
#pragma unroll
    for (int e = 0; e < MAX_CRITICS; ++e) lin.qt[e] = f[out_qt + (e < n_crit ? e : 0)];
    lin.vv = f[out_v], lin.nv = f[out_nv], lin.qv = f[net < n_crit ? net : 0];
    lin.rew = rew, lin.done = done;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int j = lj + 16 * h;
      if (h == 0 || out_dim > 16) {
        lin.mean = f[out_mean + (j < n_act ? j : n_act - 1)];
        lin.act = actv[h], lin.ls = lsv[h];
        float d3, lt, gs;
        loss_terms<BF16>(D, net, lin, fB, d3, lt, gs);
        if (j < out_dim) dz3[j * SLAB + lrow] = d3, lterm[j * SLAB + lrow] = lt, gstd[j * SLAB + lrow] = gs;
      }
    }
  }
  if (IQL_BTP_LATE == 2) {
    __builtin_amdgcn_sched_barrier(0);
    load_gemm_operands();
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();
  STAMP(1, 2);

  if (tid < SLAB) {
    float s = 0.f;
    for (int j = 0; j < out_dim; ++j) s += lterm[j * SLAB + tid];
    rowsum[tid] = s;
  }
  if (is_gauss_actor && tid >= 64 && tid < 64 + n_act) {
    const int j = tid - 64;
    float s = 0.f;
#pragma unroll
    for (int rr = 0; rr < SLAB; ++rr) s += gstd[j * SLAB + rr];
    stg(g_lsp + (size_t)slab * n_act + j, s);
  }
  for (int e = tid; e < out_dim * SLAB; e += 256) {
    const int j = e / SLAB, rr = e - j * SLAB;
    stg(g_dz3T + (size_t)net * opmax * BP + fidx<P>(j, slab * SLAB + rr, nkb), P::from_f32(dz3[j * SLAB + rr]));
  }

  // ---- dZ2 = (dZ3 W3) * relu'(h2)   (VALU: K = out_dim <= 32) ----
  {
    T *dst = g_dz2T + (size_t)net * H * BP;
    float s[SLAB];
#pragma unroll
    for (int rr = 0; rr < SLAB; ++rr) s[rr] = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      if (j < out_dim) {
        const uint4 &g = w3q[j / P::EPV];
        const uint32_t wds[4] = {g.x, g.y, g.z, g.w};
        const uint32_t wd = wds[(j % 8) >> 1];
        const float w3j = bf2f((uint16_t)((j & 1) ? (wd >> 16) : (wd & 0xffff)));
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const float4 dv = *reinterpret_cast<const float4 *>(&dz3[j * SLAB + 4 * g4]);
          s[4 * g4] += dv.x * w3j, s[4 * g4 + 1] += dv.y * w3j;
          s[4 * g4 + 2] += dv.z * w3j, s[4 * g4 + 3] += dv.w * w3j;
        }
      }
    }
    const bool lean = !drop_on;
    if (lean) {
      // rows 8 h .. 8 h + 7 of a hidden unit are 16 contiguous bytes of the feature-major plane: two
      // 16-byte stores per thread instead of four 8-byte ones
#pragma unroll
      for (int h8 = 0; h8 < 2; ++h8) {
        uint2 u2[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int g4 = 2 * h8 + e;
          float t[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) t[i] = h2v[4 * g4 + i] > 0.f ? s[4 * g4 + i] : 0.f;
          u2[e] = make_uint2(pk_bf16(t[0], t[1]), pk_bf16(t[2], t[3]));
          T *drow = dz2s + (4 * g4) * HP + c2;
          drow[0] = (T)(u2[e].x & 0xffff), drow[HP] = (T)(u2[e].x >> 16);
          drow[2 * HP] = (T)(u2[e].y & 0xffff), drow[3 * HP] = (T)(u2[e].y >> 16);
        }
        act_store16(dst, fidx<P>(c2, slab * SLAB + 8 * h8, nkb), make_uint4(u2[0].x, u2[0].y, u2[1].x, u2[1].y));
      }
    } else {
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        float outv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int rr = 4 * g4 + i;
          float sv = P::round(s[rr]);
          if (drop_on) sv = P::round(sv * drop_scale);
          outv[i] = h2v[rr] > 0.f ? sv : 0.f;
          dz2s[rr * HP + c2] = P::from_f32(outv[i]);
        }
        store4T<BF16>(dst + fidx<P>(c2, slab * SLAB + 4 * g4, nkb), outv);
      }
    }
  }
  __syncthreads();
  STAMP(1, 3);
  if (tid == 0) {
    float s = 0.f;
#pragma unroll
    for (int rr = 0; rr < SLAB; ++rr) s += rowsum[rr];
    stg(g_lossp + net * nslab + slab, s);
  }

  // ---- dZ1 = (dZ2 W2) * relu'(h1)   (MFMA; wave w: n-tiles 2w, 2w + 1 x both slabs) ----
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    f32x4 acc[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
      const T *xrow = reinterpret_cast<const T *>(smem + m * SUB_BYTES) + r * HP + P::EPV * q;
#pragma unroll
      for (int ks = 0; ks < K::NK2; ++ks) {
        const uint4 a = *reinterpret_cast<const uint4 *>(xrow + ks * P::KM);
        P::mma(a, w2t[t][ks], acc[m]);
      }
    }
    const int col = 16 * (tile0 + t) + r;
    if (!drop_on) {
      uint2 u[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        float tv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) tv[i] = h1v[t][m][i] > 0.f ? acc[m][i] : 0.f;
        u[m] = make_uint2(pk_bf16(tv[0], tv[1]), pk_bf16(tv[2], tv[3]));
      }
      act_store16_pair(g_dz1T + (size_t)net * H * BP, col, slab32 * 2 * SLAB, q, nkb, u[0], u[1]);
    } else {
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        float outv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float sv = P::round(acc[m][i]);
          sv = P::round(sv * drop_scale);
          outv[i] = h1v[t][m][i] > 0.f ? sv : 0.f;
        }
        store4T<BF16>(g_dz1T + (size_t)net * H * BP + fidx<P>(col, (slab32 * 2 + m) * SLAB + 4 * q, nkb), outv);
      }
    }
  }
  STAMP(1, 4);
}

// ========================================================================
// k_update
// ========================================================================
// (polyak, adam_apply, ipow, make_adam_coef: step_math.h)

// Adam bias corrections and the cosine actor lr of the step in flight (1-based
// count t1), computed once by one thread of a spare k_forward block.
__device__ __forceinline__ void write_adam_coef(const TrainerDesc &D, const DevArgs &A, int64_t t1,
                                                AdamCoef *out) {
  *out = make_adam_coef(D.beta1, D.beta2, D.eps, A.lr_q, A.lr_v, A.lr_a_base, D.t_max, t1);
}

// The fp32 optimiser state (masters, moments, target) a tile writes is next read one step later by
// the same tile's successor: it is stored write-through (sc0 sc1), so the bytes leave while the
// kernel still runs and nothing of them is dirty for the release at the kernel boundary (the
// boundary behind k_update costs + dirty bytes / 6 TB/s, MI355X_MICROARCH.md "boundary").  A/B on
// one box (round 3, r4b): one seed 63.4k -> 64.9k steps/s (k_update 5.8 -> 5.4 us in the launch
// tiling), 8 seeds per launch 171.2k -> 172.4k.  (`nt` stores, tried in round 2, are not
// write-through: no effect.)  -DIQL_WT_STATE=0 builds the plain-store variant.
#ifndef IQL_WT_STATE
#define IQL_WT_STATE 1
#endif
__device__ __forceinline__ void state_store(const float *base, int64_t elem, float4 v) {  // uniform base, lane element
#if IQL_WT_STATE
  stg16_wt(base, (uint32_t)elem * 4u, v);
#else
  stg16(base, (uint32_t)elem * 4u, v);
#endif
}

// s += the 8 (bf16) / 4 (fp32) values of one 16-byte operand fragment, pairwise, in a fixed order
template <bool BF16>
__device__ __forceinline__ void frag_acc(float &s, const uint4 &f) {
  if constexpr (BF16) {
    const uint32_t w[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) s += bf2f((uint16_t)(w[i] & 0xffff)) + bf2f((uint16_t)(w[i] >> 16));
  } else {
    const float4 v = __builtin_bit_cast(float4, f);
    s += (v.x + v.y) + (v.z + v.w);
  }
}

// IQL_WT_ALL (A/B build): also the compute-precision copies and the layer-1 strips' state leave
// write-through, so that k_update ends with (almost) nothing dirty in the L2s.
#ifndef IQL_WT_ALL
#define IQL_WT_ALL 0
#endif
template <bool BF16>
__device__ __forceinline__ void copy_store4(typename Prec<BF16>::T *base, size_t elem, const float v[4]) {
#if IQL_WT_ALL
  using P = Prec<BF16>;
  if constexpr (BF16) {
    uint2 u;
    u.x = (uint32_t)P::from_f32(v[0]) | ((uint32_t)P::from_f32(v[1]) << 16);
    u.y = (uint32_t)P::from_f32(v[2]) | ((uint32_t)P::from_f32(v[3]) << 16);
    stg8_wt(base, (uint32_t)elem * 2u, u);
  } else {
    stg16_wt(base, (uint32_t)elem * 4u, make_float4(v[0], v[1], v[2], v[3]));
  }
#else
  store4T<BF16>(base + elem, v);
#endif
}
// eight consecutive k of one row of a fragment-major copy: ONE 16-byte store in bf16 (a lane's whole
// fragment slot), two in fp32 (elem4: the index of the second four)
template <bool BF16>
__device__ __forceinline__ void copy_store8(typename Prec<BF16>::T *base, size_t elem, size_t elem4, const float v[8]) {
  using P = Prec<BF16>;
  if constexpr (BF16) {
    uint4 u;
    u.x = (uint32_t)P::from_f32(v[0]) | ((uint32_t)P::from_f32(v[1]) << 16);
    u.y = (uint32_t)P::from_f32(v[2]) | ((uint32_t)P::from_f32(v[3]) << 16);
    u.z = (uint32_t)P::from_f32(v[4]) | ((uint32_t)P::from_f32(v[5]) << 16);
    u.w = (uint32_t)P::from_f32(v[6]) | ((uint32_t)P::from_f32(v[7]) << 16);
#if IQL_WT_ALL
    stg16_wt(base, (uint32_t)elem * 2u, __builtin_bit_cast(float4, u));
#else
    stg16(base + elem, __builtin_bit_cast(float4, u));
#endif
  } else {
    copy_store4<BF16>(base, elem, v);
    copy_store4<BF16>(base, elem4, v + 4);
  }
}
__device__ __forceinline__ void state_store1(float *base, int64_t elem, float v) {
#if IQL_WT_ALL
  stg4_wt(base, (uint32_t)elem * 4u, v);
#else
  stg(base + elem, v);
#endif
}

constexpr int UKC = 8;  // batch k-steps per register chunk in the weight-gradient GEMM

// One work-group owns the gradient tile dW[o0..o0+64)[i0..i0+32) (192 work-groups
// at H = 256: the optimiser state is THE streaming traffic of the step and a CU
// pulls only ~30 GB/s from the Infinity Cache, so it is spread over most CUs):
//   1. every thread requests its share of the optimiser state (params, exp_avg,
//      exp_avg_sq, target) in ROW order -- a wave instruction covers 8 rows x one
//      full 128-byte line of the torch [out][in] arrays;
//   2. wave w computes dW^T = X^T dZ for one 16-wide in-feature tile x two
//      out-feature tiles on MFMA (A = layer input, B = dZ^T, fragment-major);
//   3. the tile goes through LDS so that step 1's row-ordered threads pick up
//      their gradients, apply Adam + Polyak and stream the state back;
//   4. the new weights go through LDS once more for the transposed compute copy.
// Rows of the first layer are not 16-byte aligned (in_dim 29, 37, ...): that layer
// uses a scalar path over the same tile.
// Threads of an update work-group (template parameter of k_update): 512 = 8 waves, two per SIMD,
// for one seed per launch (the Adam pass is bound by VALU issue: one wave per SIMD used half of
// it); 256 for several seeds per launch (four 126-VGPR work-groups per CU measured faster there).
constexpr int UTO = 64;          // tile: out-features
constexpr int UTI = 32;          // tile: in-features
constexpr int ULD = UTI + 4;     // LDS row stride (floats)
constexpr int UTPR = UTI / 4;    // threads per tile row (float4 each)
constexpr int UMAXI = 128;       // S + A <= 128 (host check)
constexpr int UNIT = UMAXI / 16; // in-feature tiles of a layer-1 strip, at most
#ifndef IQL_USR
#define IQL_USR 16
#endif
constexpr int USR = IQL_USR;     // rows of a layer-1 strip (16 or 32)
constexpr int UOT = USR / 16;    // out-feature tiles of a strip
constexpr int UPD_TILE = UTO * (UTI + 4) > USR * (UMAXI + 4) ? UTO * (UTI + 4) : USR * (UMAXI + 4);
// (the second region: a strip's new target weights, or -- group launches -- a layer-2 tile's)
constexpr int UPD_LDS = UPD_TILE + (UTO * (UTI + 4) > USR * (UMAXI + 4) ? UTO * (UTI + 4) : USR * (UMAXI + 4));  // floats (~18 KB)

template <bool BF16, bool LAT, int UT>
__device__ __forceinline__ void update_body(const TrainerDesc *__restrict__ Dp,
                                            const DevArgs *__restrict__ Ap, DevCtr *__restrict__ Cp,
                                            const UpdItem *__restrict__ items, int n_items,
                                            const int blk) {
  using P = Prec<BF16>;
  using T = typename P::T;
  constexpr bool AF = BF16 && IQL_ADAM_FAST != 0;  // (see adam_apply)
  constexpr int UWAVES = UT / 64;
  constexpr int URPP = UT / UTPR;                  // tile rows per pass
  constexpr int UNP = UTO / URPP;                  // passes
  constexpr int UNB = (UTO / 16) / (UWAVES / 2);   // out-feature tiles per wave in the tile GEMM
  constexpr int UWPO = UWAVES / UOT;               // waves sharing a strip's out-feature tile
  static_assert(UNP >= 1 && UNB >= 1 && UTO % URPP == 0, "update tile geometry");
  const TrainerDesc &D = *Dp;
  const DevArgs &A = *Ap;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: wave-uniform branches stay scalar
  const int r = lane & 15, q = lane >> 4;
  const int H = D.H, B = D.B, BP = D.BP;
  const int nslab = B / SLAB;
  // ---- ONE batch of scalar loads, before any branch: the work item (clamped for the misc
  // block), the Adam coefficients of the step, the arena pointers ----
  const UpdItem it = items[blk < n_items ? blk : n_items - 1];
  const int64_t t1 = Cp->coef_step;  // 1-based Adam step of this update (set by k_forward)
  const AdamCoef coef = Cp->coef;
  float *const g_params = D.params, *const g_m = D.exp_avg, *const g_v = D.exp_avg_sq;
  float *const g_target = D.target, *const g_grads = D.grads;

  // tile: [o][i] gradient, then new weights (64 x 32 tile, or a layer-1 strip of USR rows x all
  // in-features); tile2: a strip's new target weights.  One array: the misc block sums the loss
  // partials in all of it (UPD_LDS floats; host check in api.hip).
  __shared__ __attribute__((aligned(16))) float lds_upd[UPD_LDS];
  float *const tile = lds_upd, *const tile2 = lds_upd + UPD_TILE;
  __shared__ float bgrad[UTO], bgrad2[UTO];  // (bgrad2: the second half of a split batch)
  STAMP(2, 0);

  if (blk >= n_items) {
    // ---------------- misc block: log_std, logged losses, step counter ------------
    // all partials are fetched in parallel, then summed in a fixed order from LDS
    float *sred = tile;
    const int nt = D.ntrain;
    const int nl = nt * nslab, na = D.deterministic ? 0 : nslab * D.A;  // nl + na + nt fits the tile (host check)
    for (int e = tid; e < nl + na; e += UT) sred[e] = e < nl ? ldg(D.lossp + e) : ldg(D.lsp + e - nl);
    float ls = 0.f, pm = 0.f, pv = 0.f;
    if (!D.deterministic && tid < D.A) {
      const int64_t o = D.off_log_std + tid;
      ls = ldg(D.params + o), pm = ldg(D.exp_avg + o), pv = ldg(D.exp_avg_sq + o);
    }
    __syncthreads();
    if (!D.deterministic && tid < D.A) {
      float g = 0.f;
      for (int s = 0; s < nslab; ++s) g += sred[nl + s * D.A + tid];
      const int64_t o = D.off_log_std + tid;
      const float lsc = fminf(fmaxf(ls, -20.f), 2.f);
      g = g * expf(lsc) * ((ls >= -20.f && ls <= 2.f) ? 1.f : 0.f);
      float p = ls;
      adam_apply<AF>(p, pm, pv, g, coef, coef.neg_step[2]);
      stg(D.params + o, p), stg(D.exp_avg + o, pm), stg(D.exp_avg_sq + o, pv);
      if (D.grads) stg(D.grads + o, g);
    }
    if (tid >= 64 && tid < 64 + nt) {  // one lane per network: fixed-order sum of its slab partials
      const int n = tid - 64;
      float s = 0.f;
      for (int k = 0; k < nslab; ++k) s += sred[n * nslab + k];
      sred[nl + na + n] = s / (float)B;
    }
    __syncthreads();
    if (tid == 64) {
      const float vl = sred[nl + na + D.net_v];
      float ql = 0.f;  // sum(mse(q, targets) for q in qs) / len(qs)  (ref:606)
      for (int e = 0; e < D.E; ++e) ql += sred[nl + na + e];
      ql = ql / (float)D.E;
      const float al = sred[nl + na + D.net_a];
      Cp->last_losses[0] = vl, Cp->last_losses[1] = ql, Cp->last_losses[2] = al;
      Cp->loss_sum[0] += vl, Cp->loss_sum[1] += ql, Cp->loss_sum[2] += al;
      if (A.losses_out) {
        float *lo = A.losses_out + (size_t)(t1 - 1 - A.base_step) * 3;
        stg(lo, vl), stg(lo + 1, ql), stg(lo + 2, al);
      }
      Cp->ctr[0] = t1;
    }
    STAMP(2, 4);
    return;
  }

  if (it.net < 0) {
    // padding slot of the XCD-major item table: gather the NEXT step's replay rows while the
    // tiles work (random HBM rows + TLB misses leave the next k_forward's critical path); possible
    // iff that step's indices are known
    if (D.prefetch && (A.idx_mode == 0 || (A.idx_mode == 1 && t1 - A.base_step < A.n_steps)))
      stage_rows16<BF16>(D, A, t1, it.o0 * (UT / 16), it.i0 * (UT / 16), tid);
    return;
  }
  const int L = it.layer;
  const int Odim = it.Odim, Idim = it.Idim, Kw = it.Kw, Opad = it.Opad;
  const int Ipad = round_up(Idim, 16);
  const int o0 = it.o0, i0 = it.i0;
  const float neg_step =
      it.group == 0 ? coef.neg_step[0] : (it.group == 1 ? coef.neg_step[1] : coef.neg_step[2]);
  const bool has_target = it.has_target != 0;
  const int nk = BP / P::KM;
  // Long batches (16 k-steps and more: batch >= 512 in bf16): the tile GEMMs sum the batch in TWO
  // halves, k-steps [0, kh) and [kh, nk), each in ascending order, and add the halves (fp32) -- in the
  // group variant two wave pairs take a half each (the whole work-group streams operands: twice the
  // bytes in flight on a loop that is bound by load latency x depth: stamps of round 4, 32 k-steps in
  // 5.4 us on two waves), in the one-seed variant every wave runs the halves one after the other into
  // two accumulators.  The same sum either way: a seed in a group launch stays the seed alone.
  // bf16 only: precision = fp32 (the parity mode, 16 k-steps already at batch 256) keeps ONE ascending
  // sum per element -- measured there: the first 100 steps of the reference's 1,000-step trajectories
  // agree to 6e-7 with that order and to 5e-4 with the split one (tests/test_gpu_long_horizon.py).
  const int kh = (BF16 && nk >= 16) ? (nk / (2 * UKC)) * UKC : nk;  // (scalar; = nk: no split)
  const bool ksplit = kh < nk;
  // descriptor / item fields of the Adam and store phases: fetched with the first batch
  pin_s(D.tau), pin_s(D.one_m_tau), pin_s(D.polyak_convex);
  pin_s(it.off_w), pin_s(it.toff_w), pin_s(it.off_b), pin_s(it.toff_b), pin_s(it.wc), pin_s(it.tc);
  pin_s(it.w2ct), pin_s(it.Xsrc), pin_s(it.Zsrc), pin_s(Kw), pin_s(neg_step);
  pin_s(coef.one_m_b1), pin_s(coef.b2), pin_s(coef.one_m_b2), pin_s(coef.bc2_sqrt), pin_s(coef.eps);
  pin_s(coef.inv_bc2_sqrt);

  if (L == 0) {
    // =========== layer 1: a strip of 32 out-features x ALL in-features ===========
    // Rows of W1 (S + A floats) are not 16-byte aligned, so the strip's optimiser state is
    // streamed as ONE flat range o0*Idim .. (o0+32)*Idim: consecutive lanes, consecutive
    // addresses, every line fully used.  (32 rather than 64 rows: the strips are the longest
    // work-groups of the launch, twice as many halves the tail.)
    const int TLD = Ipad + 4;
    const int64_t fbase = it.off_w + (int64_t)o0 * Idim, tbase = it.toff_w + (int64_t)o0 * Idim;
    // one ELEMENT of the flat range per thread and pass (UNE passes cover 16 x 128 elements):
    // the strip's Adam work is spread over all 512 threads instead of a quarter of them
    // Group variant: FOUR consecutive elements per thread and pass -- a strip's flat range starts on a
    // 16-byte boundary and is a multiple of 16 bytes long (16 rows x Idim floats; tensors start on 128-byte
    // lines) -- so its state moves as 16-byte loads and write-through stores like a tile's, a quarter
    // of the memory instructions (stamps of round 4, 8 seeds per launch: a strip work-group took 6.5 us,
    // a 64 x 32 tile with 3.5 x its bytes 5.3).
    constexpr int UNE = LAT ? (USR * UMAXI + UT - 1) / UT : 1;
    constexpr int UN4 = LAT ? 1 : (USR * UMAXI / 4 + UT - 1) / UT;
    const int n_el = USR * Idim, n4 = n_el >> 2;
    float pf[UNE], mf[UNE], vf[UNE], tf[UNE];
    float4 pf4[UN4], mf4[UN4], vf4[UN4], tf4[UN4];
    float pb, mb, vb, tb;  // branch-free (see the tiles below)
    auto load_state = [&]() {
      if constexpr (LAT) {
#pragma unroll
        for (int k = 0; k < UNE; ++k) {
          const int e = tid + UT * k;
          const int ec = e < n_el ? e : n_el - 1;  // branch-free: lanes past the end re-read the last element
          pf[k] = ldg(g_params + fbase + ec);
          mf[k] = ldg(g_m + fbase + ec);
          vf[k] = ldg(g_v + fbase + ec);
          tf[k] = ldg((has_target ? g_target + tbase : g_params + fbase) + ec);
        }
      } else {
#pragma unroll
        for (int k = 0; k < UN4; ++k) {
          if (UT * k < n4) {  // (scalar: a pass no lane needs issues nothing)
            const int e4 = tid + UT * k;
            const int ec = 4 * (e4 < n4 ? e4 : n4 - 1);
            pf4[k] = __builtin_bit_cast(float4, ldg16(g_params + fbase + ec));
            mf4[k] = __builtin_bit_cast(float4, ldg16(g_m + fbase + ec));
            vf4[k] = __builtin_bit_cast(float4, ldg16(g_v + fbase + ec));
            tf4[k] = __builtin_bit_cast(float4, ldg16((has_target ? g_target + tbase : g_params + fbase) + ec));
          }
        }
      }
      const int ob_ = o0 + (tid & (USR - 1));  // < Odim = H always
      const int64_t eb = it.off_b + ob_;
      pb = ldg(g_params + eb), mb = ldg(g_m + eb), vb = ldg(g_v + eb);
      tb = ldg((has_target ? g_target + it.toff_b : g_params + it.off_b) + ob_);
    };
    if constexpr (LAT) load_state();  // (!LAT: behind the GEMM, see the tiles below)
    STAMP(2, 1);
    // dW1^T strip: wave w = out-feature tile (w & 1) against the in-feature tiles of parity
    // (w >> 1): every element is produced by one wave, k-steps in order (no cross-wave sums)
    // layer-1 input of THIS step (t1 - 1): plane (t1 - 1) & 1 of xT
    const T *Xsrc = reinterpret_cast<const T *>(it.Xsrc) + (size_t)((t1 - 1) & 1) * D.xrows * BP;
    const T *Zsrc = reinterpret_cast<const T *>(it.Zsrc);
    const int nit = Ipad >> 4;
    const int wo = wave % UOT, th = wave / UOT;
    constexpr int NT = UNIT / UWPO;  // in-feature tiles per wave, at most
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    // The operand fragments of this wave's first two in-feature tiles (16-row strips: all of
    // them) are requested up front, unconditionally (clamped indices): guards only around the MFMAs.
    constexpr int TB = NT < 2 ? NT : 2;
    auto chunk = [&](const int k0) {
      uint4 zf[UKC], xf[TB][UKC];
#pragma unroll
      for (int ks = 0; ks < UKC; ++ks) {
        const int kk = k0 + ks < nk ? k0 + ks : nk - 1;
        zf[ks] = ldg16(Zsrc + frag_off<P>((o0 >> 4) + wo, kk, nk, lane));
      }
#pragma unroll
      for (int tb = 0; tb < TB; ++tb) {
        // (the group variant skips in-feature tiles that do not exist -- S + A <= 64: every second one --
        // with a scalar branch: 8 KB of fragments per wave through an already full vector-memory queue)
        if (LAT || tb == 0 || th + UWPO * tb < nit) {
          const int tt = th + UWPO * tb < nit ? th + UWPO * tb : nit - 1;
#pragma unroll
          for (int ks = 0; ks < UKC; ++ks) {
            const int kk = k0 + ks < nk ? k0 + ks : nk - 1;
            xf[tb][ks] = ldg16(Xsrc + frag_off<P>(tt, kk, nk, lane));
          }
        }
      }
      if (th == 0) {
#pragma unroll
        for (int ks = 0; ks < UKC; ++ks) {
          if (k0 + ks < nk) {  // bias gradient = row sums of dZ^T
            if constexpr (BF16) {
              const uint32_t w[4] = {zf[ks].x, zf[ks].y, zf[ks].z, zf[ks].w};
#pragma unroll
              for (int i = 0; i < 4; ++i)
                bsum += bf2f((uint16_t)(w[i] & 0xffff)) + bf2f((uint16_t)(w[i] >> 16));
            } else {
              const float4 f = __builtin_bit_cast(float4, zf[ks]);
              bsum += (f.x + f.y) + (f.z + f.w);
            }
          }
        }
      }
#pragma unroll
      for (int tb = 0; tb < TB; ++tb) {
        if (th + UWPO * tb < nit) {
#pragma unroll
          for (int ks = 0; ks < UKC; ++ks)
            if (k0 + ks < nk) P::mma(xf[tb][ks], zf[ks], acc[tb]);
        }
      }
      // wider inputs (S + A > 64): the remaining tiles one at a time
#pragma unroll
      for (int tb = TB; tb < NT; ++tb) {
        if (th + UWPO * tb < nit) {
          uint4 xg[UKC];
#pragma unroll
          for (int ks = 0; ks < UKC; ++ks) {
            const int kk = k0 + ks < nk ? k0 + ks : nk - 1;
            xg[ks] = ldg16(Xsrc + frag_off<P>(th + UWPO * tb, kk, nk, lane));
          }
#pragma unroll
          for (int ks = 0; ks < UKC; ++ks)
            if (k0 + ks < nk) P::mma(xg[ks], zf[ks], acc[tb]);
        }
      }
    };
    // (waves whose in-feature tile does not exist -- S + A <= 48 at 512 threads: five of the eight -- load
    // nothing: every fragment costs the CU's L1 port 16 cycles)
    if (th < nit) {
      chunk(0);  // straight-line first chunk: no loop pre-header to drain the state loads in
#pragma unroll 1
      for (int k0 = UKC; k0 < nk; k0 += UKC) chunk(k0);
    }
    if constexpr (!LAT) load_state();
    // C/D layout: lane (r, q) of acc[tb] holds dW[o0 + 16 wo + r][16 (th + 2 tb) + 4 q + k]
#pragma unroll
    for (int tb = 0; tb < NT; ++tb)
      if (th + UWPO * tb < nit)
        *reinterpret_cast<f32x4 *>(&tile[(16 * wo + r) * TLD + 16 * (th + UWPO * tb) + 4 * q]) = acc[tb];
    if (th == 0) {
      bsum = xor32_sum(xor16_sum(bsum));
      if (q == 0) bgrad[16 * wo + r] = bsum;
    }
    __syncthreads();
    STAMP(2, 3);
    T *wc = reinterpret_cast<T *>(it.wc);
    T *tc = reinterpret_cast<T *>(it.tc);
    const int nkw = Kw / P::KM;
    // Adam + Polyak on the flat range; the new weights (and targets) go back to LDS in
    // [row][in-feature] order for the compute copies
    if constexpr (LAT) {
#pragma unroll
      for (int k = 0; k < UNE; ++k) {
        const int e = tid + UT * k;
        if (e < n_el) {
          const int ol = e / Idim, i = e - ol * Idim;
          const float g = P::round(tile[ol * TLD + i]);
          float p_ = pf[k], m_ = mf[k], v_ = vf[k];
          adam_apply<AF>(p_, m_, v_, g, coef, neg_step);
          tile[ol * TLD + i] = p_;
          state_store1(g_params, fbase + e, p_), state_store1(g_m, fbase + e, m_), state_store1(g_v, fbase + e, v_);
          if (g_grads) stg(g_grads + fbase + e, g);
          if (has_target) {
            const float t_ = polyak(D, tf[k], p_);
            tile2[ol * TLD + i] = t_;
            state_store1(g_target, tbase + e, t_);
          }
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < UN4; ++k) {
        const int e4 = tid + UT * k;
        if (UT * k < n4 && e4 < n4) {
          float p_[4] = {pf4[k].x, pf4[k].y, pf4[k].z, pf4[k].w}, m_[4] = {mf4[k].x, mf4[k].y, mf4[k].z, mf4[k].w};
          float v_[4] = {vf4[k].x, vf4[k].y, vf4[k].z, vf4[k].w}, t_[4] = {tf4[k].x, tf4[k].y, tf4[k].z, tf4[k].w};
          float g_[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int e = 4 * e4 + j;
            const int ol = e / Idim, i = e - ol * Idim;
            g_[j] = P::round(tile[ol * TLD + i]);
            adam_apply<AF>(p_[j], m_[j], v_[j], g_[j], coef, neg_step);
            tile[ol * TLD + i] = p_[j];
            if (has_target) {
              t_[j] = polyak(D, t_[j], p_[j]);
              tile2[ol * TLD + i] = t_[j];
            }
          }
          state_store(g_params, fbase + 4 * e4, make_float4(p_[0], p_[1], p_[2], p_[3]));
          state_store(g_m, fbase + 4 * e4, make_float4(m_[0], m_[1], m_[2], m_[3]));
          state_store(g_v, fbase + 4 * e4, make_float4(v_[0], v_[1], v_[2], v_[3]));
          if (g_grads) stg16(g_grads + fbase + 4 * e4, make_float4(g_[0], g_[1], g_[2], g_[3]));
          if (has_target) state_store(g_target, tbase + 4 * e4, make_float4(t_[0], t_[1], t_[2], t_[3]));
        }
      }
    }
    if (tid < USR) {
      const int64_t e = it.off_b + o0 + tid;
      const float g = P::round(bgrad[tid]);
      adam_apply<AF>(pb, mb, vb, g, coef, neg_step);
      stg(g_params + e, pb), stg(g_m + e, mb), stg(g_v + e, vb);
      if (g_grads) stg(g_grads + e, g);
      if (has_target) stg(g_target + it.toff_b + o0 + tid, polyak(D, tb, pb));
    }
    __syncthreads();
    // compute copies: 4 consecutive in-features of one row are contiguous in the fragment-major
    // image (columns Idim .. Ipad of the tile hold zero gradients = the copies' zero padding)
    const int cpr = Ipad >> 2;
    for (int e = tid; e < USR * cpr; e += UT) {
      const int ol = e / cpr, i = (e - ol * cpr) * 4;
      const float4 p4 = *reinterpret_cast<const float4 *>(&tile[ol * TLD + i]);
      float pv4[4] = {p4.x, p4.y, p4.z, p4.w};
      copy_store4<BF16>(wc, fidx<P>(o0 + ol, i, nkw), pv4);
      if (has_target) {
        const float4 t4 = *reinterpret_cast<const float4 *>(&tile2[ol * TLD + i]);
        float tv4[4] = {i < Idim ? t4.x : 0.f, i + 1 < Idim ? t4.y : 0.f, i + 2 < Idim ? t4.z : 0.f,
                        i + 3 < Idim ? t4.w : 0.f};
        copy_store4<BF16>(tc, fidx<P>(o0 + ol, i, nkw), tv4);
      }
    }
    STAMP(2, 4);
    return;
  }

  // ---- 2. dW^T tile on MFMA: A = layer input X^T, B = dZ^T (fragment-major) ----
  const T *Xsrc = reinterpret_cast<const T *>(it.Xsrc);
  const T *Zsrc = reinterpret_cast<const T *>(it.Zsrc);
  const bool do_bias = i0 == 0;
  const int ntile_o = Opad >> 4;
  // (addresses as uniform base + ONE 32-bit lane offset: global_load ... v_off, s[base:base+1];
  // a 64-bit per-lane address per fragment costs two registers each while the loads are issued)
  const uint32_t lane16 = (uint32_t)lane * 16u;

  // ---- 1. optimiser state, row order: thread -> (row tr + 16 pass, columns 4 tc .. +3) ----
  const int tr = tid / UTPR, tc4 = (tid % UTPR) * 4;
  float pw[UNP][4], mw[UNP][4], vw[UNP][4], tw[UNP][4];  // statically indexed only (registers)
  float pb, mb, vb, tb;
  auto load_state = [&]() {
#pragma unroll
    for (int ps = 0; ps < UNP; ++ps) {
      const int o = o0 + tr + URPP * ps, i = i0 + tc4;
      // Branch-free: every lane loads (rows beyond Odim -- layer 3 has 1 or A rows -- re-read the
      // last row; their results are never stored).  A guarded load becomes a branch, and the
      // branch a `s_waitcnt vmcnt(0)`: the loads of one tile would queue behind each other.
      // Idim = H here: i < Idim always, and rows of the fp32 masters are 16-byte aligned.
      const int oc = o < Odim ? o : Odim - 1;
      const int64_t e = it.off_w + (int64_t)oc * Idim + i;
      const int64_t te = (has_target ? it.toff_w : it.off_w) + (int64_t)oc * Idim + i;
      const float4 a4 = __builtin_bit_cast(float4, ldg16(g_params + e));
      const float4 b4 = __builtin_bit_cast(float4, ldg16(g_m + e));
      const float4 c4 = __builtin_bit_cast(float4, ldg16(g_v + e));
      // no target network: a dummy read of the parameters keeps the load unconditional
      const float4 d4 = __builtin_bit_cast(float4, ldg16((has_target ? g_target : g_params) + te));
      pw[ps][0] = a4.x, pw[ps][1] = a4.y, pw[ps][2] = a4.z, pw[ps][3] = a4.w;
      mw[ps][0] = b4.x, mw[ps][1] = b4.y, mw[ps][2] = b4.z, mw[ps][3] = b4.w;
      vw[ps][0] = c4.x, vw[ps][1] = c4.y, vw[ps][2] = c4.z, vw[ps][3] = c4.w;
      tw[ps][0] = d4.x, tw[ps][1] = d4.y, tw[ps][2] = d4.z, tw[ps][3] = d4.w;
    }
    // biases: every thread loads (clamped index, dummy target), threads < 64 of the i0 = 0 tile use them
    const int ob_ = o0 + (tid & (UTO - 1)) < Odim ? o0 + (tid & (UTO - 1)) : Odim - 1;
    const int64_t eb = it.off_b + ob_;
    pb = ldg(g_params + eb), mb = ldg(g_m + eb), vb = ldg(g_v + eb);
    tb = ldg((has_target ? g_target + it.toff_b : g_params + it.off_b) + ob_);
  };

  if constexpr (LAT) {
    // ======== one seed per launch: a latency chain on an idle chip, 8 waves ========
    // wave (wo, wi) = (wave >> 1, wave & 1): in-feature tile wi x out-feature tiles UNB wo .. + UNB
    const int wo = wave >> 1, wi = wave & 1;
    const int ib = i0 + 16 * wi, ob = o0 + 16 * UNB * wo;
    f32x4 acc[UNB];
    float bsum[UNB];
#pragma unroll
    for (int b = 0; b < UNB; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f}, bsum[b] = 0.f;
    const bool wave_bias = do_bias && wi == 0;
    // Branch-free operand stream: in-features always exist here (Idim = H), out-feature tiles
    // beyond Opad (layer 3: one tile) re-read the last tile -- their products land in tile rows
    // that are never stored -- and k-steps beyond nk re-read the last one and are skipped by the
    // (scalar) guard around the MFMA only.
    int ot[UNB];
#pragma unroll
    for (int b = 0; b < UNB; ++b) ot[b] = (ob >> 4) + b < ntile_o ? (ob >> 4) + b : ntile_o - 1;
    // Operand fragments are requested FIRST, the optimiser state behind them: the MFMAs (and the
    // LDS hand-over to the row-ordered Adam pass) then start as soon as the fragments are in, while
    // the state is still streaming (+0.8 % measured).  First chunk straight-line: a loop
    // pre-header would drain every pending load.
    auto load_frags = [&](const int k0, uint4(&xf)[UKC], uint4(&zf)[UKC][UNB]) {
      const char *const xb = reinterpret_cast<const char *>(Xsrc) + (size_t)((ib >> 4) * nk) * 1024;
      const char *zb[UNB];
#pragma unroll
      for (int b = 0; b < UNB; ++b) zb[b] = reinterpret_cast<const char *>(Zsrc) + (size_t)(ot[b] * nk) * 1024;
#pragma unroll
      for (int ks = 0; ks < UKC; ++ks) {
        const int kk = k0 + ks < nk ? k0 + ks : nk - 1;  // (one fragment = 64 lanes x 16 B = 1 KiB, both precisions)
        xf[ks] = ldg16(xb + (size_t)kk * 1024 + lane16);
#pragma unroll
        for (int b = 0; b < UNB; ++b) zf[ks][b] = ldg16(zb[b] + (size_t)kk * 1024 + lane16);
      }
    };
    auto mma_frags = [&](const int k0, const uint4(&xf)[UKC], const uint4(&zf)[UKC][UNB], f32x4(&ac)[UNB],
                         float(&bs)[UNB]) {
#pragma unroll
      for (int ks = 0; ks < UKC; ++ks) {
        if (k0 + ks < nk) {
#pragma unroll
          for (int b = 0; b < UNB; ++b) {
            if (wave_bias) frag_acc<BF16>(bs[b], zf[ks][b]);  // bias gradient = row sums of dZ^T
            P::mma(xf[ks], zf[ks][b], ac[b]);
          }
        }
      }
    };
    // (Tried: the operand panels of a tile through LDS-DMA -- 48 distinct fragments instead of the
    // 128 the eight waves load between them.  No gain: 3.24 vs 3.14 us per tile work-group, the
    // wait for ALL fragments plus a barrier costs what the L1 port saves.)
    uint4 xf0[UKC], zf0[UKC][UNB];
    load_frags(0, xf0, zf0);
    __builtin_amdgcn_sched_barrier(0);
    load_state();  // right behind the operand fragments: everything in flight at once
    STAMP(2, 1);
    __builtin_amdgcn_sched_barrier(0);
    mma_frags(0, xf0, zf0, acc, bsum);
#pragma unroll 1
    for (int k0 = UKC; k0 < kh; k0 += UKC) {
      uint4 xf[UKC], zf[UKC][UNB];
      load_frags(k0, xf, zf);
      mma_frags(k0, xf, zf, acc, bsum);
    }
    if (ksplit) {  // the second half of a long batch into its own accumulators, then lo + hi
      f32x4 acc_hi[UNB];
      float bsum_hi[UNB];
#pragma unroll
      for (int b = 0; b < UNB; ++b) acc_hi[b] = f32x4{0.f, 0.f, 0.f, 0.f}, bsum_hi[b] = 0.f;
#pragma unroll 1
      for (int k0 = kh; k0 < nk; k0 += UKC) {
        uint4 xf[UKC], zf[UKC][UNB];
        load_frags(k0, xf, zf);
        mma_frags(k0, xf, zf, acc_hi, bsum_hi);
      }
#pragma unroll
      for (int b = 0; b < UNB; ++b) {
        acc[b] += acc_hi[b];
        if (wave_bias) bsum[b] = xor32_sum(xor16_sum(bsum[b])), bsum_hi[b] = xor32_sum(xor16_sum(bsum_hi[b]));
        bsum[b] += bsum_hi[b];
      }
    }
    // C/D layout: lane (r, q) of acc[b] holds dW[ob + 16 b + r][ib + 4 q + k]
#pragma unroll
    for (int b = 0; b < UNB; ++b)
      *reinterpret_cast<f32x4 *>(&tile[(16 * UNB * wo + 16 * b + r) * ULD + 16 * wi + 4 * q]) = acc[b];
    if (wave_bias) {
#pragma unroll
      for (int b = 0; b < UNB; ++b) {
        float bs = bsum[b];
        if (!ksplit) bs = xor32_sum(xor16_sum(bs));  // (split: the halves were reduced before they were added)
        if (q == 0) bgrad[16 * UNB * wo + 16 * b + r] = bs;
      }
    }
  } else {
    // ======== several seeds per launch: throughput, 4 waves, four work-groups per CU ========
    // In-kernel stamps of round 3 (tools/stamps.py STAMP_ALL, 8 seeds per launch): with the chip
    // full a tile work-group spends 2-8 us just getting its loads ISSUED -- the CU's vector-memory
    // pipeline, not latency, bounds the launch (requesting the state earlier, through LDS-DMA, made
    // the launch 20 % slower).  Of the 172 KB a 64 x 32 tile moves through that pipeline, 96 KB were
    // operand fragments for 48 KB of distinct ones: every wave loaded its own copy of the X / Z
    // fragments it shared with a neighbour.  Here TWO waves do the GEMM -- wave g: out-feature tiles
    // 2g, 2g+1 x BOTH in-feature tiles, 32 KB each, 64 KB per tile -- with the k-steps pipelined
    // GIF deep so that the registers stay below the 4-waves-per-SIMD budget; the other two waves,
    // which have no operands to hold, request their half of the optimiser state right away.
    // Every accumulator still sums its k-steps in ascending order: results are bit-identical.
    #ifndef IQL_GW
#define IQL_GW 2  // GEMM waves per tile (A/B: 1 = one wave loads every distinct fragment exactly once)
#endif
#ifndef IQL_GIF
#define IQL_GIF (IQL_GW == 2 ? 4 : 3)  // k-steps in flight per GEMM wave (one more spills at the 128-register budget)
#endif
    constexpr int GW = IQL_GW, GNB = (UTO / 16) / GW, GNI = UTI / 16, GIF = IQL_GIF;
    static_assert(UT == 256 && UTO == 64 && (GW == 1 || GW == 2), "the GEMM waves cover a 64-row tile");
    // (split batch: all four waves are GEMM waves -- wave pair `half` takes k-steps [half kh, ...) )
    const bool gemm_wave = ksplit || wave < GW;  // (scalar)
    const int gw = wave & (GW - 1), half = ksplit ? wave / GW : 0;
    const int kbeg = half ? kh : 0, kend = ksplit ? (half ? nk : kh) : nk;
    const bool wave_bias = do_bias && gemm_wave;
    f32x4 acc[GNB][GNI];
    float bsum[GNB];
#pragma unroll
    for (int b = 0; b < GNB; ++b) {
      bsum[b] = 0.f;
#pragma unroll
      for (int c = 0; c < GNI; ++c) acc[b][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#ifndef IQL_G2_EARLY
#define IQL_G2_EARLY 1
#endif
    if (!gemm_wave) {
#if IQL_G2_EARLY
      load_state();
#endif
      STAMP(2, 1);
    } else {
      static_assert(GW == 2 || IQL_GW == 1, "split batches pair the waves (GW = 2)");
      const char *xb[GNI], *zb[GNB];
#pragma unroll
      for (int c = 0; c < GNI; ++c) xb[c] = reinterpret_cast<const char *>(Xsrc) + (size_t)(((i0 >> 4) + c) * nk) * 1024;
#pragma unroll
      for (int b = 0; b < GNB; ++b) {
        const int t_ = (o0 >> 4) + GNB * gw + b;
        zb[b] = reinterpret_cast<const char *>(Zsrc) + (size_t)((t_ < ntile_o ? t_ : ntile_o - 1) * nk) * 1024;
      }
      uint4 xf[UKC][GNI], zf[UKC][GNB];  // (a k-step's registers are live from its issue to its MFMAs only)
      auto issue = [&](const int k0, const int ks) {
        const int kk = k0 + ks < kend ? k0 + ks : kend - 1;
#pragma unroll
        for (int c = 0; c < GNI; ++c) xf[ks][c] = ldg16(xb[c] + (size_t)kk * 1024 + lane16);
#pragma unroll
        for (int b = 0; b < GNB; ++b) zf[ks][b] = ldg16(zb[b] + (size_t)kk * 1024 + lane16);
      };
      auto chunk = [&](const int k0, const bool first) {
#pragma unroll
        for (int ks = 0; ks < GIF; ++ks) issue(k0, ks);
        if (first) STAMP(2, 1);
#pragma unroll
        for (int ks = 0; ks < UKC; ++ks) {
          __builtin_amdgcn_sched_barrier(0);
          if (k0 + ks < kend) {
#pragma unroll
            for (int b = 0; b < GNB; ++b) {
              if (wave_bias) frag_acc<BF16>(bsum[b], zf[ks][b]);
#pragma unroll
              for (int c = 0; c < GNI; ++c) P::mma(xf[ks][c], zf[ks][b], acc[b][c]);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
          if (ks + GIF < UKC) issue(k0, ks + GIF);  // into the registers the k-step just consumed freed
        }
      };
      chunk(kbeg, true);  // straight-line first chunk (a loop pre-header would drain the pending loads)
#pragma unroll 1
      for (int k0 = kbeg + UKC; k0 < kend; k0 += UKC) chunk(k0, false);
      __builtin_amdgcn_sched_barrier(0);
      load_state();  // the operand registers are free now
      // C/D layout: lane (r, q) of acc[b][c] holds dW[o0 + 32 g + 16 b + r][i0 + 16 c + 4 q + k]
      float *const tdst = half ? tile2 : tile;  // (the second half's partial sums: added in the Adam pass)
#pragma unroll
      for (int b = 0; b < GNB; ++b)
#pragma unroll
        for (int c = 0; c < GNI; ++c)
          *reinterpret_cast<f32x4 *>(&tdst[(16 * GNB * gw + 16 * b + r) * ULD + 16 * c + 4 * q]) = acc[b][c];
      if (wave_bias) {
#pragma unroll
        for (int b = 0; b < GNB; ++b) {
          float bs = bsum[b];
          bs = xor32_sum(xor16_sum(bs));
          if (q == 0) (half ? bgrad2 : bgrad)[16 * GNB * gw + 16 * b + r] = bs;
        }
      }
    }
  }
#if !IQL_G2_EARLY
  if constexpr (!LAT) {
    if (!ksplit && wave >= 2) load_state();
  }
#endif
  __syncthreads();
  STAMP(2, 3);

  // ---- 3. Adam + Polyak in row order; gradients come back from LDS ----
  T *wc = reinterpret_cast<T *>(it.wc);
  T *tc = reinterpret_cast<T *>(it.tc);
  const int nkw = Kw / P::KM;
#pragma unroll
  for (int ps = 0; ps < UNP; ++ps) {
    const int ol = tr + URPP * ps, o = o0 + ol, i = i0 + tc4;
    const float4 g4 = *reinterpret_cast<const float4 *>(&tile[ol * ULD + tc4]);
    float g[4] = {g4.x, g4.y, g4.z, g4.w};
    if constexpr (!LAT) {
      if (ksplit) {  // lo + hi of a split batch (the one-seed variant added them in registers)
        const float4 h4 = *reinterpret_cast<const float4 *>(&tile2[ol * ULD + tc4]);
        g[0] += h4.x, g[1] += h4.y, g[2] += h4.z, g[3] += h4.w;
      }
    }
    float p[4] = {pw[ps][0], pw[ps][1], pw[ps][2], pw[ps][3]};
    float m[4] = {mw[ps][0], mw[ps][1], mw[ps][2], mw[ps][3]};
    float v[4] = {vw[ps][0], vw[ps][1], vw[ps][2], vw[ps][3]};
    float tv[4] = {tw[ps][0], tw[ps][1], tw[ps][2], tw[ps][3]};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      g[k] = P::round(g[k]);
      adam_apply<AF>(p[k], m[k], v[k], g[k], coef, neg_step);
      if (has_target) tv[k] = polyak(D, tv[k], p[k]);
    }
    // the new weights replace this thread's gradients in the tile (step 4 reads them transposed)
    if (L == 1) *reinterpret_cast<float4 *>(&tile[ol * ULD + tc4]) = make_float4(p[0], p[1], p[2], p[3]);
    // group launches: also the new target weights -- step 4 writes every copy of a layer-2 tile from
    // LDS, eight elements (16 bytes of bf16) per store instead of four
    const bool copies_from_lds = !LAT && L == 1;
    if (copies_from_lds && has_target)
      *reinterpret_cast<float4 *>(&tile2[ol * ULD + tc4]) = make_float4(tv[0], tv[1], tv[2], tv[3]);
    if (o < Odim && i < Idim) {
      const int64_t e = it.off_w + (int64_t)o * Idim + i;
      const int64_t te = it.toff_w + (int64_t)o * Idim + i;
      state_store(g_params, e, make_float4(p[0], p[1], p[2], p[3]));  // (arenas are far below 4 GB)
      state_store(g_m, e, make_float4(m[0], m[1], m[2], m[3]));
      state_store(g_v, e, make_float4(v[0], v[1], v[2], v[3]));
      if (g_grads) stg16(g_grads + e, make_float4(g[0], g[1], g[2], g[3]));
      if (has_target) state_store(g_target, te, make_float4(tv[0], tv[1], tv[2], tv[3]));
      // 4 consecutive k of one row are contiguous in the fragment-major copies
      if (!copies_from_lds) {
        copy_store4<BF16>(wc, fidx<P>(o, i, nkw), p);
        if (has_target) copy_store4<BF16>(tc, fidx<P>(o, i, nkw), tv);
      }
      if (L == 2) {  // layer 3: the [H][Opad] transposed copy k_backward reads (one unit's weights to all outputs)
        T *w3t = reinterpret_cast<T *>(it.w3t);
#pragma unroll
        for (int k = 0; k < 4; ++k) stg(w3t + (size_t)(i + k) * Opad + o, P::from_f32(p[k]));
      }
    }
  }
  STAMP(2, 5);
  // ---- bias: thread t < 64 owns out-feature o0 + t ----
  if (do_bias && tid < UTO && o0 + tid < Odim) {
    const int64_t e = it.off_b + o0 + tid;
    float gb = bgrad[tid];
    if constexpr (!LAT) {
      if (ksplit) gb += bgrad2[tid];
    }
    const float g = P::round(gb);
    adam_apply<AF>(pb, mb, vb, g, coef, neg_step);
    stg(g_params + e, pb), stg(g_m + e, mb), stg(g_v + e, vb);
    if (g_grads) stg(g_grads + e, g);
    if (has_target) stg(g_target + it.toff_b + o0 + tid, polyak(D, tb, pb));
  }
  STAMP(2, 6);
  // ---- 4. transposed compute copy of layer 2 for the backward GEMM: [in][out] ----
  if (L == 1) {
    __syncthreads();
    if constexpr (!LAT) {
      static_assert(LAT || (UT == 256 && UTO == 64 && UTI == 32), "one pass of 8-element stores covers the tile");
      // [out][in] copies (and the targets'): thread -> row tid / 4, in-features 8 (tid % 4) .. + 7
      {
        const int ol = tid >> 2, c8 = (tid & 3) * 8;
        const float4 a0 = *reinterpret_cast<const float4 *>(&tile[ol * ULD + c8]);
        const float4 a1 = *reinterpret_cast<const float4 *>(&tile[ol * ULD + c8 + 4]);
        const float pv8[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        copy_store8<BF16>(wc, fidx<P>(o0 + ol, i0 + c8, nkw), fidx<P>(o0 + ol, i0 + c8 + 4, nkw), pv8);
        if (has_target) {
          const float4 t0 = *reinterpret_cast<const float4 *>(&tile2[ol * ULD + c8]);
          const float4 t1 = *reinterpret_cast<const float4 *>(&tile2[ol * ULD + c8 + 4]);
          const float tv8[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
          copy_store8<BF16>(tc, fidx<P>(o0 + ol, i0 + c8, nkw), fidx<P>(o0 + ol, i0 + c8 + 4, nkw), tv8);
        }
      }
      // [in][out] copy: thread -> in-feature i0 + (tid & 31), out-features o0 + 8 (tid >> 5) .. + 7
      {
        const int il = tid & (UTI - 1), o8 = (tid >> 5) * 8;
        float pv8[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) pv8[k] = tile[(o8 + k) * ULD + il];
        copy_store8<BF16>(reinterpret_cast<T *>(it.w2ct), fidx<P>(i0 + il, o0 + o8, H / P::KM),
                          fidx<P>(i0 + il, o0 + o8 + 4, H / P::KM), pv8);
      }
      STAMP(2, 4);
      return;
    }
    // thread -> in-feature i0 + (tid & 31), out-features o0 + 4 (tid >> 5) .. +3 (+ UT / 8 per pass)
#pragma unroll
    for (int ps = 0; ps < (UTO / 4) / (UT / UTI); ++ps) {
      // consecutive lanes read consecutive in-features of one tile row: conflict-free LDS reads
      const int il = tid & (UTI - 1), o4 = ((tid >> 5) + (UT / UTI) * ps) * 4;
      float pv4[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) pv4[k] = tile[(o4 + k) * ULD + il];
      copy_store4<BF16>(reinterpret_cast<T *>(it.w2ct), fidx<P>(i0 + il, o0 + o4, H / P::KM), pv4);
    }
  }
  STAMP(2, 4);
}

// ------------------------------------------------------------------------
// __global__ wrappers
// ------------------------------------------------------------------------
template <bool BF16, int H, bool PRE, int PW>
__global__ __launch_bounds__(256, PRE ? 1 : (PW > 1 ? (BF16 ? 4 : 2) : (BF16 ? 6 : 3)))
void k_backward(const TrainerDesc *__restrict__ Dp, const DevArgs *__restrict__ Ap, DevCtr *__restrict__ Cp,
                const int nslab, const int ntrain) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  backward_body<BF16, H, PRE, PW>(Dp + blockIdx.y, Cp + blockIdx.y, (int)blockIdx.x, smem, nslab, ntrain);
}

template <bool BF16, bool LAT>
__global__ __launch_bounds__(LAT ? 512 : 256, LAT ? 2 : (BF16 ? 4 : 3)) void k_update(const TrainerDesc *__restrict__ Dp,
                                                             const DevArgs *__restrict__ Ap,
                                                             DevCtr *__restrict__ Cp,
                                                             const UpdItem *__restrict__ items, int n_items) {
  // group launch: blockIdx.y = seed; grid.x is padded to a multiple of 8 (so that block -> XCD
  // stays blockIdx.x & 7 for every seed), the blocks behind the misc block have nothing to do
  if ((int)blockIdx.x > n_items) return;
  update_body<BF16, LAT, (LAT ? 512 : 256)>(Dp + blockIdx.y, Ap + blockIdx.y, Cp + blockIdx.y,
                         items + (size_t)blockIdx.y * n_items, n_items, (int)blockIdx.x);
}

// ========================================================================
// k_sync_weights: rebuild every compute-precision copy from the fp32 masters
// ========================================================================
template <bool BF16>
__global__ void k_sync_weights(const TrainerDesc *__restrict__ Dp) {
  const TrainerDesc &D = *Dp;
  using P = Prec<BF16>;
  using T = typename P::T;
  const int net = blockIdx.y;
  if (net >= D.ntrain) return;
  const TrainNet &N = D.net[net];
  const int H = D.H;
  for (int L = 0; L < 3; ++L) {
    const int Odim = (L == 2) ? N.out_dim : H;
    const int Idim = (L == 0) ? N.in_dim : H;
    const int Kw = (L == 0) ? N.k1pad : H;
    const int total = Odim * Idim;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
      const int o = e / Idim, i = e - o * Idim;
      const float p = ldg(D.params + N.off_w[L] + e);
      stg(reinterpret_cast<T *>(N.wc[L]) + fidx<P>(o, i, Kw / P::KM), P::from_f32(p));
      if (L == 1) stg(reinterpret_cast<T *>(N.w2ct) + fidx<P>(i, o, H / P::KM), P::from_f32(p));
      if (L == 2) stg(reinterpret_cast<T *>(N.w3t) + (size_t)i * N.out_pad + o, P::from_f32(p));
      if (N.has_target)
        stg(reinterpret_cast<T *>(N.tc[L]) + fidx<P>(o, i, Kw / P::KM),
            P::from_f32(ldg(D.target + N.toff_w[L] + e)));
    }
  }
}

// ------------------------------------------------------------------------
// launchers (host driver: api.hip)
// ------------------------------------------------------------------------
// k_infer (unsplit 16-row slabs, mlp_slab)
size_t infer_smem_bytes(bool bf16, int H, int k1max) {
  const int es = bf16 ? 2 : 4, epv = bf16 ? 8 : 4;
  return (size_t)SLAB * (k1max + epv) * es + 2 * (size_t)SLAB * (H + epv) * es + 4 * 2 * 64 * 4 * 4;
}
// k_forward: xs [16 MT][k1max + EPV], h1 [16 MT][H + EPV], h2 [PW][16 MT][64 + EPV]
size_t fwd_smem_bytes(bool bf16, int H, int k1max, int mt, int pw) {
  const int es = bf16 ? 2 : 4, epv = bf16 ? 8 : 4;
  return (size_t)16 * mt * ((k1max + epv) + (H + epv) + pw * (64 + epv)) * es;
}
size_t bwd_smem_bytes(bool bf16, int H) {
  const int es = bf16 ? 2 : 4, epv = bf16 ? 8 : 4;
  return (size_t)SLAB * (H + epv) * es + 3 * SLAB * 32 * 4 + SLAB * 4 + SLAB * FIN_LD * 4;
}
// batch rows per forward work-group = 16 x this (more rows per work-group = fewer re-reads of
// the weights; fewer work-groups): 2 for the headline batch 256 (224 work-groups at E = 2)
// Measured (tools/group_scan.py, IQLHIP_FWD_MT / IQLHIP_FWD_PW): one seed alone is fastest with
// 16-row work-groups (448 of them at batch 256: two per CU, whose phases interleave on each
// SIMD), two seeds and more with 32 rows; from four seeds on with two layer-2 parts per
// work-group (K = 8: 154.8k steps/s against 150.2k with 64 rows x one part, 139.7k with 32 x one).
int fwd_row_tiles(int B, int n_seeds) {
  static const int forced = getenv("IQLHIP_FWD_MT") ? atoi(getenv("IQLHIP_FWD_MT")) : 0;  // A/B knob
  if ((forced == 1 || forced == 2 || forced == 4) && B % (16 * forced) == 0) return forced;
  const int64_t rows = (int64_t)B * n_seeds;
  // (round 3, on the balanced update table: 64-row work-groups -- IQLHIP_FWD_MT=4 -- gain 1.6 % for eight
  // seeds as ONE group, 202.5k -> 205.7k steps/s, and lose for everything else: K = 4 152.9k -> 147.1k,
  // K = 2 103.7k -> 98.6k, two sub-groups of eight on two streams 263.5k -> 255.4k: not the default)
  if (rows >= 512 && B % 32 == 0) return 2;
  return 1;
}
int layer2_parts(int H) { return H >= 256 ? 4 : H / 64; }
// parts of hidden layer 2 per forward work-group: 2 in a group launch that keeps the chip full
// anyway (layer 1 and its epilogue are then computed twice per slab instead of four times)
int fwd_parts_per_wg(int B, int H, int n_seeds) {
  static const int forced = getenv("IQLHIP_FWD_PW") ? atoi(getenv("IQLHIP_FWD_PW")) : 0;  // A/B knob
  const int spl = layer2_parts(H);
  if ((forced == 1 || forced == 2) && spl % forced == 0) return forced;
  return ((int64_t)B * n_seeds >= 1024 && spl % 2 == 0) ? 2 : 1;
}

#define DISPATCH_H(BF, HH, CALL)                 \
  do {                                           \
    if (BF) {                                    \
      if (HH == 256) { CALL(true, 256); }        \
      else if (HH == 128) { CALL(true, 128); }   \
      else { CALL(true, 64); }                   \
    } else {                                     \
      if (HH == 256) { CALL(false, 256); }       \
      else if (HH == 128) { CALL(false, 128); }  \
      else { CALL(false, 64); }                  \
    }                                            \
  } while (0)

// Launches with many rows (seed groups, batch-1024 ensembles) take the throughput kernels
// k_forward_tp / k_backward_tp (bf16, H = 256, batch a multiple of 64); IQLHIP_TP=0 / 1 forces the choice.
bool use_tp(bool bf16, const TrainerDesc &D, int n_seeds, bool backward) {
  static const int forced = getenv("IQLHIP_TP") ? atoi(getenv("IQLHIP_TP")) : -1;          // A/B knobs: both kernels,
  static const int forced_b = getenv("IQLHIP_TP_BWD") ? atoi(getenv("IQLHIP_TP_BWD")) : -1;  // the backward alone
  if (!bf16 || D.H != 256 || D.B % 64 != 0) return false;
  if (backward && forced_b >= 0) return forced_b != 0;
  if (forced >= 0) return forced != 0;
  // The throughput kernels are one eight-wave work-group per CU, all resident at once: they win while
  // their work-groups fill a good part of the 256 CUs and lose beyond them (two rounds) and far below
  // (nothing covers a lone work-group's latency).  Measured (round 4, tools/tp_matrix.sh and d2 / g4;
  // steps/s old kernels -> throughput forward + old backward -> both):
  //   forward work-groups = evaluations x batch / 64 x seeds, backward = trained nets x batch / 32 x seeds
  //   E = 3 / batch 256 (36 / 40) 56.4k -> 52.7k -> 47.9k     E = 4 / 256 (44 / 48) 52.7k -> 51.6k -> 46.3k
  //   2 seeds (56 / 64) 110.6k -> 102.5k -> 89.2k            E = 8 / 256 (76 / 80) 41.4k -> 44.3k -> 42.9k
  //   E = 4 / 512 (88 / 96) 46.0k -> 47.8k -> 44.1k           4 seeds (112 / 128) 167.0k -> 167.3k -> 154.7k
  //   TwinQ / 1024 (112 / 128) 48.6k -> 48.8k -> 44.4k        E = 3 / 1024 (144 / 160) 35.9k -> 38.8k -> 40.7k
  //   E = 4 / 1024 (176 / 192) 31.6k -> 34.4k -> 37.4k        8 seeds (224 / 256) 200.9k -> 223.4k -> 223.9k
  //   E = 8 / 1024 (304 / 320) 24.8k -> 24.4k -> 23.9k
  const int64_t fwd_wgs = (int64_t)D.nfwd * (D.B / 64) * n_seeds, bwd_wgs = (int64_t)D.ntrain * (D.B / 32) * n_seeds;
  const bool fwd_tp = fwd_wgs >= 72 && fwd_wgs <= 256;
  return backward ? (fwd_tp && bwd_wgs >= 144 && bwd_wgs <= 256) : fwd_tp;
}
// once per trainer (iqlhip_trainer_create, outside any stream capture): k_forward_tp's 64-row slabs
// take more than the 64 KB of dynamic LDS a kernel may use by default
hipError_t prepare_step_kernels() {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_forward_tp<256, false>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute(reinterpret_cast<const void *>(k_forward_tp<256, true>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}
size_t fwd_tp_smem_bytes(int H, int k1max) { return (size_t)64 * ((k1max + 8) + 2 * (H + 8)) * 2; }
size_t bwd_smem_bytes(bool bf16, int H);

hipError_t launch_forward(bool bf16, const TrainerDesc &D, const TrainerDesc *dD, const DevArgs *a,
                          const DevCtr *c, int n_seeds, hipStream_t st) {
  if (use_tp(bf16, D, n_seeds, false)) {
    const int nsl64 = D.B / 64;
    const int grid = 8 * ((D.nfwd + 1 + 7) / 8) * nsl64;
    // (more than 64 KB of dynamic LDS: allowed by prepare_step_kernels, at trainer creation)
    if (D.has_dropout)
      hipLaunchKernelGGL((k_forward_tp<256, true>), dim3(grid, n_seeds), dim3(512), fwd_tp_smem_bytes(256, D.k1max), st,
                         dD, a, c, nsl64, D.nfwd);
    else
      hipLaunchKernelGGL((k_forward_tp<256, false>), dim3(grid, n_seeds), dim3(512), fwd_tp_smem_bytes(256, D.k1max), st,
                         dD, a, c, nsl64, D.nfwd);
    return hipGetLastError();
  }
  const int mt = fwd_row_tiles(D.B, n_seeds), nsl = (D.B + 16 * mt - 1) / (16 * mt);
  const int pw = fwd_parts_per_wg(D.B, D.H, n_seeds);
  // nfwd evaluations + the spare job, each nsl slabs x (SPL / pw) part groups
  const int grid = 8 * ((D.nfwd + 1 + 7) / 8) * nsl * (layer2_parts(D.H) / pw);
  const size_t sm = fwd_smem_bytes(bf16, D.H, D.k1max, mt, pw);
#define LAUNCH_F(BF, HH, MTV, PWV)                                                                            \
  hipLaunchKernelGGL((k_forward<BF, HH, MTV, PWV, false>), dim3(grid, n_seeds), dim3(256), sm, st, dD, a, c, nsl, D.nfwd)
#define CALL(BF, HH)                                                   \
  do {                                                                 \
    if constexpr (HH >= 128) {                                         \
      if (pw == 2) {                                                   \
        if (mt == 4) LAUNCH_F(BF, HH, 4, 2);                           \
        else if (mt == 2) LAUNCH_F(BF, HH, 2, 2);                      \
        else LAUNCH_F(BF, HH, 1, 2);                                   \
        break;                                                         \
      }                                                                \
    }                                                                  \
    if (mt == 4) LAUNCH_F(BF, HH, 4, 1);                               \
    else if (mt == 2) LAUNCH_F(BF, HH, 2, 1);                          \
    else LAUNCH_F(BF, HH, 1, 1);                                       \
  } while (0)
  DISPATCH_H(bf16, D.H, CALL);
#undef CALL
#undef LAUNCH_F
  return hipGetLastError();
}
// parts of the dZ1 columns per backward work-group: 1 for one seed at batch 256 (every part repeats
// the loss / dZ2 phase: parallelism bought with redundant work), 2 from 512 rows per launch on.
// Measured (tools/group_scan.py, IQLHIP_BWD_PW; steps/s with 1 / 2 parts per work-group): one seed
// 65.4k / 64.1k, two seeds 97.4k / 102.2k, four 136.2k / 144.6k, eight 177.3k / 189.5k (k_backward 10.7 ->
// 8.0 us); all four parts in one work-group (512 work-groups of 128 registers, spilling): 148k.
int bwd_parts_per_wg(int B, int H, int n_seeds) {
  static const int forced = getenv("IQLHIP_BWD_PW") ? atoi(getenv("IQLHIP_BWD_PW")) : 0;  // A/B knob
  const int spl = layer2_parts(H);
  if ((forced == 1 || forced == 2 || forced == 4) && spl % forced == 0) return forced;
  return ((int64_t)B * n_seeds >= 512 && spl % 2 == 0) ? 2 : 1;
}
hipError_t launch_backward(bool bf16, const TrainerDesc &D, const TrainerDesc *dD, const DevArgs *a,
                           DevCtr *c, int n_seeds, hipStream_t st) {
  if (use_tp(bf16, D, n_seeds, true)) {
    const int nslab32 = D.B / 32;
    const int nxn = (D.ntrain <= 4 && nslab32 % 2 == 0) ? 2 : 1;  // XCDs per trained net
    const int grid = 8 * ((D.ntrain * nxn + 7) / 8) * (nslab32 / nxn);
    hipLaunchKernelGGL((k_backward_tp<256>), dim3(grid, n_seeds), dim3(512), 2 * bwd_smem_bytes(true, 256), st, dD, a, c,
                       nslab32, D.ntrain, nxn);
    return hipGetLastError();
  }
  const int pw = bwd_parts_per_wg(D.B, D.H, n_seeds);
  const int grid = 8 * ((layer2_parts(D.H) / pw * D.ntrain + 7) / 8) * (D.B / SLAB);
  const size_t sm = bwd_smem_bytes(bf16, D.H);
  static const int forced_pre = getenv("IQLHIP_BWD_PRE") ? atoi(getenv("IQLHIP_BWD_PRE")) : -1;  // A/B knob
  // measured: no difference for one seed (63.2k either way), K = 8 170.2k against 165.1k
  const bool pre = forced_pre >= 0 ? forced_pre != 0 : (int64_t)D.B * n_seeds < 1024;
#define LAUNCH_B(BF, HH, PRE_, PW_)                                                                          \
  hipLaunchKernelGGL((k_backward<BF, HH, PRE_, PW_>), dim3(grid, n_seeds), dim3(256), sm, st, dD, a, c, D.B / SLAB, \
                     D.ntrain)
#define CALL(BF, HH)                                                  \
  do {                                                                \
    if constexpr (HH >= 256) {                                        \
      if (pw == 4) { LAUNCH_B(BF, HH, false, 4); break; }             \
    }                                                                 \
    if constexpr (HH >= 128) {                                        \
      if (pw == 2) { LAUNCH_B(BF, HH, false, 2); break; }             \
    }                                                                 \
    if (pre) LAUNCH_B(BF, HH, true, 1);                               \
    else LAUNCH_B(BF, HH, false, 1);                                  \
  } while (0)
  DISPATCH_H(bf16, D.H, CALL);
#undef CALL
#undef LAUNCH_B
  return hipGetLastError();
}
hipError_t launch_stage(bool bf16, const TrainerDesc &D, const TrainerDesc *dD, const DevArgs *a,
                        const DevCtr *c, int n_seeds, hipStream_t st) {
  if (bf16)
    hipLaunchKernelGGL(k_stage<true>, dim3((D.B + 15) / 16, n_seeds), dim3(256), 0, st, dD, a, c);
  else
    hipLaunchKernelGGL(k_stage<false>, dim3((D.B + 15) / 16, n_seeds), dim3(256), 0, st, dD, a, c);
  return hipGetLastError();
}
int strip_rows() { return USR; }
int update_lds_floats() { return UPD_LDS; }
hipError_t launch_update(bool bf16, const TrainerDesc *dD, const DevArgs *a, DevCtr *c,
                         const UpdItem *items, int n_items, int n_seeds, hipStream_t st) {
  // n_items tiles + the misc block; a group launch pads grid.x to a multiple of 8
  const dim3 grid(n_seeds > 1 ? round_up(n_items + 1, 8) : n_items + 1, n_seeds);
  // The 512-thread latency variant while a lone seed's work-groups have a CU each; the four-per-CU
  // throughput variant for groups and for a seed with more work items than CUs (E = 4 critics at batch
  // 1024: 337 items, 30.7k -> 32.0k steps/s; two critics at batch 256: 65.7k -> 63.5k).  The same bits.
  static const int forced_lat = getenv("IQLHIP_UPD_LAT") ? atoi(getenv("IQLHIP_UPD_LAT")) : -1;  // A/B knob
  static const int cus = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      n = 256;
    return n;
  }();
  const bool lat = forced_lat >= 0 ? forced_lat != 0 : (n_seeds == 1 && n_items + 1 <= cus + 1);
  if (bf16 && !lat)
    hipLaunchKernelGGL((k_update<true, false>), grid, dim3(256), 0, st, dD, a, c, items, n_items);
  else if (bf16)
    hipLaunchKernelGGL((k_update<true, true>), grid, dim3(512), 0, st, dD, a, c, items, n_items);
  else if (!lat)
    hipLaunchKernelGGL((k_update<false, false>), grid, dim3(256), 0, st, dD, a, c, items, n_items);
  else
    hipLaunchKernelGGL((k_update<false, true>), grid, dim3(512), 0, st, dD, a, c, items, n_items);
  return hipGetLastError();
}
hipError_t launch_infer(bool bf16, const TrainerDesc &D, const TrainerDesc *dD, const FwdNet &N,
                        const float *s, const float *a, int64_t n, float *out, int out_stride,
                        hipStream_t st) {
  const int grid = (int)((n + SLAB - 1) / SLAB);
  const size_t sm = infer_smem_bytes(bf16, D.H, D.k1max);
#define CALL(BF, HH) \
  hipLaunchKernelGGL((k_infer<BF, HH>), dim3(grid), dim3(256), sm, st, dD, N, s, a, n, out, out_stride)
  DISPATCH_H(bf16, D.H, CALL);
#undef CALL
  return hipGetLastError();
}
hipError_t launch_sync_weights(bool bf16, const TrainerDesc *dD, hipStream_t st) {
  if (bf16)
    hipLaunchKernelGGL(k_sync_weights<true>, dim3(32, MAX_TRAIN), dim3(256), 0, st, dD);
  else
    hipLaunchKernelGGL(k_sync_weights<false>, dim3(32, MAX_TRAIN), dim3(256), 0, st, dD);
  return hipGetLastError();
}

}  // namespace iqlhip
