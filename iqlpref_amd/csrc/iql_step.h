// Descriptors shared by the host driver and the step kernels (iql_step.hip).
#pragma once
#include <stdint.h>

namespace iqlhip {

// E critics (TwinQ: E = 2; SURVEY 8 "config 5": the E-way critic ensemble), V and the actor are
// trained; one step evaluates the E critics, V(s), the actor, the E target critics and V(s').
//   trained nets : critic e = e, V = E, actor = E + 1                      (E + 2)
//   evaluations  : q_e = e, v = E, actor = E + 1, qt_e = E + 2 + e, next_v = 2E + 2   (2E + 3)
//   columns of the forward outputs (planes outs[part][column][row]):
//                  q_e = e, v = E, qt_e = E + 1 + e, next_v = 2E + 1, mean_j = 2E + 2 + j
constexpr int MAX_CRITICS = 8;
constexpr int MAX_TRAIN = MAX_CRITICS + 2;
constexpr int MAX_FWD = 2 * MAX_CRITICS + 3;

struct FwdNet {
  const void *w1c, *w2c, *w3c;  // compute-precision copies [H][k1pad] [H][H] [out_pad][H]
  const float *b1, *b2, *b3;    // fp32 biases (masters / target arena)
  int32_t in_off;               // float offset of the input inside a replay row
  int32_t in_dim, k1pad;        // layer-1 K, padded to the MFMA macro step
  int32_t out_dim, out_pad;     // layer-3 N, padded to 16
  int32_t out_col;              // first column in outs[][]
  int32_t train_slot;           // 0..3: store hidden activations for backward; -1: no
  int32_t tanh_out;             // actor
  int32_t dropout;              // actor with dropout
  int32_t stage;                // this evaluation also writes the batch staging (q1)
};

struct TrainNet {
  // fp32 arenas (element offsets are the same in params / exp_avg / exp_avg_sq / grads)
  int64_t off_w[3], off_b[3];
  int64_t toff_w[3], toff_b[3];  // target arena offsets (q nets), -1 otherwise
  void *wc[3];                   // compute copies written by the update kernel
  void *w2ct;                    // [H][H] transposed copy of layer 2 (backward GEMM)
  void *w3t;                     // [H][out_pad] row-major transposed copy of layer 3 (backward: one
                                 // hidden unit's weights to all outputs are 1-2 contiguous 16-byte reads)
  void *tc[3];                   // target compute copies (q nets) or null
  int32_t in_dim, k1pad, out_dim, out_pad;
  int32_t has_target;
};

// The scalars and workspace pointers every kernel needs BEFORE its first vector load sit at the
// front, in a few adjacent cache lines: a kernel reads them in one batch of scalar loads at its
// top (a scalar load that misses is ~0.3-0.5 us; a chain of dependent ones was the first 1.5 us
// of every kernel).
struct TrainerDesc {
  int32_t S, A, H, B, BP, OUTW, k1max;  // BP: batch leading dimension (B padded to 32)
  int32_t E, ntrain, nfwd, net_v, net_a; // critics; E + 2; 2E + 3; E; E + 1
  int32_t out_v, out_qt, out_nv, out_mean;  // column bases in outs[][]
  int32_t stage_stride, next_off;        // replay row stride; float offset of s' inside a row
  int32_t opmax, xrows;
  int32_t deterministic, has_dropout;
  int32_t prefetch;                      // k_update's idle work-groups gather the next step's batch
  int32_t polyak_convex; // 0: t + tau (p - t) (offline/iql.py), 1: (1 - tau) t + tau p (custom_offline)
  float two_over_B, inv_E;               // mse backward: (2/B) * (q - t) * (1/E)   (ref:606)
  float discount, tau, beta, iql_tau;
  float one_m_tau;       // float(1 - tau) for the convex Polyak form
  float drop_scale;      // 1/(1-p), bf16-rounded in bf16 mode (ATen _dropout_impl)
  uint32_t drop_thr;     // keep iff philox word >= thr
  int32_t pad0_;
  uint64_t seed;
  // workspace (T = compute type)
  float *stage_rows;  // [B][stage_stride] the batch of the step about to run (k_stage / k_update)
  void *xT;       // [2][xrows][B] layer-1 input (s|a), feature-major; plane = step parity
  float *rd;      // [B][2]       reward, done
  float *actf;    // [B][A]       actions (fp32, actor loss)
  void *hT;       // [ntrain][2][H][B] hidden activations (post ReLU / dropout)
  void *dz1T;     // [ntrain][H][B]
  void *dz2T;     // [ntrain][H][B]
  void *dz3T;     // [ntrain][opmax][B]
  float *outs;    // [SPL][OUTW][B] forward outputs as partial dot products, one plane per part
                  // of hidden layer 2 (SPL = 4 at H = 256), column-major (a lane's 4 rows = one
                  // 16-byte store); summed by k_backward (fin_value)
  float *lossp;   // [ntrain][nslab]   per-slab loss partial sums
  float *lsp;     // [nslab][A]   per-slab d(loss)/d(std) partial sums
  float *ls_snap; // [A] log_std as of the start of the step (written by k_forward's spare block)
  // arenas
  float *params, *exp_avg, *exp_avg_sq, *target, *grads;
  int64_t off_log_std;
  unsigned long long *dbg;  // diagnostic stamps (IQL_STAMPS builds), else null
  double beta1, beta2, eps;
  int64_t t_max;
  FwdNet fwd[MAX_FWD];
  TrainNet net[MAX_TRAIN];
};

// Host-written, read-only for the kernels during a launch sequence.
struct DevArgs {
  const float *rows;  // replay view
  int64_t n_rows;
  int32_t row_stride;
  int32_t idx_mode;          // 0 philox, 1 injected, 2 identity (explicit batch)
  const int64_t *idx;        // [n][B] when idx_mode == 1
  const uint8_t *drop_keep;  // [n][2][B][H] or null (philox masks)
  float *losses_out;         // [n][3] or null
  int64_t base_step;         // total_it of the first step of this call
  double lr_q, lr_v, lr_a_base;
  int64_t n_steps;           // steps of this call (bounds idx[] for the batch prefetch)
  uint64_t generation;       // of the replay view (host-side bookkeeping: see push_args)
};

// Device-written counters / metrics.
struct AdamCoef {
  float one_m_b1, b2, one_m_b2, neg_step[3], bc2_sqrt, eps;  // neg_step per group q / v / actor
  float inv_bc2_sqrt, pad_;                                    // 1 / sqrt(1 - beta2^t) (IQL_ADAM_FAST)
};

struct DevCtr {
  int64_t ctr[2];     // [0] steps completed (read by fwd/bwd), [1] 1-based Adam step (update)
  int64_t coef_step;  // 1-based Adam step `coef` belongs to (= ctr[0] + 1 of that forward)
  AdamCoef coef;      // written by a spare forward block for the step in flight
  float last_losses[4];
  double loss_sum[4];
};

// One work-group of k_update: everything it needs, flattened (no second,
// dependent descriptor load; no runtime-indexed struct arrays -> no scratch).
struct UpdItem {
  int32_t net, layer;      // net < 0: padding slot of the XCD-major table (o0 = its index among
                           // the padding slots, i0 = their count: they prefetch the next batch)
  int32_t o0, i0;          // tile origin: out-features [o0, +64) x in-features [i0, +64)
  int32_t Odim, Idim;      // true extents of the weight matrix
  int32_t Opad, Kw;        // out-features padded to 16; K extent of the compute copy
  int32_t has_target, group;  // group: 0 q, 1 v, 2 actor (which Adam step size)
  int64_t off_w, off_b, toff_w, toff_b;
  void *wc, *tc, *w2ct;    // compute copies to refresh (w2ct: layer 2 only, else null)
  void *w3t;               // layer 3 only: [H][Opad] transposed row-major copy, else null
  const void *Xsrc, *Zsrc; // fragment-major layer input and dZ^T of this (net, layer)
};

}  // namespace iqlhip
