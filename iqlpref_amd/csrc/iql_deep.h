// Descriptors of the general layer-wise IQL step (iql_deep.hip): any number of hidden layers
// (n_hidden = 1..6) of any width (1..1024), the shapes ref:417-449 MLP / :452-543 accept and the
// tuned three-Linear step (iql_step.hip) is not built for.
#pragma once
#include <stdint.h>

#include "iql_step.h"

namespace iqlhip {

constexpr int DEEP_MAX_LIN = 7;  // Linear layers per network (n_hidden <= 6)
constexpr int DEEP_MAX_H = 1024;

// One Linear layer as an evaluation reads it.
struct DeepLin {
  const void *w;   // compute-precision copy of W[N][K], zero padded to [Npad][Kpad], fragment-major (common.h fidx)
  const float *b;  // fp32 bias [N] (master or target arena)
  int32_t K, Kpad, N, Npad;
};

// One forward evaluation (q_e, v, actor, target q_e, next_v: iql_step.h numbering).
struct DeepEval {
  DeepLin lin[DEEP_MAX_LIN];
  int32_t in_off, in_dim;  // float offset / width of the input inside a replay row
  int32_t out_dim, out_col;
  int32_t train_slot;      // trained net whose activation planes this evaluation writes, or -1
  int32_t tanh_out, dropout, stage;
};

// One trained network: where its tensors live and what the update kernel refreshes.
struct DeepNet {
  int64_t off_w[DEEP_MAX_LIN], off_b[DEEP_MAX_LIN];    // parameter / moment / gradient arenas
  int64_t toff_w[DEEP_MAX_LIN], toff_b[DEEP_MAX_LIN];  // target arena (critics), else -1
  void *wc[DEEP_MAX_LIN];  // compute copies [Npad][Kpad], fragment-major
  void *wt[DEEP_MAX_LIN];  // transposed copies [Kpad][NKpad], fragment-major (layers >= 1: the backward GEMM's operand)
  void *tc[DEEP_MAX_LIN];  // target compute copies (critics) or null
  void *hT[DEEP_MAX_LIN];  // hT[l]: input of layer l >= 1 = hidden activations after layer l - 1,
                           // feature-major [rows64(Kpad)][BP]; hT[0] = the batch's (s | a) plane
  void *dzT[DEEP_MAX_LIN]; // d(loss)/d(pre-activation of layer l), feature-major [rows64(Npad)][BP]
  int32_t K[DEEP_MAX_LIN], Kpad[DEEP_MAX_LIN], N[DEEP_MAX_LIN], Npad[DEEP_MAX_LIN];
  int32_t NKpad[DEEP_MAX_LIN];  // N padded to the MFMA K step (row length of wt)
  int32_t has_target, group;    // group: 0 q, 1 v, 2 actor (which Adam step size)
};

struct DeepDesc {
  int32_t S, A, H, Hp, B, BP, OUTW, NL;  // NL = n_hidden + 1 Linear layers
  int32_t E, ntrain, nfwd, net_v, net_a;
  int32_t out_v, out_qt, out_nv, out_mean;
  int32_t next_off, opad, nslab, lds_w;   // lds_w: elements per LDS activation row (widest layer + pad)
  int32_t deterministic, has_dropout, polyak_convex, pad0_;
  float two_over_B, inv_E;
  float discount, tau, beta, iql_tau, one_m_tau, drop_scale;
  uint32_t drop_thr, pad1_;
  uint64_t seed;
  float *rd;     // [B][2] reward, done
  float *actf;   // [B][A] actions (fp32)
  float *outs;   // [OUTW][BP] finished forward outputs (fp32 values of the compute precision)
  float *lossp;  // [ntrain][nslab]
  float *lsp;    // [nslab][A]
  float *params, *exp_avg, *exp_avg_sq, *target, *grads;
  int64_t off_log_std;
  DeepEval ev[MAX_FWD];
  DeepNet net[MAX_TRAIN];
};

// What one step needs from the host (passed by value: the general step runs as plain launches).
struct DeepStep {
  const float *rows;  // replay view
  int64_t n_rows;
  int32_t row_stride;
  int32_t idx_mode;          // 0 philox, 1 injected, 2 identity
  const int64_t *idx;        // [B] of THIS step when idx_mode == 1
  const uint8_t *drop_keep;  // [n_hidden][B][H] of THIS step, or null (philox masks)
  float *losses_out;         // [3] of THIS step, or null
  int64_t step;              // total_it before this step
  AdamCoef coef;
};

// One tile of the update kernel.
struct DeepItem {
  int32_t net, layer;  // net < 0: the misc block (losses, log_std)
  int32_t o0, i0;      // tile origin (out-features, in-features), 64 x 64
};

}  // namespace iqlhip
