// Arithmetic of the IQL step shared by the tuned three-Linear step (iql_step.hip) and the general
// layer-wise step (iql_deep.hip): the loss terms and their derivatives (ref:581-637), the Adam
// update (torch.optim.Adam, _single_tensor_adam), the Polyak target update (ref:127-129).  One
// definition, so that both paths round in the same places.
#pragma once
#include <math.h>

#include "common.h"
#include "iql_step.h"

namespace iqlhip {

// ------------------------------------------------------------------------
// d(loss)/d(out) of output j of batch row b for network `net`, plus the logged
// loss term and (Gaussian actor) the d(loss)/d(std) term  (ref:581-637).
// ------------------------------------------------------------------------
struct LossIn {
  float qt[MAX_CRITICS];  // target critics (entries beyond E repeat the first); reduced in loss_terms
  float vv, mean, act, ls, nv, qv, rew, done;
};

template <bool BF16, class Desc>
__device__ __forceinline__ void loss_terms(const Desc &D, int net, const LossIn &x, float fB,
                                           float &dz3, float &lterm, float &gstd) {
  using P = Prec<BF16>;
  // Branch-free on purpose: all three variants are evaluated and one is selected.  With
  // `if (net == actor) {...}` the compiler sinks the loads of mean / action / log_std into the
  // branch, behind the weight stream -- one more memory round trip before the loss is known.
  float qtm = x.qt[0];
#pragma unroll
  for (int e = 1; e < MAX_CRITICS; ++e) qtm = fminf(qtm, x.qt[e]);
  const float vv = x.vv;
  const float adv = P::round(qtm - vv);                                        // ref:583-587
  // ---- actor (AWR, ref:615-637) ----
  const float mean = x.mean, act = x.act;
  const float eadv = fminf(P::round(expf(P::round(D.beta * adv))), 100.f);     // ref:622
  const float gbc = eadv / fB;
  const float lsc = fminf(fmaxf(x.ls, -20.f), 2.f);
  const float sd = expf(lsc), var = sd * sd, zg = act - mean;
  // -log_prob (torch.distributions.Normal.log_prob)
  const float bc_g = (zg * zg) / (2.f * var) + logf(sd) + 0.9189385332046727f;
  const float gm_g = P::round(-gbc * (zg / var));
  const float gstd_g = gbc * (-(zg * zg) / (var * sd) + 1.f / sd);
  const float zd = mean - act;  // ref:629
  const float bc_d = zd * zd;
  const float gm_d = P::round(gbc * 2.f * zd);
  const bool det = D.deterministic != 0;
  const float bc = det ? bc_d : bc_g, gm = det ? gm_d : gm_g;
  const float lterm_a = eadv * bc;
  const float dz3_a = P::round(gm * (1.f - mean * mean));  // tanh backward
  // ---- V (expectile, ref:404-405, 581-593) ----
  const float w = fabsf(D.iql_tau - (adv < 0.f ? 1.f : 0.f));
  const float lterm_v = w * P::round(adv * adv);
  float g;
  if constexpr (BF16)
    g = rbf(rbf(w / fB) * (2.f * adv));
  else
    g = (w / fB) * (2.f * adv);
  const float dz3_v = -g;  // adv = target_q - v
  // ---- critics (TD, ref:595-613) ----
  const float target = x.rew + (1.f - x.done) * D.discount * x.nv;  // ref:604
  const float diff = x.qv - target;
  const float lterm_q = diff * diff;
  // q_loss = sum_e mse(q_e, t) / E (ref:606): the division hands 1/E to each mse term, whose
  // backward is (2/B) * (q - t) * grad_out; for E = 2 and B a power of two = (q - t) / B
  const float dz3_q = P::round((D.two_over_B * diff) * D.inv_E);
  const bool is_a = net == D.net_a, is_v = net == D.net_v;
  dz3 = is_a ? dz3_a : (is_v ? dz3_v : dz3_q);
  lterm = is_a ? lterm_a : (is_v ? lterm_v : lterm_q);
  gstd = (is_a && !det) ? gstd_g : 0.f;
}

// Polyak update of a target weight t towards the new weight p.  Two forms, different rounding:
//   offline/iql.py:127-129        tp.lerp_(sp, tau)                       t + tau (p - t)
//   custom_offline/iql.py:85-87   copy_((1 - tau) * tp + tau * sp)        (1 - tau) t + tau p
// The roundings of these two functions are spelled out and contraction is off inside them: with
// the default -ffp-contract=fast the compiler chose DIFFERENT fused forms for the same source line
// in two call sites (v b2 + ((1 - b2) g) g became fma(b2, v, ((1 - b2) g) g) in one kernel path and
// fma((1 - b2) g, g, v b2) in another: one ulp apart), and every code path that updates a parameter
// must give the same bits (seeds in a group launch == the seed alone).
template <class Desc>
__device__ __forceinline__ float polyak(const Desc &D, float t, float p) {
#pragma clang fp contract(off)
  // convex form: two rounded products and a sum, as the tensor expression (1 - tau) * tp + tau * sp
  // evaluates; lerp form: one fused multiply-add over the rounded difference (ATen's lerp kernel)
  return D.polyak_convex ? (D.one_m_tau * t) + (D.tau * p) : __builtin_fmaf(D.tau, p - t, t);
}

// A/B on one box (round 3, r4m): one seed 64.1k -> 65.6k steps/s, 8 seeds per launch 175.0k -> 177.4k;
// every parity test unchanged at its tolerance.  -DIQL_ADAM_FAST=0 builds the IEEE form everywhere.
// FAST is the bf16 step's form only: precision = fp32 (the parity mode) keeps torch's arithmetic --
// _single_tensor_adam's correctly rounded sqrt and two divisions, operation for operation.
#ifndef IQL_ADAM_FAST
#define IQL_ADAM_FAST 1
#endif
template <bool FAST>
__device__ __forceinline__ void adam_apply(float &p, float &m, float &v, float g, const AdamCoef &c,
                                           float neg_step) {
#pragma clang fp contract(off)
  m = __builtin_fmaf(g - m, c.one_m_b1, m);                 // exp_avg.lerp_(grad, 1 - beta1)
  v = __builtin_fmaf(c.b2, v, (c.one_m_b2 * g) * g);        // mul_(beta2).addcmul_(g, g, 1 - beta2)
  if constexpr (FAST) {
    // denom = sqrt(v) / sqrt(bc2) + eps and m / denom on the hardware's 1-ulp v_sqrt_f32 / v_rcp_f32
    // and a precomputed reciprocal instead of two correctly rounded divisions and a correctly rounded
    // square root (~10 vector instructions each, three quarters of the Adam pass): the step differs
    // from the IEEE form by <= ~3e-7 of itself, i.e. <= 1e-10 absolute at lr = 3e-4
    const float denom = __builtin_fmaf(__builtin_amdgcn_sqrtf(v), c.inv_bc2_sqrt, c.eps);
    p = __builtin_fmaf(neg_step, m * __builtin_amdgcn_rcpf(denom), p);
  } else {
    const float denom = sqrtf(v) / c.bc2_sqrt + c.eps;
    p = __builtin_fmaf(neg_step, m / denom, p);             // addcdiv_(exp_avg, denom, -step_size)
  }
}

// beta^t by binary exponentiation (t <= 2^31): a few ulp, ~60 double multiplies
__host__ __device__ inline double ipow(double b, int64_t t) {
  double r = 1.0;
  while (t > 0) {
    if (t & 1) r *= b;
    b *= b;
    t >>= 1;
  }
  return r;
}

// Adam bias corrections and the cosine actor lr of the step whose 1-based count is t1 (t1 - 1
// scheduler steps have been taken, ref:636-637; CosineAnnealingLR closed form).
__host__ __device__ inline AdamCoef make_adam_coef(double beta1, double beta2, double eps, double lr_q, double lr_v,
                                                   double lr_a_base, int64_t t_max, int64_t t1) {
  const double bc1 = 1.0 - ipow(beta1, t1);
  const double bc2 = 1.0 - ipow(beta2, t1);
  const double lr_a = lr_a_base * (1.0 + cos(M_PI * (double)(t1 - 1) / (double)t_max)) * 0.5;
  AdamCoef c;
  c.one_m_b1 = (float)(1.0 - beta1);
  c.b2 = (float)beta2;
  c.one_m_b2 = (float)(1.0 - beta2);
  c.neg_step[0] = (float)(-(lr_q / bc1));
  c.neg_step[1] = (float)(-(lr_v / bc1));
  c.neg_step[2] = (float)(-(lr_a / bc1));
  c.bc2_sqrt = (float)sqrt(bc2);
  c.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2)), c.pad_ = 0.f;
  c.eps = (float)eps;
  return c;
}

}  // namespace iqlhip
