/* The drop-in boundary is a C-ABI library: this program uses it from plain C (no Python, no
 * torch), the way the binding of INTEGRATION.md section 2 does.  It fills the arenas and a
 * small replay buffer with a fixed integer recurrence, runs 12 steps (on-device Philox batch
 * indices, hipGraph replay of 4) and prints the losses as hex floats; tests/test_gpu_c_abi.py
 * builds the same state through the Python binding and expects the same bits.
 *   gcc abi_smoke.c -I../../include -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ \
 *       -L../../iqlpref_amd -liqlhip -L/opt/rocm/lib -lamdhip64 -lm                       */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "iqlhip.h"

#define CHECK(x)                                                        \
  do {                                                                  \
    int rc_ = (x);                                                      \
    if (rc_) {                                                          \
      fprintf(stderr, "%s -> %d: %s\n", #x, rc_, iqlhip_last_error()); \
      return 1;                                                         \
    }                                                                   \
  } while (0)
#define HIP(x)                                                  \
  do {                                                          \
    hipError_t e_ = (x);                                        \
    if (e_ != hipSuccess) {                                     \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));   \
      return 1;                                                 \
    }                                                           \
  } while (0)

/* same recurrence as tests/test_gpu_c_abi.py:lcg_fill */
static void lcg_fill(float *dst, int64_t n, uint32_t seed, float scale) {
  uint32_t s = seed;
  for (int64_t i = 0; i < n; ++i) {
    s = s * 1664525u + 1013904223u;
    dst[i] = ((float)(s >> 8) * (1.0f / 16777216.0f) - 0.5f) * scale;
  }
}

/* abi_smoke [n_hidden hidden_dim]: default 2 x 64 (the tuned step); any other depth / width runs on the general
 * layer-wise step behind the same entry points */
int main(int argc, char **argv) {
  enum { S = 11, A = 3, B = 32, N = 500, STEPS = 12 };
  const int NH = argc > 2 ? atoi(argv[1]) : 2, H = argc > 2 ? atoi(argv[2]) : 64, NL = NH + 1;
  iqlhip_trainer_config cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.state_dim = S, cfg.action_dim = A, cfg.hidden_dim = H, cfg.batch_size = B, cfg.n_hidden = NH;
  cfg.deterministic = 0, cfg.precision = IQLHIP_PREC_FP32, cfg.dropout_p = -1.0f;
  cfg.discount = 0.99f, cfg.tau = 0.005f, cfg.beta = 3.0f, cfg.iql_tau = 0.7f;
  cfg.lr_q = cfg.lr_v = cfg.lr_actor = 3e-4;
  cfg.adam_beta1 = 0.9, cfg.adam_beta2 = 0.999, cfg.adam_eps = 1e-8;
  cfg.cosine_t_max = 1000, cfg.seed = 2024, cfg.n_critics = 0;
  if (iqlhip_abi_version() != 6) return 2;

  int64_t off[IQLHIP_N_TENSORS], n_params = 0, n_target = 0;
  CHECK(iqlhip_arena_layout(&cfg, off, &n_params, &n_target));
  float *hp = (float *)calloc((size_t)n_params, sizeof(float));
  /* tensor k gets its own stream of the recurrence (sizes from consecutive offsets) */
  const int in_dim[4] = {S + A, S + A, S, S}, out_dim[4] = {1, 1, 1, A};
  for (int n = 0; n < 4; ++n)
    for (int l = 0; l < NL; ++l) { /* W_l [rows][cols], b_l [rows]: 2 NL offsets per network */
      const int64_t rows = l == NL - 1 ? out_dim[n] : H, cols = l == 0 ? in_dim[n] : H;
      const int k = (n * NL + l) * 2;
      lcg_fill(hp + off[k], rows * cols, 1000u + (uint32_t)k, 0.25f);
      lcg_fill(hp + off[k + 1], rows, 1000u + (uint32_t)k + 1u, 0.25f);
    }
  /* log_std stays 0 (the reference's init) */
  iqlhip_arenas ar;
  memset(&ar, 0, sizeof(ar));
  HIP(hipMalloc((void **)&ar.params, (size_t)n_params * 4));
  HIP(hipMalloc((void **)&ar.exp_avg, (size_t)n_params * 4));
  HIP(hipMalloc((void **)&ar.exp_avg_sq, (size_t)n_params * 4));
  HIP(hipMalloc((void **)&ar.target, (size_t)n_target * 4));
  HIP(hipMemcpy(ar.params, hp, (size_t)n_params * 4, hipMemcpyHostToDevice));
  HIP(hipMemset(ar.exp_avg, 0, (size_t)n_params * 4));
  HIP(hipMemset(ar.exp_avg_sq, 0, (size_t)n_params * 4));
  HIP(hipMemcpy(ar.target, hp, (size_t)n_target * 4, hipMemcpyHostToDevice)); /* deepcopy(qf), ref:565 */

  /* replay buffer */
  float *h_obs = malloc(sizeof(float) * N * S), *h_act = malloc(sizeof(float) * N * A);
  float *h_rew = malloc(sizeof(float) * N), *h_nobs = malloc(sizeof(float) * N * S), *h_done = malloc(sizeof(float) * N);
  lcg_fill(h_obs, N * S, 1u, 2.0f), lcg_fill(h_act, N * A, 2u, 2.0f), lcg_fill(h_rew, N, 3u, 1.0f);
  lcg_fill(h_nobs, N * S, 4u, 2.0f);
  for (int i = 0; i < N; ++i) h_done[i] = (i % 37 == 0) ? 1.0f : 0.0f;
  float *d_obs, *d_act, *d_rew, *d_nobs, *d_done, *rows, *d_losses;
  HIP(hipMalloc((void **)&d_obs, sizeof(float) * N * S));
  HIP(hipMalloc((void **)&d_act, sizeof(float) * N * A));
  HIP(hipMalloc((void **)&d_rew, sizeof(float) * N));
  HIP(hipMalloc((void **)&d_nobs, sizeof(float) * N * S));
  HIP(hipMalloc((void **)&d_done, sizeof(float) * N));
  HIP(hipMemcpy(d_obs, h_obs, sizeof(float) * N * S, hipMemcpyHostToDevice));
  HIP(hipMemcpy(d_act, h_act, sizeof(float) * N * A, hipMemcpyHostToDevice));
  HIP(hipMemcpy(d_rew, h_rew, sizeof(float) * N, hipMemcpyHostToDevice));
  HIP(hipMemcpy(d_nobs, h_nobs, sizeof(float) * N * S, hipMemcpyHostToDevice));
  HIP(hipMemcpy(d_done, h_done, sizeof(float) * N, hipMemcpyHostToDevice));
  const int32_t stride = iqlhip_replay_row_stride(S, A);
  HIP(hipMalloc((void **)&rows, sizeof(float) * (size_t)N * stride));
  HIP(hipMemset(rows, 0, sizeof(float) * (size_t)N * stride));
  CHECK(iqlhip_replay_pack(rows, stride, S, A, 0, N, d_obs, d_act, d_rew, d_nobs, d_done, NULL));

  iqlhip_trainer *t = NULL;
  CHECK(iqlhip_trainer_create(&t, &cfg, &ar));
  CHECK(iqlhip_trainer_sync_weights(t, NULL));
  iqlhip_replay_view view = {rows, N, stride, S, A};
  HIP(hipMalloc((void **)&d_losses, sizeof(float) * STEPS * 3));
  CHECK(iqlhip_train_steps(t, &view, STEPS, NULL, NULL, d_losses, 4, NULL));
  HIP(hipDeviceSynchronize());
  float losses[STEPS * 3];
  HIP(hipMemcpy(losses, d_losses, sizeof(losses), hipMemcpyDeviceToHost));
  int64_t it = 0;
  double lr = 0;
  CHECK(iqlhip_trainer_get_step(t, &it, &lr));
  int32_t kind = -1;
  CHECK(iqlhip_trainer_step_kind(t, &kind));
  printf("total_it %lld step_kind %d\n", (long long)it, kind);
  for (int i = 0; i < STEPS; ++i) printf("%a %a %a\n", losses[3 * i], losses[3 * i + 1], losses[3 * i + 2]);
  CHECK(iqlhip_trainer_destroy(t));
  return 0;
}
