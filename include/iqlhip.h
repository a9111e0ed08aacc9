/*
 * iqlhip.h -- C-ABI of libiqlhip.so, the MI355X (gfx950) implementation of the
 * IQL-with-preference-reward hot path of ml4ai/iqlpref.
 *
 * The reference has no FFI: its boundary for this path is the Python module
 * surface of algorithms/offline/iql.py.  Each entry point
 * below names the reference lines it replaces ("ref:" = that file).  The Python
 * package iqlpref_amd mirrors the reference classes on top of these calls via
 * ctypes (INTEGRATION.md shows the binding).
 *
 * Conventions
 *   - plain C types only; every pointer documented "device" is a HIP device
 *     pointer owned by the caller (borrowed for the duration stated);
 *   - every function returns 0 on success and a negative iqlhip_status on
 *     failure; iqlhip_last_error() returns a thread-local message;
 *   - all work is enqueued on the hipStream_t passed as `stream` (void*), nothing
 *     synchronises the host unless stated;
 *   - one trainer <-> one stream <-> one host thread (not re-entrant per trainer).
 */
#ifndef IQLHIP_H
#define IQLHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  IQLHIP_OK = 0,
  IQLHIP_ERR_INVALID = -1,     /* bad argument (Python raises ValueError)        */
  IQLHIP_ERR_HIP = -2,         /* HIP runtime error (Python raises RuntimeError) */
  IQLHIP_ERR_UNSUPPORTED = -3, /* shape outside what the kernels are built for   */
  IQLHIP_ERR_NOMEM = -4
} iqlhip_status;

const char *iqlhip_last_error(void);
/* ABI version of this header; bumped on any incompatible change. */
int iqlhip_abi_version(void);
/* 16 hex digits: sha256 prefix of the sources (csrc/ + this header + compiler flags) the
 * library was built from, stamped by iqlpref_amd/build.py.  The Python binding refuses a
 * library whose tag differs from the sources beside it.                                  */
const char *iqlhip_build_tag(void);

/* ------------------------------------------------------------------------ */
/* Shape envelope (everything outside returns IQLHIP_ERR_UNSUPPORTED with a   */
/* message; every YAML under the reference's configs/offline/iql/ fits):      */
/*   trainer   ReLU hidden activations, one width for all hidden layers,       */
/*             n_hidden 1..6, hidden_dim 1..1024, batch_size a multiple of 16, */
/*             state_dim + action_dim <= 128, action_dim <= 32, 2..8 critics,  */
/*             plain Adam (no weight decay / amsgrad), one (beta1, beta2, eps) */
/*             for the three optimisers.  n_hidden = 2 with hidden_dim 64, 128 */
/*             or 256 (the reference's default and every shipped YAML) runs on */
/*             the tuned three-kernel step; every other shape on the general   */
/*             layer-wise step (csrc/iql_deep.hip: same arithmetic, plain      */
/*             launches only, no seed groups);                                 */
/*   MLP fwd   1..8 layers, every width in [1, 1024] (beyond 256: a plain      */
/*             one-wave-per-16-rows variant);                                  */
/*   CVaR      1 <= n_tail <= S <= 2400;                                       */
/*   PT        embd_dim 64, ONE GPT-2 block, num_heads a power of two <= 16,   */
/*             inter_dim a multiple of 256 up to 1024, state_dim + action_dim  */
/*             <= 192, query_length such that the window fits the 160 KiB LDS. */
/* ------------------------------------------------------------------------ */

/* ------------------------------------------------------------------------ */
/* Replay buffer  (ref:164-226 ReplayBuffer)                                 */
/*                                                                           */
/* Storage is ONE row-major fp32 matrix [capacity][row_stride]; row i holds  */
/* transition i as  [ s(S) | a(A) | r | d | pad | s'(S) | pad ]  with s' on a */
/* 16-byte boundary (float offset iqlhip_replay_next_offset), so that one    */
/* sample is one contiguous read of aligned 16-byte pieces instead of five.  */
/* ------------------------------------------------------------------------ */
typedef struct {
  const float *rows;  /* device, [n_rows][row_stride]          */
  int64_t n_rows;     /* min(_size,_pointer) of ref:212         */
  int32_t row_stride; /* floats, multiple of 4                  */
  int32_t state_dim;
  int32_t action_dim;
  uint64_t generation; /* changes whenever the CONTENTS of rows change (a Python ReplayBuffer
                          bumps it on every load).  A call whose view equals the previous
                          call's (rows, n_rows, generation), with on-device indices and no
                          per-step outputs, continues that call: the batch its last step
                          prefetched is used and nothing is re-sent.  0 is a valid value for
                          callers that never rewrite rows in place.                        */
} iqlhip_replay_view;

/* Float offset of s' inside a row: S+A+2 rounded up to a multiple of 4; row stride (floats):
 * that offset + S, rounded up to a multiple of 4.  Views must carry exactly this stride.  */
int32_t iqlhip_replay_next_offset(int32_t state_dim, int32_t action_dim);
int32_t iqlhip_replay_row_stride(int32_t state_dim, int32_t action_dim);

/* ref:193-209 load_d4rl_dataset: interleave the five device arrays
 * obs[n][S], act[n][A], rew[n], next_obs[n][S], done[n] (fp32) into
 * rows[first_row .. first_row+n).                                           */
int iqlhip_replay_pack(float *rows, int32_t row_stride, int32_t state_dim,
                       int32_t action_dim, int64_t first_row, int64_t n,
                       const float *obs, const float *act, const float *rew,
                       const float *next_obs, const float *done, void *stream);

/* The same with the state z-scoring of ref:1438-1448 fused in: s and s' columns are written as
 * (x - mean[c]) / std[c] (fp32, the arithmetic of normalize_states ref:138-139); mean / std are
 * device fp32 [S] (iqlhip_prep_state_stats, or compute_mean_std results uploaded).            */
int iqlhip_replay_pack_normalized(float *rows, int32_t row_stride, int32_t state_dim,
                                  int32_t action_dim, int64_t first_row, int64_t n,
                                  const float *obs, const float *act, const float *rew,
                                  const float *next_obs, const float *done, const float *mean,
                                  const float *std, void *stream);

/* ------------------------------------------------------------------------ */
/* Dataset preparation on the device (the Python loops over the N transitions */
/* that run before training starts).  All pointers are device pointers unless  */
/* stated; terminals / timeouts are uint8 (0 / 1).                             */
/* ------------------------------------------------------------------------ */
/* ref:701-716 (= ref:938-951, 1127-1141, 1236-1253): keep[i] and the episode-step counter
 * ep_steps[i] seen by transition i, for i < n - 1.  timeouts == NULL: `final` is
 * (counter == max_episode_steps - 1) as in the reference.  Integer outputs: exact.          */
int iqlhip_prep_keep_mask(const uint8_t *terminals, const uint8_t *timeouts, int64_t n,
                          int32_t max_episode_steps, int32_t terminate_on_end, uint8_t *keep,
                          int64_t *ep_steps, void *stream);
/* ref:344-360 return_reward_range: an episode ends at a terminal or after max_episode_steps
 * transitions.  *min_ret / *max_ret (HOST doubles) over the complete episodes -- every return
 * is summed in double in transition order, so they equal the reference's bit for bit;
 * trj_lens[i] (device double [n]) = length of the episode transition i belongs to.
 * Synchronises `stream`.  IQLHIP_ERR_INVALID when no episode is complete.                   */
int iqlhip_prep_reward_range(const float *rewards, const uint8_t *terminals, int64_t n,
                             int32_t max_episode_steps, double *trj_lens, double *min_ret,
                             double *max_ret, void *stream);
/* ref:363-401 modify_reward, in place, with numpy's in-place float32 arithmetic:
 *   sub_first 1: r -= min_ret (fp32)   2: r = fp32(double(r) - min_ret / trj_lens[i])
 *   scale      : r /= fp32(max_ret - min_ret); r *= fp32(max_episode_steps)
 *   sub_one    : r -= 1
 * (which of them apply to which env / normalize_reward value is host logic).                */
int iqlhip_prep_modify_reward(float *rewards, int64_t n, const double *trj_lens, int32_t sub_first,
                              int32_t scale, int32_t sub_one, double min_ret, double max_ret,
                              int32_t max_episode_steps, void *stream);
/* ref:132-135 compute_mean_std: mean[c], std[c] + eps of obs[n][state_dim] (state_dim <= 256),
 * accumulated in double in a fixed order (deterministic; differs from numpy's fp32 row-order
 * sums at the 1e-6 level).                                                                  */
int iqlhip_prep_state_stats(const float *obs, int64_t n, int32_t state_dim, double eps, float *mean,
                            float *std, void *stream);

/* ref:211-221 sample.  idx == NULL: indices are drawn on device,
 * idx[b] = philox4x32_10(key=seed, ctr=(b, step, stream 0)).x % n_rows
 * (oracle/philox.py); otherwise idx is a device int64[batch] that is used as
 * given (parity runs).  Outputs are dense device fp32 tensors
 * s[B][S] a[B][A] r[B][1] s2[B][S] d[B][1]; idx_out (optional) receives the
 * indices used.                                                             */
int iqlhip_replay_sample(const iqlhip_replay_view *view, int32_t batch,
                         const int64_t *idx, uint64_t seed, uint64_t step, float *s,
                         float *a, float *r, float *s2, float *d, int64_t *idx_out,
                         void *stream);

/* ------------------------------------------------------------------------ */
/* Trainer  (ref:546-688 ImplicitQLearning; nets ref:408-543)                 */
/* ------------------------------------------------------------------------ */
#define IQLHIP_PREC_FP32 0 /* autocast disabled: exact fp32 MFMA (16x16x4 f32)    */
#define IQLHIP_PREC_BF16 1 /* ref:650 autocast: bf16 MFMA, fp32 accumulate        */

typedef struct {
  int32_t state_dim, action_dim;
  int32_t hidden_dim;    /* 1..1024 (ref:417-449 MLP: one width for all hidden layers) */
  int32_t batch_size;    /* multiple of 16                                        */
  int32_t deterministic; /* 1: DeterministicPolicy (ref:485), 0: Gaussian (ref:452)*/
  int32_t precision;     /* IQLHIP_PREC_*                                         */
  float dropout_p;       /* actor dropout (ref:436-437); < 0 = none               */
  float discount, tau, beta, iql_tau; /* ref:558-562                             */
  double lr_q, lr_v, lr_actor;        /* Adam lr; lr_actor = cosine base lr       */
  double adam_beta1, adam_beta2, adam_eps;
  int64_t cosine_t_max;  /* CosineAnnealingLR T_max = max_steps (ref:571)         */
  uint64_t seed;         /* Philox key for on-device indices and dropout          */
  int32_t n_critics;     /* 0 or 2: TwinQ (ref:517-533); 3..8: E-way critic ensemble,
                            the same MLP E times, q_target = min over E, q_loss =
                            sum(mse)/E (ref:606 generalised; SURVEY 8 "config 5")    */
  int32_t polyak_form;   /* target update: 0 = tp.lerp_(sp, tau) (offline/iql.py:127-129),
                            1 = (1 - tau) tp + tau sp (custom_offline/iql.py:85-87)      */
  int32_t n_hidden;      /* hidden layers per network (ref:458-459, 519, 538: n_hidden);
                            0 = the reference's default 2; 1..6                           */
} iqlhip_trainer_config;

/* Number of fp32 elements of the arenas (n_params, n_target: they include the
 * alignment padding, every tensor starts on a 128-byte line; padding elements
 * are never read or written) and the element offset of every tensor in the
 * parameter arena.  Order of the 2 L (E+2) + 1 offsets, L = n_hidden + 1 Linear
 * layers (25 offsets for TwinQ with two hidden layers): for net in (q1, .., qE, v,
 * actor): W_1[H][in] b_1[H] W_2[H][H] b_2[H] ... W_L[out][H] b_L[out]; then actor
 * log_std[A] (offset -1 when deterministic); unused slots are -1.
 * Torch [out][in] row-major layouts.  The target arena uses the q1..qE part of
 * the same layout.                                                           */
#define IQLHIP_MAX_CRITICS 8
#define IQLHIP_MAX_HIDDEN 6
#define IQLHIP_N_TENSORS (2 * (IQLHIP_MAX_HIDDEN + 1) * (IQLHIP_MAX_CRITICS + 2) + 1)
int iqlhip_arena_layout(const iqlhip_trainer_config *cfg, int64_t offsets[IQLHIP_N_TENSORS],
                        int64_t *n_params, int64_t *n_target);

typedef struct {
  float *params;  /* device fp32 [n_params]  master weights (torch-owned)       */
  float *exp_avg; /* device fp32 [n_params]  Adam m                             */
  float *exp_avg_sq; /* device fp32 [n_params]  Adam v                          */
  float *target;  /* device fp32 [n_target]  q_target (ref:565)                 */
  float *grads;   /* optional device fp32 [n_params]; when non-NULL every step
                     also stores the parameter gradients (tests)               */
} iqlhip_arenas;

typedef struct iqlhip_trainer iqlhip_trainer;

/* Borrows the arenas for the trainer's lifetime; allocates its own workspace
 * (compute-precision weight copies, activations) on the current device.      */
int iqlhip_trainer_create(iqlhip_trainer **out, const iqlhip_trainer_config *cfg,
                          const iqlhip_arenas *arenas);
int iqlhip_trainer_destroy(iqlhip_trainer *t);

/* Which step runs this trainer's shape: 0 = the tuned three-kernel step (csrc/iql_step.hip),
 * 1 = the general layer-wise step (csrc/iql_deep.hip).  See the shape envelope above.        */
int iqlhip_trainer_step_kind(iqlhip_trainer *t, int32_t *kind);

/* Rebuild the compute-precision weight copies from the fp32 masters (call
 * after the arenas were written from outside: init, load_state_dict).        */
int iqlhip_trainer_sync_weights(iqlhip_trainer *t, void *stream);

/* total_it (ref:575,640) = optimiser steps taken = Adam `step` = scheduler
 * last_epoch.                                                                */
int iqlhip_trainer_set_step(iqlhip_trainer *t, int64_t total_it);
int iqlhip_trainer_get_step(iqlhip_trainer *t, int64_t *total_it, double *actor_lr);
int iqlhip_trainer_set_lr(iqlhip_trainer *t, double lr_q, double lr_v, double lr_actor_base);

/* ref:1533-1536 the hot loop body, n_steps times, without host round trips:
 *   sample (ref:211-221) -> ImplicitQLearning.train (ref:639-662).
 * idx: NULL (on-device Philox indices) or device int64[n_steps][batch].
 * dropout_keep: NULL (on-device Philox masks) or device uint8
 *   [n_steps][2][batch][hidden] (1 = keep); ignored without actor dropout.
 * losses_out: NULL or device fp32[n_steps][3] = value_loss, q_loss, actor_loss
 *   of each step (ref:589,607,633).
 * graph_unroll: > 0 replays a captured hipGraph of that many steps per launch;
 *   0: plain kernel launches, three per step, from this call's loop (one seed: the
 *   faster mode, a graph launch costs ~5 us of device time; seed groups: graphs of 50).
 * Asynchronous on `stream`, except that a call never leaves more than
 * IQLHIP_MAX_INFLIGHT kernel dispatches (environment, default 6144, 0 = no bound;
 * 3 per step) queued: beyond that it waits for the oldest third of them.      */
int iqlhip_train_steps(iqlhip_trainer *t, const iqlhip_replay_view *view, int64_t n_steps,
                       const int64_t *idx, const uint8_t *dropout_keep, float *losses_out,
                       int32_t graph_unroll, void *stream);

/* ------------------------------------------------------------------------ */
/* Seed groups: K independent trainers of ONE shape (dims, batch, precision,   */
/* critics, policy kind, dropout on/off) stepped by one launch sequence -- the */
/* three kernels of a step run with gridDim.y = K.  The seeds share nothing;   */
/* each one's arithmetic is bit-identical to stepping it alone.  This is the   */
/* path's sharding unit (one (seed, dataset) run, ensemble_sweeps/launch.sh:   */
/* 12 AGENTS_PER_GPU, :84-94) used inside one GPU.                              */
/* While a trainer is a member, its own iqlhip_train_* calls keep working;     */
/* destroy the group before its members.  views[k], idx[k], dropout_keep[k],   */
/* losses_out[k] belong to member k (idx / dropout_keep / losses_out may be    */
/* NULL as a whole or per member; meaning as in iqlhip_train_steps).           */
/* ------------------------------------------------------------------------ */
#define IQLHIP_MAX_GROUP 16
typedef struct iqlhip_group iqlhip_group;
int iqlhip_group_create(iqlhip_group **out, iqlhip_trainer *const *trainers, int32_t n);
int iqlhip_group_destroy(iqlhip_group *g);
int iqlhip_group_train_steps(iqlhip_group *g, const iqlhip_replay_view *views, int64_t n_steps,
                             const int64_t *const *idx, const uint8_t *const *dropout_keep,
                             float *const *losses_out, int32_t graph_unroll, void *stream);

/* A HIP stream confined to one slice of the compute units: bit i of the CU mask belongs to
 * slice i % n_slices.  The mask enumerates the CUs round-robin over the 8 XCDs, so two slices
 * are the even and the odd XCDs: each sub-group keeps four L2s to itself and all memory channels
 * (slices that split every XCD in half instead measured 180k steps/s against 209k).  Two seed
 * groups stepped on the two slices overlap one group's HBM-bound update with the other's
 * latency-bound forward / backward: 8 seeds as 2 x 4 measured 205k steps/s against 171k as one
 * group of 8 on the whole chip (tools/group_streams.py).  No counterpart in the reference (its
 * AGENTS_PER_GPU processes share the GPU unmanaged).  Destroy with iqlhip_stream_destroy.   */
int iqlhip_stream_create_cu_slice(void **stream, int32_t slice, int32_t n_slices);
int iqlhip_stream_destroy(void *stream);

/* Per-kernel HIP-event timing of a group's launches (as iqlhip_trainer_set/get_timing). */
int iqlhip_group_set_timing(iqlhip_group *g, int32_t enable);
int iqlhip_group_get_timing(iqlhip_group *g, double avg_ms[3], int64_t *n_launches);

/* ref:639-662 train(batch) on an explicit batch of dense device tensors
 * (shapes as iqlhip_replay_sample produces them).                           */
int iqlhip_train_batch(iqlhip_trainer *t, const float *s, const float *a, const float *r,
                       const float *s2, const float *d, const uint8_t *dropout_keep,
                       float *losses_out, void *stream);

/* Forward passes on the live weights (ref:452-543), n rows (any n >= 1):
 *   which = 0: q1..qE -> out[n][E]   (needs a; E = 2 for TwinQ)
 *   which = 1: v     -> out[n]
 *   which = 2: actor mean (tanh)  -> out[n][A]   (eval mode: no dropout)
 *   which = 3: q_target1..E -> out[n][E]   (needs a)                        */
int iqlhip_forward(iqlhip_trainer *t, int32_t which, const float *s, const float *a,
                   int64_t n, float *out, void *stream);

/* ------------------------------------------------------------------------ */
/* Stand-alone fp32 MLP forward (exact f32 MFMA)                              */
/*   - nn.Module.forward() of MLP/TwinQ/ValueFunction/policies outside the   */
/*     autocast region (ref:408-543; eval_actor ref:306-319);                 */
/*   - the Markovian reward model forward of the relabel paths               */
/*     (ref:719-724 MR, ref:986-991 BNN, ref:1176-1178 MR ensemble).          */
/* ------------------------------------------------------------------------ */
#define IQLHIP_MLP_MAX_LAYERS 8
typedef struct {
  int32_t n_layers;                          /* Linear layers, 1..8                */
  int32_t dims[IQLHIP_MLP_MAX_LAYERS + 1];   /* in, hidden..., out; each <= 1024    */
  const float *weights[IQLHIP_MLP_MAX_LAYERS]; /* device fp32                      */
  const float *biases[IQLHIP_MLP_MAX_LAYERS];  /* device fp32 [out]                */
  int32_t w_in_out;   /* 0: W[out][in] (torch nn.Linear); 1: W[in][out] (x @ W)  */
  int32_t hidden_act; /* 0 relu, 1 tanh; 8 + i: entry i of reward_models/q_mlp.py:121-130
                         (cos, tanh, relu, softplus, sin, leaky_relu, swish, none)  */
  int32_t out_act;    /* 0 none, 1 tanh; 8 + i as above                          */
  /* nn.Dropout(p) behind every hidden activation, as a module in train mode applies it
   * (ref:436-437; GaussianPolicy.act in train mode, ref:476-482): dropout_p <= 0 disables.
   * Masks: Philox4x32-10 keyed by dropout_seed, counter (row, dropout_call, unit / 4 | layer << 16,
   * stream 3) -- oracle/philox.py:mlp_dropout_keep; a caller passes a fresh dropout_call per
   * forward.  Kept values are multiplied by float(1 / (1 - p)) (ATen _dropout_impl).            */
  float dropout_p;
  uint32_t dropout_call;
  uint64_t dropout_seed;
} iqlhip_mlp_desc;

/* out[n][out_stride] (first dims[n_layers] columns) = MLP(x[n][x_stride]).  */
int iqlhip_mlp_forward(const iqlhip_mlp_desc *d, const float *x, int64_t n, int32_t x_stride,
                       float *out, int32_t out_stride, void *stream);

/* ------------------------------------------------------------------------ */
/* Ensemble CVaR  (ref:1003-1011 BNN, ref:1185-1187 MR snapshots)             */
/*   out[c] = mean of the n_tail smallest of preds[0..S)[c],  c < N           */
/* preds: device fp32 [S][N] row-major (one row per posterior sample /         */
/* snapshot, filled with iqlhip_mlp_forward using out_stride = 1 into row k). */
/* n_tail = max(1, floor((1 - alpha) S)) (ref:935,1152); S <= 2400.           */
/* ------------------------------------------------------------------------ */
int iqlhip_cvar_tail_mean(const float *preds, int32_t S, int64_t N, int32_t n_tail, float *out,
                          void *stream);

/* ------------------------------------------------------------------------ */
/* Preference-transformer relabel  (ref:1223-1309 qlearning_dataset_pt)        */
/* Architecture: reward_models/pref_transformer.py:170-277 (PT), ops.py:6-117.  */
/* One GPT-2 block, embd_dim 64.  Every pointer is device fp32; "T" = stored   */
/* transposed ([in][out]) relative to the torch / state-dict [out][in] layout. */
/* ------------------------------------------------------------------------ */
typedef struct {
  int32_t state_dim, action_dim;
  int32_t embd_dim;   /* must be 64                                             */
  int32_t num_heads;  /* power of two <= 16                                     */
  int32_t inter_dim;  /* GPT2MLP width, multiple of 64, <= 1024                 */
  int32_t num_layers; /* must be 1                                              */
  int32_t n_temb;     /* rows of timestep_embed (max_episode_steps + 1)         */
  float eps;          /* LayerNorm epsilon                                      */
  const float *state_wT, *state_b;   /* state_linear  [S][64] T, [64]          */
  const float *action_wT, *action_b; /* action_linear [A][64] T, [64]          */
  const float *temb;                 /* timestep_embed.weight [n_temb][64]     */
  const float *sln_w, *sln_b;        /* stacked_layer_norm                     */
  const float *ln0_w, *ln0_b;        /* gpt.layers.0.layer_norm_0              */
  const float *qkv_w, *qkv_b;        /* attention.in_linear [192][64], [192]   */
  const float *attn_out_w, *attn_out_b; /* attention.out_linear.weight [64][64], [64] */
  const float *ln1_w, *ln1_b;        /* layer_norm_1                           */
  const float *mlp_in_w, *mlp_in_b;   /* mlp.in_linear.weight  [I][64], [I]   */
  const float *mlp_out_w, *mlp_out_b; /* mlp.out_linear.weight [64][I], [64]  */
  const float *lnf_w, *lnf_b;        /* gpt.layer_norm                         */
  const float *pref_w_last;          /* LAST row of pref_linear.weight [64]    */
  float pref_b_last;                 /* last element of pref_linear.bias       */
} iqlhip_pt_weights;

/* out[w] = value[:, 0, -1, 0] (ref:1301) of the window of win_len[w] consecutive
 * transitions obs/act[win_start[w] .. +win_len[w]), right-aligned in a
 * query_length window (ref:1269-1292); the timestep of the window's k-th transition is
 * win_t0[w] + k.  win_t0 == NULL: 0 for every window (ref:1281,1291: timesteps 0..len-1);
 * custom_offline/iql.py:183-211 passes the true episode step (win_t0[w] + len <= n_temb is the
 * caller's to guarantee).  Because attention is causal, the value at position i of ONE forward
 * over a window (custom_offline:172-192) equals the last-token value of its prefix of length
 * i + 1: per-position values are windows of growing length.
 * obs [n_rows][S], act [n_rows][A], win_start int64 [n_win], win_len int32
 * [n_win] (1 <= len <= query_length), win_t0 int32 [n_win] or NULL, all device.
 * Precondition (the window arrays live on the device, the entry point cannot read them):
 * win_start[w] >= 0, win_start[w] + win_len[w] <= n_rows, win_t0[w] >= 0,
 * win_t0[w] + win_len[w] <= n_temb.  The kernel clamps every window into these bounds, so a
 * window that violates them yields a value for the clamped window, never an out-of-bounds read. */
int iqlhip_pt_relabel(const iqlhip_pt_weights *w, const float *obs, const float *act, int64_t n_rows,
                      const int64_t *win_start, const int32_t *win_len, const int32_t *win_t0,
                      int64_t n_win, int32_t query_length, float *out, void *stream);

/* Algorithmic traffic and work of one step for this configuration
 * (SURVEY.md section 8d): bytes = 4B(2S+A+2) + 32 P_train + 8 P_q.          */
int iqlhip_step_cost(const iqlhip_trainer_config *cfg, double *bytes, double *flops);

/* HIP-event timing of the kernels of the most recent iqlhip_train_steps call
 * made with timing enabled (bench.py roofline leg).  enable != 0 brackets
 * every launch of the dominant kernel with events on `stream`.              */
int iqlhip_trainer_set_timing(iqlhip_trainer *t, int32_t enable);
/* avg_ms[3] = mean duration of the forward / backward / update kernels, with
 * the cost of an empty event pair (measured in the same pass) subtracted.    */
int iqlhip_trainer_get_timing(iqlhip_trainer *t, double avg_ms[3], int64_t *n_launches);

#ifdef __cplusplus
}
#endif
#endif /* IQLHIP_H */
