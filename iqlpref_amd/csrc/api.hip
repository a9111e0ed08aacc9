// Host side of libiqlhip.so: the extern "C" entry points of include/iqlhip.h,
// trainer workspace management, hipGraph capture of the step sequence.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/iqlhip.h"
#include "common.h"
#include "iql_deep.h"
#include "iql_step.h"
#include "step_math.h"

namespace iqlhip {
// the general layer-wise step (iql_deep.hip)
struct DeepTrainer;
bool deep_shape_ok(const iqlhip_trainer_config &, int n_hidden, const char **why);
hipError_t deep_create(DeepTrainer **, const iqlhip_trainer_config &, int n_hidden, const iqlhip_arenas &,
                       const int64_t *offsets);
void deep_destroy(DeepTrainer *);
hipError_t deep_sync_weights(DeepTrainer *, hipStream_t);
hipError_t deep_step(DeepTrainer *, const DeepStep &, hipStream_t, hipEvent_t *ev);
hipError_t deep_infer(DeepTrainer *, int which, const float *s, const float *a, int64_t n, float *out, hipStream_t);
int layer2_parts(int H);
hipError_t launch_forward(bool, const TrainerDesc &, const TrainerDesc *, const DevArgs *, const DevCtr *,
                          int n_seeds, hipStream_t);
hipError_t launch_backward(bool, const TrainerDesc &, const TrainerDesc *, const DevArgs *, DevCtr *,
                           int n_seeds, hipStream_t);
hipError_t launch_stage(bool, const TrainerDesc &, const TrainerDesc *, const DevArgs *, const DevCtr *,
                        int n_seeds, hipStream_t);
int strip_rows();
int update_lds_floats();
hipError_t launch_update(bool, const TrainerDesc *, const DevArgs *, DevCtr *, const UpdItem *, int,
                         int n_seeds, hipStream_t);
hipError_t launch_sync_weights(bool, const TrainerDesc *, hipStream_t);
hipError_t prepare_step_kernels();

hipError_t launch_infer(bool, const TrainerDesc &, const TrainerDesc *, const FwdNet &, const float *,
                        const float *, int64_t, float *, int, hipStream_t);

hipError_t launch_pack(float *rows, int stride, int S, int A, int64_t first, int64_t n, const float *obs,
                       const float *act, const float *rew, const float *nxt, const float *done,
                       hipStream_t st);
hipError_t launch_sample(const iqlhip_replay_view &v, int batch, const int64_t *idx, uint64_t seed,
                         uint64_t step, float *s, float *a, float *r, float *s2, float *d,
                         int64_t *idx_out, hipStream_t st);
hipError_t launch_mlp_f32(const iqlhip_mlp_desc &d, const float *x, int64_t n, int x_stride, float *out,
                          int out_stride, hipStream_t st);
hipError_t launch_cvar(const float *preds, int S, int64_t N, int n_tail, float *out, hipStream_t st);
hipError_t launch_keep_steps(const uint8_t *term, const uint8_t *tmo, int64_t n, int64_t M, int toe,
                             uint8_t *keep, int64_t *ep_steps, hipStream_t st);
hipError_t launch_reward_range(const float *rew, const uint8_t *term, int64_t n, int64_t M, double *trj_lens,
                               double *out3, hipStream_t st);
hipError_t launch_modify_reward(float *rew, int64_t n, const double *trj_lens, int sub_first, int scale,
                                int sub_one, double min_ret, double range, float steps, hipStream_t st);
hipError_t launch_state_stats(const float *x, int64_t n, int S, double eps, float *mean_f, float *std_f,
                              hipStream_t st);
hipError_t launch_pack_norm(float *rows, int stride, int S, int A, int64_t first, int64_t n, const float *obs,
                            const float *act, const float *rew, const float *nxt, const float *done,
                            const float *mean, const float *sd, hipStream_t st);
size_t pt_smem_bytes(const iqlhip_pt_weights &W, int ql);
hipError_t launch_pt(const iqlhip_pt_weights &W, const float *obs, const float *act, int64_t n_rows,
                     const int64_t *win_start, const int32_t *win_len, const int32_t *win_t0, int64_t n_win,
                     int ql, float *out, hipStream_t st);
}  // namespace iqlhip

using namespace iqlhip;

// ------------------------------------------------------------------ errors --
static thread_local char g_err[512] = "";
static int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess)                                                                     \
      return fail(IQLHIP_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                  __LINE__);                                                                  \
  } while (0)

extern "C" const char *iqlhip_last_error(void) { return g_err; }
// One capture-only stream per host thread and device, kept for the life of the process (stream
// capture is thread-local, nothing ever executes on it).  A stream per trainer / group, created and
// destroyed with its owner, churned the runtime's hardware queues: CU-slice streams created after
// such a destroy no longer ran side by side (both halves of the chip serialised).
static hipError_t capture_stream(hipStream_t *out) {
  static thread_local hipStream_t pool[64] = {};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  if (!pool[dev]) {
    e = hipStreamCreateWithFlags(&pool[dev], hipStreamNonBlocking);
    if (e != hipSuccess) return e;
  }
  *out = pool[dev];
  return hipSuccess;
}

extern "C" int iqlhip_abi_version(void) { return 6; }
// sha256 prefix of csrc/* + include/iqlhip.h, stamped by iqlpref_amd/build.py: the Python side
// refuses a library whose tag does not match the sources it sits beside
#ifndef IQLHIP_BUILD_TAG
#define IQLHIP_BUILD_TAG "untagged"
#endif
static const char g_build_tag[] = "IQLHIP_BUILD_TAG=" IQLHIP_BUILD_TAG;  // marker: build.py reads it from the file
extern "C" const char *iqlhip_build_tag(void) { return g_build_tag + sizeof("IQLHIP_BUILD_TAG=") - 1; }

// ------------------------------------------------------------------ replay --
// Packed replay row: [ s(S) | a(A) | r | d | pad | s'(S) | pad ], s' on a 16-byte boundary
extern "C" int32_t iqlhip_replay_next_offset(int32_t S, int32_t A) { return round_up(S + A + 2, 4); }
extern "C" int32_t iqlhip_replay_row_stride(int32_t S, int32_t A) {
  return round_up(iqlhip_replay_next_offset(S, A) + S, 4);
}

extern "C" int iqlhip_replay_pack(float *rows, int32_t row_stride, int32_t S, int32_t A, int64_t first_row,
                                  int64_t n, const float *obs, const float *act, const float *rew,
                                  const float *next_obs, const float *done, void *stream) {
  if (!rows || !obs || !act || !rew || !next_obs || !done) return fail(IQLHIP_ERR_INVALID, "null pointer");
  if (S <= 0 || A <= 0 || n < 0 || first_row < 0 || row_stride < iqlhip_replay_row_stride(S, A) || (row_stride & 3))
    return fail(IQLHIP_ERR_INVALID, "bad replay geometry S=%d A=%d stride=%d", S, A, row_stride);
  if (n == 0) return 0;
  HIP_TRY(launch_pack(rows, row_stride, S, A, first_row, n, obs, act, rew, next_obs, done,
                      (hipStream_t)stream));
  return 0;
}

extern "C" int iqlhip_replay_pack_normalized(float *rows, int32_t row_stride, int32_t S, int32_t A,
                                             int64_t first_row, int64_t n, const float *obs, const float *act,
                                             const float *rew, const float *next_obs, const float *done,
                                             const float *mean, const float *std, void *stream) {
  if (!rows || !obs || !act || !rew || !next_obs || !done || !mean || !std)
    return fail(IQLHIP_ERR_INVALID, "null pointer");
  if (S <= 0 || A <= 0 || n < 0 || first_row < 0 || row_stride < iqlhip_replay_row_stride(S, A) || (row_stride & 3))
    return fail(IQLHIP_ERR_INVALID, "bad replay geometry S=%d A=%d stride=%d", S, A, row_stride);
  if (n == 0) return 0;
  HIP_TRY(launch_pack_norm(rows, row_stride, S, A, first_row, n, obs, act, rew, next_obs, done, mean, std,
                           (hipStream_t)stream));
  return 0;
}

// ------------------------------------------------------- dataset preparation --
extern "C" int iqlhip_prep_keep_mask(const uint8_t *terminals, const uint8_t *timeouts, int64_t n,
                                     int32_t max_episode_steps, int32_t terminate_on_end, uint8_t *keep,
                                     int64_t *ep_steps, void *stream) {
  if (!terminals || !keep || !ep_steps) return fail(IQLHIP_ERR_INVALID, "null pointer");
  if (n < 1) return fail(IQLHIP_ERR_INVALID, "n must be >= 1");
  if (!timeouts && max_episode_steps < 1) return fail(IQLHIP_ERR_INVALID, "max_episode_steps must be >= 1");
  HIP_TRY(launch_keep_steps(terminals, timeouts, n, max_episode_steps, terminate_on_end != 0, keep, ep_steps,
                            (hipStream_t)stream));
  return 0;
}

extern "C" int iqlhip_prep_reward_range(const float *rewards, const uint8_t *terminals, int64_t n,
                                        int32_t max_episode_steps, double *trj_lens, double *min_ret,
                                        double *max_ret, void *stream) {
  if (!rewards || !terminals || !trj_lens || !min_ret || !max_ret) return fail(IQLHIP_ERR_INVALID, "null pointer");
  if (n < 1 || max_episode_steps < 1) return fail(IQLHIP_ERR_INVALID, "n and max_episode_steps must be >= 1");
  double out3[3];
  HIP_TRY(launch_reward_range(rewards, terminals, n, max_episode_steps, trj_lens, out3, (hipStream_t)stream));
  if (out3[2] == 0.0) return fail(IQLHIP_ERR_INVALID, "dataset holds no complete episode");
  *min_ret = out3[0], *max_ret = out3[1];
  return 0;
}

extern "C" int iqlhip_prep_modify_reward(float *rewards, int64_t n, const double *trj_lens, int32_t sub_first,
                                         int32_t scale, int32_t sub_one, double min_ret, double max_ret,
                                         int32_t max_episode_steps, void *stream) {
  if (!rewards) return fail(IQLHIP_ERR_INVALID, "null pointer");
  if (sub_first < 0 || sub_first > 2) return fail(IQLHIP_ERR_INVALID, "sub_first must be 0, 1 or 2");
  if (sub_first == 2 && !trj_lens) return fail(IQLHIP_ERR_INVALID, "sub_first = 2 needs trj_lens");
  if (n <= 0) return 0;
  HIP_TRY(launch_modify_reward(rewards, n, trj_lens, sub_first, scale != 0, sub_one != 0, min_ret,
                               max_ret - min_ret, (float)max_episode_steps, (hipStream_t)stream));
  return 0;
}

extern "C" int iqlhip_prep_state_stats(const float *obs, int64_t n, int32_t state_dim, double eps, float *mean,
                                       float *std, void *stream) {
  if (!obs || !mean || !std) return fail(IQLHIP_ERR_INVALID, "null pointer");
  if (n < 1 || state_dim < 1 || state_dim > 256) return fail(IQLHIP_ERR_UNSUPPORTED, "n >= 1, 1 <= state_dim <= 256");
  HIP_TRY(launch_state_stats(obs, n, state_dim, eps, mean, std, (hipStream_t)stream));
  return 0;
}

extern "C" int iqlhip_replay_sample(const iqlhip_replay_view *view, int32_t batch, const int64_t *idx,
                                    uint64_t seed, uint64_t step, float *s, float *a, float *r, float *s2,
                                    float *d, int64_t *idx_out, void *stream) {
  if (!view || !view->rows || !s || !a || !r || !s2 || !d) return fail(IQLHIP_ERR_INVALID, "null pointer");
  if (batch <= 0) return fail(IQLHIP_ERR_INVALID, "batch must be positive");
  if (view->n_rows <= 0) return fail(IQLHIP_ERR_INVALID, "cannot sample from an empty replay buffer");
  HIP_TRY(launch_sample(*view, batch, idx, seed, step, s, a, r, s2, d, idx_out, (hipStream_t)stream));
  return 0;
}

// ----------------------------------------------------------------- trainer --
static void group_invalidate(struct iqlhip_group *g);

// Bound on the work one caller can leave queued on a stream.  A train_steps call of n steps is
// 3n kernel dispatches (graph nodes or eager launches) and returns without a host wait; a caller
// that loops over such calls (bench.py's long regions, a seed-group sweep) used to be able to
// queue 60,000+ dispatches.  Every WINDOW dispatches the call records an event and, before it
// queues more, waits for the event of TWO windows ago: at most ~3 windows are ever outstanding,
// the device never runs dry (a window is >= 1.5 ms of work), and the host blocks only when it is
// that far ahead.  IQLHIP_MAX_INFLIGHT (dispatches) overrides the window total; 0 disables.
struct Throttle {
  static constexpr int NEV = 2;
  hipEvent_t ev[NEV] = {};
  bool used[NEV] = {};
  int head = 0;
  int64_t since = 0;  // dispatches queued since the last recorded event
  static int64_t window() {
    static const int64_t w = [] {
      const char *e = getenv("IQLHIP_MAX_INFLIGHT");
      const int64_t total = e ? atoll(e) : 6144;
      return total <= 0 ? (int64_t)0 : (total / 3 > 0 ? total / 3 : (int64_t)1);
    }();
    return w;
  }
  // call after `n` dispatches have been queued on `st`
  hipError_t queued(int64_t n, hipStream_t st) {
    const int64_t w = window();
    if (w == 0) return hipSuccess;
    since += n;
    if (since < w) return hipSuccess;
    since = 0;
    hipError_t e;
    if (!ev[head] && (e = hipEventCreateWithFlags(&ev[head], hipEventDisableTiming)) != hipSuccess) return e;
    if (used[head] && (e = hipEventSynchronize(ev[head])) != hipSuccess) return e;  // two windows ago
    if ((e = hipEventRecord(ev[head], st)) != hipSuccess) return e;
    used[head] = true;
    head = (head + 1) % NEV;
    return hipSuccess;
  }
  void destroy() {
    for (auto &e : ev)
      if (e) (void)hipEventDestroy(e), e = nullptr;
  }
};

struct iqlhip_trainer {
  iqlhip_trainer_config cfg;
  iqlhip_arenas arenas;
  // shapes outside the tuned step's (n_hidden != 2 or another width): the general layer-wise step;
  // of the members below only cfg, the learning rates, total_it, the timing events, the throttle and
  // batch_rows are in use then
  DeepTrainer *deep = nullptr;
  TrainerDesc D;
  bool bf16;
  void *ws = nullptr;
  size_t ws_bytes = 0;
  TrainerDesc *ddesc = nullptr;  // device copy of D (kernels read it through L2, not the kernarg segment)
  DevArgs *dargs = nullptr;
  DevCtr *dctr = nullptr;
  UpdItem *ditems = nullptr;
  int n_items = 0;      // of the table `ditems` points to (the group's while the trainer is a member of one)
  int own_n_items = 0;  // of the trainer's own table
  // host copy of the update kernel's work items, per trained net: first the items of the lower half of
  // every layer, then the upper half (n_half0 of them in the first part) -- tables are dealt from these
  std::vector<UpdItem> net_items[MAX_TRAIN];
  int n_half0[MAX_TRAIN] = {};
  // the slots of this trainer's own workspace; ddesc / dargs / dctr / ditems point into a
  // group's contiguous arrays while the trainer is a member of one (iqlhip_group_create)
  TrainerDesc *own_ddesc = nullptr;
  DevArgs *own_dargs = nullptr;
  DevCtr *own_dctr = nullptr;
  UpdItem *own_ditems = nullptr;
  struct iqlhip_group *group = nullptr;

  float *batch_rows = nullptr;  // [B][stride] staging for iqlhip_train_batch
  // DevArgs travel through a small ring of pinned host slots (a pageable source makes
  // hipMemcpyAsync host-blocking, which serialised the streams of a SeedGroup); a slot is
  // reused only after the copy that read it has completed (its event)
  static constexpr int ARG_RING = 8;
  DevArgs dev_args;             // what the device copy holds ...
  bool dev_args_valid = false;  // ... and whether the batch of step total_it is already staged for it
  DevArgs *harg[ARG_RING] = {};
  hipEvent_t harg_ev[ARG_RING] = {};
  bool harg_used[ARG_RING] = {};
  int harg_head = 0;

  int64_t total_it = 0;
  double lr_q, lr_v, lr_a_base;
  Throttle throttle;
  // hipGraph of `graph_unroll` steps
  hipGraphExec_t gexec = nullptr;
  int graph_unroll = 0;
  hipStream_t cap_stream = nullptr;  // capture only (the legacy default stream cannot capture); shared, see capture_stream
  // timing
  bool timing = false;
  hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  double t_acc[3] = {0, 0, 0};
  double t_empty = 0;  // interval of two back-to-back event records (event overhead)
  int64_t t_n = 0;
};

static_assert(MAX_CRITICS == IQLHIP_MAX_CRITICS, "iql_step.h and iqlhip.h disagree");
static_assert(IQLHIP_N_TENSORS == 2 * DEEP_MAX_LIN * MAX_TRAIN + 1, "offset table size");
static_assert(IQLHIP_MAX_HIDDEN + 1 == DEEP_MAX_LIN, "iql_deep.h and iqlhip.h disagree");
// number of critics: 0 in the config means the reference's TwinQ
static int n_critics(const iqlhip_trainer_config &c) { return c.n_critics > 0 ? c.n_critics : 2; }
// hidden layers: 0 in the config means the reference's default (ref:458-459, 519, 538)
static int n_hidden(const iqlhip_trainer_config &c) { return c.n_hidden > 0 ? c.n_hidden : 2; }
// which step runs this shape: the tuned three-kernel one or the general layer-wise one
static bool is_deep(const iqlhip_trainer_config &c) {
  return n_hidden(c) != 2 || (c.hidden_dim != 64 && c.hidden_dim != 128 && c.hidden_dim != 256) ||
         getenv("IQLHIP_FORCE_GENERAL") != nullptr;  // (tests: the general step on a shape the tuned one takes)
}

static int check_cfg(const iqlhip_trainer_config *c) {
  if (!c) return fail(IQLHIP_ERR_INVALID, "null config");
  if (c->n_critics < 0 || c->n_critics == 1 || c->n_critics > MAX_CRITICS)
    return fail(IQLHIP_ERR_UNSUPPORTED, "n_critics %d: 0 (= 2) or 2..%d", c->n_critics, MAX_CRITICS);
  if (c->state_dim <= 0 || c->action_dim <= 0) return fail(IQLHIP_ERR_INVALID, "bad dims");
  if (c->n_hidden < 0) return fail(IQLHIP_ERR_INVALID, "n_hidden must be >= 0");
  if (const char *why = nullptr; !deep_shape_ok(*c, n_hidden(*c), &why))
    return fail(IQLHIP_ERR_UNSUPPORTED, "n_hidden %d, hidden_dim %d: %s", n_hidden(*c), c->hidden_dim, why);
  if (c->batch_size < 16 || c->batch_size % 16)
    return fail(IQLHIP_ERR_UNSUPPORTED, "batch_size %d: must be a positive multiple of 16", c->batch_size);
  if (c->action_dim > 32) return fail(IQLHIP_ERR_UNSUPPORTED, "action_dim %d > 32", c->action_dim);
  if (c->state_dim + c->action_dim > 128) return fail(IQLHIP_ERR_UNSUPPORTED, "state_dim+action_dim > 128");
  // the misc block of k_update sums the per-slab loss partials in its LDS
  if (!is_deep(*c) && (c->batch_size / 16) * (n_critics(*c) + 2 + (c->deterministic ? 0 : c->action_dim)) + n_critics(*c) + 2 >
      update_lds_floats())
    return fail(IQLHIP_ERR_UNSUPPORTED, "batch_size %d too large for action_dim %d", c->batch_size,
                c->action_dim);
  if (c->precision != IQLHIP_PREC_FP32 && c->precision != IQLHIP_PREC_BF16)
    return fail(IQLHIP_ERR_INVALID, "precision must be IQLHIP_PREC_FP32 or IQLHIP_PREC_BF16");
  if (c->dropout_p >= 1.0f) return fail(IQLHIP_ERR_INVALID, "dropout_p must be < 1");
  if (c->cosine_t_max <= 0) return fail(IQLHIP_ERR_INVALID, "cosine_t_max must be positive");
  if (c->polyak_form != 0 && c->polyak_form != 1) return fail(IQLHIP_ERR_INVALID, "polyak_form must be 0 or 1");
  return 0;
}

struct Layout {
  int64_t off[IQLHIP_N_TENSORS];
  int64_t n_params, n_target;      // arena elements (with alignment padding)
  int64_t true_params, true_target;  // trained scalars (SURVEY 8d byte model)
};
static void net_dims(const iqlhip_trainer_config &c, int net, int *in, int *out) {
  const int E = n_critics(c);
  *in = net < E ? c.state_dim + c.action_dim : c.state_dim;
  *out = (net == E + 1) ? c.action_dim : 1;
}
static Layout make_layout(const iqlhip_trainer_config &c) {
  // every tensor starts on a 128-byte line: rows of the H-wide matrices and the flat layer-1
  // strips are then streamed in whole, aligned lines
  constexpr int64_t ALIGN = 32;
  Layout L;
  int64_t o = 0, cnt = 0;
  const int H = c.hidden_dim, NL = n_hidden(c) + 1;
  const int E = n_critics(c), ntrain = E + 2;
  for (int k = 0; k < IQLHIP_N_TENSORS; ++k) L.off[k] = -1;
  for (int n = 0; n < ntrain; ++n) {
    int in, out;
    net_dims(c, n, &in, &out);
    for (int l = 0; l < NL; ++l) {
      const int64_t rows = l == NL - 1 ? out : H, cols = l == 0 ? in : H;
      const int64_t sz[2] = {rows * cols, rows};  // W_l [rows][cols], b_l [rows]
      for (int k = 0; k < 2; ++k) {
        o = (o + ALIGN - 1) / ALIGN * ALIGN;
        L.off[(n * NL + l) * 2 + k] = o;
        o += sz[k];
        cnt += sz[k];
      }
    }
    if (n == E - 1) L.n_target = o, L.true_target = cnt;
  }
  if (c.deterministic) {
    L.off[ntrain * NL * 2] = -1;
  } else {
    o = (o + ALIGN - 1) / ALIGN * ALIGN;
    L.off[ntrain * NL * 2] = o;
    o += c.action_dim;
    cnt += c.action_dim;
  }
  L.n_params = o;
  L.true_params = cnt;
  return L;
}

extern "C" int iqlhip_arena_layout(const iqlhip_trainer_config *cfg, int64_t offsets[IQLHIP_N_TENSORS],
                                   int64_t *n_params, int64_t *n_target) {
  if (int e = check_cfg(cfg)) return e;
  const Layout L = make_layout(*cfg);
  if (offsets) memcpy(offsets, L.off, sizeof(L.off));
  if (n_params) *n_params = L.n_params;
  if (n_target) *n_target = L.n_target;
  return 0;
}

extern "C" int iqlhip_step_cost(const iqlhip_trainer_config *cfg, double *bytes, double *flops) {
  if (int e = check_cfg(cfg)) return e;
  const Layout L = make_layout(*cfg);
  const double B = cfg->batch_size, S = cfg->state_dim, A = cfg->action_dim, H = cfg->hidden_dim;
  // gather + (read p,g,m,v; write g,p,m,v) per trained parameter + target r/w
  if (bytes) *bytes = 4.0 * B * (2 * S + A + 2) + 32.0 * (double)L.true_params + 8.0 * (double)L.true_target;
  if (flops) {
    const double E = n_critics(*cfg), HH = (n_hidden(*cfg) - 1) * H * H;  // hidden-to-hidden matrices
    const double wv = S * H + HH + H, wq = (S + A) * H + HH + H, wa = S * H + HH + H * A;
    const double fwd = 2 * wv + 2 * E * wq + wa;                 // V twice, target+online critics, actor
    const double dw = wv + E * wq + wa;                          // weight gradients
    const double dx = (HH + H) * (E + 1) + (HH + H * A);         // input gradients of every layer but the first
    *flops = 2.0 * B * (fwd + dw + dx);
  }
  return 0;
}

template <typename T>
static T *carve(char *&p, size_t n) {
  T *r = reinterpret_cast<T *>(p);
  p += (n * sizeof(T) + 255) / 256 * 256;
  return r;
}

// The update kernel's item table is XCD-major: block b runs on XCD b & 7 (round-robin dispatch), so
// slot i of the table belongs to XCD i & 7, row i >> 3.  `per_xcd` lists what each XCD works on; rows
// are padded with idle slots (net = -1), which gather the next step's batch: they are numbered
// (o0 = index, i0 = count).  `min_depth`: tables of one group share their size.
static std::vector<UpdItem> flatten_table(const std::vector<UpdItem> (&per_xcd)[8], size_t min_depth, size_t *depth_out,
                                          size_t min_idle) {
  size_t depth = 0, n_real = 0;
  for (auto &v : per_xcd) depth = v.size() > depth ? v.size() : depth, n_real += v.size();
  // a table that happens to be (almost) full gets more rows of slots for the batch prefetch: one idle
  // slot per 16 batch rows (a 256-thread slot gathers 16 rows per pass; batch 1024 on 16 slots took four
  // passes of random rows out of a 288 MB buffer each -- the longest chain of the launch)
  while (8 * depth - n_real < min_idle) ++depth;
  if (depth < min_depth) depth = min_depth;
  std::vector<UpdItem> items;
  for (size_t d = 0; d < depth; ++d)
    for (int x = 0; x < 8; ++x) {
      UpdItem pad;
      memset(&pad, 0, sizeof(pad));
      pad.net = -1;
      items.push_back(d < per_xcd[x].size() ? per_xcd[x][d] : pad);
    }
  int n_pad = 0;
  for (auto &it : items)
    if (it.net < 0) it.o0 = n_pad++;
  for (auto &it : items)
    if (it.net < 0) it.i0 = n_pad;
  if (depth_out) *depth_out = depth;
  return items;
}

// Which XCD works on which items.
//   one seed per launch, TwinQ (4 trained nets): network n on XCDs 2n and 2n + 1, each taking half of
//     every layer (the panels a GEMM shares meet in two L2s, the bytes are spread over all eight);
//   one seed, more critics: the nets' items dealt out in order, an eighth of all items per XCD (6 nets
//     on pairs of XCDs left four XCDs with twice the work: E = 4 at batch 1024, round 4);
//   member k of a group launch: a WHOLE network per XCD -- (2n + k) & 7 for four nets: the eight
//     members of a group put one network of every kind on every XCD -- so that its activation / delta
//     panels are fetched from memory into ONE L2 (the balanced one-seed table fetches them into two:
//     2.9 MB of the 13.1 MB a seed's update moved in round 3).
// Measured (round 4, gpurun_out/e1, steps/s with the one-seed table / the whole-network table in every
// member): 8 seeds 219.7k / 223.6k (k_update 17.3 -> 16.7 us), 4 seeds 167.7k / 160.0k, 2 seeds 111.4k /
// 107.7k -- the rotation gives every XCD one network of every kind only for multiples of eight members;
// smaller groups keep the one-seed table.  Dealt in order (E > 2, one seed): E = 4 at batch 1024 35.9k ->
// 37.5k steps/s, E = 8 at batch 256 41.0k -> 42.4k.
static void deal_items(const iqlhip_trainer *t, int member, int group_size, std::vector<UpdItem> (&per_xcd)[8]) {
  const int NT = t->D.ntrain;
  static const int forced = getenv("IQLHIP_ITEM_TABLE") ? atoi(getenv("IQLHIP_ITEM_TABLE")) : -1;  // A/B: 0 / 1 forces
  const bool group = group_size > 1;
  const bool whole = forced >= 0 ? (forced != 0 && group) : (group && group_size % 8 == 0);
  if (whole && NT == 4) {
    for (int n = 0; n < NT; ++n)
      for (auto &it : t->net_items[n]) per_xcd[(2 * n + member) & 7].push_back(it);
    return;
  }
  if (NT == 4) {
    for (int n = 0; n < NT; ++n)
      for (size_t i = 0; i < t->net_items[n].size(); ++i)
        per_xcd[(2 * n + ((int)i < t->n_half0[n] ? 0 : 1)) & 7].push_back(t->net_items[n][i]);
    return;
  }
  size_t total = 0, g = 0;
  for (int n = 0; n < NT; ++n) total += t->net_items[n].size();
  for (int n = 0; n < NT; ++n)
    for (auto &it : t->net_items[n]) {
      const int x = (int)(g * 8 / total);
      per_xcd[(x + (group ? member : 0)) & 7].push_back(it);
      ++g;
    }
  // One seed, more items per XCD than it has CUs (E = 4: 42 on 32): the launch is resident at once and the
  // dispatcher deals an XCD's work-groups over its CUs in order, so rows 32.. of a column land on the CUs of
  // rows 0..: those d CUs carry two work-groups, and what a CU takes in per microsecond bounds a work-group
  // (DESIGN.md section 4).  Put the d lightest items first and the next d lightest last, the heavy layer-2
  // tiles in between: no CU gets two heavy ones.  (IQLHIP_ITEM_ORDER=0: the order of the networks.)
  static const bool reorder = !(getenv("IQLHIP_ITEM_ORDER") && atoi(getenv("IQLHIP_ITEM_ORDER")) == 0);
  if (!group && reorder) {
    constexpr size_t CUS_PER_XCD = 32;
    auto weight = [](const UpdItem &it) { return it.layer == 1 ? 3 : (it.layer == 0 ? 2 : 1); };  // layer-2 tile / strip / layer-3 tile
    for (auto &col : per_xcd) {
      if (col.size() <= CUS_PER_XCD || col.size() >= 2 * CUS_PER_XCD) continue;
      const size_t d = col.size() - CUS_PER_XCD;
      std::vector<UpdItem> sorted(col);
      std::stable_sort(sorted.begin(), sorted.end(), [&](const UpdItem &a, const UpdItem &b) { return weight(a) < weight(b); });
      std::vector<UpdItem> out(sorted.begin(), sorted.begin() + d);                 // rows 0 .. d-1: the lightest
      out.insert(out.end(), sorted.begin() + 2 * d, sorted.end());                  // the rest (heaviest) in the middle
      out.insert(out.end(), sorted.begin() + d, sorted.begin() + 2 * d);            // rows 32 ..: the next lightest
      col.swap(out);
    }
  }
}

extern "C" int iqlhip_trainer_create(iqlhip_trainer **out, const iqlhip_trainer_config *cfg,
                                     const iqlhip_arenas *ar) {
  if (!out) return fail(IQLHIP_ERR_INVALID, "null out");
  if (int e = check_cfg(cfg)) return e;
  if (!ar || !ar->params || !ar->exp_avg || !ar->exp_avg_sq || !ar->target)
    return fail(IQLHIP_ERR_INVALID, "null arena pointer");
  HIP_TRY(prepare_step_kernels());
  iqlhip_trainer *t = new (std::nothrow) iqlhip_trainer();
  if (!t) return fail(IQLHIP_ERR_NOMEM, "host allocation failed");
  t->cfg = *cfg;
  t->arenas = *ar;
  t->bf16 = cfg->precision == IQLHIP_PREC_BF16;
  t->lr_q = cfg->lr_q, t->lr_v = cfg->lr_v, t->lr_a_base = cfg->lr_actor;
  const Layout L = make_layout(*cfg);
  if (is_deep(*cfg)) {
    memset(&t->D, 0, sizeof(t->D));
    t->D.E = n_critics(*cfg);
    const size_t rows_bytes = (size_t)cfg->batch_size * iqlhip_replay_row_stride(cfg->state_dim, cfg->action_dim) * 4;
    hipError_t e = deep_create(&t->deep, *cfg, n_hidden(*cfg), *ar, L.off);
    if (e == hipSuccess && (e = hipMalloc((void **)&t->batch_rows, rows_bytes)) != hipSuccess) deep_destroy(t->deep);
    if (e != hipSuccess) {
      delete t;
      return fail(e == hipErrorOutOfMemory ? IQLHIP_ERR_NOMEM : IQLHIP_ERR_HIP, "general step: %s", hipGetErrorString(e));
    }
    *out = t;
    return 0;
  }
  const int S = cfg->state_dim, A = cfg->action_dim, H = cfg->hidden_dim, B = cfg->batch_size;
  const int es = t->bf16 ? 2 : 4, KM = t->bf16 ? 32 : 16;
  TrainerDesc &D = t->D;
  memset(&D, 0, sizeof(D));
  D.S = S, D.A = A, D.H = H, D.B = B, D.BP = round_up(B, 32);
  const int E = n_critics(*cfg), NT = E + 2, NF = 2 * E + 3;
  D.E = E, D.ntrain = NT, D.nfwd = NF, D.net_v = E, D.net_a = E + 1;
  D.out_v = E, D.out_qt = E + 1, D.out_nv = 2 * E + 1, D.out_mean = 2 * E + 2;
  D.two_over_B = 2.0f / (float)B, D.inv_E = 1.0f / (float)E;
  D.OUTW = round_up(D.out_mean + A, 4);
  D.deterministic = cfg->deterministic;
  D.has_dropout = cfg->dropout_p > 0.f;
  D.discount = cfg->discount, D.tau = cfg->tau, D.beta = cfg->beta, D.iql_tau = cfg->iql_tau;
  D.one_m_tau = (float)(1.0 - (double)cfg->tau);
  D.polyak_convex = cfg->polyak_form == 1;
  if (D.has_dropout) {
    const float scale = 1.0f / (float)(1.0 - (double)cfg->dropout_p);
    if (t->bf16) {  // noise.div_(1-p) happens in bf16 under autocast (ATen _dropout_impl)
      uint32_t u;
      memcpy(&u, &scale, 4);
      u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
      memcpy(&D.drop_scale, &u, 4);
    } else {
      D.drop_scale = scale;
    }
    const double thr = (double)cfg->dropout_p * 4294967296.0;
    D.drop_thr = thr >= 4294967295.0 ? 0xffffffffu : (uint32_t)thr;
  }
  D.beta1 = cfg->adam_beta1, D.beta2 = cfg->adam_beta2, D.eps = cfg->adam_eps;
  D.t_max = cfg->cosine_t_max;
  D.seed = cfg->seed;
  D.params = ar->params, D.exp_avg = ar->exp_avg, D.exp_avg_sq = ar->exp_avg_sq;
  D.target = ar->target, D.grads = ar->grads;
  D.off_log_std = L.off[NT * 6];
  D.opmax = round_up(A, 16);
  D.xrows = round_up(S + A, 64);
  D.k1max = round_up(S + A, KM);
  // ---- workspace ----
  size_t total = 0;
  auto add = [&](size_t bytes) { total += (bytes + 255) / 256 * 256; };
  int k1pad[MAX_TRAIN], outpad[MAX_TRAIN], indim[MAX_TRAIN], outdim[MAX_TRAIN];
  for (int n = 0; n < NT; ++n) {
    net_dims(*cfg, n, &indim[n], &outdim[n]);
    k1pad[n] = round_up(indim[n], KM);
    outpad[n] = round_up(outdim[n], 16);
    const int copies = (n < E) ? 2 : 1;  // + target copies
    for (int c = 0; c < copies; ++c) {
      add((size_t)H * k1pad[n] * es);
      add((size_t)H * H * es);
      add((size_t)outpad[n] * H * es);
    }
    add((size_t)H * H * es);  // w2ct
    add((size_t)H * outpad[n] * es);  // w3t
  }
  add((size_t)2 * D.xrows * D.BP * es);
  add((size_t)B * 2 * 4);
  add((size_t)B * A * 4);
  add((size_t)NT * 2 * H * D.BP * es);
  add((size_t)NT * H * D.BP * es);
  add((size_t)NT * H * D.BP * es);
  add((size_t)NT * D.opmax * D.BP * es);
  add((size_t)layer2_parts(H) * B * D.OUTW * 4);
  add((size_t)NT * (B / 16) * 4);
  add((size_t)(B / 16) * A * 4);
  add((size_t)A * 4);
  const int stride = iqlhip_replay_row_stride(S, A);
  add((size_t)B * stride * 4);
  add((size_t)B * stride * 4);  // stage_rows
  add(sizeof(DevArgs));
  add(sizeof(DevCtr));
  add(sizeof(TrainerDesc));
  // update work items (see deal_items / flatten_table): per net, the lower half of every layer first
  for (int n = 0; n < NT; ++n) {
    std::vector<UpdItem> half[2];
    auto put = [&](int layer, int o0, int i0) {
      UpdItem it;
      memset(&it, 0, sizeof(it));
      it.net = n, it.layer = layer, it.o0 = o0, it.i0 = i0;
      const int rows = layer == 2 ? outpad[n] : H;
      half[(layer == 2 && rows <= 64) ? ((i0 / 32) & 1) : (o0 >= rows / 2 ? 1 : 0)].push_back(it);
    };
    for (int o0 = 0; o0 < H; o0 += 64)
      for (int i0 = 0; i0 < H; i0 += 32) put(1, o0, i0);
    for (int o0 = 0; o0 < H; o0 += strip_rows()) put(0, o0, 0);  // layer 1: strips over all in-features
    for (int o0 = 0; o0 < outpad[n]; o0 += 64)
      for (int i0 = 0; i0 < H; i0 += 32) put(2, o0, i0);
    t->n_half0[n] = (int)half[0].size();
    t->net_items[n] = half[0];
    t->net_items[n].insert(t->net_items[n].end(), half[1].begin(), half[1].end());
  }
  size_t n_slots;
  {
    std::vector<UpdItem> per_xcd[8];
    deal_items(t, 0, 1, per_xcd);  // (sizes only: D.ntrain is set, the pointers are filled in below)
    size_t depth = 0;
    n_slots = flatten_table(per_xcd, 0, &depth, (size_t)(B + 15) / 16).size();
  }
  t->n_items = t->own_n_items = (int)n_slots;
  add(n_slots * sizeof(UpdItem));

  if (hipMalloc(&t->ws, total) != hipSuccess) {
    delete t;
    return fail(IQLHIP_ERR_NOMEM, "hipMalloc of %zu workspace bytes failed", total);
  }
  t->ws_bytes = total;
  if (hipMemset(t->ws, 0, total) != hipSuccess) {
    (void)hipFree(t->ws);
    delete t;
    return fail(IQLHIP_ERR_HIP, "hipMemset failed");
  }
  char *p = reinterpret_cast<char *>(t->ws);
  for (int n = 0; n < NT; ++n) {
    TrainNet &N = D.net[n];
    N.in_dim = indim[n], N.k1pad = k1pad[n], N.out_dim = outdim[n], N.out_pad = outpad[n];
    N.has_target = n < E;
    for (int k = 0; k < 3; ++k) {
      N.off_w[k] = L.off[n * 6 + 2 * k];
      N.off_b[k] = L.off[n * 6 + 2 * k + 1];
      N.toff_w[k] = N.has_target ? N.off_w[k] : -1;  // target arena = q1,q2 prefix of the layout
      N.toff_b[k] = N.has_target ? N.off_b[k] : -1;
    }
    N.wc[0] = carve<char>(p, (size_t)H * k1pad[n] * es);
    N.wc[1] = carve<char>(p, (size_t)H * H * es);
    N.wc[2] = carve<char>(p, (size_t)outpad[n] * H * es);
    if (N.has_target) {
      N.tc[0] = carve<char>(p, (size_t)H * k1pad[n] * es);
      N.tc[1] = carve<char>(p, (size_t)H * H * es);
      N.tc[2] = carve<char>(p, (size_t)outpad[n] * H * es);
    }
    N.w2ct = carve<char>(p, (size_t)H * H * es);
    N.w3t = carve<char>(p, (size_t)H * outpad[n] * es);
  }
  D.xT = carve<char>(p, (size_t)2 * D.xrows * D.BP * es);
  D.rd = carve<float>(p, (size_t)B * 2);
  D.actf = carve<float>(p, (size_t)B * A);
  D.hT = carve<char>(p, (size_t)NT * 2 * H * D.BP * es);
  D.dz1T = carve<char>(p, (size_t)NT * H * D.BP * es);
  D.dz2T = carve<char>(p, (size_t)NT * H * D.BP * es);
  D.dz3T = carve<char>(p, (size_t)NT * D.opmax * D.BP * es);
  D.outs = carve<float>(p, (size_t)layer2_parts(H) * B * D.OUTW);
  D.lossp = carve<float>(p, (size_t)NT * (B / 16));
  D.lsp = carve<float>(p, (size_t)(B / 16) * A);
  D.ls_snap = carve<float>(p, (size_t)A);
  t->batch_rows = carve<float>(p, (size_t)B * stride);
  D.stage_rows = carve<float>(p, (size_t)B * stride);
  D.stage_stride = stride;
  D.next_off = iqlhip_replay_next_offset(S, A);
  // the idle slots of the update kernel gather the next step's batch; without them (or with
  // IQLHIP_NO_PREFETCH, the A/B of tests/test_gpu_step.py) k_stage runs before every step
  D.prefetch = !getenv("IQLHIP_NO_PREFETCH") ? 1 : 0;  // (every table has idle slots: flatten_table)
  t->own_dargs = t->dargs = carve<DevArgs>(p, 1);
  t->own_dctr = t->dctr = carve<DevCtr>(p, 1);
  t->own_ddesc = t->ddesc = carve<TrainerDesc>(p, 1);
  t->own_ditems = t->ditems = carve<UpdItem>(p, n_slots);

  for (int n_ = 0; n_ < NT; ++n_)
  for (auto &it : t->net_items[n_]) {
    const TrainNet &N = D.net[it.net];
    const int L = it.layer;
    it.Odim = (L == 2) ? N.out_dim : H;
    it.Idim = (L == 0) ? N.in_dim : H;
    it.Opad = (L == 2) ? N.out_pad : H;
    it.Kw = (L == 0) ? N.k1pad : H;
    it.has_target = N.has_target;
    it.group = it.net == D.net_v ? 1 : (it.net == D.net_a ? 2 : 0);
    it.off_w = N.off_w[L], it.off_b = N.off_b[L], it.toff_w = N.toff_w[L], it.toff_b = N.toff_b[L];

    it.wc = N.wc[L], it.tc = N.has_target ? N.tc[L] : nullptr, it.w2ct = (L == 1) ? N.w2ct : nullptr;
    it.w3t = (L == 2) ? N.w3t : nullptr;
    const size_t plane = (size_t)H * D.BP * es;
    it.Xsrc = (L == 0) ? D.xT : reinterpret_cast<char *>(D.hT) + (size_t)(it.net * 2 + (L - 1)) * plane;
    it.Zsrc = (L == 0)   ? reinterpret_cast<char *>(D.dz1T) + (size_t)it.net * plane
              : (L == 1) ? reinterpret_cast<char *>(D.dz2T) + (size_t)it.net * plane
                         : reinterpret_cast<char *>(D.dz3T) + (size_t)it.net * D.opmax * D.BP * es;
  }
  std::vector<UpdItem> items;
  {
    std::vector<UpdItem> per_xcd[8];
    deal_items(t, 0, 1, per_xcd);
    items = flatten_table(per_xcd, 0, nullptr, (size_t)(B + 15) / 16);
  }
  if (items.size() != n_slots || hipMemcpy(t->ditems, items.data(), items.size() * sizeof(UpdItem), hipMemcpyHostToDevice) !=
      hipSuccess) {
    (void)hipFree(t->ws);
    delete t;
    return fail(IQLHIP_ERR_HIP, "hipMemcpy of the work items failed");
  }

  // ---- forward evaluations ----
  auto mk = [&](int f, int net, bool target, int in_off, int out_col, int slot) {
    FwdNet &F = D.fwd[f];
    const TrainNet &N = D.net[net];
    F.w1c = target ? N.tc[0] : N.wc[0];
    F.w2c = target ? N.tc[1] : N.wc[1];
    F.w3c = target ? N.tc[2] : N.wc[2];
    const float *base = target ? ar->target : ar->params;
    F.b1 = base + N.off_b[0], F.b2 = base + N.off_b[1], F.b3 = base + N.off_b[2];
    F.in_off = in_off, F.in_dim = N.in_dim, F.k1pad = N.k1pad;
    F.out_dim = N.out_dim, F.out_pad = N.out_pad, F.out_col = out_col;
    F.train_slot = slot;
    F.tanh_out = net == D.net_a;
    F.dropout = (net == D.net_a) && D.has_dropout;
    F.stage = f == 0;
  };
  // evaluation order: q_e, v, actor, target q_e, next_v (iql_step.h)
  for (int e = 0; e < E; ++e) mk(e, e, false, 0, e, e);
  mk(E, D.net_v, false, 0, D.out_v, D.net_v);
  mk(E + 1, D.net_a, false, 0, D.out_mean, D.net_a);
  for (int e = 0; e < E; ++e) mk(E + 2 + e, e, true, 0, D.out_qt + e, -1);
  mk(2 * E + 2, D.net_v, false, D.next_off, D.out_nv, -1);

  if (hipMemcpy(t->ddesc, &t->D, sizeof(TrainerDesc), hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(t->ws);
    delete t;
    return fail(IQLHIP_ERR_HIP, "hipMemcpy of the descriptor failed");
  }
  *out = t;
  return 0;
}

extern "C" int iqlhip_trainer_destroy(iqlhip_trainer *t) {
  if (!t) return 0;
  if (t->group) return fail(IQLHIP_ERR_INVALID, "trainer is a member of a group: destroy the group first");
  if (t->gexec) (void)hipGraphExecDestroy(t->gexec);
  for (int k = 0; k < iqlhip_trainer::ARG_RING; ++k) {
    if (t->harg_ev[k]) (void)hipEventDestroy(t->harg_ev[k]);
    if (t->harg[k]) (void)hipHostFree(t->harg[k]);
  }

  for (auto &e : t->ev)
    if (e) (void)hipEventDestroy(e);
  t->throttle.destroy();
  if (t->deep) {
    deep_destroy(t->deep);
    (void)hipFree(t->batch_rows);
  }
  if (t->ws) (void)hipFree(t->ws);
  delete t;
  return 0;
}

extern "C" int iqlhip_trainer_step_kind(iqlhip_trainer *t, int32_t *kind) {
  if (!t || !kind) return fail(IQLHIP_ERR_INVALID, "null argument");
  *kind = t->deep ? 1 : 0;
  return 0;
}

extern "C" int iqlhip_trainer_sync_weights(iqlhip_trainer *t, void *stream) {
  if (!t) return fail(IQLHIP_ERR_INVALID, "null trainer");
  if (t->deep) {
    HIP_TRY(deep_sync_weights(t->deep, (hipStream_t)stream));
    return 0;
  }
  HIP_TRY(launch_sync_weights(t->bf16, t->ddesc, (hipStream_t)stream));
  return 0;
}

extern "C" int iqlhip_trainer_set_step(iqlhip_trainer *t, int64_t total_it) {
  if (!t || total_it < 0) return fail(IQLHIP_ERR_INVALID, "bad argument");
  // rare (init, load_state_dict): drain every stream first, so that no step in flight on a
  // non-blocking stream can race the counter write below
  HIP_TRY(hipDeviceSynchronize());
  t->total_it = total_it;
  if (t->deep) return 0;  // (the general step takes its step count with every launch)
  DevCtr c;
  memset(&c, 0, sizeof(c));
  c.ctr[0] = total_it, c.ctr[1] = total_it;
  HIP_TRY(hipMemcpy(t->dctr, &c, sizeof(c), hipMemcpyHostToDevice));
  t->dev_args_valid = false;
  if (t->group) group_invalidate(t->group);
  return 0;
}

static double cosine_lr(double base, int64_t t, int64_t t_max) {
  return base * (1.0 + cos(M_PI * (double)t / (double)t_max)) * 0.5;
}

extern "C" int iqlhip_trainer_get_step(iqlhip_trainer *t, int64_t *total_it, double *actor_lr) {
  if (!t) return fail(IQLHIP_ERR_INVALID, "null trainer");
  if (total_it) *total_it = t->total_it;
  if (actor_lr) *actor_lr = cosine_lr(t->lr_a_base, t->total_it, t->cfg.cosine_t_max);
  return 0;
}

extern "C" int iqlhip_trainer_set_lr(iqlhip_trainer *t, double lr_q, double lr_v, double lr_a_base) {
  if (!t) return fail(IQLHIP_ERR_INVALID, "null trainer");
  t->lr_q = lr_q, t->lr_v = lr_v, t->lr_a_base = lr_a_base;
  return 0;
}

extern "C" int iqlhip_trainer_set_timing(iqlhip_trainer *t, int32_t enable) {
  if (!t) return fail(IQLHIP_ERR_INVALID, "null trainer");
  t->timing = enable != 0;
  t->t_acc[0] = t->t_acc[1] = t->t_acc[2] = 0;
  t->t_empty = 0;
  t->t_n = 0;
  if (t->timing)
    for (auto &e : t->ev)
      if (!e) HIP_TRY(hipEventCreate(&e));
  return 0;
}

extern "C" int iqlhip_trainer_get_timing(iqlhip_trainer *t, double avg_ms[3], int64_t *n) {
  if (!t) return fail(IQLHIP_ERR_INVALID, "null trainer");
  // event-pair overhead (measured on an empty interval in the same pass) is subtracted
  for (int k = 0; k < 3; ++k) {
    const double v = t->t_n ? (t->t_acc[k] - t->t_empty) / (double)t->t_n : 0.0;
    avg_ms[k] = v > 0 ? v : 0.0;
  }
  if (n) *n = t->t_n;
  return 0;
}

static int enqueue_step(iqlhip_trainer *t, hipStream_t st) {
  if (!t->D.prefetch) HIP_TRY(launch_stage(t->bf16, t->D, t->ddesc, t->dargs, t->dctr, 1, st));
  HIP_TRY(launch_forward(t->bf16, t->D, t->ddesc, t->dargs, t->dctr, 1, st));
  HIP_TRY(launch_backward(t->bf16, t->D, t->ddesc, t->dargs, t->dctr, 1, st));
  HIP_TRY(launch_update(t->bf16, t->ddesc, t->dargs, t->dctr, t->ditems, t->n_items, 1, st));
  return 0;
}

// A call continues the previous one when it reads the same replay contents (rows, size,
// generation) with on-device indices and dropout masks, produces no per-step outputs and has the
// same learning rates: nothing in DevArgs that the kernels read differs (base_step / n_steps only
// index idx[], drop_keep[] and losses_out[]), and the last update of the previous call has already
// staged this call's first batch.  Then nothing is sent and k_stage is not launched.
static bool continues(const DevArgs &a, const DevArgs &b) {
  return a.rows == b.rows && a.n_rows == b.n_rows && a.row_stride == b.row_stride &&
         a.generation == b.generation && a.idx_mode == 0 && b.idx_mode == 0 && !a.drop_keep && !b.drop_keep &&
         !a.losses_out && !b.losses_out && a.lr_q == b.lr_q && a.lr_v == b.lr_v && a.lr_a_base == b.lr_a_base;
}

static hipError_t push_args(iqlhip_trainer *t, const DevArgs &args_in, int64_t n_steps, hipStream_t st) {
  DevArgs args = args_in;
  args.n_steps = n_steps;
  const int k = t->harg_head;
  t->harg_head = (k + 1) % iqlhip_trainer::ARG_RING;
  hipError_t e;
  if (!t->harg[k]) {
    if ((e = hipHostMalloc((void **)&t->harg[k], sizeof(DevArgs), hipHostMallocDefault)) != hipSuccess) return e;
    if ((e = hipEventCreateWithFlags(&t->harg_ev[k], hipEventDisableTiming)) != hipSuccess) return e;
  }
  if (t->harg_used[k] && (e = hipEventSynchronize(t->harg_ev[k])) != hipSuccess) return e;
  *t->harg[k] = args;
  if ((e = hipMemcpyAsync(t->dargs, t->harg[k], sizeof(DevArgs), hipMemcpyHostToDevice, st)) != hipSuccess)
    return e;
  if ((e = hipEventRecord(t->harg_ev[k], st)) != hipSuccess) return e;
  t->harg_used[k] = true;
  t->dev_args = args;
  return hipSuccess;
}

// The general step: three plain launches per step, the step's arguments (index / mask / loss slices,
// Adam coefficients computed here in double) by value.  graph_unroll is ignored.
static int run_steps_deep(iqlhip_trainer *t, const DevArgs &a, int64_t n_steps, hipStream_t st) {
  const int64_t B = t->cfg.batch_size, H = t->cfg.hidden_dim, NH = n_hidden(t->cfg);
  for (int64_t i = 0; i < n_steps; ++i) {
    DeepStep s;
    memset(&s, 0, sizeof(s));
    s.rows = a.rows, s.n_rows = a.n_rows, s.row_stride = a.row_stride, s.idx_mode = a.idx_mode;
    s.idx = a.idx ? a.idx + i * B : nullptr;
    s.drop_keep = a.drop_keep ? a.drop_keep + i * NH * B * H : nullptr;
    s.losses_out = a.losses_out ? a.losses_out + i * 3 : nullptr;
    s.step = a.base_step + i;
    s.coef = make_adam_coef(t->cfg.adam_beta1, t->cfg.adam_beta2, t->cfg.adam_eps, a.lr_q, a.lr_v, a.lr_a_base,
                            t->cfg.cosine_t_max, s.step + 1);
    HIP_TRY(deep_step(t->deep, s, st, t->timing ? t->ev : nullptr));
    if (t->timing) {
      HIP_TRY(hipEventRecord(t->ev[4], st));
      HIP_TRY(hipEventSynchronize(t->ev[4]));
      for (int k = 0; k < 3; ++k) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, t->ev[k], t->ev[k + 1]));
        t->t_acc[k] += ms;
      }
      float ems = 0.f;
      HIP_TRY(hipEventElapsedTime(&ems, t->ev[3], t->ev[4]));
      t->t_empty += ems;
      t->t_n++;
    } else {
      HIP_TRY(t->throttle.queued(3, st));
    }
  }
  return 0;
}

static int run_steps(iqlhip_trainer *t, const DevArgs &args_in, int64_t n_steps, int graph_unroll,
                     hipStream_t st) {
  if (t->deep) return run_steps_deep(t, args_in, n_steps, st);
  if (t->group) group_invalidate(t->group);  // this call rewrites the member's slot of the group's arguments
  if (!(t->D.prefetch && t->dev_args_valid && !t->timing && continues(t->dev_args, args_in))) {
    HIP_TRY(push_args(t, args_in, n_steps, st));
    // the first step's batch (later steps are staged by the update kernel of the step before)
    if (t->D.prefetch) HIP_TRY(launch_stage(t->bf16, t->D, t->ddesc, t->dargs, t->dctr, 1, st));
  }
  // after this call the device holds these arguments and (prefetch, on-device indices) the batch
  // of the step that follows it
  t->dev_args_valid = args_in.idx_mode == 0;
  int64_t done = 0;
  if (t->timing) {
    // one event pair per kernel: serialises the stream a little; diagnostic mode only
    for (; done < n_steps; ++done) {
      if (!t->D.prefetch) HIP_TRY(launch_stage(t->bf16, t->D, t->ddesc, t->dargs, t->dctr, 1, st));
      HIP_TRY(hipEventRecord(t->ev[0], st));
      HIP_TRY(launch_forward(t->bf16, t->D, t->ddesc, t->dargs, t->dctr, 1, st));
      HIP_TRY(hipEventRecord(t->ev[1], st));
      HIP_TRY(launch_backward(t->bf16, t->D, t->ddesc, t->dargs, t->dctr, 1, st));
      HIP_TRY(hipEventRecord(t->ev[2], st));
      HIP_TRY(launch_update(t->bf16, t->ddesc, t->dargs, t->dctr, t->ditems, t->n_items, 1, st));
      HIP_TRY(hipEventRecord(t->ev[3], st));
      HIP_TRY(hipEventRecord(t->ev[4], st));  // empty interval: what a record pair costs by itself
      HIP_TRY(hipEventSynchronize(t->ev[4]));
      for (int k = 0; k < 3; ++k) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, t->ev[k], t->ev[k + 1]));
        t->t_acc[k] += ms;
      }
      float ems = 0.f;
      HIP_TRY(hipEventElapsedTime(&ems, t->ev[3], t->ev[4]));
      t->t_empty += ems;
      t->t_n++;
    }
    return 0;
  }
  if (graph_unroll > 0 && n_steps >= graph_unroll) {
    if (!t->gexec || t->graph_unroll != graph_unroll) {
      if (t->gexec) {
        (void)hipGraphExecDestroy(t->gexec);
        t->gexec = nullptr;
      }
      hipGraph_t g = nullptr;
      if (!t->cap_stream) HIP_TRY(capture_stream(&t->cap_stream));
      HIP_TRY(hipStreamBeginCapture(t->cap_stream, hipStreamCaptureModeThreadLocal));
      int rc = 0;
      for (int u = 0; u < graph_unroll && !rc; ++u) rc = enqueue_step(t, t->cap_stream);
      hipError_t ce = hipStreamEndCapture(t->cap_stream, &g);
      if (rc) return rc;
      HIP_TRY(ce);
      HIP_TRY(hipGraphInstantiate(&t->gexec, g, nullptr, nullptr, 0));
      (void)hipGraphDestroy(g);
      t->graph_unroll = graph_unroll;
    }
    for (; done + graph_unroll <= n_steps; done += graph_unroll) {
      HIP_TRY(hipGraphLaunch(t->gexec, st));
      HIP_TRY(t->throttle.queued(3 * (int64_t)graph_unroll, st));
    }
  }
  for (; done < n_steps; ++done) {
    if (int rc = enqueue_step(t, st)) return rc;
    HIP_TRY(t->throttle.queued(3, st));
  }
  return 0;
}

extern "C" int iqlhip_train_steps(iqlhip_trainer *t, const iqlhip_replay_view *view, int64_t n_steps,
                                  const int64_t *idx, const uint8_t *dropout_keep, float *losses_out,
                                  int32_t graph_unroll, void *stream) {
  if (!t || !view || !view->rows) return fail(IQLHIP_ERR_INVALID, "null argument");
  if (n_steps < 0) return fail(IQLHIP_ERR_INVALID, "n_steps must be >= 0");
  if (view->state_dim != t->cfg.state_dim || view->action_dim != t->cfg.action_dim)
    return fail(IQLHIP_ERR_INVALID, "replay dims (%d,%d) do not match the trainer (%d,%d)", view->state_dim,
                view->action_dim, t->cfg.state_dim, t->cfg.action_dim);
  if (view->n_rows <= 0) return fail(IQLHIP_ERR_INVALID, "cannot sample from an empty replay buffer");
  if (view->row_stride != iqlhip_replay_row_stride(view->state_dim, view->action_dim))
    return fail(IQLHIP_ERR_INVALID, "row_stride %d is not the packed layout's (%d)", view->row_stride,
                iqlhip_replay_row_stride(view->state_dim, view->action_dim));
  if (n_steps == 0) return 0;
  DevArgs a;
  memset(&a, 0, sizeof(a));
  a.rows = view->rows, a.n_rows = view->n_rows, a.row_stride = view->row_stride;
  a.generation = view->generation;
  a.idx_mode = idx ? 1 : 0, a.idx = idx;
  a.drop_keep = dropout_keep, a.losses_out = losses_out;
  a.base_step = t->total_it;
  a.lr_q = t->lr_q, a.lr_v = t->lr_v, a.lr_a_base = t->lr_a_base;
  if (int rc = run_steps(t, a, n_steps, graph_unroll, (hipStream_t)stream)) return rc;
  t->total_it += n_steps;
  return 0;
}

extern "C" int iqlhip_train_batch(iqlhip_trainer *t, const float *s, const float *a, const float *r,
                                  const float *s2, const float *d, const uint8_t *dropout_keep,
                                  float *losses_out, void *stream) {
  if (!t || !s || !a || !r || !s2 || !d) return fail(IQLHIP_ERR_INVALID, "null argument");
  const int S = t->cfg.state_dim, A = t->cfg.action_dim, B = t->cfg.batch_size;
  const int stride = iqlhip_replay_row_stride(S, A);
  HIP_TRY(launch_pack(t->batch_rows, stride, S, A, 0, B, s, a, r, s2, d, (hipStream_t)stream));
  DevArgs args;
  memset(&args, 0, sizeof(args));
  args.rows = t->batch_rows, args.n_rows = B, args.row_stride = stride;
  args.idx_mode = 2;
  args.drop_keep = dropout_keep, args.losses_out = losses_out;
  args.base_step = t->total_it;
  args.lr_q = t->lr_q, args.lr_v = t->lr_v, args.lr_a_base = t->lr_a_base;
  if (int rc = run_steps(t, args, 1, 0, (hipStream_t)stream)) return rc;
  t->total_it += 1;
  return 0;
}

// ------------------------------------------------------------------ groups --
// K independent trainers (seeds) of one shape stepped by ONE launch sequence: every kernel
// runs with gridDim.y = K and work-group (x, k) does for seed k exactly what work-group x of
// a solo launch does -- the arithmetic of every seed is bit-identical to running it alone.
// One seed at batch 256 is a latency chain on a fraction of the chip; K seeds per launch
// fill it and pay the kernel boundaries once per K seed-steps (the reference runs several
// agents per GPU for the same reason, ensemble_sweeps/launch.sh:12 AGENTS_PER_GPU).
struct iqlhip_group {
  int K = 0;
  int n_items = 0;  // slots of every member's item table (deal_items: a whole network per XCD, rotated by member)
  iqlhip_trainer *tr[IQLHIP_MAX_GROUP] = {};
  void *mem = nullptr;  // [K] TrainerDesc | [K] DevArgs | [K] DevCtr | [K][n_items] UpdItem
  TrainerDesc *gdesc = nullptr;
  DevArgs *gargs = nullptr;
  DevCtr *gctr = nullptr;
  UpdItem *gitems = nullptr;
  static constexpr int ARG_RING = 8;
  DevArgs *harg[ARG_RING] = {};  // pinned, [K] each
  hipEvent_t harg_ev[ARG_RING] = {};
  bool harg_used[ARG_RING] = {};
  int harg_head = 0;
  hipGraphExec_t gexec = nullptr;
  int graph_unroll = 0;
  hipStream_t cap_stream = nullptr;
  Throttle throttle;
  DevArgs dev_args[IQLHIP_MAX_GROUP];  // what the device copies hold (see `continues`)
  bool dev_args_valid = false;
  // per-kernel HIP-event timing (diagnostic mode, eager launches)
  bool timing = false;
  hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  double t_acc[3] = {0, 0, 0}, t_empty = 0;
  int64_t t_n = 0;
};

static void group_invalidate(iqlhip_group *g) { g->dev_args_valid = false; }

static bool same_shape(const iqlhip_trainer_config &a, const iqlhip_trainer_config &b) {
  return a.state_dim == b.state_dim && a.action_dim == b.action_dim && a.hidden_dim == b.hidden_dim &&
         a.batch_size == b.batch_size && a.deterministic == b.deterministic && a.precision == b.precision &&
         n_critics(a) == n_critics(b) && (a.dropout_p > 0.f) == (b.dropout_p > 0.f);  // (polyak_form: per seed)
}

extern "C" int iqlhip_group_create(iqlhip_group **out, iqlhip_trainer *const *trainers, int32_t n) {
  if (!out || !trainers) return fail(IQLHIP_ERR_INVALID, "null argument");
  if (n < 1 || n > IQLHIP_MAX_GROUP) return fail(IQLHIP_ERR_INVALID, "group size %d: 1..%d", n, IQLHIP_MAX_GROUP);
  for (int k = 0; k < n; ++k) {
    if (!trainers[k]) return fail(IQLHIP_ERR_INVALID, "null trainer");
    if (trainers[k]->deep)
      return fail(IQLHIP_ERR_UNSUPPORTED, "seed groups run on the tuned step only (n_hidden = 2, hidden_dim 64 / 128 / "
                  "256); step trainers of other shapes one by one");
    if (trainers[k]->group) return fail(IQLHIP_ERR_INVALID, "trainer %d already belongs to a group", k);
    for (int j = 0; j < k; ++j)
      if (trainers[j] == trainers[k]) return fail(IQLHIP_ERR_INVALID, "trainer %d listed twice", k);
    if (!same_shape(trainers[0]->cfg, trainers[k]->cfg) || trainers[k]->n_items != trainers[0]->n_items)
      return fail(IQLHIP_ERR_INVALID, "trainer %d differs in shape from trainer 0 (dims, batch, precision, "
                  "critics, policy kind and dropout on/off must match)", k);
  }
  iqlhip_group *g = new (std::nothrow) iqlhip_group();
  if (!g) return fail(IQLHIP_ERR_NOMEM, "host allocation failed");
  g->K = n;
  // the members' item tables, dealt for a group launch; one size for all
  std::vector<std::vector<UpdItem>> tables(n);
  {
    size_t depth = 0;
    for (int pass = 0; pass < 2; ++pass)
      for (int k = 0; k < n; ++k) {
        std::vector<UpdItem> per_xcd[8];
        deal_items(trainers[k], k, n, per_xcd);
        size_t d = 0;
        tables[k] = flatten_table(per_xcd, depth, &d, (size_t)(trainers[k]->D.B + 15) / 16);
        depth = d > depth ? d : depth;
      }
  }
  const int ni = (int)tables[0].size();
  for (int k = 0; k < n; ++k)
    if ((int)tables[k].size() != ni) {
      delete g;
      return fail(IQLHIP_ERR_INVALID, "item tables of the members differ in size");
    }
  g->n_items = ni;
  auto up = [](size_t b) { return (b + 255) / 256 * 256; };
  const size_t b_desc = up(sizeof(TrainerDesc) * n), b_args = up(sizeof(DevArgs) * n),
               b_ctr = up(sizeof(DevCtr) * n), b_items = up(sizeof(UpdItem) * (size_t)ni * n);
  HIP_TRY(hipDeviceSynchronize());  // members may have steps in flight on other streams
  if (hipMalloc(&g->mem, b_desc + b_args + b_ctr + b_items) != hipSuccess) {
    delete g;
    return fail(IQLHIP_ERR_NOMEM, "hipMalloc of the group descriptors failed");
  }
  char *p = reinterpret_cast<char *>(g->mem);
  g->gdesc = reinterpret_cast<TrainerDesc *>(p), p += b_desc;
  g->gargs = reinterpret_cast<DevArgs *>(p), p += b_args;
  g->gctr = reinterpret_cast<DevCtr *>(p), p += b_ctr;
  g->gitems = reinterpret_cast<UpdItem *>(p);
  hipError_t e = hipSuccess;
  for (int k = 0; k < n && e == hipSuccess; ++k) {
    iqlhip_trainer *t = trainers[k];
    e = hipMemcpy(g->gdesc + k, t->ddesc, sizeof(TrainerDesc), hipMemcpyDeviceToDevice);
    if (e == hipSuccess) e = hipMemcpy(g->gargs + k, t->dargs, sizeof(DevArgs), hipMemcpyDeviceToDevice);
    if (e == hipSuccess) e = hipMemcpy(g->gctr + k, t->dctr, sizeof(DevCtr), hipMemcpyDeviceToDevice);
    if (e == hipSuccess)
      e = hipMemcpy(g->gitems + (size_t)k * ni, tables[k].data(), sizeof(UpdItem) * ni, hipMemcpyHostToDevice);
  }
  if (e != hipSuccess) {
    (void)hipFree(g->mem);
    delete g;
    return fail(IQLHIP_ERR_HIP, "copying the member descriptors failed: %s", hipGetErrorString(e));
  }
  for (int k = 0; k < n; ++k) {
    iqlhip_trainer *t = trainers[k];
    g->tr[k] = t;
    t->group = g;
    t->ddesc = g->gdesc + k, t->dargs = g->gargs + k, t->dctr = g->gctr + k, t->ditems = g->gitems + (size_t)k * ni;
    t->n_items = ni;
    // the member's DevArgs slot has moved: its next solo call must send its arguments and stage
    // its first batch again (`continues` would otherwise trust what the OLD slot held)
    t->dev_args_valid = false;
    if (t->gexec) {  // the member's own graph holds the old descriptor addresses
      (void)hipGraphExecDestroy(t->gexec);
      t->gexec = nullptr;
    }
  }
  *out = g;
  return 0;
}

// Members go back to their own workspace slots (with the group's current counters) and stay
// usable on their own.
extern "C" int iqlhip_group_destroy(iqlhip_group *g) {
  if (!g) return 0;
  (void)hipDeviceSynchronize();
  for (int k = 0; k < g->K; ++k) {
    iqlhip_trainer *t = g->tr[k];
    (void)hipMemcpy(t->own_dctr, t->dctr, sizeof(DevCtr), hipMemcpyDeviceToDevice);
    t->ddesc = t->own_ddesc, t->dargs = t->own_dargs, t->dctr = t->own_dctr, t->ditems = t->own_ditems;
    t->n_items = t->own_n_items;
    t->group = nullptr;
    // own_dargs holds whatever the member's last solo call OUTSIDE the group sent (all zero if
    // there was none: rows = NULL): a solo call after the group must never continue from it
    t->dev_args_valid = false;
    if (t->gexec) {
      (void)hipGraphExecDestroy(t->gexec);
      t->gexec = nullptr;
    }
  }
  if (g->gexec) (void)hipGraphExecDestroy(g->gexec);
  g->throttle.destroy();
  for (auto &e : g->ev)
    if (e) (void)hipEventDestroy(e);
  for (int k = 0; k < iqlhip_group::ARG_RING; ++k) {
    if (g->harg_ev[k]) (void)hipEventDestroy(g->harg_ev[k]);
    if (g->harg[k]) (void)hipHostFree(g->harg[k]);
  }
  if (g->mem) (void)hipFree(g->mem);
  delete g;
  return 0;
}

static int group_enqueue_step(iqlhip_group *g, hipStream_t st) {
  iqlhip_trainer *t0 = g->tr[0];
  if (!t0->D.prefetch) HIP_TRY(launch_stage(t0->bf16, t0->D, g->gdesc, g->gargs, g->gctr, g->K, st));
  HIP_TRY(launch_forward(t0->bf16, t0->D, g->gdesc, g->gargs, g->gctr, g->K, st));
  HIP_TRY(launch_backward(t0->bf16, t0->D, g->gdesc, g->gargs, g->gctr, g->K, st));
  HIP_TRY(launch_update(t0->bf16, g->gdesc, g->gargs, g->gctr, g->gitems, g->n_items, g->K, st));
  return 0;
}

extern "C" int iqlhip_group_train_steps(iqlhip_group *g, const iqlhip_replay_view *views, int64_t n_steps,
                                        const int64_t *const *idx, const uint8_t *const *dropout_keep,
                                        float *const *losses_out, int32_t graph_unroll, void *stream) {
  if (!g || !views) return fail(IQLHIP_ERR_INVALID, "null argument");
  if (n_steps < 0) return fail(IQLHIP_ERR_INVALID, "n_steps must be >= 0");
  for (int k = 0; k < g->K; ++k) {
    const iqlhip_replay_view &v = views[k];
    const iqlhip_trainer_config &c = g->tr[k]->cfg;
    if (!v.rows) return fail(IQLHIP_ERR_INVALID, "null replay view %d", k);
    if (v.state_dim != c.state_dim || v.action_dim != c.action_dim)
      return fail(IQLHIP_ERR_INVALID, "replay dims (%d,%d) do not match the trainer (%d,%d)", v.state_dim,
                  v.action_dim, c.state_dim, c.action_dim);
    if (v.n_rows <= 0) return fail(IQLHIP_ERR_INVALID, "cannot sample from an empty replay buffer");
    if (v.row_stride != iqlhip_replay_row_stride(v.state_dim, v.action_dim))
      return fail(IQLHIP_ERR_INVALID, "row_stride %d is not the packed layout's", v.row_stride);
  }
  if (n_steps == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  // ---- K DevArgs through one pinned slot, one copy (skipped when the call continues the last) ----
  DevArgs want[IQLHIP_MAX_GROUP];
  bool same = g->tr[0]->D.prefetch && g->dev_args_valid && !g->timing, all_philox = true;
  for (int k = 0; k < g->K; ++k) {
    iqlhip_trainer *t = g->tr[k];
    DevArgs a;
    memset(&a, 0, sizeof(a));
    a.rows = views[k].rows, a.n_rows = views[k].n_rows, a.row_stride = views[k].row_stride;
    a.generation = views[k].generation;
    a.idx = idx ? idx[k] : nullptr;
    a.idx_mode = a.idx ? 1 : 0;
    a.drop_keep = dropout_keep ? dropout_keep[k] : nullptr;
    a.losses_out = losses_out ? losses_out[k] : nullptr;
    a.base_step = t->total_it;
    a.lr_q = t->lr_q, a.lr_v = t->lr_v, a.lr_a_base = t->lr_a_base;
    a.n_steps = n_steps;
    want[k] = a;
    same = same && continues(g->dev_args[k], a);
    all_philox = all_philox && a.idx_mode == 0;
    t->dev_args_valid = false;  // a member's own next call starts from scratch
  }
  if (!same) {
    const int slot = g->harg_head;
    g->harg_head = (slot + 1) % iqlhip_group::ARG_RING;
    if (!g->harg[slot]) {
      HIP_TRY(hipHostMalloc((void **)&g->harg[slot], sizeof(DevArgs) * g->K, hipHostMallocDefault));
      HIP_TRY(hipEventCreateWithFlags(&g->harg_ev[slot], hipEventDisableTiming));
    }
    if (g->harg_used[slot]) HIP_TRY(hipEventSynchronize(g->harg_ev[slot]));
    for (int k = 0; k < g->K; ++k) g->harg[slot][k] = g->dev_args[k] = want[k];
    HIP_TRY(hipMemcpyAsync(g->gargs, g->harg[slot], sizeof(DevArgs) * g->K, hipMemcpyHostToDevice, st));
    HIP_TRY(hipEventRecord(g->harg_ev[slot], st));
    g->harg_used[slot] = true;
    if (g->tr[0]->D.prefetch)
      HIP_TRY(launch_stage(g->tr[0]->bf16, g->tr[0]->D, g->gdesc, g->gargs, g->gctr, g->K, st));
  }
  g->dev_args_valid = all_philox;
  // ---- the steps: hipGraphs of `graph_unroll` steps, the remainder eagerly ----
  int64_t done = 0;
  if (g->timing) {  // one event pair per kernel; diagnostic mode only
    iqlhip_trainer *t0 = g->tr[0];
    for (; done < n_steps; ++done) {
      if (!t0->D.prefetch) HIP_TRY(launch_stage(t0->bf16, t0->D, g->gdesc, g->gargs, g->gctr, g->K, st));
      HIP_TRY(hipEventRecord(g->ev[0], st));
      HIP_TRY(launch_forward(t0->bf16, t0->D, g->gdesc, g->gargs, g->gctr, g->K, st));
      HIP_TRY(hipEventRecord(g->ev[1], st));
      HIP_TRY(launch_backward(t0->bf16, t0->D, g->gdesc, g->gargs, g->gctr, g->K, st));
      HIP_TRY(hipEventRecord(g->ev[2], st));
      HIP_TRY(launch_update(t0->bf16, g->gdesc, g->gargs, g->gctr, g->gitems, g->n_items, g->K, st));
      HIP_TRY(hipEventRecord(g->ev[3], st));
      HIP_TRY(hipEventRecord(g->ev[4], st));
      HIP_TRY(hipEventSynchronize(g->ev[4]));
      for (int k = 0; k < 3; ++k) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, g->ev[k], g->ev[k + 1]));
        g->t_acc[k] += ms;
      }
      float ems = 0.f;
      HIP_TRY(hipEventElapsedTime(&ems, g->ev[3], g->ev[4]));
      g->t_empty += ems;
      g->t_n++;
    }
    for (int k = 0; k < g->K; ++k) g->tr[k]->total_it += n_steps;
    return 0;
  }
  if (graph_unroll > 0 && n_steps >= graph_unroll) {
    if (!g->gexec || g->graph_unroll != graph_unroll) {
      if (g->gexec) {
        (void)hipGraphExecDestroy(g->gexec);
        g->gexec = nullptr;
      }
      hipGraph_t gr = nullptr;
      if (!g->cap_stream) HIP_TRY(capture_stream(&g->cap_stream));
      HIP_TRY(hipStreamBeginCapture(g->cap_stream, hipStreamCaptureModeThreadLocal));
      int rc = 0;
      for (int u = 0; u < graph_unroll && !rc; ++u) rc = group_enqueue_step(g, g->cap_stream);
      hipError_t ce = hipStreamEndCapture(g->cap_stream, &gr);
      if (rc) return rc;
      HIP_TRY(ce);
      HIP_TRY(hipGraphInstantiate(&g->gexec, gr, nullptr, nullptr, 0));
      (void)hipGraphDestroy(gr);
      g->graph_unroll = graph_unroll;
    }
    for (; done + graph_unroll <= n_steps; done += graph_unroll) {
      HIP_TRY(hipGraphLaunch(g->gexec, st));
      HIP_TRY(g->throttle.queued(3 * (int64_t)graph_unroll, st));
    }
  }
  for (; done < n_steps; ++done) {
    if (int rc = group_enqueue_step(g, st)) return rc;
    HIP_TRY(g->throttle.queued(3, st));
  }
  for (int k = 0; k < g->K; ++k) g->tr[k]->total_it += n_steps;
  return 0;
}


extern "C" int iqlhip_group_set_timing(iqlhip_group *g, int32_t enable) {
  if (!g) return fail(IQLHIP_ERR_INVALID, "null group");
  g->timing = enable != 0;
  g->t_acc[0] = g->t_acc[1] = g->t_acc[2] = 0, g->t_empty = 0, g->t_n = 0;
  if (g->timing)
    for (auto &e : g->ev)
      if (!e) HIP_TRY(hipEventCreate(&e));
  return 0;
}

extern "C" int iqlhip_stream_create_cu_slice(void **stream, int32_t slice, int32_t n_slices) {
  if (!stream || n_slices < 1 || slice < 0 || slice >= n_slices)
    return fail(IQLHIP_ERR_INVALID, "stream slice %d of %d", slice, n_slices);
  int dev = 0, cus = 0;
  HIP_TRY(hipGetDevice(&dev));
  HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  if (cus < n_slices) return fail(IQLHIP_ERR_INVALID, "%d slices of %d compute units", n_slices, cus);
  std::vector<uint32_t> mask((cus + 31) / 32, 0u);
  for (int cu = slice; cu < cus; cu += n_slices) mask[cu >> 5] |= 1u << (cu & 31);
  hipStream_t st = nullptr;
  HIP_TRY(hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()));
  *stream = st;
  return 0;
}
extern "C" int iqlhip_stream_destroy(void *stream) {
  if (stream) HIP_TRY(hipStreamDestroy((hipStream_t)stream));
  return 0;
}

extern "C" int iqlhip_group_get_timing(iqlhip_group *g, double avg_ms[3], int64_t *n) {
  if (!g) return fail(IQLHIP_ERR_INVALID, "null group");
  for (int k = 0; k < 3; ++k) {
    const double v = g->t_n ? (g->t_acc[k] - g->t_empty) / (double)g->t_n : 0.0;
    avg_ms[k] = v > 0 ? v : 0.0;
  }
  if (n) *n = g->t_n;
  return 0;
}

extern "C" int iqlhip_forward(iqlhip_trainer *t, int32_t which, const float *s, const float *a, int64_t n,
                              float *out, void *stream) {
  if (!t || !s || !out) return fail(IQLHIP_ERR_INVALID, "null argument");
  if (n <= 0) return fail(IQLHIP_ERR_INVALID, "n must be positive");
  if ((which == 0 || which == 3) && !a) return fail(IQLHIP_ERR_INVALID, "Q forward needs actions");
  hipStream_t st = (hipStream_t)stream;
  if (t->deep) {
    if (which < 0 || which > 3) return fail(IQLHIP_ERR_INVALID, "which must be 0..3");
    HIP_TRY(deep_infer(t->deep, which, s, a, n, out, st));
    return 0;
  }
  const int E = t->D.E;
  FwdNet N;
  switch (which) {
    case 0:  // q_1 .. q_E -> out[n][E]
      for (int e = 0; e < E; ++e) {
        N = t->D.fwd[e], N.out_col = e, N.train_slot = -1, N.stage = 0;
        HIP_TRY(launch_infer(t->bf16, t->D, t->ddesc, N, s, a, n, out, E, st));
      }
      return 0;
    case 1:
      N = t->D.fwd[E], N.out_col = 0, N.train_slot = -1;
      HIP_TRY(launch_infer(t->bf16, t->D, t->ddesc, N, s, a, n, out, 1, st));
      return 0;
    case 2:  // eval-mode actor: no dropout (ref:299 actor.eval())
      N = t->D.fwd[E + 1], N.out_col = 0, N.train_slot = -1, N.dropout = 0;
      HIP_TRY(launch_infer(t->bf16, t->D, t->ddesc, N, s, a, n, out, t->cfg.action_dim, st));
      return 0;
    case 3:  // target critics -> out[n][E]
      for (int e = 0; e < E; ++e) {
        N = t->D.fwd[E + 2 + e], N.out_col = e;
        HIP_TRY(launch_infer(t->bf16, t->D, t->ddesc, N, s, a, n, out, E, st));
      }
      return 0;
    default:
      return fail(IQLHIP_ERR_INVALID, "which must be 0..3");
  }
}

extern "C" int iqlhip_mlp_forward(const iqlhip_mlp_desc *d, const float *x, int64_t n, int32_t x_stride,
                                  float *out, int32_t out_stride, void *stream) {
  if (!d || !x || !out) return fail(IQLHIP_ERR_INVALID, "null argument");
  if (n <= 0) return fail(IQLHIP_ERR_INVALID, "n must be positive");
  if (d->n_layers < 1 || d->n_layers > IQLHIP_MLP_MAX_LAYERS)
    return fail(IQLHIP_ERR_INVALID, "n_layers must be in [1, %d]", IQLHIP_MLP_MAX_LAYERS);
  for (int i = 0; i <= d->n_layers; ++i)
    if (d->dims[i] <= 0 || d->dims[i] > 1024)
      return fail(IQLHIP_ERR_UNSUPPORTED, "layer width %d outside [1, 1024]", d->dims[i]);
  for (int i = 0; i < d->n_layers; ++i)
    if (!d->weights[i] || !d->biases[i]) return fail(IQLHIP_ERR_INVALID, "null weight pointer");
  for (int a : {d->hidden_act, d->out_act})
    if (a < 0 || (a > 1 && a < 8) || a > 15) return fail(IQLHIP_ERR_INVALID, "activation code %d", a);
  if (x_stride < d->dims[0] || out_stride < d->dims[d->n_layers])
    return fail(IQLHIP_ERR_INVALID, "stride smaller than the row width");
  if (!(d->dropout_p < 1.0f)) return fail(IQLHIP_ERR_INVALID, "dropout_p must be < 1");
  if (d->dropout_p > 0.f && n > 0xffffffffLL) return fail(IQLHIP_ERR_UNSUPPORTED, "dropout: n must fit 32 bits");
  HIP_TRY(launch_mlp_f32(*d, x, n, x_stride, out, out_stride, (hipStream_t)stream));
  return 0;
}

// Diagnostic: attach a device buffer [3][512][8][2] u64 for IQL_STAMPS builds.
extern "C" int iqlhip_trainer_set_debug(iqlhip_trainer *t, void *buf) {
  if (!t) return fail(IQLHIP_ERR_INVALID, "null trainer");
  if (t->deep) return fail(IQLHIP_ERR_UNSUPPORTED, "no stamps in the general step");
  t->D.dbg = reinterpret_cast<unsigned long long *>(buf);
  HIP_TRY(hipMemcpy(t->ddesc, &t->D, sizeof(TrainerDesc), hipMemcpyHostToDevice));
  if (t->gexec) {
    (void)hipGraphExecDestroy(t->gexec);
    t->gexec = nullptr;
  }
  return 0;
}

extern "C" int iqlhip_cvar_tail_mean(const float *preds, int32_t S, int64_t N, int32_t n_tail, float *out,
                                     void *stream) {
  if (!preds || !out) return fail(IQLHIP_ERR_INVALID, "null argument");
  if (S < 1 || N < 1) return fail(IQLHIP_ERR_INVALID, "S and N must be positive");
  if (n_tail < 1 || n_tail > S) return fail(IQLHIP_ERR_INVALID, "n_tail must be in [1, S]");
  if (S > 2400) return fail(IQLHIP_ERR_UNSUPPORTED, "S = %d > 2400", S);
  HIP_TRY(launch_cvar(preds, S, N, n_tail, out, (hipStream_t)stream));
  return 0;
}

extern "C" int iqlhip_pt_relabel(const iqlhip_pt_weights *w, const float *obs, const float *act,
                                 int64_t n_rows, const int64_t *win_start, const int32_t *win_len,
                                 const int32_t *win_t0, int64_t n_win, int32_t query_length, float *out,
                                 void *stream) {
  if (!w || !obs || !act || !win_start || !win_len || !out) return fail(IQLHIP_ERR_INVALID, "null argument");
  if (n_win <= 0 || n_rows <= 0 || query_length < 1) return fail(IQLHIP_ERR_INVALID, "empty problem");
  if (w->embd_dim != 64) return fail(IQLHIP_ERR_UNSUPPORTED, "embd_dim %d: the kernel is built for 64", w->embd_dim);
  if (w->num_layers != 1) return fail(IQLHIP_ERR_UNSUPPORTED, "num_layers %d: only 1 is built", w->num_layers);
  const int nh = w->num_heads;
  if (nh < 1 || nh > 16 || (nh & (nh - 1))) return fail(IQLHIP_ERR_UNSUPPORTED, "num_heads %d", nh);
  if (w->inter_dim < 64 || w->inter_dim > 1024 || w->inter_dim % 256)
    return fail(IQLHIP_ERR_UNSUPPORTED, "inter_dim %d: must be a multiple of 256 up to 1024", w->inter_dim);
  if (w->state_dim < 1 || w->action_dim < 1 || w->state_dim + w->action_dim > 192)
    return fail(IQLHIP_ERR_UNSUPPORTED, "state/action dims");
  if (w->n_temb < 1) return fail(IQLHIP_ERR_INVALID, "empty timestep table");
  if (!win_t0 && query_length > w->n_temb)
    return fail(IQLHIP_ERR_INVALID, "query_length exceeds the timestep table");
  if (pt_smem_bytes(*w, query_length) > 160 * 1024)
    return fail(IQLHIP_ERR_UNSUPPORTED, "query_length %d does not fit the 160 KiB LDS", query_length);
  HIP_TRY(launch_pt(*w, obs, act, n_rows, win_start, win_len, win_t0, n_win, query_length, out,
                    (hipStream_t)stream));
  return 0;
}
