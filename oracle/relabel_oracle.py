"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's dataset /
reward-relabel path (plain Python loops + numpy, small cases).

"ref" = /root/reference/algorithms/offline/iql.py.  Pinned against
tests/golden/dataset_ops.npz (captured from the reference) by
tests/test_relabel_oracle.py, EXCEPT the two reward networks themselves:

  * RewardMLP lives in the un-vendored submodule gp_reward-priors (ml4ai/gp-reward-priors,
    no pinned SHA in the mounted tree): its call-site contract (x @ W + b layers,
    parameter order, ref:953-972,1326-1336) is restated here -- PARITY UNPINNED for
    the class itself, pinned for everything the reference does around it.
  * The preference transformer exists in the reference only as JAX/Flax source
    (reward_models/pref_transformer.py:10-277, reward_models/ops.py:6-117; jax is
    not installed): restated from source below -- PARITY UNPINNED.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.
"""
import math

import numpy as np

from .iql_oracle import bf16

F32 = np.float32


# --------------------------------------------------------------------------- #
# keep mask + episode step (ref:701-716 and ref:1236-1253, the D4RL loop)
# --------------------------------------------------------------------------- #
def keep_mask_and_steps(terminals, timeouts, max_episode_steps, terminate_on_end=False):
    N = len(terminals)
    keep = np.ones(N - 1, dtype=bool)
    ep_steps = np.zeros(N - 1, dtype=np.int64)
    ep = 0
    for i in range(N - 1):
        ep_steps[i] = ep
        done_bool = bool(terminals[i])
        final = bool(timeouts[i]) if timeouts is not None else ep == max_episode_steps - 1
        if (not terminate_on_end) and final:
            keep[i] = False
            ep = 0
            continue
        if done_bool or final:
            ep = 0
        ep += 1
    return keep, ep_steps


# --------------------------------------------------------------------------- #
# reward post-processing (ref:344-401)
# --------------------------------------------------------------------------- #
def return_reward_range(rewards, terminals, max_episode_steps):
    returns, lengths = [], []
    ep_ret, ep_len = 0.0, 0
    i = 0
    trj_lens = np.zeros(rewards.shape[0])
    for j, (r, d) in enumerate(zip(rewards, terminals)):
        ep_ret += float(r)
        ep_len += 1
        trj_lens[i:j + 1] = ep_len
        if d or ep_len == max_episode_steps:
            i = j + 1
            returns.append(ep_ret)
            lengths.append(ep_len)
            ep_ret, ep_len = 0.0, 0
    lengths.append(ep_len)
    assert sum(lengths) == len(rewards)
    return min(returns), max(returns), trj_lens


def modify_reward(rewards, terminals, env_name, normalize_reward, max_episode_steps=1000):
    """Returns the modified copy (the reference mutates dataset["rewards"] in place)."""
    r = np.array(rewards, copy=True)
    rng = lambda: return_reward_range(r, terminals, max_episode_steps)
    if any(s in env_name for s in ("halfcheetah", "hopper", "walker2d")):
        mn, mx, _ = rng()
        r /= mx - mn
        r *= max_episode_steps
    elif "antmaze" in env_name:
        if normalize_reward == 1:
            r -= 1.0
        elif normalize_reward in (2, 3):
            mn, mx, _ = rng()
            r /= mx - mn
            r *= max_episode_steps
            if normalize_reward == 3:
                r -= 1.0
        elif normalize_reward in (4, 5):
            mn, mx, _ = rng()
            r -= mn
            r /= mx - mn
            r *= max_episode_steps
            if normalize_reward == 5:
                r -= 1.0
        else:
            mn, mx, tl = rng()
            r -= mn / tl
            r /= mx - mn
            r *= max_episode_steps
            if normalize_reward != 6:
                r -= 1.0
    return r


# --------------------------------------------------------------------------- #
# CVaR (ref:735-827, ref:1003-1011)
# --------------------------------------------------------------------------- #
def n_tail_of(alpha, S):
    return max(1, int(np.floor((1.0 - alpha) * S)))


def empirical_cvar(samples, alpha):
    if not (0.0 <= alpha < 1.0):
        raise ValueError(f"alpha must be in [0, 1), got {alpha!r}")
    s = np.sort(samples)
    return float(s[:n_tail_of(alpha, len(samples))].mean())


def cvar_tail_mean(all_preds, alpha):
    """Vectorised partition CVaR over axis 0 (ref:1009-1011 / 1185-1187)."""
    S = all_preds.shape[0]
    n_tail = n_tail_of(alpha, S)
    kth = min(n_tail, S - 1)
    part = np.partition(all_preds, kth, axis=0)
    return part[:n_tail].mean(axis=0).astype(np.float32)


def cvar_stability_check(all_preds, alpha, n_checks=50):
    if alpha == 0.0:
        return 0.0
    S, N = all_preds.shape
    rng = np.random.default_rng(seed=42)
    indices = rng.choice(N, size=min(n_checks, N), replace=False)
    ratios = []
    for i in indices:
        full = empirical_cvar(all_preds[:, i], alpha)
        half = empirical_cvar(all_preds[:S // 2, i], alpha)
        if abs(full) > 1e-8:
            ratios.append(abs(full - half) / abs(full))
    return float(np.mean(ratios)) if ratios else float("nan")


# --------------------------------------------------------------------------- #
# Markovian reward MLP (call-site contract of the absent optbnn RewardMLP)
# --------------------------------------------------------------------------- #
def reward_mlp_forward(weights, x, activation="relu"):
    """weights: [W0, b0, W1, b1, ..., Wout, bout] with W shaped [in, out]."""
    act = {"relu": lambda v: np.maximum(v, 0), "tanh": np.tanh}[activation]
    h = np.asarray(x, dtype=F32)
    n = len(weights) // 2
    for l in range(n):
        h = (h @ np.asarray(weights[2 * l], dtype=F32) + np.asarray(weights[2 * l + 1], dtype=F32)).astype(F32)
        if l < n - 1:
            h = act(h).astype(F32)
    return h


def relabel_dataset(dataset, rewards, keep):
    obs = dataset["observations"].astype(np.float32)
    act = dataset["actions"].astype(np.float32)
    return {
        "observations": obs[:-1][keep],
        "actions": act[:-1][keep],
        "next_observations": obs[1:][keep],
        "rewards": rewards[keep],
        "terminals": dataset["terminals"][:-1][keep],
    }


def qlearning_dataset_mr(dataset, weights, max_episode_steps, terminate_on_end=False, activation="relu"):
    """ref:691-732"""
    keep, _ = keep_mask_and_steps(dataset["terminals"], dataset.get("timeouts"), max_episode_steps,
                                  terminate_on_end)
    obs_act = np.concatenate([dataset["observations"][:-1], dataset["actions"][:-1]], axis=1).astype(F32)
    r = reward_mlp_forward(weights, obs_act, activation)[:, 0]
    return relabel_dataset(dataset, r, keep)


def qlearning_dataset_ensemble(dataset, weight_sets, alpha, max_episode_steps, activation="relu"):
    """ref:1085-1220 (MR snapshots) and ref:830-1044 (BNN): S forwards + tail mean."""
    keep, _ = keep_mask_and_steps(dataset["terminals"], dataset.get("timeouts"), max_episode_steps)
    obs_act = np.concatenate([dataset["observations"][:-1], dataset["actions"][:-1]], axis=1).astype(F32)
    preds = np.stack([reward_mlp_forward(w, obs_act, activation)[:, 0] for w in weight_sets])
    return relabel_dataset(dataset, cvar_tail_mean(preds, alpha), keep), preds


# --------------------------------------------------------------------------- #
# preference-transformer windows (ref:1255-1292), bug-compatible by default:
# the window rows are episode-RELATIVE indices applied to the GLOBAL arrays
# --------------------------------------------------------------------------- #
def pt_windows(obs_all, act_all, ep_steps, query_length, correct_window_offsets=False):
    n = len(ep_steps)
    s_dim, a_dim = obs_all.shape[1], act_all.shape[1]
    sts = np.zeros((n, query_length, s_dim), dtype=np.float32)
    acts = np.zeros((n, query_length, a_dim), dtype=np.float32)
    ts = np.zeros((n, query_length), dtype=np.int64)
    am = np.zeros((n, query_length), dtype=np.float32)
    for j in range(n):
        ep = int(ep_steps[j])
        if ep >= query_length:
            start = ep - query_length + 1
            if correct_window_offsets:
                start = j - query_length + 1
            sts[j] = obs_all[start:start + query_length]
            acts[j] = act_all[start:start + query_length]
            ts[j] = np.arange(query_length)
            am[j] = 1.0
        else:
            seq_len = ep + 1
            pad = query_length - seq_len
            start = (j - ep) if correct_window_offsets else 0
            sts[j, pad:] = obs_all[start:start + seq_len]
            acts[j, pad:] = act_all[start:start + seq_len]
            ts[j, pad:] = np.arange(seq_len)
            am[j, pad:] = 1.0
    return sts, acts, ts, am


# --------------------------------------------------------------------------- #
# preference transformer forward (reward_models/pref_transformer.py + ops.py)
# params: dict of float32 arrays, torch [out][in] layouts:
#   state_linear.{weight,bias} action_linear.{weight,bias} timestep_embed.weight
#   stacked_layer_norm.{weight,bias}
#   gpt.layers.{l}.layer_norm_0.{weight,bias}  gpt.layers.{l}.attention.in_linear.{weight,bias}
#   gpt.layers.{l}.attention.out_linear.{weight,bias}  gpt.layers.{l}.layer_norm_1.{weight,bias}
#   gpt.layers.{l}.mlp.in_linear.{weight,bias}  gpt.layers.{l}.mlp.out_linear.{weight,bias}
#   gpt.layer_norm.{weight,bias}  pref_linear.{weight,bias}
# --------------------------------------------------------------------------- #
def _ln(x, w, b, eps):
    mu = x.mean(-1, keepdims=True, dtype=F32)
    var = ((x - mu) ** 2).mean(-1, keepdims=True, dtype=F32)
    return ((x - mu) / np.sqrt(var + F32(eps)) * w + b).astype(F32)


def _lin(x, p, name):
    return (x @ p[name + ".weight"].T + p[name + ".bias"]).astype(F32)


def pt_value_last(p, states, actions, timesteps, attn_mask, num_heads=4, eps=1e-5, all_positions=False):
    """value[:, 0, -1, 0] of PT.__call__ (pref_transformer.py:210-277), eval mode.

    states [B,QL,S], actions [B,QL,A], timesteps [B,QL] int, attn_mask [B,QL]."""
    B, QL = states.shape[:2]
    E = p["state_linear.weight"].shape[0]
    temb = p["timestep_embed.weight"][timesteps]  # [B,QL,E]
    es = _lin(states.astype(F32), p, "state_linear") + temb
    ea = _lin(actions.astype(F32), p, "action_linear") + temb
    x = np.stack([es, ea], axis=2).reshape(B, 2 * QL, E)  # s0,a0,s1,a1,... (:221-225)
    x = _ln(x, p["stacked_layer_norm.weight"], p["stacked_layer_norm.bias"], eps)
    m2 = np.stack([attn_mask, attn_mask], axis=2).reshape(B, 2 * QL).astype(F32)
    add_mask = ((F32(1.0) - m2) * F32(-10000.0))[:, None, None, :]  # ops.py:6-11
    T = 2 * QL
    causal = np.tril(np.ones((T, T), dtype=bool))[None, None]
    hd = E // num_heads
    n_layers = 0
    while f"gpt.layers.{n_layers}.layer_norm_0.weight" in p:
        n_layers += 1
    for l in range(n_layers):
        pre = f"gpt.layers.{l}."
        res = x
        h = _ln(x, p[pre + "layer_norm_0.weight"], p[pre + "layer_norm_0.bias"], eps)
        qkv = _lin(h, p, pre + "attention.in_linear")
        q, k, v = np.split(qkv, 3, axis=2)
        sh = lambda t: t.reshape(B, T, num_heads, hd).transpose(0, 2, 1, 3)
        q, k, v = sh(q), sh(k), sh(v)
        # ops.py:74-76: query/key cast to bf16; the product is a bf16 tensor
        w = bf16((bf16(q) @ bf16(k).transpose(0, 1, 3, 2)).astype(F32))
        w = bf16(w / F32(float(hd) ** 0.5))
        w = np.where(causal, w, bf16(np.asarray(-1e4, dtype=F32)))
        w = w.astype(F32) + add_mask
        w = w - w.max(-1, keepdims=True)
        pw = np.exp(w).astype(F32)
        pw = (pw / pw.sum(-1, keepdims=True, dtype=F32)).astype(F32)
        o = (pw @ v).astype(F32).transpose(0, 2, 1, 3).reshape(B, T, E)
        x = _lin(o, p, pre + "attention.out_linear") + res
        res = x
        h = _ln(x, p[pre + "layer_norm_1.weight"], p[pre + "layer_norm_1.bias"], eps)
        h = np.maximum(_lin(h, p, pre + "mlp.in_linear"), 0).astype(F32)
        x = _lin(h, p, pre + "mlp.out_linear") + res
    x = _ln(x, p["gpt.layer_norm.weight"], p["gpt.layer_norm.bias"], eps)
    hidden = x.reshape(B, QL, 2, E)[:, :, 1]  # action tokens (:241-242)
    out = _lin(hidden, p, "pref_linear")
    if all_positions:
        return out[:, :, -1]  # value at every timestep of the window (custom_offline/iql.py:176-192)
    return out[:, -1, -1]  # value = last output column, last timestep


def qlearning_dataset_pt(dataset, p, max_episode_steps, query_length, num_heads=4, eps=1e-5,
                         correct_window_offsets=False):
    """ref:1223-1309 with the restated PT as r_model."""
    keep, ep_steps = keep_mask_and_steps(dataset["terminals"], dataset.get("timeouts"), max_episode_steps)
    obs = dataset["observations"].astype(F32)
    act = dataset["actions"].astype(F32)
    sts, acts, ts, am = pt_windows(obs, act, ep_steps, query_length, correct_window_offsets)
    r = pt_value_last(p, sts, acts, ts, am, num_heads, eps).astype(F32)
    return relabel_dataset(dataset, r, keep)


def custom_qlearning_dataset(episodes, p, query_length, num_heads=4, eps=1e-5, value_fn=None):
    """algorithms/custom_offline/iql.py:158-225 (query_length > 1) with the restated PT as r_model.
    ``episodes``: list of dicts with observations [L+1,S], actions [L,A], terminations [L].
    Episodes no longer than the query are labelled by ONE forward over the whole episode (values
    at every position); longer ones by that forward for the first query_length steps and then one
    rolling window per step with the TRUE timesteps i+1-QL .. i (custom_offline:197-211).
    ``value_fn(states, actions, timesteps, mask) -> [1, L]`` per-position values replaces the
    restated PT (tests: the stand-in model the reference's own loop was recorded with, which
    pins this loop -- windows, timesteps, which positions are kept -- to the reference)."""
    if value_fn is None:
        def value_fn(sts, acts, ts, am):
            return pt_value_last(p, sts, acts, ts, am, num_heads, eps, all_positions=True)
    obs, nxt, acts, rews, dones = [], [], [], [], []
    for ep in episodes:
        o, a = np.asarray(ep["observations"], F32), np.asarray(ep["actions"], F32)
        L = a.shape[0]
        if L <= query_length:
            r = np.asarray(value_fn(o[:-1][None], a[None], np.arange(L)[None], np.ones((1, L), F32))).reshape(L)
        else:
            r = np.zeros(L, F32)
            QL = query_length
            r[:QL] = np.asarray(value_fn(o[:-1][:QL][None], a[:QL][None], np.arange(QL)[None],
                                         np.ones((1, QL), F32))).reshape(QL)
            for i in range(QL, L):
                sl = slice(i - QL + 1, i + 1)
                r[i] = np.asarray(value_fn(o[:-1][sl][None], a[sl][None], np.arange(i + 1 - QL, i + 1)[None],
                                           np.ones((1, QL), F32))).reshape(QL)[-1]
        obs.append(o[:-1]), nxt.append(o[1:]), acts.append(a), rews.append(r.astype(F32))
        dones.append(np.asarray(ep["terminations"]))
    return {"observations": np.concatenate(obs), "actions": np.concatenate(acts),
            "next_observations": np.concatenate(nxt), "rewards": np.concatenate(rews),
            "terminals": np.concatenate(dones)}


def make_pt_params(rng, state_dim, action_dim, max_episode_steps, embd=64, pref=64, inter=256, layers=1):
    """Seeded random PT weights (tests / bench): N(0, 0.2) linears, LN weight 1 + noise."""
    p = {}
    def lin(name, o, i):
        p[name + ".weight"] = (rng.standard_normal((o, i)) * (1.0 / math.sqrt(i))).astype(F32)
        p[name + ".bias"] = (rng.standard_normal(o) * 0.1).astype(F32)
    def ln(name):
        p[name + ".weight"] = (1.0 + 0.1 * rng.standard_normal(embd)).astype(F32)
        p[name + ".bias"] = (0.1 * rng.standard_normal(embd)).astype(F32)
    lin("state_linear", embd, state_dim)
    lin("action_linear", embd, action_dim)
    p["timestep_embed.weight"] = (rng.standard_normal((max_episode_steps + 1, embd)) * 0.5).astype(F32)
    ln("stacked_layer_norm")
    for l in range(layers):
        pre = f"gpt.layers.{l}."
        ln(pre + "layer_norm_0")
        lin(pre + "attention.in_linear", 3 * embd, embd)
        lin(pre + "attention.out_linear", embd, embd)
        ln(pre + "layer_norm_1")
        lin(pre + "mlp.in_linear", inter, embd)
        lin(pre + "mlp.out_linear", embd, inter)
    ln("gpt.layer_norm")
    lin("pref_linear", 2 * pref + 1, embd)
    return p


# --------------------------------------------------------------------------- #
# eval_actor's episode accounting  (ref:296-333), as the reference loops it
# --------------------------------------------------------------------------- #
def eval_accounting(steps, n_envs, n_episodes, is_antmaze):
    """``steps`` = iterable of (rewards[n_envs], dones[n_envs]) as ``vec_env.step`` returns them.
    Returns (scores[n_episodes], steps_to_goal, number of environment steps consumed)."""
    ep_rewards = np.zeros(n_envs, dtype=np.float64)
    ep_steps = np.zeros(n_envs, dtype=np.int64)
    completed, steps_to_goal, used = [], [], 0
    it = iter(steps)
    while len(completed) < n_episodes:            # ref:304
        rewards, dones = next(it)
        used += 1
        ep_rewards += rewards                      # ref:321
        ep_steps += 1                              # ref:322
        for i in range(n_envs):                    # ref:324
            if dones[i]:
                if is_antmaze and ep_rewards[i] > 0.5:   # ref:328
                    steps_to_goal.append(int(ep_steps[i]))
                completed.append(float(ep_rewards[i]))   # ref:330
                ep_rewards[i] = 0.0
                ep_steps[i] = 0
                if len(completed) >= n_episodes:          # ref:333
                    break
    return np.asarray(completed[:n_episodes]), steps_to_goal, used
