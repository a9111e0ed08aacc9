"""Dataset preparation and reward relabelling (reference rows A12-A15 of SURVEY.md
section 8): the host logic of ``algorithms/offline/iql.py:344-401, 691-1390`` with
the network forwards and the ensemble CVaR running in HIP kernels.

"ref:" = /root/reference/algorithms/offline/iql.py.  Same function names,
signatures, return dicts and exceptions as the reference; the Python loops over N
transitions (ref:701-716, 1236-1253, 344-360) are replaced by closed-form numpy.
"""
import ctypes as C
import glob as _glob
import os
import re
import warnings
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn
import yaml

from . import _lib
from ._lib import check, ptr, stream_ptr
from .iql import mlp_forward_f32


# --------------------------------------------------------------------------- #
# keep mask / episode step  (ref:701-716, ref:1236-1253)
# --------------------------------------------------------------------------- #
def keep_mask_and_steps(terminals, timeouts, max_episode_steps: int, terminate_on_end: bool = False,
                        device=None):
    """Vectorised form of the reference's single pass over N-1 transitions.  With ``device``
    (a ROCm device) the scan runs in the kernels of csrc/prep.hip and the two arrays are
    copied back; the host form below is the same computation in numpy.

    Returns (keep[N-1] bool, ep_steps[N-1] int64).  The episode-step counter is 0
    after a dropped (final, not terminate_on_end) transition and 1 after a kept
    terminal / final one -- the D4RL quirk the reference inherits (ref:713-716)."""
    if device is not None:
        from . import prep
        keep_t, steps_t = prep.keep_mask_and_steps(terminals, timeouts, max_episode_steps, terminate_on_end,
                                                   device)
        return keep_t.cpu().numpy(), steps_t.cpu().numpy()
    term = np.asarray(terminals).reshape(-1).astype(bool)[:-1]
    n = term.shape[0]
    if timeouts is not None:
        final = np.asarray(timeouts).reshape(-1).astype(bool)[:-1]
        keep = np.ones(n, dtype=bool) if terminate_on_end else ~final
        # value of `ep` right after transition i: 0 (dropped), 1 (kept reset), else previous + 1
        reset0 = final & (not terminate_on_end)
        reset1 = (term | final) & ~reset0
        return keep, _steps_from_resets(reset0, reset1)
    # no timeouts: final = (ep == max_episode_steps - 1) depends on the counter itself:
    #   ep' = 0 if final and not terminate_on_end, 1 if done or final, else ep + 1.
    # Between two terminals the counter is an arithmetic ramp followed by a periodic
    # sawtooth, so each terminal-delimited segment has a closed form; the segments
    # (one per terminal, ~1e3-1e4 for a D4RL dataset) are chained on the host.
    M = int(max_episode_steps)
    ep = np.zeros(n, dtype=np.int64)
    keep = np.ones(n, dtype=bool)
    bounds = np.flatnonzero(term)
    seg_ends = np.concatenate([bounds + 1, [n]])
    base = 1 if terminate_on_end else 0  # counter after a `final` transition
    a, e0 = 0, 0
    for b in seg_ends:
        if b <= a:
            continue
        t = np.arange(b - a, dtype=np.int64)
        if e0 <= M - 1:
            f0 = M - 1 - e0  # offset of the first final transition
            if base <= M - 1:
                vals = np.where(t > f0, (t - f0 - 1) % (M - base) + base, e0 + t)
            else:  # M == 1 with terminate_on_end: the counter never returns to M - 1
                vals = np.where(t > f0, base + (t - f0 - 1), e0 + t)
        else:
            vals = e0 + t
        ep[a:b] = vals
        fin = vals == M - 1
        if not terminate_on_end:
            keep[a:b] = ~fin
        # counter after the segment's last transition (the terminal, when b - 1 is one)
        if fin[-1]:
            e0 = base
        elif b - 1 < n and term[b - 1]:
            e0 = 1
        else:
            e0 = int(vals[-1]) + 1
        a = b
    return keep, ep


def _steps_from_resets(reset0, reset1):
    """ep_steps[i] = counter BEFORE transition i, given per-transition resets of the
    counter AFTER it: reset0 -> 0, reset1 -> 1, otherwise previous + 1."""
    n = reset0.shape[0]
    idx = np.arange(n, dtype=np.int64)
    any_reset = reset0 | reset1
    last = np.where(any_reset, idx, -1)
    last = np.maximum.accumulate(last)
    after = np.where(last >= 0, (idx - last) + np.where(reset1[np.maximum(last, 0)], 1, 0), idx + 1)
    # `after[i]` = counter after transition i; shift to get the value before it
    ep = np.empty(n, dtype=np.int64)
    ep[0] = 0
    ep[1:] = after[:-1]
    return ep


# --------------------------------------------------------------------------- #
# reward post-processing  (ref:344-401)
# --------------------------------------------------------------------------- #
def return_reward_range(dataset, max_episode_steps):
    """ref:344-360 without the O(N L) Python loop.  Returns (min_ret, max_ret, trj_lens)."""
    rewards = np.asarray(dataset["rewards"]).reshape(-1)
    term = np.asarray(dataset["terminals"]).reshape(-1).astype(bool)
    n = rewards.shape[0]
    idx = np.arange(n, dtype=np.int64)
    # an episode ends at a terminal or after max_episode_steps transitions
    tb = np.flatnonzero(term)
    starts = np.concatenate([[0], tb + 1])
    starts = starts[starts < n]
    seg_start = starts[np.searchsorted(starts, idx, side="right") - 1]
    boundary = term | ((idx - seg_start) % max_episode_steps + 1 == max_episode_steps)
    ends = np.flatnonzero(boundary)
    assert ends.size, "dataset holds no complete episode"
    ep_starts = np.concatenate([[0], ends + 1])  # last entry: start of the trailing partial episode
    red_at = ep_starts[ep_starts < n]
    sums = np.add.reduceat(rewards.astype(np.float64), red_at)[:ends.size]  # ep_ret += float(r)
    ep_id = np.searchsorted(ep_starts, idx, side="right") - 1
    ep_end = np.concatenate([ends, [n - 1]])[ep_id]
    trj_lens = (ep_end - ep_starts[ep_id] + 1).astype(np.float64)
    return float(sums.min()), float(sums.max()), trj_lens


def modify_reward(dataset, env_name, normalize_reward, max_episode_steps=1000):
    """ref:363-401, in place on dataset["rewards"]."""
    if any(s in env_name for s in ("halfcheetah", "hopper", "walker2d")):
        min_ret, max_ret, _ = return_reward_range(dataset, max_episode_steps)
        dataset["rewards"] /= max_ret - min_ret
        dataset["rewards"] *= max_episode_steps
    elif "antmaze" in env_name:
        if normalize_reward == 1:
            dataset["rewards"] -= 1.0
            return
        min_ret, max_ret, trj_lens = return_reward_range(dataset, max_episode_steps)
        if normalize_reward in (2, 3):
            pass
        elif normalize_reward in (4, 5):
            dataset["rewards"] -= min_ret
        else:
            dataset["rewards"] -= min_ret / trj_lens
        dataset["rewards"] /= max_ret - min_ret
        dataset["rewards"] *= max_episode_steps
        if normalize_reward not in (2, 4, 6):
            dataset["rewards"] -= 1.0


# --------------------------------------------------------------------------- #
# CVaR helpers  (ref:735-827)
# --------------------------------------------------------------------------- #
def _tail_count(alpha: float, n: int) -> int:
    """Size of the CVaR tail of n samples at risk level alpha (ref:762, 935, 1152)."""
    return max(1, int(np.floor((1.0 - alpha) * n)))


def _tail_means(cols: np.ndarray, n_tail: int) -> np.ndarray:
    """Mean of the n_tail smallest entries of every column of ``cols`` [S, C], summed in
    ascending order in the input precision -- what ``np.sort(x)[:n_tail].mean()`` gives."""
    part = np.partition(cols, n_tail - 1, axis=0)[:n_tail]
    return np.sort(part, axis=0).mean(axis=0)


def empirical_cvar(samples: np.ndarray, alpha: float) -> float:
    """ref:735-763: mean of the worst (1 - alpha) fraction of the samples (at least one)."""
    if not (0.0 <= alpha < 1.0):
        raise ValueError(f"alpha must be in [0, 1), got {alpha!r}")
    x = np.asarray(samples).reshape(-1, 1)
    return float(_tail_means(x, _tail_count(alpha, x.shape[0]))[0])


def cvar_stability_check(all_preds, alpha: float, n_checks: int = 50,
                         remedy: str = "Increase bnn_n_samples") -> float:
    """ref:766-827: CVaR from all S rows against CVaR from the first S // 2 rows on up to
    ``n_checks`` columns drawn with ``default_rng(42)``; returns the mean relative gap and
    warns above 0.05.  ``all_preds`` [S, N] may be a device tensor: the probed columns are
    gathered on the device and only that [S, n_checks] block is copied to the host."""
    if alpha == 0.0:
        print("[CVaR stability] alpha=0 (posterior mean); stability check skipped")
        return 0.0
    S, N = all_preds.shape
    probe = np.random.default_rng(seed=42).choice(N, size=min(n_checks, N), replace=False)
    if torch.is_tensor(all_preds):
        block = all_preds.index_select(1, torch.as_tensor(probe, device=all_preds.device)).cpu().numpy()
    else:
        block = np.asarray(all_preds)[:, probe]
    full = _tail_means(block, _tail_count(alpha, S)).astype(np.float64)
    half = (_tail_means(block[:S // 2], _tail_count(alpha, S // 2)).astype(np.float64) if S >= 2
            else np.full(full.shape, np.nan))  # one sample: no half to compare with
    usable = np.abs(full) > 1e-8
    if not usable.any():
        return float("nan")
    gap = float(np.mean(np.abs(full[usable] - half[usable]) / np.abs(full[usable])))
    print(f"[CVaR stability] mean relative diff = {gap:.3f} (target < 0.05) [{'OK' if gap < 0.05 else 'WARN'}]")
    if gap > 0.05:
        warnings.warn(
            f"CVaR stability check: mean relative difference {gap:.3f} > 0.05. "
            f"{remedy} (current S={S}). Recommended minimum for alpha={alpha}: "
            f"S >= {int(np.ceil(30.0 / (1.0 - alpha)))}.", RuntimeWarning)
    return gap


def cvar_tail_mean_device(preds: torch.Tensor, n_tail: int) -> torch.Tensor:
    """out[c] = mean of the n_tail smallest of preds[:, c]  (iqlhip_cvar_tail_mean)."""
    lib = _lib.load()
    S, N = preds.shape
    preds = preds.contiguous()
    out = torch.empty(N, dtype=torch.float32, device=preds.device)
    with torch.cuda.device(preds.device):
        check(lib.iqlhip_cvar_tail_mean(ptr(preds), S, N, n_tail, ptr(out), stream_ptr()))
    return out


# --------------------------------------------------------------------------- #
# Markovian reward MLP (call-site contract of optbnn.bnn.nets.mlp.MLP, which is
# not vendored: ref:953-972, 1326-1336 -- x @ W + b layers, parameters ordered
# hidden (W, b) * depth then output (W, b), keys layers.0.W / layers.linear_i.W)
# --------------------------------------------------------------------------- #
class _XWb(nn.Module):
    def __init__(self, i, o):
        super().__init__()
        self.W = nn.Parameter(torch.zeros(i, o))
        self.b = nn.Parameter(torch.zeros(o))


class RewardMLP(nn.Module):
    def __init__(self, input_dim, output_dim, hidden_dims, activation_fn="relu"):
        super().__init__()
        if activation_fn not in ("relu", "tanh"):
            raise NotImplementedError(f"activation {activation_fn!r}")
        dims = [input_dim] + list(hidden_dims)
        self.layers = nn.ModuleDict()
        self.layers["0"] = _XWb(dims[0], dims[1])
        for i in range(1, len(hidden_dims)):
            self.layers[f"linear_{i}"] = _XWb(dims[i], dims[i + 1])
        self.fc_out = _XWb(dims[-1], output_dim)
        self.activation_fn = activation_fn

    def wb(self):
        mods = list(self.layers.values()) + [self.fc_out]
        return [m.W for m in mods], [m.b for m in mods]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        ws, bs = self.wb()
        return mlp_forward_f32(ws, bs, x, w_in_out=True,
                               hidden_act=0 if self.activation_fn == "relu" else 1)

    def load_state_dict(self, state, strict=True):
        """Accepts the real class's key names: every key that is not layers.0.* /
        layers.linear_i.* is taken as the output layer (its name is not visible from
        the reference: the submodule is not vendored)."""
        own = {}
        for k, v in state.items():
            if k.startswith("layers."):
                own[k] = v
        rest = sorted(k for k in state if not k.startswith("layers."))
        w = [k for k in rest if k.endswith("W") or k.endswith("weight")]
        b = [k for k in rest if k.endswith("b") or k.endswith("bias")]
        if len(w) != 1 or len(b) != 1:
            raise RuntimeError(f"cannot identify the output layer among {rest}")
        own["fc_out.W"], own["fc_out.b"] = state[w[0]], state[b[0]]
        return super().load_state_dict(own, strict=strict)


_COMPILE_PREFIX = "_orig_mod."  # torch.compile wraps the module: every key gains this (ref:1312-1323)


def _read_checkpoint(path: str, device) -> Dict[str, Any]:
    """The ``net`` state dict of a reward-model checkpoint, keys as an uncompiled module writes
    them.  Tensors only: the reference unpickles (weights_only=False); files this package did
    not write are never unpickled."""
    net = torch.load(path, map_location=device, weights_only=True)["net"]
    return {k.removeprefix(_COMPILE_PREFIX): v for k, v in net.items()}


def _strip_compile_prefix(state: Dict[str, Any]) -> Dict[str, Any]:
    return {k.removeprefix(_COMPILE_PREFIX): v for k, v in state.items()}


_HIDDEN_W = re.compile(r"layers\.(?:0|linear_(\d+))\.W")


def _build_mlp_reward_model(state: Dict[str, Any], activations: str, device: str = "cpu") -> nn.Module:
    """ref:1326-1336: the architecture is read off the hidden weight shapes ([in, out] each):
    layers.0.W, then layers.linear_1.W, layers.linear_2.W, ... for as long as they are present."""
    by_depth = {}
    for key, w in state.items():
        m = _HIDDEN_W.fullmatch(key)
        if m:
            by_depth[int(m.group(1) or 0)] = w.shape
    widths = []
    while len(widths) in by_depth:
        widths.append(by_depth[len(widths)][1])
    return RewardMLP(by_depth[0][0], 1, widths, activations).to(device)


def _run_config(model_dir: str) -> Dict[str, Any]:
    with open(os.path.join(model_dir, "config.yaml")) as f:
        return yaml.safe_load(f) or {}


def _mr_activations(model_dir: str) -> str:
    return _run_config(model_dir).get("activations", "relu")


def _torch_load(path, device):
    return torch.load(path, map_location=device, weights_only=True)


def load_mlp_reward_model(model_dir: str, device: str = "cpu") -> nn.Module:
    """ref:1345-1353: ``best_model.pt`` + ``config.yaml`` of an MR run directory."""
    state = _read_checkpoint(os.path.join(model_dir, "best_model.pt"), device)
    model = _build_mlp_reward_model(state, _mr_activations(model_dir), device)
    model.load_state_dict(state)
    return model.eval()


def _max_steps(env):
    return env._max_episode_steps


def _dataset_arrays(env, dataset, kwargs):
    if dataset is None:
        dataset = env.get_dataset(**kwargs)
    obs_all = dataset["observations"].astype(np.float32)
    act_all = dataset["actions"].astype(np.float32)
    return dataset, obs_all, act_all


def _finish(dataset, obs_all, act_all, rewards, keep):
    return {
        "observations": obs_all[:-1][keep],
        "actions": act_all[:-1][keep],
        "next_observations": obs_all[1:][keep],
        "rewards": rewards[keep],
        "terminals": dataset["terminals"][:-1][keep],
    }


def _device_of(model):
    return next(model.parameters()).device


def qlearning_dataset_mr(env, r_model, dataset=None, terminate_on_end=False, **kwargs):
    """ref:691-732: one reward-MLP forward over all N-1 transitions, on device."""
    dataset, obs_all, act_all = _dataset_arrays(env, dataset, kwargs)
    if not isinstance(r_model, RewardMLP):
        raise TypeError("r_model must be an iqlpref_amd RewardMLP (load_mlp_reward_model)")
    device = _device_of(r_model)
    keep, _ = keep_mask_and_steps(dataset["terminals"], dataset.get("timeouts"), _max_steps(env),
                                  terminate_on_end, device=device)
    obs_act = torch.from_numpy(np.concatenate([obs_all[:-1], act_all[:-1]], axis=1)).to(device)
    all_rewards = r_model(obs_act).squeeze(-1).cpu().numpy()
    return _finish(dataset, obs_all, act_all, all_rewards, keep)


def _ensemble_rewards(weight_sets, activation, obs_act_t, alpha, label, remedy):
    """S forwards into a device [S, N-1] matrix, then the tail mean kernel."""
    S, n = len(weight_sets), obs_act_t.shape[0]
    dev = obs_act_t.device
    n_tail = _tail_count(alpha, S)
    all_preds = torch.empty((S, n), dtype=torch.float32, device=dev)
    hidden_act = 0 if activation == "relu" else 1
    for k, (ws, bs) in enumerate(weight_sets):
        all_preds[k] = mlp_forward_f32(ws, bs, obs_act_t, w_in_out=True, hidden_act=hidden_act)[:, 0]
    penalized = cvar_tail_mean_device(all_preds, n_tail)
    cvar_stability_check(all_preds, alpha, remedy=remedy)
    mean_all = all_preds.mean(dim=0)
    print(f"[{label}] mean reward: {mean_all.mean().item():.4f} ± {mean_all.std().item():.4f}")
    name = "mean reward" if alpha == 0.0 else f"CVaR reward (alpha={alpha})"
    pr = penalized.cpu().numpy()
    print(f"[{label}] {name}: {pr.mean():.4f} ± {pr.std():.4f}")
    if pr.std() < 1e-6:
        warnings.warn("CVaR rewards have near-zero std — the penalty may have collapsed all "
                      "rewards to the same value. Consider a smaller alpha.", RuntimeWarning)
    return pr.astype(np.float32)


_SNAPSHOT = re.compile(r"checkpoint_(\d+)\.pt")


def _discover_mr_snapshots(reward_model_dir: str, burn_in: int = 0) -> List[str]:
    """ref:1047-1082: the per-epoch ``checkpoint_{epoch}.pt`` files of an MR run, by epoch,
    epochs below ``burn_in`` dropped (``best_model.pt`` duplicates one of them and is not a
    member).  Same errors as the reference: no snapshot at all -> FileNotFoundError, a burn-in
    that leaves none -> ValueError."""
    epochs: Dict[int, str] = {}
    if os.path.isdir(reward_model_dir):
        with os.scandir(reward_model_dir) as it:
            for entry in it:
                m = _SNAPSHOT.fullmatch(entry.name)
                if m:
                    epochs[int(m.group(1))] = entry.path
    if not epochs:
        raise FileNotFoundError(
            f"No MR snapshots found in {reward_model_dir}. Expected per-epoch "
            "checkpoint_{epoch}.pt files written by run_mr_training.py with checkpoints_path set.")
    members = [epochs[e] for e in sorted(epochs) if e >= burn_in]
    if not members:
        raise ValueError(f"mr_burn_in={burn_in} discarded all {len(epochs)} snapshot(s) in "
                         f"{reward_model_dir} (highest epoch present: {max(epochs)}).")
    if len(members) < len(epochs):
        print(f"[MR/CVaR] Burn-in {burn_in}: dropped {len(epochs) - len(members)} snapshot(s) "
              f"below epoch {burn_in}")
    return members


def qlearning_dataset_mr_ensemble(env, reward_model_dir: str, alpha: float = 0.95, burn_in: int = 0,
                                  device: str = "cpu", dataset=None, terminate_on_end: bool = False,
                                  **kwargs) -> Dict[str, np.ndarray]:
    """ref:1085-1220"""
    if not (0.0 <= alpha < 1.0):
        raise ValueError(f"mr_alpha must be in [0, 1), got {alpha!r}")
    dev = _lib.require_gpu(device)
    dataset, obs_all, act_all = _dataset_arrays(env, dataset, kwargs)
    keep, _ = keep_mask_and_steps(dataset["terminals"], dataset.get("timeouts"), _max_steps(env),
                                  terminate_on_end, device=dev)
    ckpt_paths = _discover_mr_snapshots(reward_model_dir, burn_in)
    n_total = len(ckpt_paths)
    n_tail = _tail_count(alpha, n_total)
    if n_tail < 5 and alpha > 0.0:
        warnings.warn(f"CVaR tail has only {n_tail} snapshot(s) with alpha={alpha} and S={n_total}. "
                      "Lower mr_alpha or train the reward model with more eval epochs.", RuntimeWarning)
    print(f"[MR/CVaR] S={n_total} snapshot(s), alpha={alpha}, n_tail={n_tail}")
    activations = _mr_activations(reward_model_dir)
    sets = []
    for p in ckpt_paths:
        state = _read_checkpoint(p, dev)
        net = _build_mlp_reward_model(state, activations, dev)
        net.load_state_dict(state)
        sets.append(net.wb())
    obs_act_t = torch.from_numpy(np.concatenate([obs_all[:-1], act_all[:-1]], axis=1)).to(dev)
    r = _ensemble_rewards(sets, activations, obs_act_t, alpha, "MR/CVaR",
                          "Lower mr_alpha, or retrain the reward model with more eval epochs to grow the ensemble")
    return _finish(dataset, obs_all, act_all, r, keep)


def qlearning_dataset_bnn(env, reward_model_dir: str, alpha: float = 0.95, n_samples: int = 500,
                          device: str = "cpu", dataset=None, terminate_on_end: bool = False,
                          **kwargs) -> Dict[str, np.ndarray]:
    """ref:830-1044"""
    if not (0.0 <= alpha < 1.0):
        raise ValueError(f"bnn_alpha must be in [0, 1), got {alpha!r}")
    dev = _lib.require_gpu(device)
    dataset, obs_all, act_all = _dataset_arrays(env, dataset, kwargs)
    keep, _ = keep_mask_and_steps(dataset["terminals"], dataset.get("timeouts"), _max_steps(env),
                                  terminate_on_end, device=dev)
    sampling_dir = os.path.join(reward_model_dir, "sampling_f")
    weight_files = sorted(_glob.glob(os.path.join(
        sampling_dir, "chain_*/sampled_weights/sampled_weights_0000000")))
    if not weight_files:
        raise FileNotFoundError(
            f"No BNN posterior weight files found under {sampling_dir}. "
            "Expected structure: sampling_f/chain_*/sampled_weights/sampled_weights_0000000")
    all_weights: List = []
    for wf in weight_files:
        all_weights.extend(load_bnn_weight_file(wf)["sampled_weights"])
    if not all_weights:
        raise RuntimeError(f"BNN checkpoint at {reward_model_dir} contained no sampled weights.")
    available = len(all_weights)
    if n_samples > 0 and n_samples > available:
        warnings.warn(f"bnn_n_samples={n_samples} requested but only {available} posterior samples "
                      f"are available. Using all {available} samples.", RuntimeWarning)
    if n_samples > 0 and n_samples < available:
        rng = np.random.default_rng(seed=0)
        idx = rng.choice(available, size=n_samples, replace=False)
        all_weights = [all_weights[i] for i in sorted(idx)]
    n_total = len(all_weights)
    n_tail = _tail_count(alpha, n_total)
    if n_tail < 5:
        warnings.warn(f"CVaR tail has only {n_tail} sample(s) with alpha={alpha} and S={n_total}.",
                      RuntimeWarning)
    print(f"[BNN/CVaR] S={n_total} samples from {len(weight_files)} chain(s), alpha={alpha}, n_tail={n_tail}")
    w0 = all_weights[0]
    print(f"[BNN/CVaR] Inferred architecture: input_dim={int(w0[0].shape[0])}, "
          f"width={int(w0[0].shape[1])}, depth={(len(w0) - 2) // 2}")
    sets = []
    for w in all_weights:
        t = [torch.as_tensor(np.asarray(a), dtype=torch.float32, device=dev) for a in w]
        sets.append((t[0::2], t[1::2]))
    obs_act_t = torch.from_numpy(np.concatenate([obs_all[:-1], act_all[:-1]], axis=1)).to(dev)
    r = _ensemble_rewards(sets, "relu", obs_act_t, alpha, "BNN/CVaR", "Increase bnn_n_samples")
    return _finish(dataset, obs_all, act_all, r, keep)


def load_bnn_weight_file(path):
    """The reference unpickles these files (lists of numpy arrays, ref:913-915).  Here
    only tensors, plain containers and numpy arrays are accepted (weights_only)."""
    try:
        return torch.load(path, map_location="cpu", weights_only=True)
    except Exception:
        try:
            from numpy._core.multiarray import _reconstruct
        except ImportError:  # numpy < 2
            from numpy.core.multiarray import _reconstruct
        allowed = [np.ndarray, np.dtype, _reconstruct] + [type(np.dtype(t)) for t in
                                                         (np.float32, np.float64, np.int64)]
        with torch.serialization.safe_globals(allowed):
            return torch.load(path, map_location="cpu", weights_only=True)


# --------------------------------------------------------------------------- #
# preference transformer  (ref:1356-1390 loader, ref:1223-1309 relabel)
# --------------------------------------------------------------------------- #
class RewardPT(nn.Module):
    """Parameter container with the state-dict keys ref:1363-1371 reads; forward =
    ``iqlhip_pt_relabel``.  Dropout arguments are accepted and unused (eval only)."""

    def __init__(self, state_dim, action_dim, max_episode_steps, embd_dim=64, pref_attn_embd_dim=64,
                 num_heads=4, attn_dropout=0.1, resid_dropout=0.1, intermediate_dim=256, num_layers=1,
                 embd_dropout=0.1, max_pos=1024, eps=1e-5):
        super().__init__()
        E, I = embd_dim, intermediate_dim
        self.num_heads, self.eps, self.max_pos = num_heads, eps, max_pos
        self.state_linear = nn.Linear(state_dim, E)
        self.action_linear = nn.Linear(action_dim, E)
        self.timestep_embed = nn.Embedding(max_episode_steps + 1, E)
        self.stacked_layer_norm = nn.LayerNorm(E, eps=eps)
        self.gpt = nn.Module()
        self.gpt.layers = nn.ModuleList()
        for _ in range(num_layers):
            blk = nn.Module()
            blk.layer_norm_0 = nn.LayerNorm(E, eps=eps)
            blk.attention = nn.Module()
            blk.attention.in_linear = nn.Linear(E, 3 * E)
            blk.attention.out_linear = nn.Linear(E, E)
            blk.attention.register_buffer("causal_bias",
                                          torch.tril(torch.ones(1, 1, max_pos, max_pos)), persistent=True)
            blk.layer_norm_1 = nn.LayerNorm(E, eps=eps)
            blk.mlp = nn.Module()
            blk.mlp.in_linear = nn.Linear(E, I)
            blk.mlp.out_linear = nn.Linear(I, E)
            self.gpt.layers.append(blk)
        self.gpt.layer_norm = nn.LayerNorm(E, eps=eps)
        self.pref_linear = nn.Linear(E, 2 * pref_attn_embd_dim + 1)

    def numpy_params(self):
        return {k: v.detach().cpu().numpy() for k, v in self.state_dict().items()
                if not k.endswith("causal_bias")}

    def _weights(self):
        dev = _device_of(self)
        f = lambda t: t.detach().to(torch.float32).contiguous()
        blk = self.gpt.layers[0]
        keep = dict(
            state_wT=f(self.state_linear.weight.t()), state_b=f(self.state_linear.bias),
            action_wT=f(self.action_linear.weight.t()), action_b=f(self.action_linear.bias),
            temb=f(self.timestep_embed.weight),
            sln_w=f(self.stacked_layer_norm.weight), sln_b=f(self.stacked_layer_norm.bias),
            ln0_w=f(blk.layer_norm_0.weight), ln0_b=f(blk.layer_norm_0.bias),
            qkv_w=f(blk.attention.in_linear.weight), qkv_b=f(blk.attention.in_linear.bias),
            attn_out_w=f(blk.attention.out_linear.weight), attn_out_b=f(blk.attention.out_linear.bias),
            ln1_w=f(blk.layer_norm_1.weight), ln1_b=f(blk.layer_norm_1.bias),
            mlp_in_w=f(blk.mlp.in_linear.weight), mlp_in_b=f(blk.mlp.in_linear.bias),
            mlp_out_w=f(blk.mlp.out_linear.weight), mlp_out_b=f(blk.mlp.out_linear.bias),
            lnf_w=f(self.gpt.layer_norm.weight), lnf_b=f(self.gpt.layer_norm.bias),
            pref_w_last=f(self.pref_linear.weight[-1]))
        w = _lib.PtWeights()
        w.state_dim = self.state_linear.in_features
        w.action_dim = self.action_linear.in_features
        w.embd_dim = self.state_linear.out_features
        w.num_heads = self.num_heads
        w.inter_dim = blk.mlp.in_linear.out_features
        w.num_layers = len(self.gpt.layers)
        w.n_temb = self.timestep_embed.num_embeddings
        w.eps = self.eps
        for k, t in keep.items():
            setattr(w, k, t.data_ptr())
        w.pref_b_last = float(self.pref_linear.bias[-1].item())
        return w, keep, dev

    def window_values(self, obs: torch.Tensor, act: torch.Tensor, win_start: torch.Tensor,
                      win_len: torch.Tensor, query_length: int,
                      win_t0: Optional[torch.Tensor] = None) -> torch.Tensor:
        """value[:, 0, -1, 0] of each (start, len) window over the device arrays obs/act; the
        timestep of a window's k-th transition is ``win_t0 + k`` (0 + k when None)."""
        lib = _lib.load()
        w, keep, dev = self._weights()
        obs = obs.to(torch.float32).contiguous()
        act = act.to(torch.float32).contiguous()
        win_start = win_start.to(torch.int64).contiguous()
        win_len = win_len.to(torch.int32).contiguous()
        if win_t0 is not None:
            win_t0 = win_t0.to(device=dev, dtype=torch.int32).contiguous()
            if int((win_t0 + win_len).max()) > w.n_temb:
                raise ValueError("a window's last timestep exceeds the timestep-embedding table")
        out = torch.empty(win_start.shape[0], dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            check(lib.iqlhip_pt_relabel(C.byref(w), ptr(obs), ptr(act), obs.shape[0], ptr(win_start),
                                        ptr(win_len), ptr(win_t0), win_start.shape[0], query_length,
                                        ptr(out), stream_ptr()))
        del keep
        return out


def load_pt_reward_model(model_dir: str, device: str = "cpu") -> nn.Module:
    """ref:1356-1390: every size comes from a tensor shape of ``best_model.pt``; what no shape
    shows (heads, dropouts, MLP width, LayerNorm eps) comes from the run's ``config.yaml`` with
    the reference's fall-backs."""
    cfg = _run_config(model_dir)
    state = _read_checkpoint(os.path.join(model_dir, "best_model.pt"), device)
    shape = lambda key: tuple(state[key].shape)
    embd_dim, state_dim = shape("state_linear.weight")
    depth = 0
    while f"gpt.layers.{depth}.layer_norm_0.weight" in state:
        depth += 1
    arch = dict(
        state_dim=state_dim,
        action_dim=shape("action_linear.weight")[1],
        max_episode_steps=shape("timestep_embed.weight")[0] - 1,
        embd_dim=embd_dim,
        pref_attn_embd_dim=(shape("pref_linear.weight")[0] - 1) // 2,
        num_layers=depth,
        max_pos=shape("gpt.layers.0.attention.causal_bias")[2],
        intermediate_dim=cfg.get("intermediate_dim") or 4 * embd_dim,
        num_heads=cfg.get("num_heads", 4),
        eps=cfg.get("model_eps", 1e-5),
    )
    for rate in ("attn_dropout", "resid_dropout", "embd_dropout"):
        arch[rate] = cfg.get(rate, 0.1)
    model = RewardPT(**arch).to(device)
    model.load_state_dict(state)
    return model.eval()


def qlearning_dataset_pt(env, r_model, query_length=100, dataset=None, terminate_on_end=False,
                         correct_window_offsets=False, **kwargs):
    """ref:1223-1309.

    Bug-compatible by default: the reference indexes the GLOBAL arrays with the
    episode-relative step (ref:1277-1280, 1289-1290), so a transition's window --
    and reward -- depends only on its episode step; one window per distinct step is
    evaluated and scattered.  ``correct_window_offsets=True`` evaluates the window
    that ends at each transition instead (one per transition)."""
    dataset, obs_all, act_all = _dataset_arrays(env, dataset, kwargs)
    if not isinstance(r_model, RewardPT):
        raise TypeError("r_model must be an iqlpref_amd RewardPT (load_pt_reward_model)")
    dev = _device_of(r_model)
    keep, ep_steps = keep_mask_and_steps(dataset["terminals"], dataset.get("timeouts"), _max_steps(env),
                                         terminate_on_end, device=dev)
    n = ep_steps.shape[0]
    lens = np.minimum(ep_steps + 1, query_length)
    if correct_window_offsets:
        starts = np.arange(n, dtype=np.int64) - lens + 1
        ws = torch.from_numpy(starts).to(dev)
        wl = torch.from_numpy(lens.astype(np.int32)).to(dev)
        vals = r_model.window_values(torch.from_numpy(obs_all).to(dev), torch.from_numpy(act_all).to(dev),
                                     ws, wl, query_length)
        all_rewards = vals.cpu().numpy()
    else:
        uniq, inv = np.unique(ep_steps, return_inverse=True)
        ulen = np.minimum(uniq + 1, query_length)
        ustart = np.where(uniq >= query_length, uniq - query_length + 1, 0)
        hi = int((ustart + ulen).max())
        vals = r_model.window_values(torch.from_numpy(obs_all[:hi]).to(dev),
                                     torch.from_numpy(act_all[:hi]).to(dev),
                                     torch.from_numpy(ustart.astype(np.int64)).to(dev),
                                     torch.from_numpy(ulen.astype(np.int32)).to(dev), query_length)
        all_rewards = vals.cpu().numpy()[inv]
    return _finish(dataset, obs_all, act_all, all_rewards.astype(np.float32), keep)
