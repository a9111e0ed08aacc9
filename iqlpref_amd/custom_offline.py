"""The ``algorithms/custom_offline/iql.py`` flavour of the path -- what the
pen sweeps run (pt_sweeps/sweep_pen_human_pt.yaml:2).  "cref:" = that file.

Differences from ``algorithms/offline/iql.py`` and how they map onto the same kernels:

* dataset = Minari episodes; the preference-transformer relabel is PER EPISODE with the TRUE
  timesteps (cref:158-225): one forward over the first ``query_length`` steps of an episode gives
  the reward of each of them, every later step gets the last value of its rolling window.  Causal
  attention makes position i of that first forward equal to the last-token value of the prefix
  window [0, i], so the whole relabel is ONE ``iqlhip_pt_relabel`` call over windows
  (start, len, t0) = (ep + max(0, i - QL + 1), min(i + 1, QL), max(0, i + 1 - QL));
* ``query_length == 1``: a Markovian reward MLP (reward_models/q_mlp.py) over (s, a);
* ``ReplayBuffer.sample`` draws indices with numpy's global RNG (cref:277-284) -- reproduced
  exactly: the indices are drawn on the host with the same call and uploaded;
* no autocast (``precision="fp32"``), Polyak update written as (1 - tau) t + tau s (cref:85-87,
  ``polyak_form=1``), checkpoint key ``actor_lr_scheduler`` and no ``total_it`` (cref:546-556);
* ``modify_reward``: only the locomotion range scaling and antmaze's -1 (cref:145-155).

Not built (stated): the Orbax / flax-nnx checkpoint readers ``load_PT`` / ``load_QMLP``
(reward_models/pref_transformer.py:280-327, q_mlp.py:100-168) need orbax + jax, which are absent;
``RewardPT.load_flax_params`` / ``QMLP.load_flax_params`` take the parameter pytree as numpy
arrays instead.  PT numerics stay "parity unpinned" (no runnable reference, no fixtures).
"""
import uuid
import os
from dataclasses import dataclass
from typing import Any, Dict, Iterable, List, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .iql import ImplicitQLearning as _OfflineIQL
from .iql import ReplayBuffer as _OfflineReplayBuffer
from .iql import mlp_forward_f32
from .relabel import RewardPT

ACTIVATIONS = ("cos", "tanh", "relu", "softplus", "sin", "leaky_relu", "swish", "none")  # q_mlp.py:121-130


@dataclass
class TrainConfig:
    """cref:44-78, same fields and defaults."""
    project: str = "IQL-pref"
    group: str = "IQL-Minari-pref"
    name: str = "iql-p"
    gamma: float = 0.99
    tau: float = 0.005
    beta: float = 3.0
    iql_tau: float = 0.7
    iql_deterministic: bool = False
    vf_lr: float = 3e-4
    qf_lr: float = 3e-4
    actor_lr: float = 3e-4
    actor_dropout: Optional[float] = None
    dataset_id: str = "D4RL/pen/human-v2"
    reward_model_path: str = "~/iqlpref/pen_labels/mr_pen/best_model.ckpt"
    query_length: int = 1
    update_steps: int = int(1e6)
    buffer_size: int = 2_000_000
    batch_size: int = 256
    normalize_state: bool = True
    normalize_reward: bool = False
    eval_every: int = int(5e3)
    eval_episodes: int = 10
    train_seed: int = 0
    eval_seed: int = 0
    checkpoints_path: Optional[str] = None

    def __post_init__(self):
        self.name = f"{self.name}-{self.dataset_id}-{str(uuid.uuid4())[:8]}"
        if self.checkpoints_path is not None:
            self.checkpoints_path = os.path.join(self.checkpoints_path, self.name)


# --------------------------------------------------------------------------- #
# reward models
# --------------------------------------------------------------------------- #
class QMLP(nn.Module):
    """reward_models/q_mlp.py:16-98: Linear layers over concat(s, a) (flax kernels are [in, out]),
    ``activations`` between them, ``activation_final`` on the scalar output."""

    def __init__(self, state_dim: int, action_dim: int, hidden_dims: Sequence[int] = (256, 256),
                 activations: str = "relu", activation_final: str = "none"):
        super().__init__()
        if activations not in ACTIVATIONS or activation_final not in ACTIVATIONS:
            raise ValueError(f"activations must be among {ACTIVATIONS}")
        dims = [state_dim + action_dim, *hidden_dims, 1]
        self.kernels = nn.ParameterList([nn.Parameter(torch.zeros(i, o)) for i, o in zip(dims[:-1], dims[1:])])
        self.biases = nn.ParameterList([nn.Parameter(torch.zeros(o)) for o in dims[1:]])
        self.activations, self.activation_final = activations, activation_final

    def load_flax_params(self, layers: Sequence[Dict[str, np.ndarray]]):
        """``layers``: [{"kernel": [in,out], "bias": [out]}, ...] in layer order (nnx.Linear)."""
        with torch.no_grad():
            for k, b, l in zip(self.kernels, self.biases, layers):
                k.copy_(torch.as_tensor(np.asarray(l["kernel"], np.float32)))
                b.copy_(torch.as_tensor(np.asarray(l["bias"], np.float32)))
        return self

    def forward(self, observations, actions) -> torch.Tensor:
        dev = self.kernels[0].device
        x = torch.cat([torch.as_tensor(observations, dtype=torch.float32, device=dev),
                       torch.as_tensor(actions, dtype=torch.float32, device=dev)], dim=-1)
        y = mlp_forward_f32(list(self.kernels), list(self.biases), x, w_in_out=True,
                            hidden_act=ACTIVATIONS.index(self.activations) + _lib.ACT_FLAX_BASE,
                            out_act=ACTIVATIONS.index(self.activation_final) + _lib.ACT_FLAX_BASE)
        return y.squeeze(-1)


def load_pt_flax_params(model: RewardPT, params: Dict[str, Any]) -> RewardPT:
    """Copy a flax-nnx PT parameter tree (reward_models/pref_transformer.py:170-209; numpy arrays,
    ``kernel`` [in, out], ``scale`` for LayerNorm weights, ``embedding`` for nnx.Embed) into the
    torch-layout container.  Keys: state_linear, action_linear, timestep_embed,
    stacked_layer_norm, gpt.layers.<i>.{layer_norm_0, attention.{in_linear,out_linear},
    layer_norm_1, mlp.{in_linear,out_linear}}, gpt.layer_norm, pref_linear."""
    sd = {}

    def walk(prefix, node):
        for k, v in node.items():
            name = f"{prefix}.{k}" if prefix else str(k)
            if isinstance(v, dict):
                walk(name, v)
            else:
                a = np.asarray(v, np.float32)
                if k == "kernel":
                    sd[prefix + ".weight"] = torch.from_numpy(np.ascontiguousarray(a.T))
                elif k in ("scale", "embedding"):
                    sd[prefix + ".weight"] = torch.from_numpy(a)
                elif k == "bias":
                    sd[prefix + ".bias"] = torch.from_numpy(a)
    walk("", params)
    missing = model.load_state_dict(sd, strict=False)
    bad = [k for k in missing.missing_keys if not k.endswith("causal_bias")]
    if bad or missing.unexpected_keys:
        raise KeyError(f"flax parameter tree does not match: missing {bad}, unexpected {missing.unexpected_keys}")
    return model


# --------------------------------------------------------------------------- #
# dataset (cref:158-225)
# --------------------------------------------------------------------------- #
def _episode_arrays(ep):
    get = (lambda k: ep[k]) if isinstance(ep, dict) else (lambda k: getattr(ep, k))
    return (np.asarray(get("observations"), np.float32), np.asarray(get("actions"), np.float32),
            np.asarray(get("terminations")))


def episode_windows(lengths: Sequence[int], query_length: int):
    """(start, len, t0) of the relabel window of every step of every episode; ``start`` indexes
    the concatenated per-step arrays.  Closed form of the rolling loop at cref:172-211."""
    L = np.asarray(lengths, dtype=np.int64)
    ep_start = np.concatenate([[0], np.cumsum(L)[:-1]])
    step = np.arange(int(L.sum()), dtype=np.int64) - np.repeat(ep_start, L)  # step i inside its episode
    t0 = np.maximum(0, step + 1 - query_length)
    return np.repeat(ep_start, L) + t0, np.minimum(step + 1, query_length).astype(np.int32), t0.astype(np.int32)


def qlearning_dataset(dataset: Iterable, r_model, query_length: int = 1) -> Dict[str, np.ndarray]:
    """cref:158-225.  ``dataset`` iterates episodes (Minari ``EpisodeData`` or dicts) with
    ``observations`` [L+1, S], ``actions`` [L, A], ``terminations`` [L]."""
    eps = [_episode_arrays(e) for e in dataset]
    obs = np.concatenate([o[:-1] for o, _, _ in eps])
    nxt = np.concatenate([o[1:] for o, _, _ in eps])
    act = np.concatenate([a for _, a, _ in eps])
    dones = np.concatenate([d for _, _, d in eps])
    if query_length > 1:
        if not isinstance(r_model, RewardPT):
            raise TypeError("query_length > 1 needs an iqlpref_amd RewardPT")
        dev = next(r_model.parameters()).device
        start, length, t0 = episode_windows([a.shape[0] for _, a, _ in eps], query_length)
        up = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
        rewards = r_model.window_values(up(obs), up(act), up(start), up(length), query_length,
                                        win_t0=up(t0)).cpu().numpy()
    else:
        rewards = r_model(obs, act).cpu().numpy()
    return {"observations": obs, "actions": act, "next_observations": nxt,
            "rewards": rewards.astype(np.float32), "terminals": dones}


def return_reward_range(dataset, max_episode_steps):
    """cref:127-142 (same loop as the offline flavour, without trajectory lengths)."""
    from .relabel import return_reward_range as rr
    lo, hi, _ = rr(dataset, max_episode_steps)
    return lo, hi


def modify_reward(dataset: Dict[str, np.ndarray], env_name: str, max_episode_steps: int = 1000):
    """cref:145-155."""
    if any(s in env_name for s in ("halfcheetah", "hopper", "walker2d")):
        lo, hi = return_reward_range(dataset, max_episode_steps)
        dataset["rewards"] /= hi - lo
        dataset["rewards"] *= max_episode_steps
    elif "antmaze" in env_name:
        dataset["rewards"] -= 1.0


def evaluate(env, actor: nn.Module, num_episodes: int, seed: int, device: str) -> np.ndarray:
    """cref:559-579: ``num_episodes`` sequential episodes of a gymnasium-API environment
    (``reset(seed=seed + i)``), greedy actions from ``actor.act`` (one exact-fp32 forward on the GPU
    per step), undiscounted returns.  The actor is handed back in train mode (cref:578)."""
    actor.eval()
    episode_rewards = []
    try:
        for i in range(num_episodes):
            done = False
            state, _ = env.reset(seed=seed + i)
            episode_reward = 0.0
            while not done:
                state, reward, terminated, truncated, _ = env.step(actor.act(np.asarray(state), device))
                done = terminated or truncated
                episode_reward += reward
            episode_rewards.append(episode_reward)
    finally:
        actor.train()
    return np.asarray(episode_rewards)


# --------------------------------------------------------------------------- #
# buffer and trainer
# --------------------------------------------------------------------------- #
class ReplayBuffer(_OfflineReplayBuffer):
    """cref:228-290: ``load_dataset`` + a sampler on numpy's GLOBAL generator -- after
    ``np.random.seed(s)`` the index stream is the reference's, draw for draw."""

    def load_dataset(self, data: Dict[str, np.ndarray]):
        self.load_d4rl_dataset(data)

    def draw_indices(self, batch_size: int, n_batches: Optional[int] = None) -> np.ndarray:
        """cref:278: ``np.random.randint(0, min(size, pointer), size=batch_size)``, once per batch."""
        hi = min(self._size, self._pointer)
        if n_batches is None:
            return np.random.randint(0, hi, size=batch_size)
        return np.stack([np.random.randint(0, hi, size=batch_size) for _ in range(n_batches)])

    def sample(self, batch_size: int, indices=None):
        if indices is None:
            indices = torch.from_numpy(self.draw_indices(batch_size)).to(self._dev)
        return super().sample(batch_size, indices)


class ImplicitQLearning(_OfflineIQL):
    """cref:438-556 on the same kernels: no autocast, convex Polyak form, the scheduler handed in."""

    def __init__(self, max_action, actor, actor_optimizer, actor_lr_scheduler, q_network, q_optimizer,
                 v_network, v_optimizer, iql_tau: float = 0.7, beta: float = 3.0, gamma: float = 0.99,
                 tau: float = 0.005, device: str = "cpu", *, seed: Optional[int] = None,
                 keep_grads: bool = False):
        super().__init__(max_action, actor, actor_optimizer, q_network, q_optimizer, v_network, v_optimizer,
                         iql_tau=iql_tau, beta=beta, max_steps=int(actor_lr_scheduler.T_max), discount=gamma,
                         tau=tau, device=device, precision="fp32", seed=seed, keep_grads=keep_grads,
                         polyak_form=1)
        self.actor_lr_scheduler = self.actor_lr_schedule = actor_lr_scheduler
        self.gamma = gamma

    def train_on_buffer(self, replay_buffer: ReplayBuffer, n_steps: int, batch_size: int):
        """``n_steps`` x (sample with numpy's generator, train) in one library call."""
        idx = torch.from_numpy(replay_buffer.draw_indices(batch_size, n_steps)).to(self._dev)
        return self.train_steps(replay_buffer, n_steps, batch_size, indices=idx)

    def state_dict(self) -> Dict[str, Any]:
        sd = super().state_dict()
        sd["actor_lr_scheduler"] = sd.pop("actor_lr_schedule")
        sd.pop("total_it")
        return sd  # cref:546-556

    def load_state_dict(self, state_dict: Dict[str, Any]):
        sd = dict(state_dict)
        sd["actor_lr_schedule"] = sd.pop("actor_lr_scheduler")
        sd.setdefault("total_it", int(sd["actor_lr_schedule"]["last_epoch"]))
        super().load_state_dict(sd)
