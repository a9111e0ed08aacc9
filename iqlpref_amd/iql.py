"""Drop-in surface of the reference's ``algorithms/offline/iql.py`` on MI355X.

Same class / function names, constructor signatures, state-dict keys and error
behaviour as the reference ("ref:" = /root/reference/algorithms/offline/iql.py);
the arithmetic runs in hand-written HIP kernels behind libiqlhip.so
(include/iqlhip.h).  PyTorch is the container for parameters, optimiser state and
checkpoints only.  There is no CPU path: constructing a buffer or trainer on a
non-ROCm device raises.
"""
import copy
import ctypes as C
import math
import os
import random
import uuid
from dataclasses import asdict, dataclass, fields
from typing import Any, Callable, Dict, List, Optional, Tuple, Union

import numpy as np
import torch
import torch.nn as nn
from torch.distributions import Normal
from torch.optim.lr_scheduler import CosineAnnealingLR

from . import _lib
from ._lib import PREC_BF16, PREC_FP32, check, ptr, stream_ptr

TensorBatch = List[torch.Tensor]

EXP_ADV_MAX = 100.0  # ref:38
LOG_STD_MIN = -20.0  # ref:39
LOG_STD_MAX = 2.0  # ref:40


# --------------------------------------------------------------------------- #
# config (ref:43-124, plus reward_model_root of iql_eval.py:105-110,143-146)
# --------------------------------------------------------------------------- #
@dataclass
class TrainConfig:
    project: str = "IQL-pref"
    group: str = "IQL-D4RL"
    name: str = "IQL"
    env: str = "halfcheetah-medium-expert-v2"
    discount: float = 0.99
    tau: float = 0.005
    beta: float = 3.0
    iql_tau: float = 0.7
    iql_deterministic: bool = False
    max_timesteps: int = int(1e6)
    buffer_size: int = 2_000_000
    batch_size: int = 256
    normalize: bool = True
    normalize_reward: int = 0
    vf_lr: float = 3e-4
    qf_lr: float = 3e-4
    actor_lr: float = 3e-4
    actor_dropout: Optional[float] = None
    log_freq: int = 250
    eval_freq: int = int(5e3)
    n_episodes: int = 10
    checkpoints_path: Optional[str] = None
    load_model: str = ""
    reward_model_path: str = ""
    query_length: int = 1
    bnn_reward_model: bool = False
    bnn_alpha: float = 0.95
    bnn_n_samples: int = 500
    mr_ensemble: bool = False
    mr_alpha: float = 0.95
    mr_burn_in: int = 0
    reward_model_root: Optional[str] = None
    # not in the reference: number of critics (2 = the reference's TwinQ; 3..8 = E-way critic
    # ensemble of BASELINE config 5, see EnsembleQ)
    n_critics: int = 2
    seed: int = 0
    device: str = "cuda"

    def __post_init__(self):
        self.name = f"{self.name}-{self.env}-{str(uuid.uuid4())[:8]}"
        if self.checkpoints_path is not None:
            self.checkpoints_path = os.path.join(self.checkpoints_path, self.name)
        if self.reward_model_root:
            self.reward_model_path = f"{self.reward_model_root}_{self.seed}"


def load_config(config_path: Optional[str] = None, **overrides) -> TrainConfig:
    """The pyrallis ``--config_path`` behaviour (ref:1393): YAML keys = field names.

    Unknown keys raise, values are coerced to the field's declared type (the
    reference YAMLs write ``3e-4`` and ``false`` for int fields)."""
    import yaml

    raw: Dict[str, Any] = {}
    if config_path:
        with open(config_path) as f:
            raw.update(yaml.safe_load(f) or {})
    raw.update(overrides)
    known = {f.name: f for f in fields(TrainConfig)}
    kwargs = {}
    for k, v in raw.items():
        if k not in known:
            raise ValueError(f"unknown TrainConfig field {k!r}")
        kwargs[k] = _coerce(v, known[k].type)
    return TrainConfig(**kwargs)


def _coerce(v, typ):
    t = str(typ)
    if v is None:
        return None
    if "float" in t:
        return float(v)
    if "int" in t and "bool" not in t:
        return int(v)
    if "bool" in t:
        if isinstance(v, str):
            return v.strip().lower() in ("1", "true", "yes")
        return bool(v)
    return v


# --------------------------------------------------------------------------- #
# small helpers of the reference module
# --------------------------------------------------------------------------- #
def soft_update(target: nn.Module, source: nn.Module, tau: float):
    """ref:127-129 (host-side helper; the training step fuses this into k_update)."""
    for tp, sp in zip(target.parameters(), source.parameters()):
        tp.data.lerp_(sp.data, tau)


def compute_mean_std(states: np.ndarray, eps: float) -> Tuple[np.ndarray, np.ndarray]:
    """ref:132-135"""
    return states.mean(0), states.std(0) + eps


def normalize_states(states: np.ndarray, mean: np.ndarray, std: np.ndarray):
    """ref:138-139"""
    return (states - mean) / std


def asymmetric_l2_loss(u: torch.Tensor, tau: float) -> torch.Tensor:
    """ref:404-405"""
    return torch.mean(torch.abs(tau - (u < 0).float()) * u**2)


def set_seed(seed: int, env=None, deterministic_torch: bool = False):
    """ref:229-239"""
    if env is not None:
        env.seed(seed)
        env.action_space.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    np.random.seed(seed)
    random.seed(seed)
    torch.manual_seed(seed)
    torch.use_deterministic_algorithms(deterministic_torch)


# --------------------------------------------------------------------------- #
# replay buffer (ref:164-226)
# --------------------------------------------------------------------------- #
_buffer_generation = [0]  # process-wide: every load of any buffer gets a fresh number


class ReplayBuffer:
    """Device-resident replay buffer with packed rows [s|a|r|d|pad|s'|pad] (s' 16-byte aligned).

    ``_states`` ... ``_dones`` are strided views into the packed matrix with the
    reference's shapes.  Storage is allocated for the rows actually loaded, not
    for ``buffer_size`` (the YAMLs ask for 1e7 rows = 2.7 GB of zeros)."""

    def __init__(self, state_dim: int, action_dim: int, buffer_size: int, device: str = "cpu"):
        self._lib = _lib.load()
        self._dev = _lib.require_gpu(device)
        self._buffer_size = buffer_size
        self._pointer = 0
        self._size = 0
        self._state_dim, self._action_dim = state_dim, action_dim
        self._device = device
        self._stride = self._lib.iqlhip_replay_row_stride(state_dim, action_dim)
        self._next_off = self._lib.iqlhip_replay_next_offset(state_dim, action_dim)
        self._alloc(0)
        self._sample_calls = 0
        self._sample_seed = None
        self._generation = 0

    def _alloc(self, n):
        S, A = self._state_dim, self._action_dim
        self._rows = torch.zeros((n, self._stride), dtype=torch.float32, device=self._dev)
        self._states = self._rows[:, :S]
        self._actions = self._rows[:, S:S + A]
        self._rewards = self._rows[:, S + A:S + A + 1]
        self._dones = self._rows[:, S + A + 1:S + A + 2]
        self._next_states = self._rows[:, self._next_off:self._next_off + S]

    def _to_tensor(self, data: np.ndarray) -> torch.Tensor:
        return torch.tensor(data, dtype=torch.float32, device=self._dev)

    def load_d4rl_dataset(self, data: Dict[str, np.ndarray]):
        """ref:193-209: the five arrays are uploaded once and interleaved on the device."""
        self.load_device_arrays(
            self._to_tensor(data["observations"]), self._to_tensor(data["actions"]),
            self._to_tensor(data["rewards"]), self._to_tensor(data["next_observations"]),
            self._to_tensor(data["terminals"]))

    def load_device_arrays(self, obs: torch.Tensor, act: torch.Tensor, rew: torch.Tensor, nxt: torch.Tensor,
                           done: torch.Tensor, state_mean: Optional[torch.Tensor] = None,
                           state_std: Optional[torch.Tensor] = None):
        """``load_d4rl_dataset`` for arrays that already live on the device (iqlpref_amd.prep).
        With ``state_mean`` / ``state_std`` ([S] float32 device tensors) the z-scoring of
        ref:1438-1448 is fused into the interleave: s and s' are stored as (x - mean) / std."""
        if self._size != 0:
            raise ValueError("Trying to load data into non-empty replay buffer")
        n = obs.shape[0]
        if n > self._buffer_size:
            raise ValueError("Replay buffer is smaller than the dataset you are trying to load!")
        f = lambda t, shape: t.to(device=self._dev, dtype=torch.float32).reshape(shape).contiguous()
        S, A = self._state_dim, self._action_dim
        obs, act, nxt = f(obs, (n, S)), f(act, (n, A)), f(nxt, (n, S))
        rew, done = f(rew, (n,)), f(done, (n,))
        self._alloc(n)
        with torch.cuda.device(self._dev):
            if state_mean is None:
                check(self._lib.iqlhip_replay_pack(
                    ptr(self._rows), self._stride, S, A, 0, n,
                    ptr(obs), ptr(act), ptr(rew), ptr(nxt), ptr(done), stream_ptr()))
            else:
                mean, std = f(state_mean, (S,)), f(state_std, (S,))
                check(self._lib.iqlhip_replay_pack_normalized(
                    ptr(self._rows), self._stride, S, A, 0, n,
                    ptr(obs), ptr(act), ptr(rew), ptr(nxt), ptr(done), ptr(mean), ptr(std), stream_ptr()))
        torch.cuda.current_stream(self._dev).synchronize()  # the staging tensors die here
        self._size += n
        self._pointer = min(self._size, n)
        _buffer_generation[0] += 1
        self._generation = _buffer_generation[0]  # the contents changed: no prefetched batch survives
        print(f"Dataset size: {n}")

    def touch(self):
        """Call after editing the contents in place through ``_states`` ... ``_dones`` (they are
        views of the packed rows): the library may hold a batch prefetched from the old contents,
        and only a new generation number makes the next ``train_steps`` call stage afresh."""
        _buffer_generation[0] += 1
        self._generation = _buffer_generation[0]

    def view(self) -> _lib.ReplayView:
        key = (self._rows.data_ptr(), self._size, self._pointer, self._generation)
        if getattr(self, "_view_key", None) != key:  # (the struct is rebuilt only when something changed)
            self._view = _lib.ReplayView(ptr(self._rows), min(self._size, self._pointer), self._stride,
                                         self._state_dim, self._action_dim, self._generation)
            self._view_key = key
        return self._view

    def sample(self, batch_size: int, indices: Optional[torch.Tensor] = None) -> TensorBatch:
        """ref:211-221.  Indices are drawn on device (Philox keyed by the torch seed
        and the call count) unless ``indices`` (int64, device) is given."""
        S, A, dev = self._state_dim, self._action_dim, self._dev
        out = [torch.empty((batch_size, w), dtype=torch.float32, device=dev) for w in (S, A, 1, S, 1)]
        seed = torch.initial_seed()
        if seed != self._sample_seed:
            self._sample_seed, self._sample_calls = seed, 0
        v = self.view()
        with torch.cuda.device(dev):
            check(self._lib.iqlhip_replay_sample(
                C.byref(v), batch_size, ptr(indices), seed & 0xFFFFFFFFFFFFFFFF,
                self._sample_calls, *[ptr(t) for t in out], None, stream_ptr()))
        self._sample_calls += 1
        return out

    def add_transition(self):
        raise NotImplementedError


# --------------------------------------------------------------------------- #
# networks (ref:408-543): torch modules hold the parameters; forward() outside
# the trainer runs the stand-alone exact-fp32 MFMA kernel (iqlhip_mlp_forward)
# --------------------------------------------------------------------------- #
class Squeeze(nn.Module):
    def __init__(self, dim=-1):
        super().__init__()
        self.dim = dim

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return x.squeeze(dim=self.dim)


class MLP(nn.Module):
    def __init__(self, dims, activation_fn: Callable[[], nn.Module] = nn.ReLU,
                 output_activation_fn: Callable[[], nn.Module] = None,
                 squeeze_output: bool = False, dropout: Optional[float] = None):
        super().__init__()
        n_dims = len(dims)
        if n_dims < 2:
            raise ValueError("MLP requires at least two dims (input and output)")
        if activation_fn not in (nn.ReLU, nn.Tanh):
            raise NotImplementedError("hidden activation must be nn.ReLU or nn.Tanh")
        if output_activation_fn not in (None, nn.Tanh):
            raise NotImplementedError("output activation must be None or nn.Tanh")
        layers = []
        for i in range(n_dims - 2):
            layers.append(nn.Linear(dims[i], dims[i + 1]))
            layers.append(activation_fn())
            if dropout is not None:
                layers.append(nn.Dropout(dropout))
        layers.append(nn.Linear(dims[-2], dims[-1]))
        if output_activation_fn is not None:
            layers.append(output_activation_fn())
        if squeeze_output:
            if dims[-1] != 1:
                raise ValueError("Last dim must be 1 when squeezing")
            layers.append(Squeeze(-1))
        self.net = nn.Sequential(*layers)
        self._dims = list(dims)
        self._hidden_act = 0 if activation_fn is nn.ReLU else 1
        self._out_act = 0 if output_activation_fn is None else 1
        self._squeeze = squeeze_output
        self._dropout = dropout
        # key of this module's stand-alone dropout masks: the torch seed in force when the module was
        # BUILT (train(seeds_per_gpu=K) builds the K actors after K successive set_seed calls: each
        # keeps its own run's seed, as it would running alone).  The per-module call counter is
        # neither checkpointed nor reset: a resumed run draws its masks from call 0 again.
        self._drop_seed = torch.initial_seed() & 0xFFFFFFFFFFFFFFFF

    def linears(self) -> List[nn.Linear]:
        return [m for m in self.net if isinstance(m, nn.Linear)]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        lin = self.linears()
        drop = None
        if self._dropout is not None and self.training and self._dropout > 0:
            # a module in train mode applies its Dropout layers (ref:436-437; GaussianPolicy.act in
            # train mode samples through them, ref:476-482): masks from the library's Philox stream
            # keyed by the torch seed the module was built under, a fresh one per call
            self._drop_calls = getattr(self, "_drop_calls", 0) + 1
            drop = (float(self._dropout), self._drop_seed, (self._drop_calls - 1) & 0xFFFFFFFF)
        y = mlp_forward_f32([l.weight for l in lin], [l.bias for l in lin], x,
                            w_in_out=False, hidden_act=self._hidden_act, out_act=self._out_act, dropout=drop)
        return y.squeeze(-1) if self._squeeze else y


def mlp_forward_f32(weights, biases, x, *, w_in_out, hidden_act=0, out_act=0, dropout=None) -> torch.Tensor:
    """Exact-fp32 MLP forward of ``x`` [n, in] on the GPU (iqlhip_mlp_forward).
    ``dropout`` = (p, seed, call): nn.Dropout(p) behind every hidden activation (train mode)."""
    lib = _lib.load()
    dev = _lib.require_gpu(x.device)
    lead = x.shape[:-1]
    x2 = x.detach().reshape(-1, x.shape[-1]).to(torch.float32).contiguous()
    n = x2.shape[0]
    d = _lib.MlpDesc()
    d.n_layers = len(weights)
    keep = []
    for i, (w, b) in enumerate(zip(weights, biases)):
        w = w.detach().to(torch.float32).contiguous()
        b = b.detach().to(torch.float32).contiguous()
        keep += [w, b]
        d.dims[i] = w.shape[0] if w_in_out else w.shape[1]
        d.dims[i + 1] = w.shape[1] if w_in_out else w.shape[0]
        d.weights[i], d.biases[i] = w.data_ptr(), b.data_ptr()
    d.w_in_out, d.hidden_act, d.out_act = int(w_in_out), hidden_act, out_act
    if dropout is not None:
        d.dropout_p, d.dropout_seed, d.dropout_call = dropout
    if x2.shape[1] != d.dims[0]:
        raise RuntimeError(f"input width {x2.shape[1]} does not match the first layer ({d.dims[0]})")
    n_out = d.dims[len(weights)]
    out = torch.empty((n, n_out), dtype=torch.float32, device=dev)
    if n:
        with torch.cuda.device(dev):
            check(lib.iqlhip_mlp_forward(C.byref(d), ptr(x2), n, x2.shape[1], ptr(out), n_out,
                                         stream_ptr()))
    return out.reshape(*lead, n_out)


class GaussianPolicy(nn.Module):
    def __init__(self, state_dim: int, act_dim: int, max_action: float, hidden_dim: int = 256,
                 n_hidden: int = 2, dropout: Optional[float] = None):
        super().__init__()
        self.net = MLP([state_dim, *([hidden_dim] * n_hidden), act_dim],
                       output_activation_fn=nn.Tanh, dropout=dropout)
        self.log_std = nn.Parameter(torch.zeros(act_dim, dtype=torch.float32))
        self.max_action = max_action

    def forward(self, obs: torch.Tensor) -> Normal:
        mean = self.net(obs)
        std = torch.exp(self.log_std.detach().clamp(LOG_STD_MIN, LOG_STD_MAX))
        return Normal(mean, std)

    @torch.inference_mode()
    def act(self, state: np.ndarray, device: str = "cpu"):
        state = torch.tensor(state.reshape(1, -1), device=device, dtype=torch.float32)
        dist = self(state)
        action = dist.mean if not self.training else dist.sample()
        action = torch.clamp(self.max_action * action, -self.max_action, self.max_action)
        return action.cpu().data.numpy().flatten()


class DeterministicPolicy(nn.Module):
    def __init__(self, state_dim: int, act_dim: int, max_action: float, hidden_dim: int = 256,
                 n_hidden: int = 2, dropout: Optional[float] = None):
        super().__init__()
        self.net = MLP([state_dim, *([hidden_dim] * n_hidden), act_dim],
                       output_activation_fn=nn.Tanh, dropout=dropout)
        self.max_action = max_action

    def forward(self, obs: torch.Tensor) -> torch.Tensor:
        return self.net(obs)

    @torch.inference_mode()
    def act(self, state: np.ndarray, device: str = "cpu"):
        state = torch.tensor(state.reshape(1, -1), device=device, dtype=torch.float32)
        return (torch.clamp(self(state) * self.max_action, -self.max_action, self.max_action)
                .cpu().data.numpy().flatten())


class TwinQ(nn.Module):
    def __init__(self, state_dim: int, action_dim: int, hidden_dim: int = 256, n_hidden: int = 2):
        super().__init__()
        dims = [state_dim + action_dim, *([hidden_dim] * n_hidden), 1]
        self.q1 = MLP(dims, squeeze_output=True)
        self.q2 = MLP(dims, squeeze_output=True)

    def both(self, state: torch.Tensor, action: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        sa = torch.cat([state, action], 1)
        return self.q1(sa), self.q2(sa)

    def forward(self, state: torch.Tensor, action: torch.Tensor) -> torch.Tensor:
        return torch.min(*self.both(state, action))

    def critics(self) -> List[MLP]:
        return [self.q1, self.q2]


class EnsembleQ(TwinQ):
    """E-way generalisation of TwinQ (BASELINE config 5 "vectorised critic
    ensemble"): the same critic MLP E times as submodules q1..qE.  ``both`` returns all E
    outputs, ``forward`` their minimum; the trainer's q_loss is sum(mse)/E as ref:606 reads
    for ``len(qs)`` critics.  E = 2 is exactly TwinQ (same state_dict keys)."""

    def __init__(self, state_dim: int, action_dim: int, hidden_dim: int = 256, n_hidden: int = 2,
                 n_critics: int = 2):
        if not 2 <= n_critics <= _lib.MAX_CRITICS:
            raise ValueError(f"n_critics must be in [2, {_lib.MAX_CRITICS}]")
        super().__init__(state_dim, action_dim, hidden_dim, n_hidden)
        dims = [state_dim + action_dim, *([hidden_dim] * n_hidden), 1]
        for e in range(2, n_critics):
            setattr(self, f"q{e + 1}", MLP(dims, squeeze_output=True))
        self.n_critics = n_critics

    def critics(self) -> List[MLP]:
        return [getattr(self, f"q{e + 1}") for e in range(self.n_critics)]

    def both(self, state: torch.Tensor, action: torch.Tensor) -> Tuple[torch.Tensor, ...]:
        sa = torch.cat([state, action], 1)
        return tuple(q(sa) for q in self.critics())

    def forward(self, state: torch.Tensor, action: torch.Tensor) -> torch.Tensor:
        qs = self.both(state, action)
        out = qs[0]
        for q in qs[1:]:
            out = torch.min(out, q)
        return out


class ValueFunction(nn.Module):
    def __init__(self, state_dim: int, hidden_dim: int = 256, n_hidden: int = 2):
        super().__init__()
        dims = [state_dim, *([hidden_dim] * n_hidden), 1]
        self.v = MLP(dims, squeeze_output=True)

    def forward(self, state: torch.Tensor) -> torch.Tensor:
        return self.v(state)


# --------------------------------------------------------------------------- #
# trainer (ref:546-688)
# --------------------------------------------------------------------------- #
def _linears(mlp: MLP, what: str) -> List[nn.Linear]:
    """The Linear layers of one network (n_hidden + 1 of them, ref:417-449)."""
    lin = mlp.linears()
    if not 2 <= len(lin) <= _lib.MAX_HIDDEN + 1:
        raise NotImplementedError(f"{what}: n_hidden must be in 1..{_lib.MAX_HIDDEN} (got {len(lin) - 1})")
    if mlp._hidden_act != 0:
        raise NotImplementedError(f"{what}: hidden activation must be ReLU")
    return lin


class ImplicitQLearning:
    """ref:546-688.  Extra keyword-only arguments (not in the reference):

    precision  "bf16" (default; the reference's ``torch.amp.autocast(bfloat16)``
               region, ref:650) or "fp32" (autocast disabled, exact fp32 MFMA).
    seed       Philox key for on-device batch indices / dropout masks
               (default: ``torch.initial_seed()``).
    keep_grads also store parameter gradients into ``p.grad`` (tests).
    polyak_form 0: ``tp.lerp_(sp, tau)`` (ref:127-129); 1: ``(1 - tau) tp + tau sp``
               (algorithms/custom_offline/iql.py:85-87; iqlpref_amd.custom_offline sets it).
    """

    def __init__(self, max_action: float, actor: nn.Module, actor_optimizer: torch.optim.Optimizer,
                 q_network: nn.Module, q_optimizer: torch.optim.Optimizer, v_network: nn.Module,
                 v_optimizer: torch.optim.Optimizer, iql_tau: float = 0.7, beta: float = 3.0,
                 max_steps: int = 1000000, discount: float = 0.99, tau: float = 0.005,
                 device: str = "cpu", *, precision: str = "bf16", seed: Optional[int] = None,
                 keep_grads: bool = False, polyak_form: int = 0):
        self._lib = _lib.load()
        self._dev = _lib.require_gpu(device)
        if precision not in ("bf16", "fp32"):
            raise ValueError("precision must be 'bf16' or 'fp32'")
        self.max_action = max_action
        self.qf = q_network
        self.vf = v_network
        self.actor = actor
        self.v_optimizer = v_optimizer
        self.q_optimizer = q_optimizer
        self.actor_optimizer = actor_optimizer
        self.actor_lr_schedule = CosineAnnealingLR(self.actor_optimizer, max_steps)
        self.iql_tau = iql_tau
        self.beta = beta
        self.discount = discount
        self.tau = tau
        self.total_it = 0
        self.device = device
        self._device_type = device.split(":")[0]
        self._max_steps = int(max_steps)
        self._precision = PREC_BF16 if precision == "bf16" else PREC_FP32
        self._seed = int(torch.initial_seed() if seed is None else seed) & 0xFFFFFFFFFFFFFFFF
        self._keep_grads = keep_grads
        self._polyak_form = int(polyak_form)
        self._handle = None
        self._handle_batch = None
        self._group_owner = None  # the SeedGroup whose device-side group holds this trainer
        # steps per hipGraph of train_steps when the caller names none; 0 = plain kernel launches from
        # the library's C loop.  Measured (round 3, one seed): plain launches 66.7k steps/s, graphs of 50
        # steps 66.0k, graphs of 8 steps 63.6k -- every graph launch costs ~5 us of device time that
        # back-to-back kernels do not, and the host's 3 us per launch stay hidden behind 5 us kernels.
        self._graph_unroll = 0

        if not isinstance(q_network, TwinQ) or not isinstance(v_network, ValueFunction) or \
                not isinstance(actor, (GaussianPolicy, DeterministicPolicy)):
            raise TypeError("networks must be iqlpref_amd TwinQ / ValueFunction / *Policy modules")
        self._deterministic = isinstance(actor, DeterministicPolicy)
        crit = q_network.critics()
        self._n_critics = E = len(crit)
        self._nets = [(f"q{e + 1}", _linears(c, f"TwinQ.q{e + 1}")) for e, c in enumerate(crit)]
        self._nets += [("v", _linears(v_network.v, "ValueFunction")),
                       ("actor", _linears(actor.net, "actor"))]
        self._state_dim = self._nets[E][1][0].in_features
        self._action_dim = self._nets[E + 1][1][-1].out_features
        self._hidden = self._nets[E][1][0].out_features
        self._n_hidden = len(self._nets[E][1]) - 1
        for _, lin in self._nets:
            if len(lin) != self._n_hidden + 1:
                raise NotImplementedError("all networks must have the same number of hidden layers")
            if any(l.out_features != self._hidden for l in lin[:-1]) or \
                    any(l.in_features != self._hidden for l in lin[1:]):
                raise NotImplementedError("all hidden layers must share one width")
        if self._nets[0][1][0].in_features != self._state_dim + self._action_dim:
            raise ValueError("TwinQ input width must be state_dim + action_dim")
        self._dropout = actor.net._dropout
        for p in list(q_network.parameters()) + list(v_network.parameters()) + list(actor.parameters()):
            if p.device.type != "cuda":
                raise ValueError("networks must live on the trainer's device before construction")
        self._build_arenas()
        self.q_target = copy.deepcopy(self.qf).requires_grad_(False).to(device)
        self._bind_target()

    # -- arenas ------------------------------------------------------------- #
    def _cfg(self, batch_size: int) -> _lib.TrainerConfig:
        g = lambda opt, k: opt.param_groups[0][k]
        for opt in (self.q_optimizer, self.v_optimizer, self.actor_optimizer):
            if not isinstance(opt, torch.optim.Adam) or g(opt, "weight_decay") != 0 or g(opt, "amsgrad") \
                    or g(opt, "maximize"):
                raise NotImplementedError("optimisers must be plain torch.optim.Adam "
                                          "(no weight decay / amsgrad / maximize)")
        b1, b2 = g(self.q_optimizer, "betas")
        for opt in (self.v_optimizer, self.actor_optimizer):
            if tuple(g(opt, "betas")) != (b1, b2) or g(opt, "eps") != g(self.q_optimizer, "eps"):
                raise NotImplementedError("the three Adam optimisers must share betas and eps")
        c = _lib.TrainerConfig()
        c.state_dim, c.action_dim, c.hidden_dim = self._state_dim, self._action_dim, self._hidden
        c.batch_size = batch_size
        c.deterministic = int(self._deterministic)
        c.precision = self._precision
        c.dropout_p = -1.0 if not self._dropout else float(self._dropout)
        c.discount, c.tau, c.beta, c.iql_tau = self.discount, self.tau, self.beta, self.iql_tau
        c.lr_q = float(g(self.q_optimizer, "lr"))
        c.lr_v = float(g(self.v_optimizer, "lr"))
        c.lr_actor = float(self.actor_lr_schedule.base_lrs[0])
        c.adam_beta1, c.adam_beta2, c.adam_eps = b1, b2, float(g(self.q_optimizer, "eps"))
        c.cosine_t_max = int(self.actor_lr_schedule.T_max)
        c.seed = self._seed
        c.n_critics = self._n_critics
        c.polyak_form = self._polyak_form
        c.n_hidden = self._n_hidden
        return c

    def _tensor_list(self) -> List[nn.Parameter]:
        """Parameters in the arena order of iqlhip_arena_layout."""
        out = []
        for _, lin in self._nets:
            for l in lin:
                out += [l.weight, l.bias]
        if not self._deterministic:
            out.append(self.actor.log_std)
        return out

    def _build_arenas(self):
        cfg = self._cfg(32)
        offs = (C.c_int64 * _lib.N_TENSORS)()
        n_params, n_target = C.c_int64(), C.c_int64()
        check(self._lib.iqlhip_arena_layout(C.byref(cfg), C.byref(offs), C.byref(n_params),
                                            C.byref(n_target)))
        self._offsets = [int(o) for o in offs]
        self._n_params, self._n_target = n_params.value, n_target.value
        dev = self._dev
        z = lambda n: torch.zeros(n, dtype=torch.float32, device=dev)
        self._params, self._exp_avg, self._exp_avg_sq = z(self._n_params), z(self._n_params), z(self._n_params)
        self._target = z(self._n_target)
        self._grads = z(self._n_params) if self._keep_grads else None
        tensors = self._tensor_list()
        offsets = [o for o in self._offsets if o >= 0]
        assert len(tensors) == len(offsets)
        self._views = []
        with torch.no_grad():
            for p, o in zip(tensors, offsets):
                view = self._params[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view  # same Parameter object: the optimisers keep their references
                self._views.append((p, o))
                if self._grads is not None:
                    p.grad = self._grads[o:o + p.numel()].view(p.shape)

    def _bind_target(self):
        """q_target parameters become views of the target arena (ref:565)."""
        tl = []
        for mlp in self.q_target.critics():
            for l in mlp.linears():
                tl += [l.weight, l.bias]
        with torch.no_grad():
            for p, o in zip(tl, self._offsets[:2 * (self._n_hidden + 1) * self._n_critics]):
                view = self._target[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view

    def _bind_optimizer_state(self, rebuild: bool = False):
        """Adam moments live in the arenas; expose them through optimizer.state so
        ``optimizer.state_dict()`` stays what torch.optim.Adam would write.  The views are made
        once (and again after ``load_state_dict``, which replaces the state tensors); afterwards a
        call only moves the step counters."""
        steps = getattr(self, "_step_tensors", None)
        if steps is None or rebuild:
            steps = []
            for p, o in self._views:
                for opt in (self.q_optimizer, self.v_optimizer, self.actor_optimizer):
                    if any(p is q for q in opt.param_groups[0]["params"]):
                        st = opt.state[p]
                        st["step"] = torch.tensor(float(self.total_it))
                        st["exp_avg"] = self._exp_avg[o:o + p.numel()].view(p.shape)
                        st["exp_avg_sq"] = self._exp_avg_sq[o:o + p.numel()].view(p.shape)
                        steps.append(st["step"])
            self._step_tensors = steps
            return
        t = float(self.total_it)
        for st in steps:
            st.fill_(t)

    def _ensure_handle(self, batch_size: int):
        if self._handle is not None and self._handle_batch == batch_size:
            return
        self._destroy_handle()
        cfg = self._cfg(batch_size)
        ar = _lib.Arenas(ptr(self._params), ptr(self._exp_avg), ptr(self._exp_avg_sq),
                         ptr(self._target), ptr(self._grads))
        h = C.c_void_p()
        with torch.cuda.device(self._dev):
            check(self._lib.iqlhip_trainer_create(C.byref(h), C.byref(cfg), C.byref(ar)))
            self._handle, self._handle_batch = h, batch_size
            check(self._lib.iqlhip_trainer_set_step(h, self.total_it))
            check(self._lib.iqlhip_trainer_sync_weights(h, stream_ptr()))

    def _destroy_handle(self):
        if getattr(self, "_handle", None) is not None:
            owner = getattr(self, "_group_owner", None)
            if owner is not None:  # the device-side group refers to this handle: dissolve it first
                owner._drop_group()
            torch.cuda.synchronize(self._dev)
            self._lib.iqlhip_trainer_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._destroy_handle()
        except Exception:
            pass

    def step_kind(self, batch_size: int) -> str:
        """"tuned" (the three-kernel step of csrc/iql_step.hip: n_hidden = 2, hidden_dim 64 / 128 / 256) or
        "general" (the layer-wise step of csrc/iql_deep.hip: every other depth / width)."""
        self._ensure_handle(batch_size)
        kind = C.c_int32()
        check(self._lib.iqlhip_trainer_step_kind(self._handle, C.byref(kind)))
        return "general" if kind.value else "tuned"

    def sync_weights(self):
        """Call after writing parameters from outside (the compute-precision copies
        the kernels read are rebuilt from the fp32 masters)."""
        if self._handle is not None:
            with torch.cuda.device(self._dev):
                check(self._lib.iqlhip_trainer_sync_weights(self._handle, stream_ptr()))

    def _after_steps(self, n: int):
        self.total_it += n
        # CosineAnnealingLR bookkeeping (ref:637) in closed form
        sch = self.actor_lr_schedule
        lr = sch.eta_min + (sch.base_lrs[0] - sch.eta_min) * (1 + math.cos(math.pi * self.total_it / sch.T_max)) / 2
        sch.last_epoch = self.total_it
        sch._step_count = self.total_it + 1
        sch._last_lr = [lr]
        self.actor_optimizer.param_groups[0]["lr"] = lr
        self._bind_optimizer_state()

    def _refresh_lrs(self):
        lrs = (self._handle.value, float(self.q_optimizer.param_groups[0]["lr"]),
               float(self.v_optimizer.param_groups[0]["lr"]), float(self.actor_lr_schedule.base_lrs[0]))
        if lrs != getattr(self, "_lrs_sent", None):
            check(self._lib.iqlhip_trainer_set_lr(self._handle, *lrs[1:]))
            self._lrs_sent = lrs

    # -- the step ----------------------------------------------------------- #
    def train(self, batch: TensorBatch, dropout_keep: Optional[torch.Tensor] = None) -> Dict[str, float]:
        """ref:639-662 on an explicit batch [s, a, r, s', d]."""
        s, a, r, s2, d = [t.detach().to(torch.float32).contiguous() for t in batch]
        if a.shape != (s.shape[0], self._action_dim):
            raise RuntimeError("Actions shape missmatch")  # ref:627-628
        self._ensure_handle(s.shape[0])
        self._refresh_lrs()
        losses = torch.empty(3, dtype=torch.float32, device=self._dev)
        with torch.cuda.device(self._dev):
            check(self._lib.iqlhip_train_batch(self._handle, ptr(s), ptr(a), ptr(r), ptr(s2), ptr(d),
                                               ptr(dropout_keep), ptr(losses), stream_ptr()))
        self._after_steps(1)
        v, q, p = losses.tolist()  # the reference's three .item() syncs, as one
        return {"value_loss": v, "q_loss": q, "actor_loss": p}

    def train_steps(self, replay_buffer: ReplayBuffer, n_steps: int, batch_size: int, *,
                    indices: Optional[torch.Tensor] = None,
                    dropout_keep: Optional[torch.Tensor] = None,
                    return_losses: bool = True, graph_unroll: Optional[int] = None):
        """ref:1533-1536 fused: ``n_steps`` x (sample + train) without host syncs.

        Returns a float32 device tensor [n_steps, 3] (value, q, actor loss per step)
        when ``return_losses``; nothing is copied to the host."""
        self._ensure_handle(batch_size)
        self._refresh_lrs()
        losses = (torch.empty((n_steps, 3), dtype=torch.float32, device=self._dev)
                  if return_losses else None)
        if indices is not None:
            if indices.dtype != torch.int64 or tuple(indices.shape) != (n_steps, batch_size):
                raise ValueError("indices must be int64 [n_steps, batch_size]")
            indices = indices.contiguous()
        if dropout_keep is not None:
            dropout_keep = dropout_keep.to(torch.uint8).contiguous()
        v = replay_buffer.view()
        unroll = self._graph_unroll if graph_unroll is None else graph_unroll
        if torch.cuda.current_device() == (self._dev.index or 0):  # (the context manager costs ~5 us a call)
            check(self._lib.iqlhip_train_steps(self._handle, C.byref(v), n_steps, ptr(indices),
                                               ptr(dropout_keep), ptr(losses), unroll, stream_ptr()))
        else:
            with torch.cuda.device(self._dev):
                check(self._lib.iqlhip_train_steps(self._handle, C.byref(v), n_steps, ptr(indices),
                                                   ptr(dropout_keep), ptr(losses), unroll, stream_ptr()))
        self._after_steps(n_steps)
        return losses

    # -- checkpoints (ref:664-688) ------------------------------------------ #
    def state_dict(self) -> Dict[str, Any]:
        torch.cuda.synchronize(self._dev)
        # (fresh views / step counters: whatever was done to the optimisers' state from outside, the
        # checkpoint holds the arenas' moments and this trainer's step count)
        if self.total_it > 0:  # (before the first step the optimisers hold no state, as the reference's)
            self._bind_optimizer_state(rebuild=True)
        return {
            "qf": self.qf.state_dict(),
            "q_optimizer": self.q_optimizer.state_dict(),
            "vf": self.vf.state_dict(),
            "v_optimizer": self.v_optimizer.state_dict(),
            "actor": self.actor.state_dict(),
            "actor_optimizer": self.actor_optimizer.state_dict(),
            "actor_lr_schedule": self.actor_lr_schedule.state_dict(),
            "total_it": self.total_it,
        }

    def load_state_dict(self, state_dict: Dict[str, Any]):
        # nothing of this trainer may still be in flight (SeedGroup streams are non-blocking)
        torch.cuda.synchronize(self._dev)
        # checkpoints written after torch.compile wrapped the nets (ref:1523-1528) carry this prefix
        strip = lambda sd: {k.removeprefix("_orig_mod."): v for k, v in sd.items()}
        self.qf.load_state_dict(strip(state_dict["qf"]))  # copies into the arena views
        self.q_optimizer.load_state_dict(state_dict["q_optimizer"])
        self.vf.load_state_dict(strip(state_dict["vf"]))
        self.v_optimizer.load_state_dict(state_dict["v_optimizer"])
        self.actor.load_state_dict(strip(state_dict["actor"]))
        self.actor_optimizer.load_state_dict(state_dict["actor_optimizer"])
        self.actor_lr_schedule.load_state_dict(state_dict["actor_lr_schedule"])
        self.total_it = state_dict["total_it"]
        # optimizer.load_state_dict made fresh moment tensors: move them into the arenas
        # (a checkpoint taken before the first step has no moments: they are zero, not stale)
        with torch.no_grad():
            self._exp_avg.zero_()
            self._exp_avg_sq.zero_()
            for p, o in self._views:
                for opt in (self.q_optimizer, self.v_optimizer, self.actor_optimizer):
                    st = opt.state.get(p)
                    if st and "exp_avg" in st:
                        self._exp_avg[o:o + p.numel()].copy_(st["exp_avg"].reshape(-1))
                        self._exp_avg_sq[o:o + p.numel()].copy_(st["exp_avg_sq"].reshape(-1))
            self._target.copy_(self._params[:self._n_target])  # ref:679 deepcopy(qf)
        self._bind_optimizer_state(rebuild=True)
        if self._handle is not None:
            with torch.cuda.device(self._dev):
                check(self._lib.iqlhip_trainer_set_step(self._handle, self.total_it))
        self.sync_weights()

    # -- forward passes at the trainer's precision (tests, diagnostics) ------ #
    def forward(self, which: str, states: torch.Tensor, actions: Optional[torch.Tensor] = None):
        """q / v / actor / q_target on the live weights inside the autocast region."""
        idx = {"q": 0, "v": 1, "actor": 2, "q_target": 3}[which]
        self._ensure_handle(self._handle_batch or 32)
        s = states.detach().to(torch.float32).contiguous()
        a = None if actions is None else actions.detach().to(torch.float32).contiguous()
        width = {0: self._n_critics, 1: 1, 2: self._action_dim, 3: self._n_critics}[idx]
        out = torch.empty((s.shape[0], width), dtype=torch.float32, device=self._dev)
        with torch.cuda.device(self._dev):
            check(self._lib.iqlhip_forward(self._handle, idx, ptr(s), ptr(a), s.shape[0], ptr(out),
                                           stream_ptr()))
        return out
