#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING the reference.

Run in the build container only (the reference never travels to the GPU box):

    python tests/golden/make_fixtures.py [--ref /root/reference]

The script imports /root/reference/algorithms/offline/iql.py with inert stub
modules for the third-party packages that are absent here (d4rl, gym, wandb,
pyrallis, optbnn -- SURVEY.md section 8c), drives the reference's own functions on
seeded synthetic inputs and stores ONLY inputs and outputs (.npz).  No reference
source or bytecode is written anywhere.

Two families are written for the training step:
  *_bf16.npz  the reference as written (torch.amp.autocast bf16 on CPU)
  *_fp32.npz  the same run with the autocast context replaced by a no-op
"""
import argparse
import contextlib
import importlib.util
import os
import sys
import tempfile
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from tests import fake_envs  # noqa: E402  (our deterministic stand-in environments)
from tests.helpers import regen_inputs, synth_dataset, tensor_checks  # noqa: E402


# --------------------------------------------------------------------------- #
# stand-ins for the absent gp_reward-priors submodule (call-site contract only,
# iql.py:953-972, 1326-1336): x @ W + b layers, keys layers.0.W / layers.linear_i.W
# --------------------------------------------------------------------------- #
class _RLayer(nn.Module):
    def __init__(self, i, o):
        super().__init__()
        self.W = nn.Parameter(torch.zeros(i, o))
        self.b = nn.Parameter(torch.zeros(o))

    def forward(self, x):
        return x @ self.W + self.b


class StandInRewardMLP(nn.Module):
    def __init__(self, input_dim, output_dim, hidden_dims, activation_fn="relu"):
        super().__init__()
        dims = [input_dim] + list(hidden_dims)
        self.layers = nn.ModuleDict()
        self.layers["0"] = _RLayer(dims[0], dims[1])
        for i in range(1, len(hidden_dims)):
            self.layers[f"linear_{i}"] = _RLayer(dims[i], dims[i + 1])
        self.out = _RLayer(dims[-1], output_dim)
        self.act = {"relu": torch.relu, "tanh": torch.tanh}[activation_fn]

    def forward(self, x):
        for k in self.layers:
            x = self.act(self.layers[k](x))
        return self.out(x)

    # parameters() order = hidden (W,b)*depth then output (W,b): iql.py:953-961


def import_reference(ref_root):
    def stub(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    stub("d4rl")
    # gym: what eval_actor / _make_eval_env / wrap_env touch (ref:140-161, 253-262, 289-295), backed
    # by our deterministic stand-in environments (tests/fake_envs.py)
    stub("gym", Env=object, make=lambda name: fake_envs.FakeGymEnv(name),
         wrappers=types.SimpleNamespace(TransformObservation=fake_envs.TransformObservation,
                                        TransformReward=fake_envs.TransformReward),
         vector=types.SimpleNamespace(AsyncVectorEnv=fake_envs.SyncVectorEnv))
    stub("wandb")
    stub("pyrallis", wrap=lambda *a, **k: (lambda f: f))
    stub("optbnn")
    stub("optbnn.bnn")
    stub("optbnn.bnn.nets")
    stub("optbnn.bnn.nets.mlp", MLP=StandInRewardMLP)
    stub("optbnn.bnn.nets.pref_trans", PT=object)
    path = os.path.join(ref_root, "algorithms", "offline", "iql.py")
    spec = importlib.util.spec_from_file_location("ref_iql", path)
    mod = importlib.util.module_from_spec(spec)
    sys.dont_write_bytecode = True
    spec.loader.exec_module(mod)
    return mod


def params_of(module):
    return {k: v.detach().cpu().numpy().copy() for k, v in module.state_dict().items()}


def flat(prefix, d):
    return {f"{prefix}/{k}": v for k, v in d.items()}


def adam_state(opt, module):
    """exp_avg / exp_avg_sq / step keyed by the module's parameter names."""
    names = [n for n, _ in module.named_parameters()]
    out = {}
    for n, p in zip(names, opt.param_groups[0]["params"]):
        st = opt.state.get(p, None)
        if st:
            out[f"{n}/exp_avg"] = st["exp_avg"].detach().numpy().copy()
            out[f"{n}/exp_avg_sq"] = st["exp_avg_sq"].detach().numpy().copy()
            out[f"{n}/step"] = np.asarray(float(st["step"]))
    return out


def run_trajectory(ref, *, s_dim, a_dim, hidden, batch, n_rows, k_steps, seed,
                   beta, iql_tau, discount, tau, deterministic, dropout, max_steps,
                   fp32, reward_kind, snap_steps=(), ckpt_at=None, n_hidden=2):
    rng = np.random.default_rng(seed)
    data = synth_dataset(rng, n_rows, s_dim, a_dim, reward_kind)
    buf = ref.ReplayBuffer(s_dim, a_dim, n_rows + 7, "cpu")
    buf.load_d4rl_dataset(data)

    torch.manual_seed(seed)
    q = ref.TwinQ(s_dim, a_dim, hidden_dim=hidden, n_hidden=n_hidden)
    v = ref.ValueFunction(s_dim, hidden_dim=hidden, n_hidden=n_hidden)
    pol_cls = ref.DeterministicPolicy if deterministic else ref.GaussianPolicy
    actor = pol_cls(s_dim, a_dim, 1.0, hidden_dim=hidden, n_hidden=n_hidden, dropout=dropout)
    if not deterministic:
        with torch.no_grad():  # non-trivial log_std so its gradient path is exercised
            actor.log_std.copy_(torch.linspace(-0.5, 0.3, a_dim))
    vo = torch.optim.Adam(v.parameters(), lr=3e-4)
    qo = torch.optim.Adam(q.parameters(), lr=3e-4)
    ao = torch.optim.Adam(actor.parameters(), lr=3e-4)
    trainer = ref.ImplicitQLearning(
        max_action=1.0, actor=actor, actor_optimizer=ao, q_network=q, q_optimizer=qo,
        v_network=v, v_optimizer=vo, iql_tau=iql_tau, beta=beta, max_steps=max_steps,
        discount=discount, tau=tau, device="cpu")

    out = {}
    out.update(flat("init/qf", params_of(q)))
    out.update(flat("init/vf", params_of(v)))
    out.update(flat("init/actor", params_of(actor)))
    for k in ("observations", "actions", "rewards", "next_observations"):
        out[f"data/{k}"] = data[k]
    out["data/terminals"] = data["terminals"].astype(np.float32)

    # dropout-mask capture: keep/drop decision of the reference's own F.dropout
    masks = []
    hooks = []
    if dropout is not None:
        def hook(_m, inp, outp):
            keep = (outp != 0) | (inp[0] == 0)
            masks.append(keep.detach().numpy().astype(np.uint8))
        for m in actor.modules():
            if isinstance(m, nn.Dropout):
                hooks.append(m.register_forward_hook(hook))

    idx_all = np.zeros((k_steps, batch), dtype=np.int64)
    losses = np.zeros((k_steps, 3), dtype=np.float64)
    lrs = np.zeros(k_steps, dtype=np.float64)

    torch.set_autocast_cache_enabled(False)  # SURVEY 8c gotcha: eager + cache fails
    real_autocast = torch.amp.autocast
    if fp32:
        torch.amp.autocast = lambda *a, **k: contextlib.nullcontext()
    try:
        g = torch.Generator().manual_seed(seed + 1)
        for t in range(k_steps):
            idx = torch.randint(0, n_rows, (batch,), generator=g)
            idx_all[t] = idx.numpy()
            b = [buf._states[idx], buf._actions[idx], buf._rewards[idx],
                 buf._next_states[idx], buf._dones[idx]]
            lrs[t] = ao.param_groups[0]["lr"]  # lr used BY this step
            log = trainer.train(b)
            losses[t] = [log["value_loss"], log["q_loss"], log["actor_loss"]]
            if t == 0:
                out.update(flat("step1/qf", params_of(q)))
                out.update(flat("step1/vf", params_of(v)))
                out.update(flat("step1/actor", params_of(actor)))
                out.update(flat("step1/q_target", params_of(trainer.q_target)))
            if ckpt_at is not None and t + 1 == ckpt_at:
                # the reference's own checkpoint dict (ref:664-674) after ckpt_at steps, as arrays, and
                # the target network it does NOT checkpoint (the run continues on its own target)
                sd = trainer.state_dict()
                for net in ("qf", "vf", "actor"):
                    out.update(flat(f"ckpt/{net}", {k: v.detach().numpy().copy() for k, v in sd[net].items()}))
                for oname, opt, mod in (("q_optimizer", qo, q), ("v_optimizer", vo, v), ("actor_optimizer", ao, actor)):
                    st = sd[oname]["state"]
                    for i, ent in st.items():
                        out[f"ckpt/{oname}/{i}/exp_avg"] = ent["exp_avg"].numpy().copy()
                        out[f"ckpt/{oname}/{i}/exp_avg_sq"] = ent["exp_avg_sq"].numpy().copy()
                        out[f"ckpt/{oname}/{i}/step"] = np.asarray(float(ent["step"]))
                    out[f"ckpt/{oname}/lr"] = np.asarray(sd[oname]["param_groups"][0]["lr"])
                    out[f"ckpt/{oname}/n_params"] = np.asarray(len(sd[oname]["param_groups"][0]["params"]))
                sch = sd["actor_lr_schedule"]
                out["ckpt/actor_lr_schedule"] = np.asarray([sch["T_max"], sch["eta_min"], sch["base_lrs"][0],
                                                            sch["last_epoch"], sch["_step_count"], sch["_last_lr"][0]],
                                                           dtype=np.float64)
                out["ckpt/total_it"] = np.asarray(int(sd["total_it"]))
                out.update(flat("ckpt/q_target", params_of(trainer.q_target)))
            if t + 1 in snap_steps:  # (long trajectories: the state after t + 1 steps)
                pre = f"at{t + 1}"
                out.update(flat(f"{pre}/qf", params_of(q)))
                out.update(flat(f"{pre}/vf", params_of(v)))
                out.update(flat(f"{pre}/actor", params_of(actor)))
                out.update(flat(f"{pre}/q_target", params_of(trainer.q_target)))
                out.update(flat(f"{pre}/q_adam", adam_state(qo, q)))
                out.update(flat(f"{pre}/actor_adam", adam_state(ao, actor)))
    finally:
        torch.amp.autocast = real_autocast
        for h in hooks:
            h.remove()

    out["indices"] = idx_all
    out["losses"] = losses
    out["actor_lr"] = lrs
    out["final_actor_lr"] = np.asarray(ao.param_groups[0]["lr"])
    out.update(flat("final/qf", params_of(q)))
    out.update(flat("final/vf", params_of(v)))
    out.update(flat("final/actor", params_of(actor)))
    out.update(flat("final/q_target", params_of(trainer.q_target)))
    out.update(flat("final/q_adam", adam_state(qo, q)))
    out.update(flat("final/v_adam", adam_state(vo, v)))
    out.update(flat("final/actor_adam", adam_state(ao, actor)))
    if masks:
        m = np.stack(masks).reshape(k_steps, n_hidden, batch, hidden)
        out["dropout_keep"] = np.packbits(m, axis=-1)
    out["hyper"] = np.asarray(
        [s_dim, a_dim, hidden, batch, n_rows, k_steps, beta, iql_tau, discount, tau,
         float(deterministic), -1.0 if dropout is None else dropout, max_steps, n_hidden],
        dtype=np.float64)
    sd = trainer.state_dict()
    out["state_dict_keys"] = np.asarray(sorted(sd.keys()))
    out["actor_keys"] = np.asarray(list(sd["actor"].keys()))
    out["qf_keys"] = np.asarray(list(sd["qf"].keys()))
    out["vf_keys"] = np.asarray(list(sd["vf"].keys()))
    return out


# Depths and widths away from the default n_hidden = 2 / hidden_dim 256 (ref:417-449 MLP, :458-459, 519, 538):
# what the general layer-wise step (csrc/iql_deep.hip) runs.  Small on purpose: full parameter sets are kept.
SHAPES = {
    # three hidden layers of a width that is no multiple of 32, Gaussian policy, antmaze hyper-parameters
    "traj_deep3_w96": dict(s_dim=29, a_dim=8, hidden=96, n_hidden=3, batch=64, n_rows=1000, k_steps=8, seed=11,
                           beta=10.0, iql_tau=0.9, discount=0.99, tau=0.005, deterministic=False, dropout=None,
                           max_steps=1000, reward_kind="sparse"),
    # ONE hidden layer of a width that is no multiple of 16, pen shapes with actor dropout
    "traj_shallow1_w40_drop": dict(s_dim=45, a_dim=24, hidden=40, n_hidden=1, batch=32, n_rows=500, k_steps=6, seed=12,
                                   beta=3.0, iql_tau=0.8, discount=0.99, tau=0.005, deterministic=False, dropout=0.1,
                                   max_steps=50, reward_kind="normal"),
    # four hidden layers, deterministic policy with dropout in every one of them, halfcheetah shapes
    "traj_deep4_w72_det": dict(s_dim=17, a_dim=6, hidden=72, n_hidden=4, batch=48, n_rows=700, k_steps=6, seed=13,
                               beta=3.0, iql_tau=0.7, discount=0.99, tau=0.005, deterministic=True, dropout=0.2,
                               max_steps=100, reward_kind="normal"),
}


def big_summary(ref, fp32):
    """H=256, B=256 antmaze shapes: losses + strided parameter samples per mode;
    the (mode-independent) initial parameters and data go to a shared file."""
    full = run_trajectory(ref, s_dim=29, a_dim=8, hidden=256, batch=256, n_rows=1024,
                          k_steps=10, seed=7, beta=10.0, iql_tau=0.9, discount=0.99,
                          tau=0.005, deterministic=False, dropout=None,
                          max_steps=1_000_000, fp32=fp32, reward_kind="sparse")
    keep, common = {}, {}
    for k, v in full.items():
        if k.startswith(("init/", "data/")):
            common[k] = v
        elif k.startswith(("step1/", "final/")) and v.size > 2048:
            keep[k + "#stride37"] = v.reshape(-1)[::37].copy()
            keep[k + "#sum"] = np.asarray(v.astype(np.float64).sum())
            keep[k + "#abssum"] = np.asarray(np.abs(v.astype(np.float64)).sum())
        else:
            keep[k] = v
    return keep, common


# BASELINE configs at their full widths (H = 256): the reference's trajectories as strided
# summaries.  Initial parameters and data are NOT stored: they are functions of the seed
# (torch.manual_seed + the module constructors, numpy default_rng) and tests/helpers.regen_inputs
# rebuilds them with OUR constructors; the fixture keeps per-tensor checksums of what the
# reference started from, and the tests refuse to run on anything else.
BIG = {
    # config 1: halfcheetah-medium-v2 (configs/offline/iql/halfcheetah/medium_v2.yaml: beta 3, iql_tau 0.7)
    "traj_cheetah_h256": dict(s_dim=17, a_dim=6, hidden=256, batch=256, n_rows=1024, k_steps=10, seed=21,
                              beta=3.0, iql_tau=0.7, discount=0.99, tau=0.005, deterministic=False,
                              dropout=None, max_steps=1_000_000, reward_kind="normal"),
    # config 3: pen-human-v1 (configs/offline/iql/pen/human_v1.yaml: actor_dropout 0.1, beta 3, iql_tau 0.8)
    "traj_pen_h256": dict(s_dim=45, a_dim=24, hidden=256, batch=256, n_rows=1024, k_steps=10, seed=22,
                          beta=3.0, iql_tau=0.8, discount=0.99, tau=0.005, deterministic=False,
                          dropout=0.1, max_steps=1_000_000, reward_kind="normal"),
    # config 5's batch with the reference's TwinQ (E = 2), antmaze hyper-parameters
    "traj_antmaze_b1024": dict(s_dim=29, a_dim=8, hidden=256, batch=1024, n_rows=4096, k_steps=10, seed=23,
                               beta=10.0, iql_tau=0.9, discount=0.99, tau=0.005, deterministic=False,
                               dropout=None, max_steps=1_000_000, reward_kind="sparse"),
}


def big_regen(ref, name, fp32):
    cfg = BIG[name]
    full = run_trajectory(ref, fp32=fp32, **cfg)
    keep = {"regen_seed": np.asarray(cfg["seed"]), "reward_kind": np.asarray(cfg["reward_kind"])}
    for k, v in full.items():
        if k.startswith(("init/", "data/")):
            keep["check/" + k] = tensor_checks(v)
        elif k.startswith(("step1/", "final/")) and v.size > 2048:
            keep[k + "#stride37"] = v.reshape(-1)[::37].copy()
            keep[k + "#sum"] = np.asarray(v.astype(np.float64).sum())
            keep[k + "#abssum"] = np.asarray(np.abs(v.astype(np.float64)).sum())
        else:
            keep[k] = v
    # what the tests will rebuild must be what the reference started from, bit for bit
    data, nets = regen_inputs(keep)
    for k, v in data.items():
        assert np.array_equal(v, full["data/" + k]), k
    for pre, net in zip(("qf", "vf", "actor"), nets):
        for k, v in net.items():
            assert np.array_equal(v, full[f"init/{pre}/{k}"]), (pre, k)
    return keep


# BASELINE config 1 says "1k steps": K = 1,000-step trajectories of the reference for configs 1 and 2
# (VERDICT r3 item 2 ii): every step's losses, strided summaries of parameters / target / Adam
# moments after 100 and 1,000 steps.  Inputs AND indices are functions of the seed (the index
# stream is torch.Generator().manual_seed(seed + 1) -> randint, regenerated by the test and
# checked against the checksum kept here).
LONG = {
    "traj_long_cheetah": dict(s_dim=17, a_dim=6, hidden=256, batch=256, n_rows=4096, k_steps=1000, seed=41,
                              beta=3.0, iql_tau=0.7, discount=0.99, tau=0.005, deterministic=False,
                              dropout=None, max_steps=1_000_000, reward_kind="normal"),
    "traj_long_antmaze": dict(s_dim=29, a_dim=8, hidden=256, batch=256, n_rows=4096, k_steps=1000, seed=42,
                              beta=10.0, iql_tau=0.9, discount=0.99, tau=0.005, deterministic=False,
                              dropout=None, max_steps=1_000_000, reward_kind="sparse"),
}


# Long-horizon arithmetic pinned without the chaos of a free-running comparison: the reference's
# checkpoint after 990 steps (+ its target net), then its next 10 steps.  The test loads the
# checkpoint into OUR trainer (load_state_dict on a reference-written dict), sets the target and
# replays the 10 batches: Adam's bias corrections at t ~ 1000, the cosine schedule in mid-flight
# (max_steps 2000) and the Polyak average meet the reference's to rounding.
RESUME = dict(s_dim=29, a_dim=8, hidden=64, batch=64, n_rows=1000, k_steps=1000, seed=43, beta=10.0, iql_tau=0.9,
              discount=0.99, tau=0.005, deterministic=False, dropout=None, max_steps=2000, reward_kind="sparse")


def resume_fixture(ref, fp32):
    full = run_trajectory(ref, fp32=fp32, ckpt_at=990, **RESUME)
    keep = {"regen_seed": np.asarray(RESUME["seed"]), "reward_kind": np.asarray(RESUME["reward_kind"])}
    for k, v in full.items():
        if k.startswith(("init/", "data/")):
            keep["check/" + k] = tensor_checks(v)
        elif k.startswith("step1/"):
            continue
        elif k == "indices":
            keep["check/indices"] = tensor_checks(v)
        elif k == "losses":
            keep["losses"] = v  # (all 1000: the free-running part of the comparison is the test's choice)
        else:
            keep[k] = v
    from tests.helpers import regen_indices
    data, nets = regen_inputs(keep)
    assert np.array_equal(regen_indices(keep), full["indices"])
    return keep


def long_regen(ref, name, fp32):
    cfg = LONG[name]
    full = run_trajectory(ref, fp32=fp32, snap_steps=(100, 1000), **cfg)
    keep = {"regen_seed": np.asarray(cfg["seed"]), "reward_kind": np.asarray(cfg["reward_kind"])}
    for k, v in full.items():
        if k.startswith(("init/", "data/")):
            keep["check/" + k] = tensor_checks(v)
        elif k.startswith(("step1/", "final/")):
            continue  # (at100 / at1000 carry the comparison points)
        elif k == "indices":
            keep["check/indices"] = tensor_checks(v)
        elif k.startswith("at") and v.size > 2048:
            keep[k + "#stride37"] = v.reshape(-1)[::37].copy()
            keep[k + "#sum"] = np.asarray(v.astype(np.float64).sum())
            keep[k + "#abssum"] = np.asarray(np.abs(v.astype(np.float64)).sum())
        else:
            keep[k] = v
    data, nets = regen_inputs(keep)
    for k, v in data.items():
        assert np.array_equal(v, full["data/" + k]), k
    for pre, net in zip(("qf", "vf", "actor"), nets):
        for k, v in net.items():
            assert np.array_equal(v, full[f"init/{pre}/{k}"]), (pre, k)
    from tests.helpers import regen_indices
    assert np.array_equal(regen_indices(keep), full["indices"])
    return keep


# --------------------------------------------------------------------------- #
# A16: the reference's OWN train() (ref:1393-1570) run as written.  What it imports from absent
# packages lands in stand-ins: d4rl.qlearning_dataset -> a seeded synthetic dataset, gym.make ->
# tests/fake_envs.py, wandb.init / wandb.log -> recorders, pyrallis.dump -> a marker file,
# torch.compile left in (inductor on the CPU) unless it fails here.  Recorded: the index stream its
# ReplayBuffer.sample drew, every wandb.log payload with its step, the checkpoint files it wrote
# (names, total_it, strided parameter summaries), the first actions of every evaluation.
# --------------------------------------------------------------------------- #
TRAIN_RUNS = {
    # config 2 hyper-parameters (configs/offline/iql/antmaze/medium_diverse_v2.yaml)
    "antmaze": dict(env="antmaze-medium-diverse-v2", normalize_reward=1, beta=10.0, iql_tau=0.9, seed=11,
                    n_rows=4000, data_seed=51, reward_kind="sparse01"),
    # config 1 hyper-parameters (configs/offline/iql/halfcheetah/medium_v2.yaml); normalize_reward=1 takes
    # the return-range branch of modify_reward for this environment name (ref:364-367)
    "cheetah": dict(env="halfcheetah-medium-v2", normalize_reward=1, beta=3.0, iql_tau=0.7, seed=12,
                    n_rows=3000, data_seed=52, reward_kind="normal"),
}
TRAIN_SCHEDULE = dict(max_timesteps=60, log_freq=20, eval_freq=30, n_episodes=7, batch_size=256, buffer_size=5000)


def train_dataset(spec):
    """The d4rl.qlearning_dataset stand-in of a TRAIN_RUNS entry (shared with the GPU test)."""
    S, A = fake_envs.DIMS[spec["env"]]
    rng = np.random.default_rng(spec["data_seed"])
    n = spec["n_rows"]
    d = {"observations": (rng.standard_normal((n, S)) * 2 + 0.5).astype(np.float32),
         "actions": rng.uniform(-1, 1, (n, A)).astype(np.float32),
         "next_observations": (rng.standard_normal((n, S)) * 2 + 0.5).astype(np.float32)}
    if spec["reward_kind"] == "sparse01":
        d["rewards"] = (rng.uniform(size=n) < 0.05).astype(np.float32)
    else:
        d["rewards"] = rng.standard_normal(n).astype(np.float32)
    d["terminals"] = rng.uniform(size=n) < 0.01
    return d


def train_run(ref, tag, fp32, use_compile=True):
    spec = TRAIN_RUNS[tag]
    out = {}
    raw = train_dataset(spec)
    logs, inits, idx_rec, trainers, first_actions = [], [], [], [], []
    sys.modules["d4rl"].qlearning_dataset = lambda env: {k: v.copy() for k, v in raw.items()}
    sys.modules["wandb"].init = lambda **kw: inits.append(kw)
    sys.modules["wandb"].log = lambda payload, step=None: logs.append((int(step), dict(payload)))
    sys.modules["pyrallis"].dump = lambda cfg, f: f.write("# written by the pyrallis stand-in\n")

    real_sample = ref.ReplayBuffer.sample
    def sample(self, batch_size):
        st = torch.random.get_rng_state()
        batch = real_sample(self, batch_size)
        after = torch.random.get_rng_state()
        torch.random.set_rng_state(st)
        idx = torch.randint(0, min(self._size, self._pointer), size=(batch_size,))  # ref:212-214 replayed
        assert torch.equal(torch.random.get_rng_state(), after) and torch.equal(self._states[idx], batch[0])
        idx_rec.append(idx.numpy().copy())
        return batch
    real_iql = ref.ImplicitQLearning
    def make_iql(**kw):
        trainers.append(real_iql(**kw))
        return trainers[-1]
    real_vec = sys.modules["gym"].vector.AsyncVectorEnv
    def spy_vec(fns):
        v = real_vec(fns)
        real_step, seen = v.step, []
        def step(a):
            if not seen:
                first_actions.append(np.asarray(a).copy())
                seen.append(1)
            return real_step(a)
        v.step = step
        return v
    real_compile, real_autocast = torch.compile, torch.amp.autocast
    ref.ReplayBuffer.sample, ref.ImplicitQLearning = sample, make_iql
    sys.modules["gym"].vector.AsyncVectorEnv = spy_vec
    torch.set_autocast_cache_enabled(False)  # (eager + autocast cache fails on this torch)
    if not use_compile:
        torch.compile = lambda m, *a, **k: m
    if fp32:
        torch.amp.autocast = lambda *a, **k: contextlib.nullcontext()
    try:
        with tempfile.TemporaryDirectory() as td:
            cfg = ref.TrainConfig(env=spec["env"], seed=spec["seed"], normalize_reward=spec["normalize_reward"],
                                  beta=spec["beta"], iql_tau=spec["iql_tau"], device="cpu", checkpoints_path=td,
                                  **TRAIN_SCHEDULE)
            out["run_name"] = np.asarray(cfg.name)
            ref.train(cfg)
            files = sorted(os.listdir(cfg.checkpoints_path))
            out["checkpoint_files"] = np.asarray(files)
            for f in files:
                if not f.endswith(".pt"):
                    continue
                ck = torch.load(os.path.join(cfg.checkpoints_path, f), map_location="cpu", weights_only=True)
                out[f"ckpt/{f}/total_it"] = np.asarray(int(ck["total_it"]))
                out[f"ckpt/{f}/keys"] = np.asarray(sorted(ck.keys()))
                out[f"ckpt/{f}/actor_keys"] = np.asarray(list(ck["actor"].keys()))
                out[f"ckpt/{f}/last_epoch"] = np.asarray(int(ck["actor_lr_schedule"]["last_epoch"]))
                for net in ("qf", "vf", "actor"):
                    for k, v in ck[net].items():
                        a = v.numpy()
                        key = f"ckpt/{f}/{net}/{k.removeprefix('_orig_mod.')}"
                        if a.size > 2048:
                            out[key + "#stride37"] = a.reshape(-1)[::37].copy()
                            out[key + "#sum"] = np.asarray(a.astype(np.float64).sum())
                        else:
                            out[key] = a.copy()
    finally:
        ref.ReplayBuffer.sample, ref.ImplicitQLearning = real_sample, real_iql
        sys.modules["gym"].vector.AsyncVectorEnv = real_vec
        torch.compile, torch.amp.autocast = real_compile, real_autocast
    tr = trainers[0]
    tgt = tr.q_target
    for k, v in tgt.state_dict().items():
        a = v.detach().numpy()
        key = "final/q_target/" + k.removeprefix("_orig_mod.")
        out[key + "#stride37" if a.size > 2048 else key] = a.reshape(-1)[::37].copy() if a.size > 2048 else a.copy()
    out["total_it"] = np.asarray(int(tr.total_it))
    out["indices"] = np.stack(idx_rec).astype(np.int32)
    out["log_steps"] = np.asarray([s for s, _ in logs])
    out["log_keys"] = np.asarray(["|".join(p.keys()) for _, p in logs])
    out["log_values"] = np.asarray([list(p.values()) + [np.nan] * (3 - len(p)) for _, p in logs], dtype=np.float64)
    out["wandb_init_keys"] = np.asarray(sorted(inits[0].keys()))
    out["wandb_config_keys"] = np.asarray(sorted(inits[0]["config"].keys()))
    out["first_actions"] = np.stack(first_actions)
    out["compiled"] = np.asarray(bool(use_compile))
    for k, v in raw.items():
        out["check/data/" + k] = tensor_checks(v)
    return out


def train_fixtures(ref):
    out = {}
    for tag in TRAIN_RUNS:
        for fp32 in (False, True):
            if tag != "antmaze" and fp32:
                continue
            try:
                r = train_run(ref, tag, fp32, use_compile=True)
            except Exception as e:  # inductor unavailable here: the eager modules (identical losses)
                print(f"torch.compile path failed ({type(e).__name__}: {str(e)[:200]}); eager")
                torch._dynamo.reset()
                r = train_run(ref, tag, fp32, use_compile=False)
            out.update(flat(f"{tag}_{'fp32' if fp32 else 'bf16'}", r))
    return out


def per_op(ref):
    """G1: single-op vectors in both autocast modes."""
    out = {}
    torch.manual_seed(3)
    rng = np.random.default_rng(3)
    for (s_dim, a_dim) in ((17, 6), (29, 8), (45, 24)):
        tag = f"S{s_dim}A{a_dim}"
        B, H = 32, 64
        q = ref.TwinQ(s_dim, a_dim, hidden_dim=H)
        v = ref.ValueFunction(s_dim, hidden_dim=H)
        g = ref.GaussianPolicy(s_dim, a_dim, 1.0, hidden_dim=H)
        d = ref.DeterministicPolicy(s_dim, a_dim, 1.0, hidden_dim=H)
        with torch.no_grad():
            g.log_std.copy_(torch.linspace(-0.7, 0.4, a_dim))
        s = torch.from_numpy(rng.standard_normal((B, s_dim)).astype(np.float32))
        a = torch.from_numpy(rng.uniform(-1, 1, (B, a_dim)).astype(np.float32))
        out.update(flat(f"{tag}/qf", params_of(q)))
        out.update(flat(f"{tag}/vf", params_of(v)))
        out.update(flat(f"{tag}/gauss", params_of(g)))
        out.update(flat(f"{tag}/det", params_of(d)))
        out[f"{tag}/s"] = s.numpy()
        out[f"{tag}/a"] = a.numpy()
        for mode in ("fp32", "bf16"):
            ctx = (contextlib.nullcontext() if mode == "fp32"
                   else torch.amp.autocast("cpu", dtype=torch.bfloat16))
            with torch.no_grad(), ctx:
                q1, q2 = q.both(s, a)
                out[f"{tag}/{mode}/q1"] = q1.float().numpy()
                out[f"{tag}/{mode}/q2"] = q2.float().numpy()
                out[f"{tag}/{mode}/qmin"] = q(s, a).float().numpy()
                out[f"{tag}/{mode}/v"] = v(s).float().numpy()
                dist = g(s)
                out[f"{tag}/{mode}/mean"] = dist.mean.float().numpy()
                out[f"{tag}/{mode}/std"] = dist.stddev.float().numpy()
                out[f"{tag}/{mode}/logp"] = dist.log_prob(a).sum(-1).float().numpy()
                out[f"{tag}/{mode}/det"] = d(s).float().numpy()
    u = torch.from_numpy(rng.standard_normal(257).astype(np.float32))
    out["asym/u"] = u.numpy()
    for tau in (0.7, 0.8, 0.9):
        out[f"asym/tau{tau}"] = np.asarray(ref.asymmetric_l2_loss(u, tau).item())
        with torch.amp.autocast("cpu", dtype=torch.bfloat16):
            ub = u.to(torch.bfloat16)
            out[f"asym/bf16/tau{tau}"] = np.asarray(
                ref.asymmetric_l2_loss(ub, tau).float().item())
    # soft_update
    src, tgt = nn.Linear(5, 3), nn.Linear(5, 3)
    out["soft/src_w"] = src.weight.detach().numpy().copy()
    out["soft/tgt_w"] = tgt.weight.detach().numpy().copy()
    ref.soft_update(tgt, src, 0.005)
    out["soft/out_w"] = tgt.weight.detach().numpy().copy()
    return out


class _FakeEnv:
    def __init__(self, max_steps):
        self._max_episode_steps = max_steps


def dataset_ops(ref):
    """G3/G4/G5: buffer gather, reward post-processing, CVaR, relabel plumbing."""
    out = {}
    rng = np.random.default_rng(11)

    # ---- G3 gather + torch CPU randint stream -------------------------------
    d = synth_dataset(rng, 300, 5, 3)
    buf = ref.ReplayBuffer(5, 3, 400, "cpu")
    buf.load_d4rl_dataset(d)
    torch.manual_seed(123)
    smp = buf.sample(16)
    for k, v in zip(("s", "a", "r", "s2", "d"), smp):
        out[f"g3/sample/{k}"] = v.numpy()
    for k in d:
        out[f"g3/data/{k}"] = d[k].astype(np.float32)
    out["g3/size_pointer"] = np.asarray([buf._size, buf._pointer])

    # ---- G4 reward range / modify_reward / mean-std ------------------------
    n = 64
    rew = rng.standard_normal(n).astype(np.float32)
    term = np.zeros(n, dtype=bool)
    term[[9, 30, 31, 50]] = True
    base = {"rewards": rew, "terminals": term}
    out["g4/rewards"] = rew
    out["g4/terminals"] = term
    mn, mx, tl = ref.return_reward_range({"rewards": rew.copy(), "terminals": term}, 12)
    out["g4/range"] = np.asarray([mn, mx])
    out["g4/trj_lens"] = tl
    for nr in range(1, 9):
        ds = {"rewards": rew.copy(), "terminals": term}
        ref.modify_reward(ds, "antmaze-medium-diverse-v2", nr, max_episode_steps=12)
        out[f"g4/antmaze_nr{nr}"] = ds["rewards"]
    ds = {"rewards": rew.copy(), "terminals": term}
    ref.modify_reward(ds, "halfcheetah-medium-v2", 1, max_episode_steps=12)
    out["g4/halfcheetah"] = ds["rewards"]
    ds = {"rewards": rew.copy(), "terminals": term}
    ref.modify_reward(ds, "pen-human-v1", 1, max_episode_steps=12)
    out["g4/pen_untouched"] = ds["rewards"]
    st = rng.standard_normal((40, 6)).astype(np.float32) * 3 + 1
    m, s = ref.compute_mean_std(st, 1e-3)
    out["g4/states"] = st
    out["g4/mean"], out["g4/std"] = m, s
    out["g4/normalized"] = ref.normalize_states(st, m, s)

    # ---- CVaR --------------------------------------------------------------
    preds = rng.standard_normal((10, 33)).astype(np.float32)
    out["cvar/preds"] = preds
    for alpha in (0.0, 0.5, 0.9, 0.95):
        n_tail = max(1, int(np.floor((1.0 - alpha) * preds.shape[0])))
        kth = min(n_tail, preds.shape[0] - 1)
        part = np.partition(preds, kth, axis=0)
        out[f"cvar/vec_alpha{alpha}"] = part[:n_tail].mean(axis=0).astype(np.float32)
        out[f"cvar/emp_alpha{alpha}"] = np.asarray(
            [ref.empirical_cvar(preds[:, i], alpha) for i in range(preds.shape[1])])
        out[f"cvar/stab_alpha{alpha}"] = np.asarray(
            ref.cvar_stability_check(preds, alpha, n_checks=20))
    out["cvar/single"] = np.asarray(ref.empirical_cvar(preds[:1, 0], 0.9))

    # ---- G5 relabel plumbing on a 3-episode dataset ------------------------
    s_dim, a_dim, N = 4, 2, 40
    ds = {
        "observations": rng.standard_normal((N, s_dim)).astype(np.float32),
        "actions": rng.uniform(-1, 1, (N, a_dim)).astype(np.float32),
        "rewards": np.zeros(N, dtype=np.float32),
        "terminals": np.zeros(N, dtype=bool),
        "timeouts": np.zeros(N, dtype=bool),
    }
    ds["terminals"][13] = True
    ds["timeouts"][27] = True
    for k, v in ds.items():
        out[f"g5/ds/{k}"] = v.astype(np.float32) if v.dtype == bool else v
    env = _FakeEnv(15)

    torch.manual_seed(5)
    rm = StandInRewardMLP(s_dim + a_dim, 1, [8, 8], "relu")
    with torch.no_grad():
        for p in rm.parameters():
            p.copy_(torch.randn_like(p) * 0.5)
    out.update(flat("g5/rm", params_of(rm)))
    for use_to in (True, False):
        dsu = dict(ds)
        if not use_to:
            dsu.pop("timeouts")
        for toe in (False, True):
            r = ref.qlearning_dataset_mr(env, rm, dataset=dsu, terminate_on_end=toe)
            tag = f"g5/mr/timeouts{int(use_to)}_toe{int(toe)}"
            for k, v in r.items():
                out[f"{tag}/{k}"] = np.asarray(v, dtype=np.float32)

    # PT: record exactly what the reference hands to r_model, chunk by chunk
    class Recorder(nn.Module):
        def __init__(self):
            super().__init__()
            self.p = nn.Parameter(torch.zeros(1))
            self.calls = []

        def forward(self, st, ac, ts, am):
            self.calls.append((st.numpy().copy(), ac.numpy().copy(),
                               ts.numpy().copy(), am.numpy().copy()))
            w = torch.arange(1, st.shape[1] + 1, dtype=torch.float32)
            val = ((st.sum(-1) + 2.0 * ac.sum(-1) + 0.01 * ts.float()) * am * w)
            val = val.cumsum(1)
            return {"value": val[:, None, :, None]}, None

    for ql in (5, 20):
        rec = Recorder()
        r = ref.qlearning_dataset_pt(env, rec, query_length=ql, dataset=ds)
        tag = f"g5/pt/ql{ql}"
        for k, v in r.items():
            out[f"{tag}/{k}"] = np.asarray(v, dtype=np.float32)
        out[f"{tag}/win_states"] = np.concatenate([c[0] for c in rec.calls])
        out[f"{tag}/win_actions"] = np.concatenate([c[1] for c in rec.calls])
        out[f"{tag}/win_timesteps"] = np.concatenate([c[2] for c in rec.calls])
        out[f"{tag}/win_mask"] = np.concatenate([c[3] for c in rec.calls])

    # MR snapshot ensemble + BNN: files we write ourselves, read by the reference
    with tempfile.TemporaryDirectory() as td:
        with open(os.path.join(td, "config.yaml"), "w") as f:
            f.write("activations: relu\n")
        snaps = []
        for ep in range(6):
            torch.manual_seed(100 + ep)
            m = StandInRewardMLP(s_dim + a_dim, 1, [8, 8], "relu")
            with torch.no_grad():
                for p in m.parameters():
                    p.copy_(torch.randn_like(p) * 0.5)
            sd = m.state_dict()
            if ep % 2:
                sd = {"_orig_mod." + k: v for k, v in sd.items()}
            torch.save({"net": sd}, os.path.join(td, f"checkpoint_{ep}.pt"))
            snaps.append(params_of(m))
        torch.save({"net": snaps and m.state_dict()}, os.path.join(td, "best_model.pt"))
        for i, sn in enumerate(snaps):
            out.update(flat(f"g5/ens/snap{i}", sn))
        for alpha, burn in ((0.0, 0), (0.5, 1), (0.9, 2)):
            r = ref.qlearning_dataset_mr_ensemble(env, td, alpha=alpha, burn_in=burn,
                                                  device="cpu", dataset=ds)
            tag = f"g5/ens/alpha{alpha}_burn{burn}"
            for k, v in r.items():
                out[f"{tag}/{k}"] = np.asarray(v, dtype=np.float32)
        out["g5/ens/discovered_burn2"] = np.asarray(
            [os.path.basename(p) for p in ref._discover_mr_snapshots(td, 2)])

        # BNN posterior layout: sampling_f/chain_*/sampled_weights/sampled_weights_0000000
        all_w = []
        for c in range(2):
            cdir = os.path.join(td, "sampling_f", f"chain_{c}", "sampled_weights")
            os.makedirs(cdir)
            ws = []
            for k in range(4):
                ws.append([rng.standard_normal(sh).astype(np.float32) * 0.5
                           for sh in ((s_dim + a_dim, 8), (8,), (8, 8), (8,), (8, 1), (1,))])
            torch.save({"sampled_weights": ws},
                       os.path.join(cdir, "sampled_weights_0000000"))
            all_w.extend(ws)
        for i, w in enumerate(all_w):
            for j, arr in enumerate(w):
                out[f"g5/bnn/w{i}/{j}"] = arr
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for alpha, ns in ((0.5, 500), (0.75, 5), (0.0, 0)):
                r = ref.qlearning_dataset_bnn(env, td, alpha=alpha, n_samples=ns,
                                              device="cpu", dataset=ds)
                tag = f"g5/bnn/alpha{alpha}_n{ns}"
                for k, v in r.items():
                    out[f"{tag}/{k}"] = np.asarray(v, dtype=np.float32)
    return out


def checkpoint_compat(ref, ckpt_path):
    """Reference-side read of a checkpoint written by OUR trainer (tools/make_checkpoint.py on the
    GPU box; the file is our own, loaded tensors-only): ["actor"] goes into the reference's
    GaussianPolicy with strict=False exactly as evaluation/d4rl/iql_eval_median.py:252-262 does;
    recorded: what load_state_dict reports and the policy's eval-mode output on fixed observations."""
    ck = torch.load(ckpt_path, map_location="cpu", weights_only=True)
    sd = ck["actor"]
    s_dim = sd["net.net.0.weight"].shape[1]
    hidden = sd["net.net.0.weight"].shape[0]
    a_dim = sd["log_std"].shape[0]
    actor = ref.GaussianPolicy(s_dim, a_dim, 1.0, hidden_dim=hidden, dropout=0.1)
    res = actor.load_state_dict(sd, strict=False)
    actor.eval()
    obs = np.random.default_rng(11).standard_normal((33, s_dim)).astype(np.float32)
    with torch.no_grad():
        dist = actor(torch.from_numpy(obs))
        act = actor.act(obs[0], "cpu")
    return {"obs": obs, "mean": dist.mean.numpy(), "std": dist.stddev.numpy()[0], "act0": act,
            "missing": np.asarray(list(res.missing_keys), dtype="U"),
            "unexpected": np.asarray(list(res.unexpected_keys), dtype="U"),
            "total_it": np.asarray(int(ck["total_it"])),
            "keys": np.asarray(sorted(ck.keys()), dtype="U")}


def import_custom_reference(ref_root):
    """algorithms/custom_offline/iql.py ("cref") with inert stubs for what is absent here
    (gymnasium, minari, orbax, flax, jax, pyrallis, wandb, the iqlpref.reward_models loaders): its
    torch / numpy parts -- ReplayBuffer, networks, ImplicitQLearning, soft_update, modify_reward,
    qlearning_dataset, evaluate -- run as written."""
    def stub(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    stub("gymnasium", Env=object,
         wrappers=types.SimpleNamespace(TransformObservation=fake_envs.TransformObservation,
                                        TransformReward=fake_envs.TransformReward))
    stub("minari", MinariDataset=object)
    stub("orbax")
    stub("orbax.checkpoint")
    sys.modules["orbax"].checkpoint = sys.modules["orbax.checkpoint"]
    stub("pyrallis", wrap=lambda *a, **k: (lambda f: f))
    stub("wandb")
    stub("flax", nnx=types.SimpleNamespace())
    stub("jax")
    stub("iqlpref")
    stub("iqlpref.reward_models")
    stub("iqlpref.reward_models.pref_transformer", load_PT=None)
    stub("iqlpref.reward_models.q_mlp", load_QMLP=None)
    path = os.path.join(ref_root, "algorithms", "custom_offline", "iql.py")
    spec = importlib.util.spec_from_file_location("ref_custom_iql", path)
    mod = importlib.util.module_from_spec(spec)
    sys.dont_write_bytecode = True
    keep_path = list(sys.path)
    spec.loader.exec_module(mod)
    sys.path[:] = keep_path  # (the file prepends "../" to sys.path on import)
    return mod


def custom_offline_fixtures(cref):
    """f4: the torch / numpy half of the custom_offline flavour, run by the reference itself."""
    out = {}
    # ---- K-step trajectory: fp32 (no autocast), convex Polyak, numpy-global-RNG sampler ----
    S, A, H, B, N, K, seed = 29, 8, 64, 64, 1000, 10, 5
    data = synth_dataset(np.random.default_rng(seed), N, S, A, "sparse")
    buf = cref.ReplayBuffer(S, A, N + 7, "cpu")
    buf.load_dataset(data)
    torch.manual_seed(seed)
    q, v = cref.TwinQ(S, A, hidden_dim=H), cref.ValueFunction(S, hidden_dim=H)
    actor = cref.GaussianPolicy(S, A, 1.0, hidden_dim=H)
    with torch.no_grad():
        actor.log_std.copy_(torch.linspace(-0.5, 0.3, A))
    vo = torch.optim.Adam(v.parameters(), lr=3e-4)
    qo = torch.optim.Adam(q.parameters(), lr=3e-4)
    ao = torch.optim.Adam(actor.parameters(), lr=3e-4)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(ao, 1000)
    trainer = cref.ImplicitQLearning(max_action=1.0, actor=actor, actor_optimizer=ao, actor_lr_scheduler=sched,
                                     q_network=q, q_optimizer=qo, v_network=v, v_optimizer=vo, iql_tau=0.9,
                                     beta=10.0, gamma=0.99, tau=0.005, device="cpu")
    out.update(flat("traj/init/qf", params_of(q)))
    out.update(flat("traj/init/vf", params_of(v)))
    out.update(flat("traj/init/actor", params_of(actor)))
    for k in ("observations", "actions", "rewards", "next_observations"):
        out[f"traj/data/{k}"] = data[k]
    out["traj/data/terminals"] = data["terminals"].astype(np.float32)
    idx_all, losses, lrs = np.zeros((K, B), np.int64), np.zeros((K, 3)), np.zeros(K)
    np.random.seed(seed)
    for t in range(K):
        st = np.random.get_state()
        batch = buf.sample(B)                       # the reference's own sampler (cref:277-284)
        after = np.random.get_state()
        np.random.set_state(st)
        idx = np.random.randint(0, N, size=B)       # ... replayed to learn which rows it drew
        assert all(np.array_equal(x, y) for x, y in zip(np.random.get_state()[1:3], after[1:3]))
        assert np.array_equal(batch[0].numpy(), data["observations"][idx])
        idx_all[t] = idx
        lrs[t] = ao.param_groups[0]["lr"]
        log = trainer.train(batch)
        losses[t] = [log["value_loss"], log["q_loss"], log["actor_loss"]]
    out["traj/hyper"] = np.asarray([S, A, H, B, N, K, 10.0, 0.9, 0.99, 0.005, 0.0, -1.0, 1000], dtype=np.float64)
    out["traj/np_seed"] = np.asarray(seed)
    out["traj/indices"], out["traj/losses"], out["traj/actor_lr"] = idx_all, losses, lrs
    out["traj/final_actor_lr"] = np.asarray(ao.param_groups[0]["lr"])
    out.update(flat("traj/final/qf", params_of(q)))
    out.update(flat("traj/final/vf", params_of(v)))
    out.update(flat("traj/final/actor", params_of(actor)))
    out.update(flat("traj/final/q_target", params_of(trainer.q_target)))
    out.update(flat("traj/final/q_adam", adam_state(qo, q)))
    sd = trainer.state_dict()
    out["traj/state_dict_keys"] = np.asarray(sorted(sd.keys()))
    out["traj/scheduler_last_epoch"] = np.asarray(sd["actor_lr_scheduler"]["last_epoch"])

    # ---- soft_update (cref:85-87): (1 - tau) t + tau s ----
    src, tgt = nn.Linear(5, 3), nn.Linear(5, 3)
    out["soft/src_w"], out["soft/tgt_w"] = src.weight.detach().numpy().copy(), tgt.weight.detach().numpy().copy()
    cref.soft_update(tgt, src, 0.005)
    out["soft/out_w"] = tgt.weight.detach().numpy().copy()

    # ---- modify_reward / return_reward_range (cref:127-155) ----
    rng = np.random.default_rng(12)
    rew = rng.standard_normal(64).astype(np.float32)
    term = np.zeros(64, dtype=bool)
    term[[9, 30, 31, 50]] = True
    out["mr/rewards"], out["mr/terminals"] = rew, term
    out["mr/range"] = np.asarray(cref.return_reward_range({"rewards": rew.copy(), "terminals": term}, 12))
    for name in ("halfcheetah-medium-v2", "hopper-medium-v2", "antmaze-medium-diverse-v2", "D4RL/pen/human-v2"):
        ds = {"rewards": rew.copy(), "terminals": term}
        cref.modify_reward(ds, name, max_episode_steps=12)
        out[f"mr/{name.replace('/', '_')}"] = ds["rewards"]

    # ---- qlearning_dataset (cref:158-225) with stand-in reward models: what the loop hands the
    # model call by call, and the dataset it builds from the answers ----
    lengths = (3, 8, 9, 25, 1)
    out["qd/lengths"] = np.asarray(lengths)
    for ql in (8, 1):
        eps = fake_envs.make_episodes(31, 6, 2, lengths)
        calls = []
        if ql > 1:
            def r_model(sts, acts, ts, am, training=False):
                calls.append((np.asarray(sts).copy(), np.asarray(acts).copy(), np.asarray(ts).copy(),
                              np.asarray(am).copy()))
                return {"value": fake_envs.fake_pt_values(sts, acts, ts, am)[..., None]}, None
        else:
            r_model = fake_envs.fake_markov_reward
        ds = cref.qlearning_dataset(eps, r_model, ql)
        for k, val in ds.items():
            out[f"qd/ql{ql}/{k}"] = np.asarray(val)
        if ql > 1:
            out[f"qd/ql{ql}/call_len"] = np.asarray([c[0].shape[1] for c in calls])
            out[f"qd/ql{ql}/call_t0"] = np.asarray([int(c[2][0, 0]) for c in calls])
            out[f"qd/ql{ql}/call_ts_last"] = np.asarray([int(c[2][0, -1]) for c in calls])
            out[f"qd/ql{ql}/call_first_state"] = np.stack([c[0][0, 0] for c in calls])
            out[f"qd/ql{ql}/call_last_action"] = np.stack([c[1][0, -1] for c in calls])
            out[f"qd/ql{ql}/call_mask_sum"] = np.asarray([float(c[3].sum()) for c in calls])

    # ---- evaluate (cref:559-579): sequential episodes, gymnasium API ----
    S, A = fake_envs.DIMS["pen-human-v1"]
    torch.manual_seed(9)
    pol = cref.GaussianPolicy(S, A, 0.8, hidden_dim=64, dropout=0.1)
    with torch.no_grad():
        pol.log_std.copy_(torch.linspace(-0.4, 0.2, A))
    mean, std = np.linspace(-0.2, 0.2, S), np.linspace(0.8, 1.3, S)
    env = cref.wrap_env(_SpacedEnv(fake_envs.FakeGymnasiumEnv("pen-human-v1")), state_mean=mean, state_std=std)
    seen = []
    real_step = env.step

    def step(a):
        seen.append(np.asarray(a).copy())
        return real_step(a)
    env.step = step
    scores = cref.evaluate(env, pol, num_episodes=6, seed=40, device="cpu")
    assert pol.training  # cref:578
    out.update(flat("ev/actor", params_of(pol)))
    out["ev/mean"], out["ev/std"] = mean, std
    out["ev/scores"], out["ev/actions"] = scores, np.stack(seen)
    return out


class _SpacedEnv:
    """cref:118 passes env.observation_space to TransformObservation: give the stand-in one."""

    def __init__(self, env):
        self.env, self.observation_space = env, None

    def __getattr__(self, name):
        return getattr(self.env, name)


def eval_fixtures(ref):
    """f1: the reference's eval_actor (ref:265-341) driving the deterministic stand-in vector
    environment (its gym.make / wrap_env / AsyncVectorEnv calls land in tests/fake_envs.py)."""
    out = {}
    for tag, name, det in (("antmaze", "antmaze-medium-diverse-v2", False),
                           ("cheetah", "halfcheetah-medium-v2", True)):
        S, A = fake_envs.DIMS[name]
        torch.manual_seed(4)
        cls = ref.DeterministicPolicy if det else ref.GaussianPolicy
        actor = cls(S, A, 0.7, hidden_dim=64, dropout=0.1)
        mean, std = np.linspace(-0.2, 0.2, S), np.linspace(0.8, 1.3, S)
        made = []
        real = sys.modules["gym"].vector.AsyncVectorEnv

        def spy(fns):
            made.append(real(fns))
            return made[-1]
        sys.modules["gym"].vector.AsyncVectorEnv = spy
        try:
            scores, steps = ref.eval_actor(name, actor, 0.7, mean, std, "cpu", n_episodes=17, seed=100, n_envs=5)
        finally:
            sys.modules["gym"].vector.AsyncVectorEnv = real
        assert actor.training and made[0].closed
        out.update(flat(f"{tag}/actor", params_of(actor)))
        out[f"{tag}/mean"], out[f"{tag}/std"] = mean, std
        out[f"{tag}/scores"], out[f"{tag}/steps_to_goal"] = scores, np.asarray(steps, dtype=np.int64)
        out[f"{tag}/actions"] = np.stack(made[0].actions_seen)
        out[f"{tag}/args"] = np.asarray([0.7, 17, 100, 5, float(det)])
    return out


def save(name, d):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **d)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.0f} KiB, {len(d)} arrays")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--checkpoint-compat", default=None,
                    help="only write checkpoint_compat.npz from this checkpoint of our trainer")
    ap.add_argument("--only", default=None,
                    help="comma-separated subset: small, h256, big, per_op, dataset_ops, eval, custom, train, long, resume, shapes")
    args = ap.parse_args()
    torch.set_num_threads(1)  # fixed summation order for the captured vectors
    ref = import_reference(args.ref)
    if args.checkpoint_compat:
        save("checkpoint_compat.npz", checkpoint_compat(ref, args.checkpoint_compat))
        return
    only = set(args.only.split(",")) if args.only else None
    want = lambda what: only is None or what in only
    if want("eval"):
        save("eval_actor.npz", eval_fixtures(ref))
    if want("custom"):
        save("custom_offline.npz", custom_offline_fixtures(import_custom_reference(args.ref)))
    if want("train"):
        save("train_runs.npz", train_fixtures(ref))
    if want("resume"):
        for fp32 in (True, False):
            save(f"traj_resume_{'fp32' if fp32 else 'bf16'}.npz", resume_fixture(ref, fp32))
    if want("long"):
        for name in LONG:
            for fp32 in (True, False):
                save(f"{name}_{'fp32' if fp32 else 'bf16'}.npz", long_regen(ref, name, fp32))
    if want("shapes"):
        for name, cfg in SHAPES.items():
            for fp32 in (True, False):
                save(f"{name}_{'fp32' if fp32 else 'bf16'}.npz", run_trajectory(ref, fp32=fp32, **cfg))
    if want("big"):
        for name in BIG:
            for fp32 in (True, False):
                save(f"{name}_{'fp32' if fp32 else 'bf16'}.npz", big_regen(ref, name, fp32))
    if only is not None and not (only & {"small", "h256", "per_op", "dataset_ops"}):
        return

    cfgs = {
        # antmaze hyper-parameters (configs/offline/iql/antmaze/medium_diverse_v2.yaml)
        "traj_antmaze": dict(s_dim=29, a_dim=8, hidden=64, batch=64, n_rows=1000,
                             k_steps=10, seed=0, beta=10.0, iql_tau=0.9, discount=0.99,
                             tau=0.005, deterministic=False, dropout=None,
                             max_steps=1000, reward_kind="sparse"),
        # pen: Gaussian policy with actor_dropout=0.1 (configs/offline/iql/pen/human_v1.yaml)
        "traj_pen_dropout": dict(s_dim=45, a_dim=24, hidden=64, batch=32, n_rows=500,
                                 k_steps=6, seed=1, beta=3.0, iql_tau=0.8,
                                 discount=0.99, tau=0.005, deterministic=False,
                                 dropout=0.1, max_steps=50, reward_kind="normal"),
        # halfcheetah shapes with the deterministic policy branch (iql.py:626-629)
        "traj_cheetah_det": dict(s_dim=17, a_dim=6, hidden=64, batch=48, n_rows=700,
                                 k_steps=6, seed=2, beta=3.0, iql_tau=0.7,
                                 discount=0.99, tau=0.005, deterministic=True,
                                 dropout=None, max_steps=100, reward_kind="normal"),
    }
    for name, cfg in cfgs.items():
        for fp32 in (True, False):
            save(f"{name}_{'fp32' if fp32 else 'bf16'}.npz",
                 run_trajectory(ref, fp32=fp32, **cfg))
    for fp32 in (True, False):
        keep, common = big_summary(ref, fp32)
        save(f"traj_antmaze_h256_{'fp32' if fp32 else 'bf16'}.npz", keep)
    save("traj_antmaze_h256_common.npz", common)
    save("per_op.npz", per_op(ref))
    save("dataset_ops.npz", dataset_ops(ref))


if __name__ == "__main__":
    main()
