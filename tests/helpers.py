"""Shared helpers for the parity tests (oracle side + golden-file access)."""
import os

import numpy as np

from oracle import iql_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

TRAJ = ["traj_antmaze", "traj_cheetah_det", "traj_pen_dropout", "traj_antmaze_h256"]
# BASELINE configs 1 / 3 and config 5's batch at H = 256 (initial parameters and data rebuilt from the seed)
TRAJ_BIG = ["traj_cheetah_h256", "traj_pen_h256", "traj_antmaze_b1024"]
# depths / widths away from the default (make_fixtures.SHAPES): the general layer-wise step
TRAJ_SHAPES = ["traj_deep3_w96", "traj_shallow1_w40_drop", "traj_deep4_w72_det"]


def synth_dataset(rng, n, s_dim, a_dim, reward="normal"):
    """Seeded synthetic transitions (shared with tests/golden/make_fixtures.py, which feeds the
    same arrays to the reference)."""
    obs = rng.standard_normal((n, s_dim)).astype(np.float32)
    nxt = rng.standard_normal((n, s_dim)).astype(np.float32)
    act = rng.uniform(-1, 1, (n, a_dim)).astype(np.float32)
    if reward == "normal":
        rew = rng.standard_normal(n).astype(np.float32)
    else:  # antmaze-like sparse, then -1 (normalize_reward=1)
        rew = (rng.uniform(size=n) < 0.05).astype(np.float32) - 1.0
    term = (rng.uniform(size=n) < 0.02)
    return {
        "observations": obs,
        "actions": act,
        "rewards": rew,
        "next_observations": nxt,
        "terminals": term,
    }


def tensor_checks(v):
    """[sum, sum of magnitudes, sum of index-weighted values] in float64: identifies a tensor."""
    x = np.asarray(v, dtype=np.float64).reshape(-1)
    return np.asarray([x.sum(), np.abs(x).sum(), (x * (np.arange(x.size) % 997 + 1)).sum()])


def regen_inputs(d):
    """Initial parameters and data of a seed-regenerated golden trajectory: the dataset from
    numpy's default_rng(seed), the networks from torch.manual_seed(seed) and OUR module
    constructors in the reference's order (q, v, actor; make_fixtures.run_trajectory) -- the same
    torch calls in the same order give the reference's initial weights, which is itself part of the
    drop-in contract.  Verified against the checksums the fixture holds of the reference's own
    arrays: a mismatch raises, nothing is compared on other inputs."""
    import torch
    import iqlpref_amd as ia
    h = d["hyper"]
    S, A, H, n = int(h[0]), int(h[1]), int(h[2]), int(h[4])
    det, drop = bool(h[10]), (None if h[11] < 0 else float(h[11]))
    seed = int(d["regen_seed"])
    data = synth_dataset(np.random.default_rng(seed), n, S, A, str(d["reward_kind"]))
    data["terminals"] = data["terminals"].astype(np.float32)
    state = torch.random.get_rng_state()
    try:
        torch.manual_seed(seed)
        q = ia.TwinQ(S, A, hidden_dim=H)
        v = ia.ValueFunction(S, hidden_dim=H)
        actor = (ia.DeterministicPolicy if det else ia.GaussianPolicy)(S, A, 1.0, hidden_dim=H, dropout=drop)
    finally:
        torch.random.set_rng_state(state)
    if not det:
        with torch.no_grad():
            actor.log_std.copy_(torch.linspace(-0.5, 0.3, A))
    nets = tuple({k: t.detach().numpy().copy() for k, t in m.state_dict().items()} for m in (q, v, actor))
    for k, arr in data.items():
        if not np.array_equal(tensor_checks(arr), d[f"check/data/{k}"]):
            raise AssertionError(f"regenerated data/{k} is not what the reference was given")
    for pre, net in zip(("qf", "vf", "actor"), nets):
        for k, arr in net.items():
            if not np.array_equal(tensor_checks(arr), d[f"check/init/{pre}/{k}"]):
                raise AssertionError(f"regenerated init/{pre}/{k} differs from the reference's initial weights")
    return data, nets


def regen_indices(d):
    """The index stream of a seed-regenerated trajectory: make_fixtures.run_trajectory draws
    torch.randint(0, n_rows, (batch,)) from torch.Generator().manual_seed(seed + 1) per step."""
    import torch
    h = d["hyper"]
    B, n, K = int(h[3]), int(h[4]), int(h[5])
    g = torch.Generator().manual_seed(int(d["regen_seed"]) + 1)
    idx = np.stack([torch.randint(0, n, (B,), generator=g).numpy() for _ in range(K)])
    if "check/indices" in getattr(d, "files", d) and not np.array_equal(tensor_checks(idx), d["check/indices"]):
        raise AssertionError("regenerated index stream is not the one the reference run drew")
    return idx


def load_traj(name, mode):
    d = np.load(os.path.join(GOLDEN, f"{name}_{mode}.npz"))
    common = d
    if "h256" in name and "regen_seed" not in d.files:
        common = np.load(os.path.join(GOLDEN, f"{name}_common.npz"))
    h = d["hyper"]
    hyper = dict(
        s_dim=int(h[0]), a_dim=int(h[1]), hidden=int(h[2]), batch=int(h[3]),
        n_rows=int(h[4]), k_steps=int(h[5]), beta=float(h[6]), iql_tau=float(h[7]),
        discount=float(h[8]), tau=float(h[9]), deterministic=bool(h[10]),
        dropout=None if h[11] < 0 else float(h[11]), max_steps=int(h[12]),
        n_hidden=int(h[13]) if len(h) > 13 else 2)
    if "regen_seed" in d.files:
        data, nets = regen_inputs(d)
        return d, hyper, data, nets
    data = {k.split("/")[1]: common[k] for k in common.files if k.startswith("data/")}
    qf, vf, actor = orc.split_init(common)
    return d, hyper, data, (qf, vf, actor)


def keep_masks(d, hyper, t):
    if hyper["dropout"] is None:
        return None
    m = np.unpackbits(d["dropout_keep"][t], axis=-1)[..., :hyper["hidden"]].astype(bool)
    return [m[l] for l in range(m.shape[0])]


def make_oracle(hyper, nets, mode):
    qf, vf, actor = nets
    return orc.IQLOracle(qf, vf, actor, iql_tau=hyper["iql_tau"], beta=hyper["beta"],
                         max_steps=hyper["max_steps"], discount=hyper["discount"],
                         tau=hyper["tau"], mode=mode, dropout=hyper["dropout"])


def golden_param(d, key, arr):
    """Compare ``arr`` against golden entry ``key`` (full or strided summary)."""
    if key in d.files:
        return d[key], arr
    if key + "#stride37" in d.files:
        return d[key + "#stride37"], np.asarray(arr).reshape(-1)[::37]
    return None, None


def resume_state(d):
    """The reference's checkpoint after `ckpt/total_it` steps out of a traj_resume_* fixture:
    (qf, vf, actor, q_target) parameter dicts, {group: {param name: (exp_avg, exp_avg_sq)}} with the
    parameter names in the reference's optimizer order, the step count."""
    pick = lambda pre: {k[len(pre):]: d[k] for k in d.files if k.startswith(pre)}
    nets = (pick("ckpt/qf/"), pick("ckpt/vf/"), pick("ckpt/actor/"))
    target = pick("ckpt/q_target/")
    moments = {}
    for group, opt, net in (("q", "q_optimizer", nets[0]), ("v", "v_optimizer", nets[1]),
                            ("actor", "actor_optimizer", nets[2])):
        names = list(net.keys())  # state_dict order = named_parameters order = the optimizer's index order
        if group == "actor" and "log_std" in names:  # (registered first in GaussianPolicy: index 0 either way)
            assert names[0] == "log_std" or names[-1] == "log_std"
        assert int(d[f"ckpt/{opt}/n_params"]) == len(names)
        moments[group] = {n: (d[f"ckpt/{opt}/{i}/exp_avg"], d[f"ckpt/{opt}/{i}/exp_avg_sq"])
                          for i, n in enumerate(names)}
        for i, n in enumerate(names):
            assert d[f"ckpt/{opt}/{i}/exp_avg"].shape == net[n].shape, (opt, i, n)
            assert float(d[f"ckpt/{opt}/{i}/step"]) == float(d["ckpt/total_it"])
    return nets, target, moments, int(d["ckpt/total_it"])
