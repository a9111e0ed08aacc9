"""Shared helpers for the parity tests (oracle side + golden-file access)."""
import os

import numpy as np

from oracle import iql_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

TRAJ = ["traj_antmaze", "traj_cheetah_det", "traj_pen_dropout", "traj_antmaze_h256"]


def load_traj(name, mode):
    d = np.load(os.path.join(GOLDEN, f"{name}_{mode}.npz"))
    common = d
    if "h256" in name:
        common = np.load(os.path.join(GOLDEN, f"{name}_common.npz"))
    h = d["hyper"]
    hyper = dict(
        s_dim=int(h[0]), a_dim=int(h[1]), hidden=int(h[2]), batch=int(h[3]),
        n_rows=int(h[4]), k_steps=int(h[5]), beta=float(h[6]), iql_tau=float(h[7]),
        discount=float(h[8]), tau=float(h[9]), deterministic=bool(h[10]),
        dropout=None if h[11] < 0 else float(h[11]), max_steps=int(h[12]))
    data = {k.split("/")[1]: common[k] for k in common.files if k.startswith("data/")}
    qf, vf, actor = orc.split_init(common)
    return d, hyper, data, (qf, vf, actor)


def keep_masks(d, hyper, t):
    if hyper["dropout"] is None:
        return None
    m = np.unpackbits(d["dropout_keep"][t], axis=-1)[..., :hyper["hidden"]].astype(bool)
    return [m[0], m[1]]


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
