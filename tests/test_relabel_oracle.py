"""oracle/relabel_oracle.py pinned against tests/golden/dataset_ops.npz (captured
from the reference's own functions).  CPU only."""
import numpy as np
import pytest

from oracle import relabel_oracle as ro
from tests import helpers


@pytest.fixture(scope="module")
def g():
    return np.load(helpers.GOLDEN + "/dataset_ops.npz")


def g5_dataset(g, use_timeouts=True):
    ds = {k.split("/")[-1]: g[k] for k in g.files if k.startswith("g5/ds/")}
    ds["terminals"] = ds["terminals"].astype(bool)
    ds["timeouts"] = ds["timeouts"].astype(bool)
    if not use_timeouts:
        ds.pop("timeouts")
    return ds


def mlp_weights(g, prefix):
    """[W0,b0,W1,b1,...,Wout,bout] from the stand-in state-dict names."""
    keys = [k[len(prefix):] for k in g.files if k.startswith(prefix)]
    hidden = sorted({k.split(".")[1] for k in keys if k.startswith("layers.")},
                    key=lambda s: 0 if s == "0" else int(s.split("_")[1]))
    out = []
    for h in hidden:
        out += [g[f"{prefix}layers.{h}.W"], g[f"{prefix}layers.{h}.b"]]
    out += [g[prefix + "out.W"], g[prefix + "out.b"]]
    return out


def test_reward_range_and_modify_reward(g):
    rew, term = g["g4/rewards"], g["g4/terminals"]
    mn, mx, tl = ro.return_reward_range(rew, term, 12)
    np.testing.assert_allclose([mn, mx], g["g4/range"], rtol=1e-12)
    np.testing.assert_array_equal(tl, g["g4/trj_lens"])
    for nr in range(1, 9):
        np.testing.assert_allclose(ro.modify_reward(rew, term, "antmaze-medium-diverse-v2", nr, 12),
                                   g[f"g4/antmaze_nr{nr}"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(ro.modify_reward(rew, term, "halfcheetah-medium-v2", 1, 12),
                               g["g4/halfcheetah"], rtol=1e-6)
    np.testing.assert_array_equal(ro.modify_reward(rew, term, "pen-human-v1", 1, 12), g["g4/pen_untouched"])


def test_cvar(g):
    preds = g["cvar/preds"]
    for alpha in (0.0, 0.5, 0.9, 0.95):
        np.testing.assert_allclose(ro.cvar_tail_mean(preds, alpha), g[f"cvar/vec_alpha{alpha}"], rtol=1e-6)
        emp = [ro.empirical_cvar(preds[:, i], alpha) for i in range(preds.shape[1])]
        np.testing.assert_allclose(emp, g[f"cvar/emp_alpha{alpha}"], rtol=1e-6)
        np.testing.assert_allclose(ro.cvar_stability_check(preds, alpha, 20), g[f"cvar/stab_alpha{alpha}"],
                                   rtol=1e-6)
    assert ro.empirical_cvar(preds[:1, 0], 0.9) == pytest.approx(float(g["cvar/single"]))
    with pytest.raises(ValueError):
        ro.empirical_cvar(preds[:, 0], 1.0)


@pytest.mark.parametrize("use_to", [True, False])
@pytest.mark.parametrize("toe", [False, True])
def test_mr_relabel(g, use_to, toe):
    ds = g5_dataset(g, use_to)
    out = ro.qlearning_dataset_mr(ds, mlp_weights(g, "g5/rm/"), 15, terminate_on_end=toe)
    tag = f"g5/mr/timeouts{int(use_to)}_toe{int(toe)}"
    for k, v in out.items():
        want = g[f"{tag}/{k}"]
        assert v.shape == want.shape, k
        np.testing.assert_allclose(np.asarray(v, dtype=np.float32), want, rtol=2e-6, atol=2e-6)


@pytest.mark.parametrize("ql", [5, 20])
def test_pt_windows_bug_compatible(g, ql):
    ds = g5_dataset(g)
    keep, ep_steps = ro.keep_mask_and_steps(ds["terminals"], ds["timeouts"], 15)
    sts, acts, ts, am = ro.pt_windows(ds["observations"], ds["actions"], ep_steps, ql)
    tag = f"g5/pt/ql{ql}"
    np.testing.assert_array_equal(sts, g[f"{tag}/win_states"])
    np.testing.assert_array_equal(acts, g[f"{tag}/win_actions"])
    np.testing.assert_array_equal(ts, g[f"{tag}/win_timesteps"])
    np.testing.assert_array_equal(am, g[f"{tag}/win_mask"])
    # the kept rows of the relabelled dataset
    np.testing.assert_array_equal(ds["observations"][:-1][keep], g[f"{tag}/observations"])
    np.testing.assert_array_equal(ds["terminals"][:-1][keep].astype(np.float32), g[f"{tag}/terminals"])
    # rewards of the recording fake model (see tests/golden/make_fixtures.py Recorder)
    w = np.arange(1, ql + 1, dtype=np.float32)
    val = ((sts.sum(-1) + 2.0 * acts.sum(-1) + 0.01 * ts.astype(np.float32)) * am * w).cumsum(1)[:, -1]
    np.testing.assert_allclose(val[keep], g[f"{tag}/rewards"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("alpha,burn", [(0.0, 0), (0.5, 1), (0.9, 2)])
def test_mr_ensemble(g, alpha, burn):
    ds = g5_dataset(g)
    sets = [mlp_weights(g, f"g5/ens/snap{i}/") for i in range(burn, 6)]
    out, _ = ro.qlearning_dataset_ensemble(ds, sets, alpha, 15)
    tag = f"g5/ens/alpha{alpha}_burn{burn}"
    for k, v in out.items():
        np.testing.assert_allclose(np.asarray(v, dtype=np.float32), g[f"{tag}/{k}"], rtol=2e-6, atol=2e-6)


@pytest.mark.parametrize("alpha,ns", [(0.5, 500), (0.75, 5), (0.0, 0)])
def test_bnn(g, alpha, ns):
    ds = g5_dataset(g)
    all_w = [[g[f"g5/bnn/w{i}/{j}"] for j in range(6)] for i in range(8)]
    if 0 < ns < len(all_w):  # ref:929-932 subsample with default_rng(0)
        idx = np.random.default_rng(seed=0).choice(len(all_w), size=ns, replace=False)
        all_w = [all_w[i] for i in sorted(idx)]
    out, _ = ro.qlearning_dataset_ensemble(ds, all_w, alpha, 15)
    tag = f"g5/bnn/alpha{alpha}_n{ns}"
    for k, v in out.items():
        np.testing.assert_allclose(np.asarray(v, dtype=np.float32), g[f"{tag}/{k}"], rtol=2e-6, atol=2e-6)


def test_pt_forward_self_consistency():
    """Parity unpinned (no runnable reference): structural checks of the restatement."""
    rng = np.random.default_rng(0)
    S, A, QL = 5, 3, 6
    p = ro.make_pt_params(rng, S, A, 20, embd=16, pref=8, inter=32, layers=2)
    st = rng.standard_normal((4, QL, S)).astype(np.float32)
    ac = rng.standard_normal((4, QL, A)).astype(np.float32)
    ts = np.tile(np.arange(QL), (4, 1))
    am = np.ones((4, QL), np.float32)
    v = ro.pt_value_last(p, st, ac, ts, am, num_heads=4)
    assert v.shape == (4,) and np.isfinite(v).all()
    # left padding with mask 0 == the shorter unpadded window with the same timesteps
    st2 = np.zeros((4, QL + 3, S), np.float32); st2[:, 3:] = st
    ac2 = np.zeros((4, QL + 3, A), np.float32); ac2[:, 3:] = ac
    ts2 = np.zeros((4, QL + 3), np.int64); ts2[:, 3:] = ts
    am2 = np.zeros((4, QL + 3), np.float32); am2[:, 3:] = 1
    v2 = ro.pt_value_last(p, st2, ac2, ts2, am2, num_heads=4)
    np.testing.assert_allclose(v2, v, rtol=1e-5, atol=1e-5)
