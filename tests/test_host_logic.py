"""CPU tests of the host-side logic of iqlpref_amd (no GPU, no compute calls):
vectorised dataset preparation vs the oracle loops and the reference goldens,
config loading, the C-ABI export list."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from oracle import relabel_oracle as ro
from tests import helpers


@pytest.fixture(scope="module")
def g():
    return np.load(helpers.GOLDEN + "/dataset_ops.npz")


def test_library_exports_every_declared_symbol():
    import __graft_entry__
    __graft_entry__.build()
    from iqlpref_amd import _lib
    lib = _lib.load()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "iqlhip.h")).read()
    declared = set(re.findall(r"\b(iqlhip_[a-z0-9_]+)\s*\(", header))
    declared -= {"iqlhip_status"}
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"libiqlhip.so does not export {name}"
        assert name in _lib.SYMBOLS, f"_lib.SYMBOLS lacks {name}"
    assert lib.iqlhip_abi_version() == _lib.ABI_VERSION
    assert lib.iqlhip_replay_row_stride(29, 8) == 72 and lib.iqlhip_replay_next_offset(29, 8) == 40
    assert lib.iqlhip_replay_row_stride(17, 6) == 48  # [17 + 6 + 2 -> 28] + 17 -> 48


def test_arena_layout_and_step_cost_match_survey():
    from iqlpref_amd import _lib
    lib = _lib.load()
    c = _lib.TrainerConfig()
    c.state_dim, c.action_dim, c.hidden_dim, c.batch_size = 29, 8, 256, 256
    c.precision, c.cosine_t_max, c.dropout_p = 1, 10, -1.0
    offs = (ctypes.c_int64 * _lib.N_TENSORS)()
    n_p, n_t = ctypes.c_int64(), ctypes.c_int64()
    assert lib.iqlhip_arena_layout(ctypes.byref(c), ctypes.byref(offs), ctypes.byref(n_p), ctypes.byref(n_t)) == 0
    # arena sizes include the 128-byte alignment padding of each tensor; SURVEY 8 parameter counts:
    true_t, true_p = 151_554, 151_554 + 73_729 + 75_536
    assert true_t <= n_t.value <= true_t + 12 * 31 and true_p <= n_p.value <= true_p + 25 * 31
    sizes = []
    for in_dim, out in ((37, 1), (37, 1), (29, 1), (29, 8)):
        sizes += [256 * in_dim, 256, 256 * 256, 256, out * 256, out]
    sizes.append(8)
    for k in range(25):
        assert offs[k] % 32 == 0
        if k:
            assert offs[k] >= offs[k - 1] + sizes[k - 1]
    assert n_p.value == offs[24] + 8
    b, f = ctypes.c_double(), ctypes.c_double()
    assert lib.iqlhip_step_cost(ctypes.byref(c), ctypes.byref(b), ctypes.byref(f)) == 0
    assert b.value == 10_908_272
    # any depth 1..6 and width 1..1024 (ref:417-449 MLP): 2 (n_hidden + 1) tensors per network, then log_std
    c.hidden_dim, c.n_hidden = 100, 3
    assert lib.iqlhip_arena_layout(ctypes.byref(c), ctypes.byref(offs), ctypes.byref(n_p), ctypes.byref(n_t)) == 0
    sizes = []
    for in_dim, out in ((37, 1), (37, 1), (29, 1), (29, 8)):
        sizes += [100 * in_dim, 100, 100 * 100, 100, 100 * 100, 100, out * 100, out]
    sizes.append(8)
    for k in range(33):
        assert offs[k] % 32 == 0 and (k == 0 or offs[k] >= offs[k - 1] + sizes[k - 1])
    assert n_p.value == offs[32] + 8 and offs[33] == -1 and n_t.value <= offs[16]
    assert lib.iqlhip_step_cost(ctypes.byref(c), ctypes.byref(b), ctypes.byref(f)) == 0
    assert b.value == 4 * 256 * (2 * 29 + 8 + 2) + 32 * sum(sizes) + 8 * sum(sizes[:16])
    for hd, nh, word in ((2000, 2, b"hidden_dim"), (256, 7, b"n_hidden")):
        c.hidden_dim, c.n_hidden = hd, nh
        assert lib.iqlhip_arena_layout(ctypes.byref(c), None, None, None) == _lib.ERR_UNSUPPORTED
        assert word in lib.iqlhip_last_error()


@pytest.mark.parametrize("seed", range(6))
def test_keep_mask_vectorised_matches_loop(seed):
    import iqlpref_amd as ia
    rng = np.random.default_rng(seed)
    n = int(rng.integers(2, 400))
    term = rng.uniform(size=n) < 0.03
    tout = rng.uniform(size=n) < 0.05
    for toe in (False, True):
        for use_to, M in ((True, 50), (False, int(rng.integers(1, 40)))):
            want = ro.keep_mask_and_steps(term, tout if use_to else None, M, toe)
            got = ia.keep_mask_and_steps(term, tout if use_to else None, M, toe)
            np.testing.assert_array_equal(got[0], want[0], err_msg=f"keep n={n} M={M} toe={toe} to={use_to}")
            np.testing.assert_array_equal(got[1], want[1], err_msg=f"steps n={n} M={M} toe={toe} to={use_to}")


def test_reward_range_and_modify_reward_match_reference(g):
    import iqlpref_amd as ia
    rew, term = g["g4/rewards"], g["g4/terminals"]
    mn, mx, tl = ia.return_reward_range({"rewards": rew.copy(), "terminals": term}, 12)
    np.testing.assert_allclose([mn, mx], g["g4/range"], rtol=1e-12)
    np.testing.assert_array_equal(tl, g["g4/trj_lens"])
    for nr in range(1, 9):
        ds = {"rewards": rew.copy(), "terminals": term}
        ia.modify_reward(ds, "antmaze-medium-diverse-v2", nr, max_episode_steps=12)
        np.testing.assert_allclose(ds["rewards"], g[f"g4/antmaze_nr{nr}"], rtol=1e-6, atol=1e-7)
    ds = {"rewards": rew.copy(), "terminals": term}
    ia.modify_reward(ds, "halfcheetah-medium-v2", 1, max_episode_steps=12)
    np.testing.assert_allclose(ds["rewards"], g["g4/halfcheetah"], rtol=1e-6)
    ds = {"rewards": rew.copy(), "terminals": term}
    ia.modify_reward(ds, "pen-human-v1", 1, max_episode_steps=12)
    np.testing.assert_array_equal(ds["rewards"], g["g4/pen_untouched"])
    # random cases against the oracle loop
    rng = np.random.default_rng(3)
    for _ in range(5):
        n = int(rng.integers(30, 300))
        r = rng.standard_normal(n).astype(np.float32)
        t = rng.uniform(size=n) < 0.04
        M = int(rng.integers(3, 25))
        a = ia.return_reward_range({"rewards": r, "terminals": t}, M)
        b = ro.return_reward_range(r, t, M)
        np.testing.assert_allclose(a[:2], b[:2], rtol=1e-12)
        np.testing.assert_array_equal(a[2], b[2])


def test_mean_std_and_cvar_helpers(g):
    import iqlpref_amd as ia
    m, s = ia.compute_mean_std(g["g4/states"], 1e-3)
    np.testing.assert_array_equal(m, g["g4/mean"])
    np.testing.assert_array_equal(s, g["g4/std"])
    np.testing.assert_array_equal(ia.normalize_states(g["g4/states"], m, s), g["g4/normalized"])
    preds = g["cvar/preds"]
    for alpha in (0.0, 0.5, 0.9, 0.95):
        emp = [ia.empirical_cvar(preds[:, i], alpha) for i in range(preds.shape[1])]
        np.testing.assert_allclose(emp, g[f"cvar/emp_alpha{alpha}"], rtol=1e-6)
        np.testing.assert_allclose(ia.cvar_stability_check(preds, alpha, n_checks=20),
                                   g[f"cvar/stab_alpha{alpha}"], rtol=1e-6)
    with pytest.raises(ValueError):
        ia.empirical_cvar(preds[:, 0], -0.1)


def test_train_config_schema_and_yaml(tmp_path):
    import iqlpref_amd as ia
    c = ia.TrainConfig()
    assert c.name.startswith("IQL-halfcheetah-medium-expert-v2-") and len(c.name.split("-")[-1]) == 8
    c2 = ia.TrainConfig(checkpoints_path="/tmp/x", reward_model_root="/r/base", seed=3)
    assert c2.checkpoints_path == os.path.join("/tmp/x", c2.name)
    assert c2.reward_model_path == "/r/base_3"  # iql_eval.py:143-146
    # the reference's own YAML for the headline config (values as written there)
    y = tmp_path / "c.yaml"
    y.write_text("actor_lr: 3e-4\nbatch_size: 256\nbeta: 10.0\nbuffer_size: 10000000\n"
                 "checkpoints_path: null\ndevice: cuda\ndiscount: 0.99\nenv: antmaze-medium-diverse-v2\n"
                 "eval_freq: 5000\ngroup: g\niql_deterministic: false\niql_tau: 0.9\nload_model: ''\n"
                 "max_timesteps: 1000000\nn_episodes: 100\nname: IQL\nnormalize: true\n"
                 "normalize_reward: 1\nqf_lr: 3e-4\nproject: IQL-pref\nseed: 0\ntau: 0.005\nvf_lr: 3e-4\n")
    c3 = ia.load_config(str(y), seed=5)
    assert c3.actor_lr == 3e-4 and c3.beta == 10.0 and c3.normalize_reward == 1 and c3.seed == 5
    assert c3.buffer_size == 10_000_000 and c3.iql_deterministic is False and c3.checkpoints_path is None
    with pytest.raises(ValueError):
        ia.load_config(None, not_a_field=1)
    # pen YAML writes `normalize_reward: false` for an int field
    assert ia.load_config(None, normalize_reward=False).normalize_reward == 0


def test_no_cpu_path():
    import iqlpref_amd as ia
    with pytest.raises(RuntimeError, match="no CPU path"):
        ia.ReplayBuffer(3, 2, 10, "cpu")


def test_build_tag_ties_library_to_sources():
    """The library carries the hash of the sources it was built from; the binding refuses another."""
    from iqlpref_amd import _lib, build
    assert build.built_tag() == build.source_tag() == _lib.build_tag()
    assert len(_lib.build_tag()) == 16 and not build.needs_build()
    assert build.source_tag(["-DX"]) != build.source_tag()


def test_snapshot_discovery_matches_reference(g, tmp_path):
    import iqlpref_amd as ia
    from iqlpref_amd import relabel
    # the directory the reference was given when the golden was recorded (make_fixtures.py):
    # checkpoint_0..5.pt + best_model.pt, burn-in 2
    ref_dir = tmp_path / "as_recorded"
    ref_dir.mkdir()
    for name in [f"checkpoint_{e}.pt" for e in range(6)] + ["best_model.pt"]:
        (ref_dir / name).write_bytes(b"")
    got = [os.path.basename(p) for p in relabel._discover_mr_snapshots(str(ref_dir), 2)]
    assert got == [str(x) for x in g["g5/ens/discovered_burn2"]]
    # numeric (not lexicographic) epoch order, and only exact checkpoint_<digits>.pt names
    for name in ("checkpoint_0.pt", "checkpoint_1.pt", "checkpoint_2.pt", "checkpoint_3.pt", "checkpoint_10.pt",
                 "best_model.pt", "checkpoint_x.pt", "checkpoint_4.pt.bak", "xcheckpoint_5.pt"):
        (tmp_path / name).write_bytes(b"")
    found = relabel._discover_mr_snapshots(str(tmp_path), 2)
    assert [os.path.basename(p) for p in found] == ["checkpoint_2.pt", "checkpoint_3.pt", "checkpoint_10.pt"]
    with pytest.raises(ValueError, match="discarded all 5"):
        relabel._discover_mr_snapshots(str(tmp_path), 11)
    with pytest.raises(FileNotFoundError, match="No MR snapshots"):
        relabel._discover_mr_snapshots(str(tmp_path / "missing"))


def test_bnn_weight_file_with_numpy_arrays(tmp_path):
    """The reference's posterior files hold lists of numpy arrays (ref:905-915); they are read
    with the restricted unpickler plus an allow-list for ndarray reconstruction."""
    from iqlpref_amd import relabel
    rng = np.random.default_rng(0)
    ws = [[rng.standard_normal(s).astype(np.float32) for s in ((5, 8), (8,), (8, 1), (1,))] for _ in range(3)]
    f = tmp_path / "sampled_weights_0000000"
    torch.save({"sampled_weights": ws}, f)
    got = relabel.load_bnn_weight_file(str(f))["sampled_weights"]
    assert len(got) == 3 and all(isinstance(a, np.ndarray) for w in got for a in w)
    for w, g_ in zip(ws, got):
        for a, b in zip(w, g_):
            np.testing.assert_array_equal(a, b)

    class Evil:
        def __reduce__(self):
            return (os.system, ("true",))
    torch.save({"sampled_weights": [Evil()]}, f)
    with pytest.raises(Exception):  # anything outside the allow-list is refused, not executed
        relabel.load_bnn_weight_file(str(f))


@pytest.mark.parametrize("goal_env", [True, False])
def test_episode_ledger_matches_reference_loop(goal_env):
    """EpisodeLedger (iqlpref_amd/train.py) against the reference's per-environment loop
    (ref:296-333), restated in oracle/relabel_oracle.py."""
    from iqlpref_amd.train import EpisodeLedger
    rng = np.random.default_rng(3)
    for trial in range(20):
        n_envs, n_eps = int(rng.integers(1, 7)), int(rng.integers(1, 30))
        steps = [(rng.uniform(-0.2, 0.4, n_envs), rng.uniform(size=n_envs) < 0.3) for _ in range(400)]
        led = EpisodeLedger(n_envs, n_eps, goal_env)
        used = 0
        while not led.full:
            led.record(*steps[used])
            used += 1
        scores, goal, used_ref = ro.eval_accounting(steps, n_envs, n_eps, goal_env)
        assert used == used_ref
        np.testing.assert_array_equal(np.asarray(led.scores[:n_eps]), scores)
        assert led.steps_to_goal == goal


def test_custom_offline_window_closed_form():
    """iqlpref_amd.custom_offline.episode_windows against the rolling loop of
    algorithms/custom_offline/iql.py:172-211 (window start, length and first TRUE timestep of every step)."""
    from iqlpref_amd.custom_offline import episode_windows
    for lengths, QL in (((3, 8, 9, 25, 1), 8), ((100,), 100), ((101, 2), 100), ((5, 5), 10)):
        start, length, t0 = episode_windows(lengths, QL)
        want, base = [], 0
        for L in lengths:
            for i in range(L):
                if i < QL:          # one forward over the first min(L, QL) steps: position i sees 0..i
                    want.append((base, i + 1, 0))
                else:               # rolling window i+1-QL .. i with its true timesteps
                    want.append((base + i + 1 - QL, QL, i + 1 - QL))
            base += L
        got = list(zip(start.tolist(), length.tolist(), t0.tolist()))
        assert got == want


def test_reward_model_follows_the_ranks_seed(monkeypatch):
    """iql_eval.py:143-146 ties the reward model to the run's seed.  Rank 1 of a
    one-seed-per-GPU launch trains config.seed + 1 and must relabel with {root}_{seed + 1}, exactly as
    slot 1 of a two-seeds-per-GPU process does -- not with the path TrainConfig.__post_init__ derived
    from the base seed."""
    import torch.distributed as dist
    import iqlpref_amd as ia
    from iqlpref_amd import distributed as D
    from iqlpref_amd.train import seed_configs
    cfg = ia.TrainConfig(env="antmaze-medium-diverse-v2", seed=7, reward_model_root="/models/mr")
    assert cfg.reward_model_path == "/models/mr_7"
    monkeypatch.setattr(dist, "is_initialized", lambda: True)
    monkeypatch.setattr(dist, "get_rank", lambda: 1)
    for K in (1, 2):
        seeds = [D.rank_seed(cfg.seed, K) + k for k in range(K)]
        assert seeds == [7 + K + k for k in range(K)]
        per_seed = seed_configs(cfg, seeds)
        assert [c.reward_model_path for c in per_seed] == [f"/models/mr_{s}" for s in seeds]
        assert [c.seed for c in per_seed] == seeds and cfg.reward_model_path == "/models/mr_7"
    plain = ia.TrainConfig(env="antmaze-medium-diverse-v2", seed=7, reward_model_path="/models/one")
    assert all(c is plain for c in seed_configs(plain, [8, 9]))


def test_standalone_dropout_key_is_the_seed_the_module_was_built_under():
    """train(seeds_per_gpu=K) builds the K actors after K successive set_seed calls; each
    keeps ITS seed as the key of its stand-alone dropout masks (a module built later must not change it)."""
    import iqlpref_amd as ia
    ia.set_seed(11)
    a = ia.GaussianPolicy(5, 2, 1.0, hidden_dim=64, dropout=0.1)
    ia.set_seed(12)
    b = ia.GaussianPolicy(5, 2, 1.0, hidden_dim=64, dropout=0.1)
    assert a.net._drop_seed == 11 and b.net._drop_seed == 12
    import torch
    assert torch.initial_seed() == 12  # (what the old key, read at call time, would have given both)
