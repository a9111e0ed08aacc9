"""Host logic and oracle pinned against what the REFERENCE itself produced for the rows that used
to be restatement-only: the custom_offline flavour's torch / numpy half
(tests/golden/custom_offline.npz) and eval_actor (tests/golden/eval_actor.npz), both recorded by
tests/golden/make_fixtures.py from the reference's own functions driving tests/fake_envs.py.
CPU only; the HIP side of the same fixtures is in tests/test_gpu_reference_runs.py."""
import numpy as np
import pytest
import torch

from oracle import iql_oracle as orc
from oracle import relabel_oracle as ro
from tests import fake_envs, helpers


@pytest.fixture(scope="module")
def gc():
    return np.load(helpers.GOLDEN + "/custom_offline.npz")


@pytest.fixture(scope="module")
def ge():
    return np.load(helpers.GOLDEN + "/eval_actor.npz")


def custom_traj(gc):
    sub = {k[len("traj/"):]: gc[k] for k in gc.files if k.startswith("traj/")}
    h = sub["hyper"]
    hyper = dict(s_dim=int(h[0]), a_dim=int(h[1]), hidden=int(h[2]), batch=int(h[3]), n_rows=int(h[4]),
                 k_steps=int(h[5]), beta=float(h[6]), iql_tau=float(h[7]), discount=float(h[8]), tau=float(h[9]),
                 deterministic=False, dropout=None, max_steps=int(h[12]))
    data = {k.split("/")[1]: v for k, v in sub.items() if k.startswith("data/")}
    nets = tuple({k.split("/", 2)[2]: v for k, v in sub.items() if k.startswith(f"init/{n}/")}
                 for n in ("qf", "vf", "actor"))
    return sub, hyper, data, nets


def test_oracle_convex_polyak_trajectory_matches_the_custom_reference(gc):
    """cref:438-556 run by the reference (fp32, (1 - tau) t + tau s) vs oracle(polyak_convex=True)."""
    d, hyper, data, nets = custom_traj(gc)
    o = orc.IQLOracle(*nets, iql_tau=hyper["iql_tau"], beta=hyper["beta"], max_steps=hyper["max_steps"],
                      discount=hyper["discount"], tau=hyper["tau"], mode="fp32", polyak_convex=True)
    for t in range(hyper["k_steps"]):
        assert abs(o.lr["actor"] - d["actor_lr"][t]) <= 1e-12 * d["actor_lr"][t]
        out = o.train(orc.gather_batch(data, d["indices"][t]))
        np.testing.assert_allclose([out["value_loss"], out["q_loss"], out["actor_loss"]], d["losses"][t], rtol=1e-5)
    for net, pd in (("qf", o.qf), ("vf", o.vf), ("actor", o.actor), ("q_target", o.q_target)):
        for k, v in pd.items():
            np.testing.assert_allclose(v, d[f"final/{net}/{k}"], atol=1e-6, rtol=0, err_msg=f"{net}/{k}")
    # the lerp form of offline/iql.py rounds differently: it must NOT give this target
    o2 = orc.IQLOracle(*nets, iql_tau=hyper["iql_tau"], beta=hyper["beta"], max_steps=hyper["max_steps"],
                       discount=hyper["discount"], tau=hyper["tau"], mode="fp32", polyak_convex=False)
    for t in range(hyper["k_steps"]):
        o2.train(orc.gather_batch(data, d["indices"][t]))
    k0 = "q1.net.2.weight"
    err_c = np.abs(o.q_target[k0] - d[f"final/q_target/{k0}"]).max()
    assert err_c < 1e-6 and not np.array_equal(o2.q_target[k0], o.q_target[k0])
    tgt = {"w": gc["soft/tgt_w"].copy()}
    orc.soft_update(tgt, {"w": gc["soft/src_w"]}, 0.005, convex=True)
    np.testing.assert_array_equal(tgt["w"], gc["soft/out_w"])


def test_custom_sampler_draws_the_reference_index_stream(gc):
    """cref:277-284: after np.random.seed(s) the reference's ReplayBuffer.sample drew these rows."""
    from iqlpref_amd import custom_offline as co
    d, hyper, data, nets = custom_traj(gc)
    buf = co.ReplayBuffer.__new__(co.ReplayBuffer)  # the index stream is host logic: no device needed
    buf._size = buf._pointer = hyper["n_rows"]
    np.random.seed(int(d["np_seed"]))
    np.testing.assert_array_equal(buf.draw_indices(hyper["batch"], hyper["k_steps"]), d["indices"])
    np.random.seed(int(d["np_seed"]))
    np.testing.assert_array_equal(buf.draw_indices(hyper["batch"]), d["indices"][0])


def test_custom_modify_reward_and_range_match_the_reference(gc):
    from iqlpref_amd import custom_offline as co
    rew, term = gc["mr/rewards"], gc["mr/terminals"]
    lo, hi = co.return_reward_range({"rewards": rew.copy(), "terminals": term}, 12)
    np.testing.assert_allclose([lo, hi], gc["mr/range"], rtol=1e-12)
    for name in ("halfcheetah-medium-v2", "hopper-medium-v2", "antmaze-medium-diverse-v2", "D4RL/pen/human-v2"):
        ds = {"rewards": rew.copy(), "terminals": term}
        co.modify_reward(ds, name, max_episode_steps=12)
        np.testing.assert_array_equal(ds["rewards"], gc[f"mr/{name.replace('/', '_')}"], err_msg=name)
    assert np.array_equal(gc["mr/D4RL_pen_human-v2"], rew)  # untouched


def test_custom_relabel_windows_match_the_reference_loop(gc):
    """cref:158-225 recorded call by call with a stand-in model.  Pins (i) the closed-form
    (start, len, t0) windows the product hands to ONE kernel launch, (ii) the oracle's restated
    loop, (iii) the Markovian (query_length 1) path of the product."""
    from iqlpref_amd import custom_offline as co
    lengths = tuple(int(x) for x in gc["qd/lengths"])
    eps = fake_envs.make_episodes(31, 6, 2, lengths)
    as_dict = [{"observations": e.observations, "actions": e.actions, "terminations": e.terminations} for e in eps]
    QL = 8
    obs = np.concatenate([e.observations[:-1] for e in eps])
    act = np.concatenate([e.actions for e in eps])
    start, length, t0 = co.episode_windows(lengths, QL)
    assert len(start) == sum(lengths)
    # (i) every step's reward = the last-position value of ITS window with ITS true timesteps
    got = np.zeros(len(start))
    for i, (s0, ln, tt) in enumerate(zip(start, length, t0)):
        sl = slice(int(s0), int(s0) + int(ln))
        got[i] = fake_envs.fake_pt_values(obs[sl][None], act[sl][None], (tt + np.arange(ln))[None],
                                          np.ones((1, ln)))[0, -1]
    np.testing.assert_allclose(got, gc["qd/ql8/rewards"], rtol=1e-12, atol=1e-12)
    # ... and the calls the reference made: one per episode for its first QL steps, then one per step
    ep_start = np.concatenate([[0], np.cumsum(lengths)[:-1]])
    want_len, want_t0, want_first = [], [], []
    for e0, L in zip(ep_start, lengths):
        want_len.append(min(L, QL)), want_t0.append(0), want_first.append(obs[e0])
        for i in range(QL, L):
            k = e0 + i
            assert length[k] == QL and t0[k] == i + 1 - QL and start[k] == e0 + i + 1 - QL
            want_len.append(QL), want_t0.append(int(t0[k])), want_first.append(obs[int(start[k])])
    np.testing.assert_array_equal(want_len, gc["qd/ql8/call_len"])
    np.testing.assert_array_equal(want_t0, gc["qd/ql8/call_t0"])
    np.testing.assert_array_equal(np.stack(want_first), gc["qd/ql8/call_first_state"])
    np.testing.assert_array_equal(gc["qd/ql8/call_ts_last"] - gc["qd/ql8/call_t0"] + 1, gc["qd/ql8/call_len"])
    # (ii) the oracle's loop with the same stand-in model
    want = ro.custom_qlearning_dataset(as_dict, None, QL, value_fn=fake_envs.fake_pt_values)
    for k in ("observations", "actions", "next_observations", "terminals"):
        np.testing.assert_array_equal(want[k], gc[f"qd/ql8/{k}"].astype(want[k].dtype), err_msg=k)
    np.testing.assert_allclose(want["rewards"], gc["qd/ql8/rewards"], rtol=1e-6)
    # (iii) query_length 1: the product's own function, a stand-in Markovian model
    ds = co.qlearning_dataset(as_dict, lambda o, a: torch.from_numpy(fake_envs.fake_markov_reward(o, a)), 1)
    for k in ("observations", "actions", "next_observations", "rewards", "terminals"):
        np.testing.assert_array_equal(ds[k], gc[f"qd/ql1/{k}"].astype(ds[k].dtype), err_msg=k)


def replay_eval(ge, tag, name):
    """Fresh stand-in environments stepped with the actions the reference's eval_actor chose."""
    import iqlpref_amd as ia
    _, n_eps, seed, n_envs, _ = ge[f"{tag}/args"]
    n_envs, n_eps, seed = int(n_envs), int(n_eps), int(seed)

    def make(i):
        def thunk():
            e = ia.wrap_env(fake_envs.FakeGymEnv(name), state_mean=ge[f"{tag}/mean"], state_std=ge[f"{tag}/std"])
            e.seed(seed + i)
            return e
        return thunk
    env = fake_envs.SyncVectorEnv([make(i) for i in range(n_envs)])
    env.reset()
    return env, n_envs, n_eps


@pytest.mark.parametrize("tag,name", [("antmaze", "antmaze-medium-diverse-v2"), ("cheetah", "halfcheetah-medium-v2")])
def test_episode_ledger_and_oracle_accounting_match_the_reference_eval(ge, tag, name):
    """ref:296-333 as the reference ran it: same actions -> same episodes -> its scores, its
    steps-to-goal, no step beyond the last one it took."""
    import iqlpref_amd as ia
    env, n_envs, n_eps = replay_eval(ge, tag, name)
    ledger = ia.EpisodeLedger(n_envs, n_eps, goal_env="antmaze" in name)
    steps = []
    for a in ge[f"{tag}/actions"]:
        assert not ledger.full
        _, rew, done, _ = env.step(a)
        ledger.record(rew, done)
        steps.append((rew, done))
    assert ledger.full
    np.testing.assert_array_equal(np.asarray(ledger.scores[:n_eps]), ge[f"{tag}/scores"])
    np.testing.assert_array_equal(np.asarray(ledger.steps_to_goal, dtype=np.int64), ge[f"{tag}/steps_to_goal"])
    assert (len(ge[f"{tag}/steps_to_goal"]) > 0) == ("antmaze" in name)
    scores, stg, used = ro.eval_accounting(steps, n_envs, n_eps, "antmaze" in name)
    np.testing.assert_array_equal(scores, ge[f"{tag}/scores"])
    assert stg == list(ge[f"{tag}/steps_to_goal"]) and used == len(steps)


def test_our_constructors_give_the_reference_initial_weights():
    """torch.manual_seed(s) + TwinQ / ValueFunction / GaussianPolicy built in the reference's order
    = the reference's initial parameters (checksums recorded from the reference's own modules)."""
    for name in helpers.TRAJ_BIG:
        d, hyper, data, nets = helpers.load_traj(name, "fp32")  # raises on any mismatch
        assert hyper["hidden"] == 256 and data["observations"].shape == (hyper["n_rows"], hyper["s_dim"])
        assert set(nets[2]) >= {"log_std", "net.net.0.weight"}
