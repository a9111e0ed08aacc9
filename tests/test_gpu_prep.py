"""Device-side dataset preparation (csrc/prep.hip) against the reference goldens
(tests/golden/dataset_ops.npz) and the oracle loops.  -m gpu.

Bars: masks, counters and trajectory lengths exact; episode return range exact (double sums in
transition order); normalised rewards bit-identical to the reference's float32 results; state
statistics rtol 1e-6 (accumulated in double, numpy sums float32 rows in order); the fused
z-scoring bit-identical to numpy given the same mean / std."""
import os
import time

import numpy as np
import pytest
import torch

from oracle import relabel_oracle as ro
from tests import helpers

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def g():
    return np.load(helpers.GOLDEN + "/dataset_ops.npz")


def _diag(line):
    path = os.environ.get("IQL_TEST_DIAG")
    if path:
        with open(path, "a") as f:
            f.write(line + "\n")


@pytest.mark.parametrize("seed", range(8))
def test_keep_mask_and_steps_exact(seed):
    from iqlpref_amd import prep
    rng = np.random.default_rng(seed)
    # sizes around the scan's chunk (1024) and wave (64) boundaries, and tiny ones
    n = int([2, 3, 65, 1023, 1025, 4097, 20_000, 70_001][seed])
    term = rng.uniform(size=n) < (0.03 if seed % 2 else 0.002)
    tout = rng.uniform(size=n) < (0.05 if seed % 3 else 0.001)
    for toe in (False, True):
        for use_to, M in ((True, 50), (False, int(rng.integers(1, 40))), (False, 1), (False, 1000)):
            want = ro.keep_mask_and_steps(term, tout if use_to else None, M, toe)
            keep, steps = prep.keep_mask_and_steps(term, tout if use_to else None, M, toe, DEV)
            tag = f"n={n} M={M} toe={toe} timeouts={use_to}"
            assert keep.dtype == torch.bool and steps.dtype == torch.int64
            np.testing.assert_array_equal(keep.cpu().numpy(), want[0], err_msg="keep " + tag)
            np.testing.assert_array_equal(steps.cpu().numpy(), want[1], err_msg="steps " + tag)
    # degenerate inputs: no reset at all (look-back reaches the start), a reset at every transition
    z = np.zeros(5000, bool)
    for t_, o_ in ((z, z), (~z, z), (z, ~z)):
        for toe in (False, True):
            want = ro.keep_mask_and_steps(t_, o_, 100, toe)
            keep, steps = prep.keep_mask_and_steps(t_, o_, 100, toe, DEV)
            np.testing.assert_array_equal(keep.cpu().numpy(), want[0])
            np.testing.assert_array_equal(steps.cpu().numpy(), want[1])


def test_keep_mask_in_the_relabel_functions_matches_reference(g):
    """The relabel entry points take their keep mask from the device scan: the goldens of
    qlearning_dataset_mr (recorded from the reference) cover timeouts / no timeouts x
    terminate_on_end through tests/test_gpu_relabel.py; here the mask alone."""
    import iqlpref_amd as ia
    from tests.test_relabel_oracle import g5_dataset
    for use_to in (True, False):
        ds = g5_dataset(g, use_to)
        for toe in (False, True):
            got = ia.keep_mask_and_steps(ds["terminals"], ds.get("timeouts"), 15, toe, device=DEV)
            want = ia.keep_mask_and_steps(ds["terminals"], ds.get("timeouts"), 15, toe)
            np.testing.assert_array_equal(got[0], want[0])
            np.testing.assert_array_equal(got[1], want[1])


def test_reward_range_and_modify_reward_match_reference(g):
    from iqlpref_amd import prep
    rew, term = g["g4/rewards"], g["g4/terminals"]
    rt = torch.from_numpy(rew.copy()).to(DEV)
    mn, mx, tl = prep.return_reward_range(rt, term, 12)
    assert [mn, mx] == [float(x) for x in g["g4/range"]]  # double sums in order: exact
    np.testing.assert_array_equal(tl.cpu().numpy(), g["g4/trj_lens"])
    for nr in range(1, 9):
        r = torch.from_numpy(rew.copy()).to(DEV)
        prep.modify_reward(r, term, "antmaze-medium-diverse-v2", nr, max_episode_steps=12)
        np.testing.assert_array_equal(r.cpu().numpy(), g[f"g4/antmaze_nr{nr}"], err_msg=f"normalize_reward={nr}")
    r = torch.from_numpy(rew.copy()).to(DEV)
    prep.modify_reward(r, term, "halfcheetah-medium-v2", 1, max_episode_steps=12)
    np.testing.assert_array_equal(r.cpu().numpy(), g["g4/halfcheetah"])
    r = torch.from_numpy(rew.copy()).to(DEV)
    prep.modify_reward(r, term, "pen-human-v1", 1, max_episode_steps=12)
    np.testing.assert_array_equal(r.cpu().numpy(), g["g4/pen_untouched"])
    # random cases against the oracle loop, incl. a trailing partial episode and M = 1
    rng = np.random.default_rng(3)
    for _ in range(6):
        n = int(rng.integers(30, 5000))
        r = rng.standard_normal(n).astype(np.float32)
        t = rng.uniform(size=n) < 0.02
        M = int(rng.choice([1, 3, 25, 1000]))
        want = ro.return_reward_range(r, t, M)
        mn, mx, tl = prep.return_reward_range(torch.from_numpy(r).to(DEV), t, M)
        assert (mn, mx) == (want[0], want[1])
        np.testing.assert_array_equal(tl.cpu().numpy(), want[2])
    with pytest.raises(AssertionError):  # no complete episode (the reference fails on min([]))
        prep.return_reward_range(torch.zeros(5, device=DEV), np.zeros(5, bool), 10)


def test_state_stats_and_fused_normalisation(g):
    import iqlpref_amd as ia
    from iqlpref_amd import prep
    m, s = prep.state_stats(torch.from_numpy(g["g4/states"]).to(DEV), 1e-3)
    np.testing.assert_allclose(m.cpu().numpy(), g["g4/mean"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(s.cpu().numpy(), g["g4/std"], rtol=1e-6)
    rng = np.random.default_rng(0)
    for n, S, A in ((3000, 29, 8), (70_000, 45, 24), (5, 3, 2), (2048, 256, 1)):
        obs = (rng.standard_normal((n, S)) * rng.uniform(0.1, 30, S) + rng.uniform(-50, 50, S)).astype(np.float32)
        nxt = rng.standard_normal((n, S)).astype(np.float32)
        act = rng.uniform(-1, 1, (n, A)).astype(np.float32)
        rew, done = rng.standard_normal(n).astype(np.float32), (rng.uniform(size=n) < 0.1)
        m, s = prep.state_stats(torch.from_numpy(obs).to(DEV), 1e-3)
        m64, s64 = obs.astype(np.float64).mean(0), obs.astype(np.float64).std(0) + 1e-3
        np.testing.assert_allclose(m.cpu().numpy(), m64, rtol=1e-6, atol=1e-6 * np.abs(obs).max())
        np.testing.assert_allclose(s.cpu().numpy(), s64, rtol=1e-6)
        # fused z-scoring == numpy's normalize_states with the SAME statistics, bit for bit
        mh, sh = ia.compute_mean_std(obs, 1e-3)
        buf = ia.ReplayBuffer(S, A, n, DEV)
        buf.load_device_arrays(*[torch.from_numpy(np.ascontiguousarray(a)).to(DEV) for a in
                                 (obs, act, rew, nxt, done.astype(np.float32))],
                               torch.from_numpy(mh).to(DEV), torch.from_numpy(sh).to(DEV))
        np.testing.assert_array_equal(buf._states.cpu().numpy(), ia.normalize_states(obs, mh, sh))
        np.testing.assert_array_equal(buf._next_states.cpu().numpy(), ia.normalize_states(nxt, mh, sh))
        np.testing.assert_array_equal(buf._actions.cpu().numpy(), act)
        np.testing.assert_array_equal(buf._rewards.cpu().numpy()[:, 0], rew)
        np.testing.assert_array_equal(buf._dones.cpu().numpy()[:, 0], done.astype(np.float32))


@pytest.mark.parametrize("env,nr", [("antmaze-medium-diverse-v2", 1), ("antmaze-large-diverse-v2", 7),
                                    ("halfcheetah-medium-v2", 1), ("pen-human-v1", 0)])
def test_prepare_replay_matches_the_host_pipeline(env, nr):
    """ref:1435-1456 end to end: device pipeline (one upload) vs the reference's numpy sequence."""
    import iqlpref_amd as ia
    from iqlpref_amd import prep
    rng = np.random.default_rng(5)
    n, S, A, M = 40_000, 17, 6, 1000
    ds = {"observations": (rng.standard_normal((n, S)) * 3 + 1).astype(np.float32),
          "actions": rng.uniform(-1, 1, (n, A)).astype(np.float32),
          "rewards": (rng.uniform(size=n) < 0.01).astype(np.float32),
          "next_observations": (rng.standard_normal((n, S)) * 3 + 1).astype(np.float32),
          "terminals": rng.uniform(size=n) < 0.002}
    host = {k: v.copy() for k, v in ds.items()}
    if nr:
        ia.modify_reward(host, env, nr, max_episode_steps=M)
    mh, sh = ia.compute_mean_std(host["observations"], 1e-3)
    for stats in ("host", "device"):
        buf = ia.ReplayBuffer(S, A, n, DEV)
        before = {k: v.copy() for k, v in ds.items()}
        m, s = prep.prepare_replay(ds, buf, env_name=env, normalize_reward=nr, normalize=True, eps=1e-3,
                                   max_episode_steps=M, stats=stats)
        for k in ds:  # the caller's dataset is not rewritten
            np.testing.assert_array_equal(ds[k], before[k])
        np.testing.assert_array_equal(buf._rewards.cpu().numpy()[:, 0], host["rewards"])
        np.testing.assert_array_equal(buf._dones.cpu().numpy()[:, 0], host["terminals"].astype(np.float32))
        if stats == "host":
            np.testing.assert_array_equal(m, mh)
            np.testing.assert_array_equal(buf._states.cpu().numpy(), ia.normalize_states(host["observations"], mh, sh))
        else:
            np.testing.assert_allclose(m, mh, rtol=2e-5, atol=2e-5)  # numpy's own float32 row sums drift
            np.testing.assert_allclose(s, sh, rtol=2e-5)
            np.testing.assert_allclose(buf._states.cpu().numpy(), ia.normalize_states(host["observations"], mh, sh),
                                       rtol=1e-4, atol=1e-4)


def test_start_up_time_at_one_million_transitions():
    """The Python / numpy preparation of a 1M-transition antmaze dataset against the device
    pipeline (figures go to the diagnostics file; the assertion is only that the device path is
    not slower)."""
    import iqlpref_amd as ia
    from iqlpref_amd import prep
    rng = np.random.default_rng(1)
    n, S, A = 1_000_000, 29, 8
    ds = {"observations": rng.standard_normal((n, S), dtype=np.float32),
          "actions": rng.uniform(-1, 1, (n, A)).astype(np.float32),
          "rewards": (rng.uniform(size=n) < 0.01).astype(np.float32),
          "next_observations": rng.standard_normal((n, S), dtype=np.float32),
          "terminals": rng.uniform(size=n) < 1e-3, "timeouts": np.zeros(n, bool)}
    ds["timeouts"][999::1000] = True
    env = "antmaze-medium-diverse-v2"
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    host = {k: v.copy() for k, v in ds.items()}
    ia.keep_mask_and_steps(host["terminals"], host["timeouts"], 1000)
    ia.modify_reward(host, env, 7)
    mh, sh = ia.compute_mean_std(host["observations"], 1e-3)
    host["observations"] = ia.normalize_states(host["observations"], mh, sh)
    host["next_observations"] = ia.normalize_states(host["next_observations"], mh, sh)
    b0 = ia.ReplayBuffer(S, A, n, DEV)
    b0.load_d4rl_dataset(host)
    torch.cuda.synchronize()
    t_host = time.perf_counter() - t0
    times = []
    for _ in range(3):
        b1 = ia.ReplayBuffer(S, A, n, DEV)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        prep.keep_mask_and_steps(ds["terminals"], ds["timeouts"], 1000, False, DEV)
        prep.prepare_replay(ds, b1, env_name=env, normalize_reward=7, normalize=True, eps=1e-3)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    _diag(f"prep at N=1M: numpy + upload {t_host * 1e3:.0f} ms; device pipeline {min(times) * 1e3:.0f} ms "
          f"(runs: {[round(t * 1e3) for t in times]})")
    np.testing.assert_array_equal(b1._rewards.cpu().numpy()[:, 0], host["rewards"])
    # At N = 1M numpy's float32 row-order sums have drifted (1e-4 relative on the mean of a
    # zero-mean column): the device statistics (double accumulation) are checked against float64
    # numpy, the numpy-float32 pipeline only loosely
    m64 = ds["observations"].astype(np.float64).mean(0)
    s64 = ds["observations"].astype(np.float64).std(0) + 1e-3
    want = (ds["observations"][:4096] - m64.astype(np.float32)) / s64.astype(np.float32)
    np.testing.assert_allclose(b1._states[:4096].cpu().numpy(), want, rtol=1e-6, atol=1e-6)
    drift = float(np.abs(b1._states[:4096].cpu().numpy() - host["observations"][:4096]).max())
    _diag(f"prep at N=1M: max |device - numpy float32 pipeline| on normalised states {drift:.2e} "
          f"(numpy mean drift {float(np.abs(mh - m64).max()):.2e})")
    assert drift < 5e-3
    assert min(times) < t_host
