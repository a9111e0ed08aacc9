"""The custom_offline flavour (algorithms/custom_offline/iql.py = "cref") on the HIP
path: per-episode PT relabel with true timesteps, numpy-generator sampler, convex Polyak form, no
autocast.  -m gpu.  PT numerics are parity unpinned (no runnable reference): the checker is the
numpy restatement oracle/relabel_oracle.py follows line by line."""
import numpy as np
import pytest
import torch

from oracle import iql_oracle as orc
from oracle import relabel_oracle as ro
from tests import helpers

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _episodes(rng, S, A, lengths):
    return [{"observations": rng.standard_normal((L + 1, S)).astype(np.float32),
             "actions": rng.uniform(-1, 1, (L, A)).astype(np.float32),
             "terminations": (np.arange(L) == L - 1) & (rng.uniform() < 0.5)} for L in lengths]


@pytest.mark.parametrize("S,A,QL,lengths", [(11, 3, 8, (3, 8, 9, 25, 1)), (45, 24, 20, (20, 61, 7))])
def test_episode_relabel_with_true_timesteps(S, A, QL, lengths):
    """cref:158-225: one forward over an episode's first QL steps, then rolling windows with the
    TRUE timesteps -- as (start, len, t0) windows of one kernel launch."""
    import iqlpref_amd as ia
    from iqlpref_amd import custom_offline as co
    from tests.test_gpu_relabel import make_pt
    rng = np.random.default_rng(QL)
    eps = _episodes(rng, S, A, lengths)
    p = ro.make_pt_params(rng, S, A, 200, embd=64, pref=16, inter=256, layers=1)
    model = make_pt(p, S, A, 200, 4, 256)
    got = co.qlearning_dataset(eps, model, QL)
    want = ro.custom_qlearning_dataset(eps, p, QL, num_heads=4)
    for k in ("observations", "actions", "next_observations", "terminals"):
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)
    np.testing.assert_allclose(got["rewards"], want["rewards"], rtol=5e-3, atol=5e-3)
    # the true timestep matters: with t0 = 0 the rolling windows of the long episodes differ
    start, length, t0 = co.episode_windows(lengths, QL)
    assert t0.max() > 0
    up = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(DEV)
    flat = model.window_values(up(got["observations"]), up(got["actions"]), up(start), up(length), QL)
    assert np.abs(flat.cpu().numpy() - got["rewards"]).max() > 1e-3


def test_qmlp_activations_match_numpy():
    """reward_models/q_mlp.py: every activation of its table, hidden and final."""
    from iqlpref_amd import custom_offline as co
    rng = np.random.default_rng(0)
    S, A, n = 7, 3, 513
    obs, act = rng.standard_normal((n, S)).astype(np.float32), rng.uniform(-1, 1, (n, A)).astype(np.float32)
    fns = {"cos": np.cos, "tanh": np.tanh, "relu": lambda v: np.maximum(v, 0),
           "softplus": lambda v: np.logaddexp(v, 0), "sin": np.sin,
           "leaky_relu": lambda v: np.where(v >= 0, v, 0.01 * v), "swish": lambda v: v / (1 + np.exp(-v)),
           "none": lambda v: v}
    for hidden in co.ACTIVATIONS:
        final = co.ACTIVATIONS[(co.ACTIVATIONS.index(hidden) + 3) % len(co.ACTIVATIONS)]
        layers = [{"kernel": (rng.standard_normal(s) / np.sqrt(s[0])).astype(np.float32),
                   "bias": (rng.standard_normal(s[1]) * 0.1).astype(np.float32)}
                  for s in ((S + A, 32), (32, 48), (48, 1))]
        m = co.QMLP(S, A, (32, 48), hidden, final).load_flax_params(layers).to(DEV)
        x = np.concatenate([obs, act], 1).astype(np.float64)
        for i, l in enumerate(layers):
            x = x @ l["kernel"].astype(np.float64) + l["bias"]
            x = fns[hidden](x) if i < 2 else fns[final](x)
        np.testing.assert_allclose(m(obs, act).cpu().numpy(), x[:, 0], rtol=2e-5, atol=2e-5, err_msg=hidden)
    ds = co.qlearning_dataset(_episodes(rng, S, A, (5, 9)), m, 1)
    assert ds["rewards"].shape == (14,) and ds["observations"].shape == (14, S)


def test_numpy_sampler_and_convex_polyak_step():
    """cref:277-284 + cref:85-87, 438-556: after np.random.seed(s) the batches are the reference's
    draw for draw; the step (fp32, (1 - tau) t + tau s target update) follows the oracle."""
    from tests import gpu_helpers as gh
    from iqlpref_amd import custom_offline as co
    import iqlpref_amd as ia
    from torch.optim.lr_scheduler import CosineAnnealingLR
    d, hyper, data, nets = helpers.load_traj("traj_antmaze", "fp32")
    B, K, n = hyper["batch"], 8, hyper["n_rows"]
    q, v, actor = gh.make_nets(hyper, nets)
    ao = torch.optim.Adam(actor.parameters(), lr=3e-4)
    tr = co.ImplicitQLearning(1.0, actor, ao, CosineAnnealingLR(ao, hyper["max_steps"]), q,
                              torch.optim.Adam(q.parameters(), lr=3e-4), v,
                              torch.optim.Adam(v.parameters(), lr=3e-4), iql_tau=hyper["iql_tau"],
                              beta=hyper["beta"], gamma=hyper["discount"], tau=hyper["tau"], device=DEV)
    buf = co.ReplayBuffer(hyper["s_dim"], hyper["a_dim"], n + 7, DEV)
    buf.load_dataset({k: np.asarray(x) for k, x in data.items()})
    np.random.seed(123)
    first = buf.sample(B)  # one reference-style draw ...
    np.random.seed(123)
    idx0 = np.random.randint(0, n, size=B)
    np.testing.assert_array_equal(first[0].cpu().numpy(), data["observations"][idx0])
    np.random.seed(5)
    losses = tr.train_on_buffer(buf, K, B).cpu().numpy()  # ... and K of them fused with the steps
    np.random.seed(5)
    o = orc.IQLOracle(*nets, iql_tau=hyper["iql_tau"], beta=hyper["beta"], max_steps=hyper["max_steps"],
                      discount=hyper["discount"], tau=hyper["tau"], mode="fp32", polyak_convex=True)
    for t in range(K):
        out = o.train(orc.gather_batch(data, np.random.randint(0, n, size=B)))
        np.testing.assert_allclose(losses[t], [out["value_loss"], out["q_loss"], out["actor_loss"]], rtol=2e-5)
    for k, t_ in tr.q_target.state_dict().items():
        np.testing.assert_allclose(t_.cpu().numpy(), o.q_target[k], atol=2e-6, rtol=0, err_msg=k)
    # the two Polyak forms round differently: the lerp form must NOT reproduce this target bit for bit
    o2 = orc.IQLOracle(*nets, iql_tau=hyper["iql_tau"], beta=hyper["beta"], max_steps=hyper["max_steps"],
                       discount=hyper["discount"], tau=hyper["tau"], mode="fp32", polyak_convex=False)
    np.random.seed(5)
    for t in range(K):
        o2.train(orc.gather_batch(data, np.random.randint(0, n, size=B)))
    k0 = next(iter(o.q_target))
    assert not np.array_equal(o.q_target[k0], o2.q_target[k0])
    sd = tr.state_dict()
    assert "actor_lr_scheduler" in sd and "total_it" not in sd and len(sd) == 7  # cref:546-556
    tr.load_state_dict(sd)
    assert tr.total_it == K
