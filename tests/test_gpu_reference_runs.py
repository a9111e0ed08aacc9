"""HIP path (through the C-ABI) against fixtures the REFERENCE produced for the rows that were
restatement-only in round 2: custom_offline's trainer (cref:438-556), its evaluate loop
(cref:559-579), and offline/iql.py's eval_actor (ref:265-341) -- tests/golden/custom_offline.npz
and eval_actor.npz, recorded by make_fixtures.py from the reference's own functions.  -m gpu."""
import numpy as np
import pytest
import torch

from oracle import philox
from tests import fake_envs, helpers
from tests.test_reference_runs import custom_traj

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def gc():
    return np.load(helpers.GOLDEN + "/custom_offline.npz")


@pytest.fixture(scope="module")
def ge():
    return np.load(helpers.GOLDEN + "/eval_actor.npz")


def test_custom_offline_trainer_matches_the_reference_run(gc):
    """fp32, convex Polyak, numpy-global-RNG sampler: K steps of OUR custom_offline trainer under
    np.random.seed(s) = the reference's K steps (losses, parameters, target, Adam state, keys)."""
    from torch.optim.lr_scheduler import CosineAnnealingLR
    from iqlpref_amd import custom_offline as co
    from tests import gpu_helpers as gh
    d, hyper, data, nets = custom_traj(gc)
    K, B, n = hyper["k_steps"], hyper["batch"], hyper["n_rows"]
    q, v, actor = gh.make_nets(hyper, nets)
    ao = torch.optim.Adam(actor.parameters(), lr=3e-4)
    qo = torch.optim.Adam(q.parameters(), lr=3e-4)
    tr = co.ImplicitQLearning(1.0, actor, ao, CosineAnnealingLR(ao, hyper["max_steps"]), q, qo, v,
                              torch.optim.Adam(v.parameters(), lr=3e-4), iql_tau=hyper["iql_tau"],
                              beta=hyper["beta"], gamma=hyper["discount"], tau=hyper["tau"], device=DEV)
    buf = co.ReplayBuffer(hyper["s_dim"], hyper["a_dim"], n + 7, DEV)
    buf.load_dataset({k: np.asarray(x) for k, x in data.items()})
    np.random.seed(int(d["np_seed"]))
    first = buf.sample(B)  # the reference's first draw ...
    np.testing.assert_array_equal(first[0].cpu().numpy(), data["observations"][d["indices"][0]])
    np.testing.assert_array_equal(first[2].cpu().numpy()[:, 0], data["rewards"][d["indices"][0]])
    np.random.seed(int(d["np_seed"]))
    losses = tr.train_on_buffer(buf, K, B).cpu().numpy()  # ... and all K fused with the steps
    np.testing.assert_allclose(losses, d["losses"], rtol=2e-5)
    assert abs(tr.actor_optimizer.param_groups[0]["lr"] - float(d["final_actor_lr"])) < 1e-12
    for net, mod in (("qf", tr.qf), ("vf", tr.vf), ("actor", tr.actor), ("q_target", tr.q_target)):
        for k, t in mod.state_dict().items():
            np.testing.assert_allclose(t.cpu().numpy(), d[f"final/{net}/{k}"], atol=2e-6, rtol=0, err_msg=f"{net}/{k}")
    for name, p in tr.qf.named_parameters():
        st = qo.state[p]
        for mk in ("exp_avg", "exp_avg_sq"):
            want = d[f"final/q_adam/{name}/{mk}"]
            err = np.abs(st[mk].cpu().numpy() - want).max() / (np.abs(want).max() + 1e-30)
            assert err < 1e-4, (name, mk, err)
    sd = tr.state_dict()
    assert sorted(sd.keys()) == list(d["state_dict_keys"])  # cref:546-556: actor_lr_scheduler, no total_it
    assert sd["actor_lr_scheduler"]["last_epoch"] == int(d["scheduler_last_epoch"]) == K


def test_custom_offline_evaluate_matches_the_reference_run(gc):
    """cref:559-579 on the stand-in gymnasium environment: same seeds, our actor on the GPU."""
    import iqlpref_amd as ia
    from iqlpref_amd import custom_offline as co
    S, A = fake_envs.DIMS["pen-human-v1"]
    actor = ia.GaussianPolicy(S, A, 0.8, hidden_dim=64, dropout=0.1)
    actor.load_state_dict({k[len("ev/actor/"):]: torch.from_numpy(gc[k]) for k in gc.files if k.startswith("ev/actor/")})
    actor = actor.to(DEV)
    env = ia.wrap_env(fake_envs.FakeGymnasiumEnv("pen-human-v1"), state_mean=gc["ev/mean"], state_std=gc["ev/std"])
    seen = []
    real = env.step

    class Spy:
        def __getattr__(self, name):
            return getattr(env, name)

        def step(self, a):
            seen.append(np.asarray(a).copy())
            return real(a)
    scores = co.evaluate(Spy(), actor, num_episodes=6, seed=40, device=DEV)
    assert actor.training
    want = gc["ev/actions"]
    assert len(seen) == len(want)
    np.testing.assert_allclose(np.stack(seen), want, atol=5e-6, rtol=0)
    np.testing.assert_allclose(scores, gc["ev/scores"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("tag,name", [("antmaze", "antmaze-medium-diverse-v2"), ("cheetah", "halfcheetah-medium-v2")])
def test_eval_actor_matches_the_reference_run(ge, tag, name):
    """ref:265-341 run by the reference on the stand-in vector environment; here OUR eval_actor
    (batched fp32 actor forward on the GPU) drives a fresh copy built from the same seeds."""
    import iqlpref_amd as ia
    max_action, n_eps, seed, n_envs, det = ge[f"{tag}/args"]
    S, A = fake_envs.DIMS[name]
    cls = ia.DeterministicPolicy if det else ia.GaussianPolicy
    actor = cls(S, A, float(max_action), hidden_dim=64, dropout=0.1)
    pre = f"{tag}/actor/"
    actor.load_state_dict({k[len(pre):]: torch.from_numpy(ge[k]) for k in ge.files if k.startswith(pre)})
    actor = actor.to(DEV)
    made = []

    def factory(env_name, seeds, mean, std):
        assert env_name == name and list(seeds) == [int(seed) + i for i in range(int(n_envs))]

        def make(s):
            def thunk():
                e = ia.wrap_env(fake_envs.FakeGymEnv(env_name), state_mean=mean, state_std=std)
                e.seed(s)
                return e
            return thunk
        made.append(fake_envs.SyncVectorEnv([make(s) for s in seeds]))
        return made[-1]

    scores, steps = ia.eval_actor(name, actor, float(max_action), ge[f"{tag}/mean"], ge[f"{tag}/std"], DEV,
                                  int(n_eps), int(seed), n_envs=int(n_envs), vector_env=factory)
    assert actor.training and made[0].closed
    want_actions = ge[f"{tag}/actions"]
    assert len(made[0].actions_seen) == len(want_actions)  # no environment step beyond the reference's last
    np.testing.assert_allclose(np.stack(made[0].actions_seen), want_actions, atol=5e-6, rtol=0)
    assert list(steps) == list(ge[f"{tag}/steps_to_goal"])
    if "antmaze" in name:
        np.testing.assert_array_equal(scores, ge[f"{tag}/scores"])  # sparse 0 / 1 returns
    else:
        np.testing.assert_allclose(scores, ge[f"{tag}/scores"], rtol=1e-5, atol=1e-5)


def test_group_close_then_solo_without_outputs():
    """a member stepped solo WITHOUT per-step outputs while in a group, then again
    after the group is dissolved, must stage its batch again (its argument slot moved twice)."""
    import iqlpref_amd as ia
    from tests import gpu_helpers as gh
    d, hyper, data, nets = helpers.load_traj("traj_antmaze", "bf16")
    B = hyper["batch"]
    buf = gh.make_buffer(hyper, data)
    alone = gh.make_trainer(hyper, nets, "bf16", seed=4)
    alone.train_steps(buf, 6 + 3 + 4 + 5, B, return_losses=False, graph_unroll=0)
    members = [gh.make_trainer(hyper, nets, "bf16", seed=s) for s in (3, 4)]
    group = ia.SeedGroup(members, mode="group")
    group.train_steps(buf, 6, B, graph_unroll=3)
    members[1].train_steps(buf, 3, B, return_losses=False, graph_unroll=0)  # solo, inside the group
    members[1].train_steps(buf, 4, B, return_losses=False, graph_unroll=2)  # continues it
    group.close()
    members[1].train_steps(buf, 5, B, return_losses=False, graph_unroll=0)  # solo, after the group
    torch.cuda.synchronize()
    assert members[1].total_it == alone.total_it == 18
    assert torch.equal(members[1]._params, alone._params) and torch.equal(members[1]._exp_avg_sq, alone._exp_avg_sq)
    assert torch.equal(members[1]._target, alone._target)
    # a member that never ran solo before the group: its own slot was all zero
    fresh = [gh.make_trainer(hyper, nets, "bf16", seed=s) for s in (8, 9)]
    g2 = ia.SeedGroup(fresh, mode="group")
    g2.train_steps(buf, 4, B, graph_unroll=0)
    g2.close()
    fresh[0].train_steps(buf, 3, B, return_losses=False, graph_unroll=0)
    ref8 = gh.make_trainer(hyper, nets, "bf16", seed=8)
    ref8.train_steps(buf, 7, B, return_losses=False, graph_unroll=0)
    torch.cuda.synchronize()
    assert torch.equal(fresh[0]._params, ref8._params)


def test_replay_touch_restages_after_in_place_edit():
    """Edits through the strided views need ReplayBuffer.touch(): the prefetched batch is dropped."""
    from tests import gpu_helpers as gh
    d, hyper, data, nets = helpers.load_traj("traj_antmaze", "bf16")
    B = hyper["batch"]
    a, b = gh.make_trainer(hyper, nets, "bf16", seed=2), gh.make_trainer(hyper, nets, "bf16", seed=2)
    buf_a, buf_b = gh.make_buffer(hyper, data), gh.make_buffer(hyper, data)
    a.train_steps(buf_a, 5, B, return_losses=False, graph_unroll=0)
    buf_a._rewards -= 1.0
    buf_a.touch()
    a.train_steps(buf_a, 5, B, return_losses=False, graph_unroll=0)
    b.train_steps(buf_b, 5, B, return_losses=True, graph_unroll=0)
    buf_b._rewards -= 1.0
    b.train_steps(buf_b, 5, B, return_losses=True, graph_unroll=0)  # (returns losses: always restaged)
    torch.cuda.synchronize()
    assert torch.equal(a._params, b._params)


def test_mlp_forward_with_active_dropout_matches_the_philox_oracle():
    """MLP.forward in train mode (ref:436-437; GaussianPolicy.act in train mode, ref:476-482):
    masks from oracle/philox.py:mlp_dropout_keep, kept values times float(1 / (1 - p))."""
    import iqlpref_amd as ia
    S, A, n, p = 11, 3, 133, 0.25
    scale = np.float32(1.0) / np.float32(1.0 - p)
    # (the default shape; three hidden layers of a width only the wide variant of the kernel takes)
    for H, NH in ((64, 2), (320, 3)):
        torch.manual_seed(77)
        actor = ia.GaussianPolicy(S, A, 1.0, hidden_dim=H, n_hidden=NH, dropout=p).to(DEV)
        x = torch.randn(n, S, device=DEV)
        lin = [(l.weight.detach().cpu().numpy().astype(np.float64), l.bias.detach().cpu().numpy().astype(np.float64))
               for l in actor.net.linears()]
        outs = []
        for call in range(2):
            got = actor(x).mean.cpu().numpy()  # train mode: dropout active
            h = x.cpu().numpy().astype(np.float64)
            for li, (W, b) in enumerate(lin[:-1]):
                keep = philox.mlp_dropout_keep(77, call, li, n, H, p)
                h = np.maximum(h @ W.T + b, 0.0) * keep * np.float64(scale)
            want = np.tanh(h @ lin[-1][0].T + lin[-1][1])
            np.testing.assert_allclose(got, want, atol=3e-6, rtol=0)
            outs.append(got)
        assert not np.array_equal(outs[0], outs[1])  # a fresh mask per call
    keep = philox.mlp_dropout_keep(77, 0, 0, 4096, 256, p)
    assert abs(keep.mean() - (1 - p)) < 5e-3
    actor.eval()
    e1, e2 = actor(x).mean.cpu().numpy(), actor(x).mean.cpu().numpy()
    np.testing.assert_array_equal(e1, e2)  # eval mode: no dropout
    actor.train()
    act = actor.act(x[0].cpu().numpy(), DEV)  # ref:476-482: samples through the active-dropout net
    assert act.shape == (A,) and np.all(np.abs(act) <= 1.0)


def test_pt_relabel_clamps_windows_handed_in_by_a_c_caller():
    """the window arrays live on the device, the C entry point cannot check them.
    The kernel clamps every window into the arrays (include/iqlhip.h): a window that violates the
    precondition gives the value of the clamped window -- never an out-of-bounds read."""
    import ctypes as C
    from iqlpref_amd import _lib
    from iqlpref_amd._lib import check, ptr, stream_ptr
    from oracle import relabel_oracle as ro
    from tests.test_gpu_relabel import make_pt
    S, A, QL, max_ep, n_rows = 5, 3, 6, 20, 40
    rng = np.random.default_rng(1)
    m = make_pt(ro.make_pt_params(rng, S, A, max_ep, embd=64, pref=8, inter=256, layers=1), S, A, max_ep, 4, 256)
    obs = torch.from_numpy(rng.standard_normal((n_rows, S)).astype(np.float32)).to(DEV)
    act = torch.from_numpy(rng.uniform(-1, 1, (n_rows, A)).astype(np.float32)).to(DEV)
    dev = lambda a, dt: torch.from_numpy(np.asarray(a, dtype=dt)).to(DEV)
    #            start          len       t0     (bad)            ->   the window the kernel uses
    bad = [(n_rows - 2, 5, 0), (-3, 4, 2), (7, 0, 1), (3, 9, 0), (10, 3, max_ep), (5, 4, -7)]
    good = [(n_rows - 5, 5, 0), (0, 4, 2), (7, 1, 1), (3, 6, 0), (10, 3, max_ep + 1 - 3), (5, 4, 0)]
    w, keep, _ = m._weights()
    lib = _lib.load()

    def run(wins):
        out = torch.empty(len(wins), dtype=torch.float32, device=DEV)
        st, ln, t0 = (dev([x[i] for x in wins], dt) for i, dt in ((0, np.int64), (1, np.int32), (2, np.int32)))
        check(lib.iqlhip_pt_relabel(C.byref(w), ptr(obs), ptr(act), n_rows, ptr(st), ptr(ln), ptr(t0), len(wins), QL,
                                    ptr(out), stream_ptr()))
        return out.cpu().numpy()
    got, want = run(bad), run(good)
    assert np.isfinite(got).all()
    np.testing.assert_array_equal(got, want)
    del keep


def test_queue_depth_bound_does_not_change_results(tmp_path):
    """IQLHIP_MAX_INFLIGHT (read once per process): a tiny bound makes the library wait on its own
    events every few dispatches; the run is the same bits as with the default bound."""
    import subprocess
    import sys
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "depth.py"
    script.write_text('''
import sys, hashlib
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
from tests import helpers, gpu_helpers as gh
d, hyper, data, nets = helpers.load_traj("traj_antmaze", "bf16")
tr = gh.make_trainer(hyper, nets, "bf16", seed=5)
buf = gh.make_buffer(hyper, data)
for n, u in ((40, 8), (13, 0), (40, 8)):
    tr.train_steps(buf, n, hyper["batch"], return_losses=False, graph_unroll=u)
torch.cuda.synchronize()
print("DIGEST", hashlib.sha256(tr._params.cpu().numpy().tobytes() + tr._target.cpu().numpy().tobytes()).hexdigest())
''')
    digests = []
    for depth in ("9", "0", None):
        env = dict(os.environ)
        env.pop("IQLHIP_MAX_INFLIGHT", None)
        if depth is not None:
            env["IQLHIP_MAX_INFLIGHT"] = depth
        p = subprocess.run([sys.executable, str(script), root], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-1500:]
        digests.append([l for l in p.stdout.splitlines() if l.startswith("DIGEST")][0])
    assert digests[0] == digests[1] == digests[2]
