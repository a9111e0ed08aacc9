"""GPU tests added in round 2: the gaps VERDICT r01 names (config-3 pipeline, checkpoint
corner cases) and the evaluation loop.  -m gpu."""
import os

import numpy as np
import pytest
import torch

from oracle import iql_oracle as orc
from oracle import philox
from oracle import relabel_oracle as ro
from tests import helpers

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _diag(line):
    path = os.environ.get("IQL_TEST_DIAG")
    if path:
        with open(path, "a") as f:
            f.write(line + "\n")


class FakeEnv:
    def __init__(self, m):
        self._max_episode_steps = m


@pytest.fixture(scope="module")
def gh():
    from tests import gpu_helpers
    assert torch.cuda.is_available()
    return gpu_helpers


# ----------------------------------------------------------------------------- #
# BASELINE config 3 as ONE pipeline at its stated size: pen-human shapes (S 45, A 24),
# N = 5000 (25 episodes x 200), preference-transformer relabel with QL = 100, state
# normalisation, then IQL with actor_dropout = 0.1 (configs/offline/iql/pen/human_v1.yaml)
# ----------------------------------------------------------------------------- #
@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_config3_pipeline_pt_relabel_then_train(gh, mode):
    import iqlpref_amd as ia
    from tests.test_gpu_relabel import make_pt
    S, A, N, EP, QL, H, B, K = 45, 24, 5000, 200, 100, 256, 256, 50
    rng = np.random.default_rng(3)
    raw = {"observations": rng.standard_normal((N, S)).astype(np.float32),
           "actions": rng.uniform(-1, 1, (N, A)).astype(np.float32),
           "rewards": np.zeros(N, np.float32),
           "terminals": np.zeros(N, bool), "timeouts": np.zeros(N, bool)}
    raw["timeouts"][EP - 1::EP] = True
    raw["terminals"][rng.choice(N, 6, replace=False)] = True
    p = ro.make_pt_params(rng, S, A, EP, embd=64, pref=64, inter=256, layers=1)
    model = make_pt(p, S, A, EP, 4, 256)
    ds = ia.qlearning_dataset_pt(FakeEnv(EP), model, QL, dataset=dict(raw))
    want = ro.qlearning_dataset_pt(dict(raw), p, EP, QL, num_heads=4)
    assert ds["rewards"].shape == want["rewards"].shape and ds["rewards"].shape[0] > N - 40
    for k in want:
        np.testing.assert_allclose(np.asarray(ds[k], np.float32), np.asarray(want[k], np.float32),
                                   rtol=5e-3, atol=5e-3, err_msg=k)
    assert float(np.std(ds["rewards"])) > 1e-3  # the relabel produced a signal, not a constant
    # ---- normalise and train on the GPU-relabelled data; the oracle steps the same data ----
    mean, std = ia.compute_mean_std(ds["observations"], 1e-3)
    data = {"observations": ia.normalize_states(ds["observations"], mean, std).astype(np.float32),
            "next_observations": ia.normalize_states(ds["next_observations"], mean, std).astype(np.float32),
            "actions": ds["actions"], "rewards": ds["rewards"].astype(np.float32),
            "terminals": np.asarray(ds["terminals"], np.float32)}
    n = data["rewards"].shape[0]
    torch.manual_seed(11)
    q, v = ia.TwinQ(S, A, hidden_dim=H), ia.ValueFunction(S, hidden_dim=H)
    actor = ia.GaussianPolicy(S, A, 1.0, hidden_dim=H, dropout=0.1)
    sd = lambda m: {k: t.detach().numpy().copy() for k, t in m.state_dict().items()}
    hyper = dict(s_dim=S, a_dim=A, hidden=H, deterministic=False, dropout=0.1, iql_tau=0.8, beta=3.0,
                 max_steps=1_000_000, discount=0.99, tau=0.005, n_rows=n)
    nets = (sd(q), sd(v), sd(actor))
    tr = gh.make_trainer(hyper, nets, mode, seed=21)
    buf = gh.make_buffer(hyper, data)
    got = tr.train_steps(buf, K, B, graph_unroll=10).cpu().numpy()
    o = helpers.make_oracle(hyper, nets, mode)
    worst = 0.0
    for t in range(K):
        km = [philox.dropout_keep(21, t, 1, B, H, 0.1), philox.dropout_keep(21, t, 2, B, H, 0.1)]
        out = o.train(orc.gather_batch(data, philox.sample_indices(21, t, B, n)), km)
        w = np.array([out["value_loss"], out["q_loss"], out["actor_loss"]])
        worst = max(worst, float(np.max(np.abs(got[t] - w) / np.abs(w))))
        np.testing.assert_allclose(got[t], w, rtol=2e-4 if mode == "fp32" else 1e-2, err_msg=f"step {t}")
    _diag(f"config3 {mode}: worst relative loss error over {K} steps {worst:.3e}")
    assert np.isfinite(got).all() and got[-1, 1] < got[0, 1]  # the critics learn the relabelled reward


# ----------------------------------------------------------------------------- #
# checkpoints (ref:664-688): keys written after torch.compile wrapped the nets
# (ref:1523-1528) and a checkpoint taken before the first optimiser step
# ----------------------------------------------------------------------------- #
def test_checkpoint_with_compile_prefix_and_empty_optimizer_state(gh):
    d, hyper, data, nets = helpers.load_traj("traj_antmaze", "fp32")
    B = hyper["batch"]
    buf = gh.make_buffer(hyper, data)
    import copy
    a = gh.make_trainer(hyper, nets, "fp32", seed=9)
    ck0 = copy.deepcopy(a.state_dict())  # before any step: the optimisers hold no state
    assert ck0["total_it"] == 0 and not ck0["q_optimizer"]["state"]
    first = a.train_steps(buf, 6, B, graph_unroll=0).cpu().numpy()
    ck6 = copy.deepcopy(a.state_dict())  # (state_dict() returns live views of the arenas)
    # 1. a checkpoint whose network keys carry torch.compile's prefix loads like a plain one
    pref = dict(ck6)
    for net in ("qf", "vf", "actor"):
        pref[net] = {"_orig_mod." + k: v for k, v in ck6[net].items()}
    b, c = gh.make_trainer(hyper, nets, "fp32", seed=9), gh.make_trainer(hyper, nets, "fp32", seed=9)
    b.load_state_dict(pref)
    c.load_state_dict(ck6)
    assert b.total_it == c.total_it == 6
    cont_a = a.train_steps(buf, 4, B, graph_unroll=0).cpu().numpy()
    cont_b = b.train_steps(buf, 4, B, graph_unroll=0).cpu().numpy()
    cont_c = c.train_steps(buf, 4, B, graph_unroll=0).cpu().numpy()
    np.testing.assert_array_equal(cont_b, cont_c)
    for (k, vb), (_, vc) in zip(b.actor.state_dict().items(), c.actor.state_dict().items()):
        assert torch.equal(vb, vc), k
    # the target network is not checkpointed: on load it is a copy of qf (ref:679), so the resumed
    # run is not the uninterrupted one; the first q_loss (reads V and the online critics only) is
    assert cont_a[0, 1] == cont_b[0, 1] and np.isfinite(cont_b).all()
    # 2. loading the step-0 checkpoint into a trainer that has already trained resets the Adam
    # moments too (no stale arena moments behind a reset total_it): the run repeats bit for bit
    a.load_state_dict(ck0)
    assert a.total_it == 0 and float(a._exp_avg.abs().max()) == 0.0 and float(a._exp_avg_sq.abs().max()) == 0.0
    again = a.train_steps(buf, 6, B, graph_unroll=0).cpu().numpy()
    np.testing.assert_array_equal(first, again)


# ----------------------------------------------------------------------------- #
# eval_actor (ref:265-341) driven by a fake vector environment (no gym)
# ----------------------------------------------------------------------------- #
class FakeVecEnv:
    """n lock-stepped environments with auto-reset: observation = f(env, t), the reward depends
    on the action, episode i of env e ends after a scheduled number of steps."""

    def __init__(self, n, S, A, seed, mean, std):
        self.n, self.S, self.A = n, S, A
        self.rng = np.random.default_rng(seed)
        self.mean, self.std = mean, std
        self.t = np.zeros(n, dtype=np.int64)
        self.horizon = self.rng.integers(3, 12, size=n)
        self.closed = False
        self.actions_seen = []

    def _obs(self):
        raw = np.sin(0.37 * self.t[:, None] + np.arange(self.S)[None, :] * (1 + np.arange(self.n)[:, None]))
        return ((raw - self.mean) / self.std).astype(np.float64)

    def reset(self):
        self.t[:] = 0
        return self._obs()

    def step(self, actions):
        assert actions.shape == (self.n, self.A) and actions.dtype == np.float32
        self.actions_seen.append(actions.copy())
        self.t += 1
        rewards = actions.astype(np.float64).sum(axis=1) * 0.1 + (self.t == self.horizon) * 1.0
        dones = self.t >= self.horizon
        for e in np.flatnonzero(dones):
            self.t[e] = 0
            self.horizon[e] = self.rng.integers(3, 12)
        return self._obs(), rewards, dones, [{}] * self.n

    def close(self):
        self.closed = True


@pytest.mark.parametrize("env_name,det", [("antmaze-medium-diverse-v2", False), ("pen-human-v1", True)])
def test_eval_actor_with_fake_vector_env(gh, env_name, det):
    import iqlpref_amd as ia
    S, A, n_eps, n_envs = 11, 3, 17, 5
    torch.manual_seed(4)
    cls = ia.DeterministicPolicy if det else ia.GaussianPolicy
    actor = cls(S, A, 0.7, hidden_dim=64, dropout=0.1).to(DEV)
    mean, std = np.linspace(-0.2, 0.2, S), np.linspace(0.8, 1.3, S)
    made = []

    def factory(name, seeds, m, s):
        assert name == env_name and list(seeds) == [100 + i for i in range(n_envs)]
        made.append(FakeVecEnv(len(seeds), S, A, 5, m, s))
        return made[-1]

    scores, steps = ia.eval_actor(env_name, actor, 0.7, mean, std, DEV, n_eps, 100, n_envs=n_envs,
                                  vector_env=factory)
    assert actor.training and made[0].closed and scores.shape == (n_eps,)
    # restatement of ref:296-333 on a second copy of the environment, the policy evaluated in
    # plain torch fp32 on the CPU (eval mode: no dropout)
    env = FakeVecEnv(n_envs, S, A, 5, mean, std)
    cpu = cls(S, A, 0.7, hidden_dim=64, dropout=0.1)
    cpu.load_state_dict({k: v.cpu() for k, v in actor.state_dict().items()})
    lin = cpu.net.linears()
    obs = env.reset()
    ret, length, want_scores, want_steps = np.zeros(n_envs), np.zeros(n_envs, dtype=np.int64), [], []
    k = 0
    while len(want_scores) < n_eps:
        x = torch.tensor(obs, dtype=torch.float32)
        with torch.no_grad():
            h = torch.relu(torch.nn.functional.linear(x, lin[0].weight, lin[0].bias))
            h = torch.relu(torch.nn.functional.linear(h, lin[1].weight, lin[1].bias))
            m_ = torch.tanh(torch.nn.functional.linear(h, lin[2].weight, lin[2].bias))
        act = torch.clamp(0.7 * m_, -0.7, 0.7).numpy()
        np.testing.assert_allclose(made[0].actions_seen[k], act, atol=2e-6)
        k += 1
        obs, rew, dones, _ = env.step(made[0].actions_seen[k - 1])  # same actions: same trajectory
        ret += rew
        length += 1
        for i in range(n_envs):
            if dones[i]:
                if "antmaze" in env_name and ret[i] > 0.5:
                    want_steps.append(int(length[i]))
                want_scores.append(float(ret[i]))
                ret[i], length[i] = 0.0, 0
                if len(want_scores) >= n_eps:
                    break
    np.testing.assert_array_equal(scores, np.asarray(want_scores[:n_eps]))
    assert steps == want_steps and (len(steps) > 0) == ("antmaze" in env_name)
    assert len(made[0].actions_seen) == k  # no environment step beyond the last needed one


# ----------------------------------------------------------------------------- #
# checkpoint compatibility with the reference's readers (evaluation/d4rl/iql_eval_median.py:252-262):
# tests/golden/our_checkpoint.pt was written by OUR trainer on an MI355X (tools/make_checkpoint.py);
# tests/golden/make_fixtures.py --checkpoint-compat loaded its ["actor"] into the REFERENCE's
# GaussianPolicy (strict=False) and recorded the policy's output -- here our actor must give the same
# ----------------------------------------------------------------------------- #
def test_checkpoint_read_by_the_reference_matches_our_actor():
    import iqlpref_amd as ia
    g = np.load(os.path.join(helpers.GOLDEN, "checkpoint_compat.npz"))
    assert list(g["missing"]) == [] and list(g["unexpected"]) == []  # the reference found every key it wanted
    assert sorted(g["keys"]) == ["actor", "actor_lr_schedule", "actor_optimizer", "q_optimizer", "qf",
                                 "total_it", "v_optimizer", "vf"]  # ref:664-674
    ck = torch.load(os.path.join(helpers.GOLDEN, "our_checkpoint.pt"), map_location="cpu", weights_only=True)
    assert int(g["total_it"]) == ck["total_it"] == 25
    S, A, H = ck["actor"]["net.net.0.weight"].shape[1], ck["actor"]["log_std"].shape[0], ck["actor"]["net.net.0.weight"].shape[0]
    actor = ia.GaussianPolicy(S, A, 1.0, hidden_dim=H, dropout=0.1)
    res = actor.load_state_dict(ck["actor"], strict=False)
    assert not res.missing_keys and not res.unexpected_keys
    actor = actor.to(DEV).eval()
    obs = torch.from_numpy(g["obs"]).to(DEV)
    dist = actor(obs)
    np.testing.assert_allclose(dist.mean.cpu().numpy(), g["mean"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(dist.stddev.cpu().numpy()[0], g["std"], rtol=1e-6)
    np.testing.assert_allclose(actor.act(g["obs"][0], DEV), g["act0"], rtol=0, atol=2e-6)
    # and the whole checkpoint resumes a trainer (ref:676-688)
    q, v = ia.TwinQ(S, 6, hidden_dim=H).to(DEV), ia.ValueFunction(S, hidden_dim=H).to(DEV)
    a2 = ia.GaussianPolicy(S, A, 1.0, hidden_dim=H, dropout=0.1).to(DEV)
    tr = ia.ImplicitQLearning(
        max_action=1.0, actor=a2, actor_optimizer=torch.optim.Adam(a2.parameters(), lr=3e-4), q_network=q,
        q_optimizer=torch.optim.Adam(q.parameters(), lr=3e-4), v_network=v,
        v_optimizer=torch.optim.Adam(v.parameters(), lr=3e-4), iql_tau=0.8, beta=3.0, max_steps=1000, device=DEV)
    tr.load_state_dict(ck)
    assert tr.total_it == 25 and torch.equal(a2.log_std.cpu(), ck["actor"]["log_std"])
    st = tr.actor_optimizer.state[a2.log_std]
    assert float(st["step"]) == 25 and torch.equal(st["exp_avg"].cpu(),
                                                   ck["actor_optimizer"]["state"][0]["exp_avg"])


# ----------------------------------------------------------------------------- #
# The multi-rank launcher of bench.py, end to end on ONE GPU: two rank processes (fresh
# interpreters, started before this process has touched the GPU in them) sharing cuda:0, the
# collectives over gloo (IQL_BENCH_BACKEND=gloo; the driver's SCALE runs use the default, RCCL).
# ----------------------------------------------------------------------------- #
def test_bench_two_ranks_on_one_gpu_over_gloo():
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR",
                                                              "MASTER_PORT")}
    env["IQL_BENCH_BACKEND"] = "gloo"
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout  # rank 0 prints ONE JSON line
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 20 and rec["warmup"] == 5 and rec["scaling"] == "weak"
    assert rec["metric"] == "iql_grad_steps_per_sec" and rec["unit"] == "steps/s"
    ranks = rec["ranks"]
    assert [r["rank"] for r in ranks] == [0.0, 1.0] and ranks[0]["seed"] != ranks[1]["seed"]
    assert all(np.isfinite([r["value_loss"], r["q_loss"], r["actor_loss"]]).all() for r in ranks)
    assert ranks[0]["q_loss"] != ranks[1]["q_loss"]  # two different runs
    # whole-job value = 2 x K steps / the slower rank's block time
    assert abs(rec["value"] - 2 * 20 / (rec["ms_per_step"] * 20 / 1e3)) / rec["value"] < 1e-9
    assert rec["value"] > 2_000  # two ranks sharing one GPU still run thousands of steps/s


# ----------------------------------------------------------------------------- #
# BASELINE configs[3]'s product path, rehearsed on ONE GPU: train() under WORLD_SIZE = 2
# (ensemble_sweeps/launch.sh:84-94: one process per GPU, each its own seed).  Two fresh child
# processes (tests/train_rank_child.py) with IQL_DIST_BACKEND=gloo -- both ranks on cuda:0, the
# metric all-gather over gloo; the default is one GPU per rank and RCCL, nothing else differs.
# ----------------------------------------------------------------------------- #
@pytest.mark.parametrize("k_seeds", [1, 2])
def test_train_two_ranks_on_one_gpu(tmp_path, k_seeds):
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK")}
    base.update(WORLD_SIZE="2", LOCAL_WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                IQL_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(root, "tests", "train_rank_child.py"), str(tmp_path),
                               str(k_seeds)], env=dict(base, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    recs = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(2)]
    K = k_seeds
    for r, rec in enumerate(recs):
        # rank r trains seeds base + r K .. base + r K + K - 1 (distributed.rank_seed), all on cuda:0 here
        assert rec["rank"] == r and rec["device"] == "cuda:0"
        assert rec["seeds"] == [100 + r * K + k for k in range(K)] and rec["total_it"] == [40] * K
        loss_logs = [d for _, d in rec["logs"] if "value_loss" in d]
        eval_logs = [d for _, d in rec["logs"] if "mean_score" in d and "n_seeds" not in d]
        assert len(loss_logs) == 4 * K and len(eval_logs) == 2 * K
        if K > 1:
            assert sorted({int(d["seed"]) for d in loss_logs}) == rec["seeds"]
        summaries = [d for _, d in rec["logs"] if "n_seeds" in d]
        # the end-of-eval all-gather reached every rank; rank 0 logs the summary of all 2 K seeds (2 evaluations)
        assert len(summaries) == (2 if r == 0 else 0)
        for smry in summaries:
            assert smry["n_seeds"] == 2 * K and smry["steps_per_sec_total"] > 0 and np.isfinite(smry["mean_score_mean"])
    assert len({ps for rec in recs for ps in rec["param_sum"]}) == 2 * K  # 2 K different runs
    # each seed as a rank / slot of this job == that seed's solo run (same parameters to the last bit)
    import iqlpref_amd as ia  # noqa: F401  (the solo run happens in a third child: this process keeps its GPU state)
    env1 = {k: v for k, v in base.items() if k not in ("WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                        "IQL_DIST_BACKEND")}
    solo_dir = tmp_path / "solo"
    solo_dir.mkdir()
    code = ("import sys, json, os; sys.argv = ['x', %r, '1']; sys.path.insert(0, %r); "
            "import iqlpref_amd as ia; import tests.train_rank_child as c; "
            "real = ia.TrainConfig; "
            "ia.TrainConfig = lambda **kw: real(**dict(kw, seed=%d)); c.main()" % (str(solo_dir), root, 100 + 2 * K - 1))
    p = subprocess.run([sys.executable, "-c", code], env=env1, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    solo = json.load(open(solo_dir / "rank0.json"))
    assert solo["seeds"] == [100 + 2 * K - 1] and solo["param_sum"][0] == recs[1]["param_sum"][K - 1]
