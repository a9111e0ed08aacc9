"""End-to-end: train(config) on synthetic data (the reference's loop, ref:1393-1570)
checked against the oracle driven with the same seeds, plus the relabel branches of
build_dataset.  -m gpu."""
import os

import numpy as np
import pytest
import torch

from oracle import iql_oracle as orc
from oracle import philox
from oracle import relabel_oracle as ro

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def synth(n, S, A, seed=0):
    rng = np.random.default_rng(seed)
    return {
        "observations": (rng.standard_normal((n, S)) * 2 + 0.5).astype(np.float32),
        "actions": rng.uniform(-1, 1, (n, A)).astype(np.float32),
        "rewards": (rng.uniform(size=n) < 0.05).astype(np.float32),
        "next_observations": (rng.standard_normal((n, S)) * 2 + 0.5).astype(np.float32),
        "terminals": (rng.uniform(size=n) < 0.01).astype(np.float32),
    }


def test_train_loop_matches_oracle(tmp_path):
    import iqlpref_amd as ia
    S, A, N, B = 29, 8, 3000, 64
    data = synth(N, S, A)
    cfg = ia.TrainConfig(env="antmaze-medium-diverse-v2", max_timesteps=40, log_freq=10, eval_freq=20,
                         batch_size=B, normalize_reward=1, beta=10.0, iql_tau=0.9, seed=5, device=DEV,
                         buffer_size=10_000_000, checkpoints_path=str(tmp_path))
    logs = []
    tr = ia.train(cfg, dataset={k: v.copy() for k, v in data.items()}, state_dim=S, action_dim=A,
                  max_action=1.0, logger=lambda d, step: logs.append((step, d)),
                  evaluate=lambda actor, t: (np.array([1.0, 0.0]), [17]), precision="fp32")
    assert tr.total_it == 40
    loss_logs = [d for _, d in logs if "value_loss" in d]
    eval_logs = [d for _, d in logs if "mean_score" in d]
    assert len(loss_logs) == 4 and len(eval_logs) == 2
    assert eval_logs[0] == {"mean_score": 0.5, "avg_steps_to_goal": 17.0}
    for step in (19, 39):  # ref:1561 names the file after the 0-based step
        ck = torch.load(os.path.join(cfg.checkpoints_path, f"checkpoint_{step}.pt"), weights_only=True)
        assert sorted(ck) == ["actor", "actor_lr_schedule", "actor_optimizer", "q_optimizer", "qf",
                              "total_it", "v_optimizer", "vf"]
    # ---- the same run in the oracle: dataset prep (ref:1435-1448), init (ref:1467-1480), 10 steps ----
    d = {k: v.copy() for k, v in data.items()}
    d["rewards"] -= 1.0
    mean, std = d["observations"].mean(0), d["observations"].std(0) + 1e-3
    d["observations"] = (d["observations"] - mean) / std
    d["next_observations"] = (d["next_observations"] - mean) / std
    ia.set_seed(5)
    q, v, a = ia.TwinQ(S, A), ia.ValueFunction(S), ia.GaussianPolicy(S, A, 1.0)
    sd = lambda m: {k: t.detach().numpy() for k, t in m.state_dict().items()}
    o = orc.IQLOracle(sd(q), sd(v), sd(a), iql_tau=0.9, beta=10.0, max_steps=40, discount=0.99, tau=0.005,
                      mode="fp32")
    acc = np.zeros(3)
    for t in range(10):
        out = o.train(orc.gather_batch(d, philox.sample_indices(5, t, B, N)))
        acc += [out["value_loss"], out["q_loss"], out["actor_loss"]]
    got = loss_logs[0]
    np.testing.assert_allclose([got["value_loss"], got["q_loss"], got["actor_loss"]], acc / 10, rtol=5e-5)


def test_build_dataset_branches(tmp_path):
    """Config-driven relabel selection (ref:1402-1429) for the MR and PT models."""
    import iqlpref_amd as ia
    S, A, N = 6, 3, 400
    rng = np.random.default_rng(1)
    raw = {"observations": rng.standard_normal((N, S)).astype(np.float32),
           "actions": rng.uniform(-1, 1, (N, A)).astype(np.float32),
           "rewards": np.zeros(N, np.float32), "terminals": rng.uniform(size=N) < 0.02,
           "timeouts": np.zeros(N, bool)}
    raw["timeouts"][[99, 199, 299]] = True

    class Env:
        _max_episode_steps = 100

    # MR model directory written the way the reference's trainer would
    mr = tmp_path / "mr"
    mr.mkdir()
    (mr / "config.yaml").write_text("activations: tanh\n")
    w = [rng.standard_normal(s).astype(np.float32) * 0.4 for s in ((S + A, 16), (16,), (16, 16), (16,), (16, 1), (1,))]
    torch.save({"net": {"_orig_mod.layers.0.W": torch.from_numpy(w[0]), "_orig_mod.layers.0.b": torch.from_numpy(w[1]),
                        "_orig_mod.layers.linear_1.W": torch.from_numpy(w[2]),
                        "_orig_mod.layers.linear_1.b": torch.from_numpy(w[3]),
                        "_orig_mod.output.W": torch.from_numpy(w[4]), "_orig_mod.output.b": torch.from_numpy(w[5])}},
               mr / "best_model.pt")
    cfg = ia.TrainConfig(env="pen-human-v1", reward_model_path=str(mr), query_length=1, device=DEV)
    out = ia.build_dataset(cfg, Env(), dict(raw))
    want = ro.qlearning_dataset_mr(dict(raw), w, 100, activation="tanh")
    for k in want:
        np.testing.assert_allclose(np.asarray(out[k], np.float32), np.asarray(want[k], np.float32),
                                   rtol=2e-5, atol=2e-5)

    # PT model directory
    ptd = tmp_path / "pt"
    ptd.mkdir()
    (ptd / "config.yaml").write_text("num_heads: 4\nintermediate_dim: 256\nmodel_eps: 1.0e-5\n")
    p = ro.make_pt_params(rng, S, A, 100, embd=64, pref=16, inter=256, layers=1)
    state = {k: torch.from_numpy(v) for k, v in p.items()}
    state["gpt.layers.0.attention.causal_bias"] = torch.tril(torch.ones(1, 1, 32, 32))
    torch.save({"net": state}, ptd / "best_model.pt")
    cfg = ia.TrainConfig(env="pen-human-v1", reward_model_path=str(ptd), query_length=10, device=DEV)
    out = ia.build_dataset(cfg, Env(), dict(raw))
    want = ro.qlearning_dataset_pt(dict(raw), p, 100, 10, num_heads=4)
    for k in want:
        np.testing.assert_allclose(np.asarray(out[k], np.float32), np.asarray(want[k], np.float32),
                                   rtol=5e-3, atol=5e-3)


def test_seeds_per_gpu_equals_solo_runs_bit_for_bit(tmp_path):
    """ensemble_sweeps/launch.sh:12,84-94 (AGENTS_PER_GPU agents per GPU, each its own run):
    train(cfg, seeds_per_gpu=3) = three solo train() runs of seeds s, s+1, s+2 -- every logged loss
    window, every evaluation call (own actor, own seed), every checkpoint tensor."""
    import iqlpref_amd as ia
    S, A, N, B = 29, 8, 3000, 64
    data = synth(N, S, A)

    def run(seed, k_seeds, sub):
        cfg = ia.TrainConfig(env="antmaze-medium-diverse-v2", max_timesteps=60, log_freq=10, eval_freq=30,
                             batch_size=B, normalize_reward=1, beta=10.0, iql_tau=0.9, seed=seed, device=DEV,
                             buffer_size=10_000_000, checkpoints_path=str(tmp_path / sub))
        logs, evals = [], []

        def evaluate(actor, t):  # a value that depends on the actor handed in
            w = actor.net.linears()[2].weight
            evals.append((t, float(w.detach().double().sum())))
            return np.array([float(w[0, 0]), 1.0]), [t]
        out = ia.train(cfg, dataset={k: v.copy() for k, v in data.items()}, state_dim=S, action_dim=A,
                       max_action=1.0, logger=lambda d, step: logs.append((step, dict(d))), evaluate=evaluate,
                       precision="bf16", seeds_per_gpu=k_seeds)
        return cfg, out, logs, evals

    cfg_g, trainers, logs_g, evals_g = run(7, 3, "group")
    assert isinstance(trainers, list) and [t._seed for t in trainers] == [7, 8, 9]
    assert all(t.total_it == 60 for t in trainers)
    assert [e[0] for e in evals_g] == [30, 30, 30, 60, 60, 60]
    for k in range(3):
        cfg_s, tr, logs_s, evals_s = run(7 + k, 1, f"solo{k}")
        mine = [(st, {n: v for n, v in d.items() if n != "seed"}) for st, d in logs_g if d.get("seed") == 7 + k]
        assert len(mine) == len(logs_s) == 6 + 2 and mine == logs_s  # 6 loss windows + 2 evaluations, equal floats
        assert [e for i, e in enumerate(evals_g) if i % 3 == k] == evals_s
        assert torch.equal(tr._params, trainers[k]._params) and torch.equal(tr._target, trainers[k]._target)
        for step in (29, 59):
            a = torch.load(os.path.join(cfg_g.checkpoints_path, f"seed_{7 + k}", f"checkpoint_{step}.pt"), weights_only=True)
            b = torch.load(os.path.join(cfg_s.checkpoints_path, f"checkpoint_{step}.pt"), weights_only=True)
            assert a["total_it"] == b["total_it"] == step + 1
            for net in ("qf", "vf", "actor"):
                for name in a[net]:
                    assert torch.equal(a[net][name], b[net][name]), (net, name)
            for opt in ("q_optimizer", "v_optimizer", "actor_optimizer"):
                for i, st in a[opt]["state"].items():
                    assert torch.equal(st["exp_avg"], b[opt]["state"][i]["exp_avg"])
                    assert torch.equal(st["exp_avg_sq"], b[opt]["state"][i]["exp_avg_sq"])
            assert a["actor_lr_schedule"]["last_epoch"] == b["actor_lr_schedule"]["last_epoch"]
    # different seeds did train differently
    assert not torch.equal(trainers[0]._params, trainers[1]._params)


def test_seed_tied_reward_models_give_every_seed_its_own_dataset(tmp_path):
    """iql_eval.py:143-146: reward_model_path = f"{root}_{seed}": with K seeds per GPU seed s trains
    on the dataset relabelled by ITS reward model."""
    import iqlpref_amd as ia
    S, A, N = 6, 3, 400
    rng = np.random.default_rng(1)
    raw = {"observations": rng.standard_normal((N, S)).astype(np.float32),
           "actions": rng.uniform(-1, 1, (N, A)).astype(np.float32),
           "rewards": np.zeros(N, np.float32), "terminals": rng.uniform(size=N) < 0.02,
           "timeouts": np.zeros(N, bool)}
    raw["timeouts"][[99, 199, 299]] = True

    class Env:
        _max_episode_steps = 100

    for seed in (3, 4):
        mr = tmp_path / f"mr_{seed}"
        mr.mkdir()
        (mr / "config.yaml").write_text("activations: relu\n")
        w = [np.random.default_rng(seed).standard_normal(s).astype(np.float32) * 0.4
             for s in ((S + A, 16), (16,), (16, 16), (16,), (16, 1), (1,))]
        torch.save({"net": {"layers.0.W": torch.from_numpy(w[0]), "layers.0.b": torch.from_numpy(w[1]),
                            "layers.linear_1.W": torch.from_numpy(w[2]), "layers.linear_1.b": torch.from_numpy(w[3]),
                            "output.W": torch.from_numpy(w[4]), "output.b": torch.from_numpy(w[5])}},
                   mr / "best_model.pt")

    def run(seed, k):
        cfg = ia.TrainConfig(env="pen-human-v1", reward_model_root=str(tmp_path / "mr"), query_length=1, seed=seed,
                             max_timesteps=12, log_freq=6, eval_freq=12, batch_size=32, device=DEV)
        assert cfg.reward_model_path == str(tmp_path / "mr") + f"_{seed}"
        return ia.train(cfg, env=Env(), raw_dataset=dict(raw), state_dim=S, action_dim=A, max_action=1.0,
                        logger=lambda d, step: None, precision="fp32", seeds_per_gpu=k)

    both = run(3, 2)
    for k, tr in enumerate(both):
        solo = run(3 + k, 1)
        assert torch.equal(solo._params, tr._params), k


def test_train_default_prep_gives_the_reference_normalised_states():
    """train()'s default path z-scores on the device with numpy's statistics
    (ref:132-139, 1438-1448): the buffer holds the reference's normalised states bit for bit."""
    import iqlpref_amd as ia
    from iqlpref_amd import prep
    S, A, N = 17, 6, 5000
    data = synth(N, S, A, seed=3)
    buf = ia.ReplayBuffer(S, A, N, DEV)
    mean, std = prep.prepare_replay({k: v.copy() for k, v in data.items()}, buf, env_name="halfcheetah-medium-v2",
                                    normalize_reward=0, normalize=True, eps=1e-3, stats="host")
    m, s = ia.compute_mean_std(data["observations"], 1e-3)
    np.testing.assert_array_equal(mean, m)
    np.testing.assert_array_equal(std, s)
    np.testing.assert_array_equal(buf._states.cpu().numpy(), ia.normalize_states(data["observations"], m, s))
    np.testing.assert_array_equal(buf._next_states.cpu().numpy(), ia.normalize_states(data["next_observations"], m, s))
    np.testing.assert_array_equal(buf._rewards.cpu().numpy()[:, 0], data["rewards"])


@pytest.mark.parametrize("tag", ["antmaze_fp32", "antmaze_bf16", "cheetah_bf16"])
def test_train_replays_the_references_own_train_run(tmp_path, tag):
    """A16 pinned to the reference: tests/golden/train_runs.npz holds runs of the reference's own
    train() (ref:1393-1570; d4rl / gym / wandb / pyrallis calls landing in stand-ins, torch.compile left
    in) -- the index stream its sampler drew, every wandb.log payload with its step, its checkpoint
    files.  OUR train() is given the same stand-in dataset, the same config and the recorded index
    stream, and must log the same records at the same steps, write the same files, and end on the
    same parameters.  fp32 (the reference with autocast disabled) to 10-step tolerances; bf16 (as
    written) to the bf16 ones."""
    import iqlpref_amd as ia
    from tests import fake_envs
    from tests.golden.make_fixtures import TRAIN_RUNS, TRAIN_SCHEDULE, train_dataset
    from tests.helpers import GOLDEN, tensor_checks
    d = np.load(f"{GOLDEN}/train_runs.npz")
    g = lambda k: d[f"{tag}/{k}"]
    run, mode = tag.rsplit("_", 1)
    spec = TRAIN_RUNS[run]
    raw = train_dataset(spec)
    for k, v in raw.items():  # the dataset the reference's d4rl stand-in returned
        assert np.array_equal(tensor_checks(v), g(f"check/data/{k}")), k
    S, A = fake_envs.DIMS[spec["env"]]
    cfg = ia.TrainConfig(env=spec["env"], seed=spec["seed"], normalize_reward=spec["normalize_reward"],
                         beta=spec["beta"], iql_tau=spec["iql_tau"], device=DEV, checkpoints_path=str(tmp_path),
                         **TRAIN_SCHEDULE)
    idx = torch.from_numpy(g("indices").astype(np.int64))
    logs, first_actions = [], []

    def vector_env(name, seeds, mean, std):  # what the reference's gym.vector stand-in built (make_fixtures)
        def make(seed):
            def thunk():
                e = fake_envs.TransformObservation(fake_envs.FakeGymEnv(name), lambda o: (o - mean) / std)
                e.seed(seed)
                return e
            return thunk
        v = fake_envs.SyncVectorEnv([make(s) for s in seeds])
        real_step, seen = v.step, []

        def step(a):
            if not seen:
                first_actions.append(np.asarray(a).copy())
                seen.append(1)
            return real_step(a)
        v.step = step
        return v

    tr = ia.train(cfg, env=fake_envs.FakeGymEnv(spec["env"]), dataset={k: v.copy() for k, v in raw.items()},
                  logger=lambda rec, step: logs.append((step, dict(rec))), precision=mode, vector_env=vector_env,
                  index_stream=lambda k, t, n: idx[t:t + n])
    assert tr.total_it == int(g("total_it")) == 60
    # ---- the records, in order: same steps, same keys, same values ----
    assert [s for s, _ in logs] == list(g("log_steps"))
    assert ["|".join(r.keys()) for _, r in logs] == list(g("log_keys"))
    ltol = 2e-5 if mode == "fp32" else 2e-2
    for (step, rec), want in zip(logs, g("log_values")):
        vals = np.asarray(list(rec.values()))
        if "mean_score" in rec:
            # (antmaze stand-in: the seed decides success, so the scores are exact; cheetah's reward is a
            # smooth function of the actions: the actor's outputs to rounding)
            np.testing.assert_allclose(vals, want[:len(vals)], rtol=1e-9 if "antmaze" in tag else (1e-5 if mode == "fp32" else 2e-2))
        else:
            np.testing.assert_allclose(vals, want, rtol=ltol, err_msg=f"loss window ending at step {step}")
    # ---- evaluations saw the reference's actor: its first actions of each evaluation ----
    want_fa = g("first_actions")
    assert len(first_actions) == len(want_fa) == 2
    for got, want in zip(first_actions, want_fa):
        np.testing.assert_allclose(got, want, atol=5e-6 if mode == "fp32" else 2e-2)
    # ---- checkpoints: same file names (ref:1561: the 0-based step), same contents ----
    files = sorted(os.listdir(cfg.checkpoints_path))
    assert files == list(g("checkpoint_files"))
    for f in ("checkpoint_29.pt", "checkpoint_59.pt"):
        ck = torch.load(os.path.join(cfg.checkpoints_path, f), weights_only=True)
        assert sorted(ck.keys()) == list(g(f"ckpt/{f}/keys"))
        assert ck["total_it"] == int(g(f"ckpt/{f}/total_it"))
        assert ck["actor_lr_schedule"]["last_epoch"] == int(g(f"ckpt/{f}/last_epoch"))
        # (the reference's keys carry the torch.compile prefix, ref:1523-1528; ours are the plain names, which is
        # what its own pre-compile load path and evaluation/d4rl/iql_eval_median.py:252-262 read)
        assert [k.removeprefix("_orig_mod.") for k in g(f"ckpt/{f}/actor_keys")] == list(ck["actor"].keys())
        for net in ("qf", "vf", "actor"):
            for name, t in ck[net].items():
                a = t.cpu().numpy()
                key = f"{tag}/ckpt/{f}/{net}/{name}"
                want, got = (d[key + "#stride37"], a.reshape(-1)[::37]) if key + "#stride37" in d.files else (d[key], a)
                # fp32: the 60-step parameter tolerance of the trajectory tests (all but a sliver of a tensor
                # within 2e-6, none beyond 60 lr); bf16: within the run's movement (60 x lr = 0.018)
                diff = np.abs(got - want.reshape(got.shape))
                if mode == "fp32":
                    assert (diff > 2e-6).mean() < 1e-2 and diff.max() < 1e-4, (f, net, name, diff.max())
                else:
                    assert diff.max() < 60 * 3e-4 * 1.01 and (diff > 3e-3).mean() < 2e-2, (f, net, name, diff.max())
    for name, t in tr.q_target.state_dict().items():
        a = t.cpu().numpy()
        key = f"{tag}/final/q_target/{name}"
        want, got = (d[key + "#stride37"], a.reshape(-1)[::37]) if key + "#stride37" in d.files else (d[key], a)
        assert np.abs(got - want.reshape(got.shape)).max() < (2e-6 if mode == "fp32" else 60 * 0.005 * 3e-4 * 60)


def test_seeds_per_gpu_with_load_model_warns(tmp_path):
    """K seeds started from ONE checkpoint differ only in their sample streams -- say so."""
    import iqlpref_amd as ia
    S, A, N = 29, 8, 2000
    data = synth(N, S, A)
    cfg = ia.TrainConfig(env="antmaze-medium-diverse-v2", max_timesteps=4, log_freq=2, eval_freq=4, batch_size=32,
                         seed=3, device=DEV, checkpoints_path=str(tmp_path))
    ia.train(cfg, dataset={k: v.copy() for k, v in data.items()}, state_dim=S, action_dim=A, max_action=1.0,
             logger=lambda d, step: None, evaluate=None)
    ck = os.path.join(cfg.checkpoints_path, "checkpoint_3.pt")
    assert os.path.exists(ck)
    cfg2 = ia.TrainConfig(env="antmaze-medium-diverse-v2", max_timesteps=2, log_freq=2, eval_freq=2, batch_size=32,
                          seed=3, device=DEV, load_model=ck)
    with pytest.warns(UserWarning, match="SAME checkpoint"):
        trs = ia.train(cfg2, dataset={k: v.copy() for k, v in data.items()}, state_dim=S, action_dim=A,
                       max_action=1.0, logger=lambda d, step: None, evaluate=None, seeds_per_gpu=2)
    assert [t.total_it for t in trs] == [6, 6]  # both resumed at step 4 and ran 2 more
