"""BASELINE.json's full sizes (1M-transition buffer, batch 256, hidden 256): the oracle cannot
step through these in seconds, so the HIP path is checked through size-independent properties
-- bit-exact gathers against numpy on samples and on the extremes, the Philox index stream,
run-to-run determinism, graph == eager, chunk invariance of the relabel forward, order
statistics of the CVaR tail mean -- plus the oracle itself on the first steps.  -m gpu."""
import numpy as np
import pytest
import torch

from oracle import iql_oracle as orc
from oracle import philox

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
S, A, H, B, N = 29, 8, 256, 256, 1_000_000


@pytest.fixture(scope="module")
def big():
    import bench
    import iqlpref_amd as ia
    data = bench.synth_dataset(7, N)
    buf = ia.ReplayBuffer(S, A, N, DEV)
    buf.load_d4rl_dataset(data)
    return ia, bench, data, buf


def test_full_size_buffer_gather_and_index_stream(big):
    ia, bench, data, buf = big
    assert buf._size == N and buf._pointer == N
    rng = np.random.default_rng(0)
    idx = np.concatenate([[0, N - 1, N - 1, 0], rng.integers(0, N, 4092)]).astype(np.int64)
    got = buf.sample(len(idx), indices=torch.from_numpy(idx).to(DEV))
    for g, w in zip(got, orc.gather_batch(data, idx)):
        np.testing.assert_array_equal(g.cpu().numpy(), w)  # bit exact
    # whole-buffer checksum of checksums: every stored word equals the source (xor of the bit images)
    for view, key in ((buf._states, "observations"), (buf._actions, "actions"), (buf._next_states, "next_observations")):
        dev = view.contiguous().view(torch.int32)
        ref = torch.from_numpy(np.ascontiguousarray(data[key])).view(torch.int32)
        assert int(dev.sum(dtype=torch.int64)) == int(ref.sum(dtype=torch.int64))
    # on-device Philox indices == oracle stream at N = 1M (unbiased range, in bounds)
    torch.manual_seed(123)
    for call in range(3):
        b = buf.sample(B)
        ix = philox.sample_indices(123, call, B, N)
        assert ix.min() >= 0 and ix.max() < N
        np.testing.assert_array_equal(b[0].cpu().numpy(), data["observations"][ix])
        np.testing.assert_array_equal(b[4].cpu().numpy()[:, 0], data["terminals"][ix])


def test_full_size_training_is_deterministic_and_matches_the_oracle_at_the_start(big):
    ia, bench, data, buf = big
    K = 300
    runs = []
    for unroll in (50, 0, 50):
        tr = bench.build_trainer(ia, torch, DEV, 5, "bf16")
        runs.append((tr, tr.train_steps(buf, K, B, graph_unroll=unroll).cpu().numpy()))
    np.testing.assert_array_equal(runs[0][1], runs[2][1])  # same seed, same bits
    np.testing.assert_array_equal(runs[0][1], runs[1][1])  # hipGraph replay == eager launches
    for (k, va), (_, vb) in zip(runs[0][0].qf.state_dict().items(), runs[1][0].qf.state_dict().items()):
        assert torch.equal(va, vb), k
    assert np.isfinite(runs[0][1]).all()
    # the oracle on the same index stream for the first steps (bf16 arithmetic of the reference)
    torch.manual_seed(5)
    q, v, a = ia.TwinQ(S, A), ia.ValueFunction(S), ia.GaussianPolicy(S, A, 1.0)
    sd = lambda m: {k: t.detach().numpy() for k, t in m.state_dict().items()}
    o = orc.IQLOracle(sd(q), sd(v), sd(a), mode="bf16", **bench.HYPER)
    for t in range(3):
        out = o.train(orc.gather_batch(data, philox.sample_indices(5, t, B, N)))
        np.testing.assert_allclose(runs[0][1][t], [out["value_loss"], out["q_loss"], out["actor_loss"]],
                                   rtol=6e-3)  # bf16 tolerance of tests/test_gpu_step.py


def test_full_size_relabel_properties(big):
    ia, bench, data, buf = big
    from iqlpref_amd.relabel import cvar_tail_mean_device
    torch.manual_seed(0)
    x = torch.cat([buf._states, buf._actions], 1)  # [1M, 37]
    ws = [torch.randn(37, 256, device=DEV) * 0.1, torch.randn(256, 256, device=DEV) * 0.05,
          torch.randn(256, 1, device=DEV) * 0.05]
    bs = [torch.randn(256, device=DEV) * 0.1, torch.randn(256, device=DEV) * 0.1, torch.randn(1, device=DEV)]
    whole = ia.mlp_forward_f32(ws, bs, x, w_in_out=True)
    assert whole.shape == (N, 1) and torch.isfinite(whole).all()
    # chunk invariance: any split of the rows gives the same bits
    parts = torch.cat([ia.mlp_forward_f32(ws, bs, x[a:b], w_in_out=True)
                       for a, b in ((0, 1), (1, 4097), (4097, 600_001), (600_001, N))])
    assert torch.equal(whole, parts)
    # a sample of rows against fp64 numpy
    rows = np.random.default_rng(1).integers(0, N, 512)
    h = x[rows].double().cpu().numpy()
    for i, (w, b) in enumerate(zip(ws, bs)):
        h = h @ w.double().cpu().numpy() + b.double().cpu().numpy()
        if i < 2:
            h = np.maximum(h, 0)
    np.testing.assert_allclose(whole[rows].cpu().numpy(), h, rtol=2e-5, atol=2e-6)
    # CVaR tail mean over a [20, 1M] prediction matrix: order statistics
    Sn = 20
    preds = torch.randn(Sn, N, device=DEV)
    preds[3] = preds[7]  # ties
    full = cvar_tail_mean_device(preds, Sn)
    np.testing.assert_allclose(full.cpu().numpy(), preds.mean(0).cpu().numpy(), rtol=1e-5, atol=1e-6)
    one = cvar_tail_mean_device(preds, 1)
    assert torch.equal(one, preds.min(0).values)  # n_tail = 1: the minimum, exactly
    prev = one
    for n_tail in (2, 5, 19):
        cur = cvar_tail_mean_device(preds, n_tail)
        assert (cur >= prev - 1e-6).all()  # the tail mean grows with the tail
        prev = cur
    srt = preds[:, :4096].sort(0).values[:5].mean(0)
    np.testing.assert_allclose(cvar_tail_mean_device(preds, 5)[:4096].cpu().numpy(), srt.cpu().numpy(),
                               rtol=1e-5, atol=1e-6)


def test_full_length_run_of_a_million_updates():
    """BASELINE run length: 1,000,000 updates (configs/offline/iql/antmaze/medium_diverse_v2.yaml:9) at the
    headline shapes -- ~15 s on an MI355X.  The run stays finite, counts its steps, and ends the actor's
    cosine schedule where torch's CosineAnnealingLR does (lr -> 0 at T_max, ref:571,637); a checkpoint
    taken at the end holds the step count in every Adam state entry."""
    import iqlpref_amd as ia
    S, A, N = 29, 8, 1_000_000
    rng = np.random.default_rng(0)
    data = {"observations": rng.standard_normal((N, S), dtype=np.float32),
            "actions": rng.uniform(-1, 1, (N, A)).astype(np.float32),
            "rewards": (rng.uniform(size=N) < 0.01).astype(np.float32) - 1.0,
            "next_observations": rng.standard_normal((N, S), dtype=np.float32),
            "terminals": (rng.uniform(size=N) < 1e-3).astype(np.float32)}
    buf = ia.ReplayBuffer(S, A, N, DEV)
    buf.load_d4rl_dataset(data)
    torch.manual_seed(0)
    q, v, actor = ia.TwinQ(S, A).to(DEV), ia.ValueFunction(S).to(DEV), ia.GaussianPolicy(S, A, 1.0).to(DEV)
    T = 1_000_000
    tr = ia.ImplicitQLearning(
        max_action=1.0, actor=actor, actor_optimizer=torch.optim.Adam(actor.parameters(), lr=3e-4), q_network=q,
        q_optimizer=torch.optim.Adam(q.parameters(), lr=3e-4), v_network=v,
        v_optimizer=torch.optim.Adam(v.parameters(), lr=3e-4), iql_tau=0.9, beta=10.0, max_steps=T, device=DEV, seed=0)
    import time
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(T // 50_000 - 1):
        tr.train_steps(buf, 50_000, 256, return_losses=False)
    last = tr.train_steps(buf, 50_000, 256).cpu().numpy()
    dt = time.perf_counter() - t0
    assert tr.total_it == T and np.isfinite(last).all()
    assert torch.isfinite(tr._params).all() and torch.isfinite(tr._exp_avg_sq).all() and torch.isfinite(tr._target).all()
    # the schedule: lr(T_max) = 0, and the last step ran at lr(T_max - 1) > 0 (the scheduler steps AFTER the optimiser)
    assert abs(tr.actor_optimizer.param_groups[0]["lr"]) < 1e-18
    assert tr.actor_lr_schedule.last_epoch == T
    sd = tr.state_dict()
    assert sd["total_it"] == T and all(float(st["step"]) == T for st in sd["q_optimizer"]["state"].values())
    # value / q losses of a fitted run are small and positive; the critic's target has moved with it
    assert 0 < last[:, 0].mean() < 1.0 and 0 < last[:, 1].mean() < 10.0
    assert not torch.equal(tr._target, tr._params[:tr._n_target])  # Polyak average, not a copy
    assert dt < 120, f"1M updates took {dt:.0f} s"


def test_general_step_runs_a_full_schedule_to_the_end():
    """The general layer-wise step (three hidden layers of 96 units with actor dropout: a shape the tuned step does not
    take) through a whole cosine schedule of 50,000 updates: finite state, step counts, lr -> 0 at T_max, a critic
    that fits (losses small and positive), a Polyak-averaged target."""
    import iqlpref_amd as ia
    S, A, N, T = 17, 6, 200_000, 50_000
    rng = np.random.default_rng(1)
    data = {"observations": rng.standard_normal((N, S), dtype=np.float32),
            "actions": rng.uniform(-1, 1, (N, A)).astype(np.float32),
            "rewards": rng.standard_normal(N).astype(np.float32) * 0.1,
            "next_observations": rng.standard_normal((N, S), dtype=np.float32),
            "terminals": (rng.uniform(size=N) < 1e-2).astype(np.float32)}
    buf = ia.ReplayBuffer(S, A, N, DEV)
    buf.load_d4rl_dataset(data)
    torch.manual_seed(1)
    kw = dict(hidden_dim=96, n_hidden=3)
    q, v = ia.TwinQ(S, A, **kw).to(DEV), ia.ValueFunction(S, **kw).to(DEV)
    actor = ia.GaussianPolicy(S, A, 1.0, dropout=0.1, **kw).to(DEV)
    tr = ia.ImplicitQLearning(
        max_action=1.0, actor=actor, actor_optimizer=torch.optim.Adam(actor.parameters(), lr=3e-4), q_network=q,
        q_optimizer=torch.optim.Adam(q.parameters(), lr=3e-4), v_network=v,
        v_optimizer=torch.optim.Adam(v.parameters(), lr=3e-4), iql_tau=0.7, beta=3.0, max_steps=T, device=DEV, seed=1)
    assert tr.step_kind(256) == "general"
    first = tr.train_steps(buf, 1_000, 256).cpu().numpy()
    tr.train_steps(buf, T - 2_000, 256, return_losses=False)
    last = tr.train_steps(buf, 1_000, 256).cpu().numpy()
    assert tr.total_it == T and np.isfinite(first).all() and np.isfinite(last).all()
    assert torch.isfinite(tr._params).all() and torch.isfinite(tr._exp_avg_sq).all() and torch.isfinite(tr._target).all()
    assert abs(tr.actor_optimizer.param_groups[0]["lr"]) < 1e-18 and tr.actor_lr_schedule.last_epoch == T
    sd = tr.state_dict()
    assert sd["total_it"] == T and all(float(st["step"]) == T for st in sd["actor_optimizer"]["state"].values())
    # the critic settles at the noise floor of its TD targets (rewards are N(0, 0.1^2): 0.01), below where it started
    assert 0 < last[:, 1].mean() < first[:10, 1].mean() and last[:, 1].mean() < 0.02 and 0 < last[:, 0].mean() < 1.0
    assert not torch.equal(tr._target, tr._params[:tr._n_target])
