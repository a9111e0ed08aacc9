"""GPU parity of the relabel kernels (reward MLP forward, ensemble CVaR, preference
transformer) against the oracle and the reference goldens.  -m gpu."""
import os

import numpy as np
import pytest
import torch

from oracle import relabel_oracle as ro
from tests import helpers
from tests.test_relabel_oracle import g5_dataset, mlp_weights

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class FakeEnv:
    def __init__(self, m):
        self._max_episode_steps = m


@pytest.fixture(scope="module")
def g():
    return np.load(helpers.GOLDEN + "/dataset_ops.npz")


def make_reward_mlp(weights, activation="relu"):
    import iqlpref_amd as ia
    n = len(weights) // 2
    dims = [weights[0].shape[0]] + [weights[2 * l].shape[1] for l in range(n)]
    m = ia.RewardMLP(dims[0], dims[-1], dims[1:-1], activation)
    ws, bs = m.wb()
    with torch.no_grad():
        for l in range(n):
            ws[l].copy_(torch.from_numpy(weights[2 * l]))
            bs[l].copy_(torch.from_numpy(weights[2 * l + 1]))
    return m.to(DEV)


def test_mlp_forward_kernel_shapes():
    """iqlhip_mlp_forward: both weight layouts, odd widths, row counts around the 64-row tile."""
    import iqlpref_amd as ia
    rng = np.random.default_rng(0)
    for dims, w_in_out, act in (([37, 256, 256, 1], True, "relu"), ([6, 8, 8, 1], True, "relu"),
                                ([45, 256, 256, 24], False, "relu"), ([13, 100, 7], True, "tanh"),
                                # widths beyond 256: the one-wave-per-16-rows variant (k_mlp_wide)
                                ([37, 384, 384, 1], False, "relu"), ([29, 1000, 8], True, "tanh"),
                                ([300, 17, 1024, 33, 5], False, "relu")):
        ws = [rng.standard_normal((dims[i], dims[i + 1])).astype(np.float32) / np.sqrt(dims[i])
              for i in range(len(dims) - 1)]
        bs = [rng.standard_normal(dims[i + 1]).astype(np.float32) * 0.1 for i in range(len(dims) - 1)]
        flat = [x for pair in zip(ws, bs) for x in pair]
        for n in (1, 63, 64, 65, 1000):
            x = rng.standard_normal((n, dims[0])).astype(np.float32)
            want = ro.reward_mlp_forward(flat, x, act)
            tw = [torch.from_numpy(w if w_in_out else np.ascontiguousarray(w.T)).to(DEV) for w in ws]
            tb = [torch.from_numpy(b).to(DEV) for b in bs]
            got = ia.mlp_forward_f32(tw, tb, torch.from_numpy(x).to(DEV), w_in_out=w_in_out,
                                     hidden_act=0 if act == "relu" else 1).cpu().numpy()
            np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("use_to", [True, False])
@pytest.mark.parametrize("toe", [False, True])
def test_qlearning_dataset_mr_matches_reference(g, use_to, toe):
    import iqlpref_amd as ia
    ds = g5_dataset(g, use_to)
    rm = make_reward_mlp(mlp_weights(g, "g5/rm/"))
    out = ia.qlearning_dataset_mr(FakeEnv(15), rm, dataset=ds, terminate_on_end=toe)
    tag = f"g5/mr/timeouts{int(use_to)}_toe{int(toe)}"
    assert set(out) == {"observations", "actions", "next_observations", "rewards", "terminals"}
    for k, v in out.items():
        np.testing.assert_allclose(np.asarray(v, dtype=np.float32), g[f"{tag}/{k}"], rtol=2e-5, atol=2e-5)


def test_cvar_kernel_vs_partition():
    import iqlpref_amd as ia
    from iqlpref_amd.relabel import cvar_tail_mean_device
    rng = np.random.default_rng(1)
    for S, N in ((1, 10), (10, 33), (37, 1000), (500, 257), (700, 65)):
        preds = rng.standard_normal((S, N)).astype(np.float32)
        preds[:, : N // 3] = np.round(preds[:, : N // 3], 1)  # ties
        if S > 2:
            preds[1, 0] = preds[0, 0]
        preds[:, N - 1] = 0.25           # one column of S equal values
        preds[: S // 2, N - 2] = -3.0     # and one whose smaller half is one value
        for alpha in (0.0, 0.5, 0.9, 0.95):
            n_tail = ro.n_tail_of(alpha, S)
            got = cvar_tail_mean_device(torch.from_numpy(preds).to(DEV), n_tail).cpu().numpy()
            want = ro.cvar_tail_mean(preds, alpha)
            np.testing.assert_allclose(got, want, rtol=2e-6, atol=2e-6, err_msg=f"S={S} N={N} a={alpha}")


def _write_snapshots(tmp_path, g):
    (tmp_path / "config.yaml").write_text("activations: relu\n")
    names = ["layers.0", "layers.linear_1", "out"]
    for ep in range(6):
        w = mlp_weights(g, f"g5/ens/snap{ep}/")
        sd = {}
        for l, nme in enumerate(names):
            sd[f"{nme}.W"] = torch.from_numpy(w[2 * l])
            sd[f"{nme}.b"] = torch.from_numpy(w[2 * l + 1])
        if ep % 2:
            sd = {"_orig_mod." + k: v for k, v in sd.items()}
        torch.save({"net": sd}, tmp_path / f"checkpoint_{ep}.pt")
    torch.save({"net": sd}, tmp_path / "best_model.pt")


@pytest.mark.parametrize("alpha,burn", [(0.0, 0), (0.5, 1), (0.9, 2)])
def test_mr_ensemble_matches_reference(g, tmp_path, alpha, burn):
    import warnings
    import iqlpref_amd as ia
    _write_snapshots(tmp_path, g)
    ds = g5_dataset(g)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out = ia.qlearning_dataset_mr_ensemble(FakeEnv(15), str(tmp_path), alpha=alpha, burn_in=burn,
                                               device=DEV, dataset=ds)
    tag = f"g5/ens/alpha{alpha}_burn{burn}"
    for k, v in out.items():
        np.testing.assert_allclose(np.asarray(v, dtype=np.float32), g[f"{tag}/{k}"], rtol=2e-5, atol=2e-5)
    # loader + error paths of the reference
    m = ia.load_mlp_reward_model(str(tmp_path), DEV)
    assert isinstance(m, ia.RewardMLP)
    with pytest.raises(ValueError):
        ia.qlearning_dataset_mr_ensemble(FakeEnv(15), str(tmp_path), alpha=1.0, device=DEV, dataset=ds)
    with pytest.raises(ValueError):
        ia.qlearning_dataset_mr_ensemble(FakeEnv(15), str(tmp_path), alpha=0.5, burn_in=99, device=DEV, dataset=ds)
    with pytest.raises(FileNotFoundError):
        ia.qlearning_dataset_mr_ensemble(FakeEnv(15), str(tmp_path / "nope"), device=DEV, dataset=ds)


@pytest.mark.parametrize("as_numpy", [True, False])
@pytest.mark.parametrize("alpha,ns", [(0.5, 500), (0.75, 5), (0.0, 0)])
def test_bnn_matches_reference(g, tmp_path, alpha, ns, as_numpy):
    """as_numpy: the posterior files hold lists of NUMPY arrays -- the form the reference's sampler
    writes and ref:913-915 reads (and the form tests/golden/make_fixtures.py gave the reference);
    otherwise lists of tensors."""
    import warnings
    import iqlpref_amd as ia
    all_w = [[g[f"g5/bnn/w{i}/{j}"] for j in range(6)] for i in range(8)]
    for c in range(2):
        cdir = tmp_path / "sampling_f" / f"chain_{c}" / "sampled_weights"
        os.makedirs(cdir)
        conv = (lambda a: np.array(a)) if as_numpy else torch.from_numpy
        torch.save({"sampled_weights": [[conv(a) for a in w] for w in all_w[4 * c:4 * c + 4]]},
                   cdir / "sampled_weights_0000000")
    ds = g5_dataset(g)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out = ia.qlearning_dataset_bnn(FakeEnv(15), str(tmp_path), alpha=alpha, n_samples=ns, device=DEV,
                                       dataset=ds)
    tag = f"g5/bnn/alpha{alpha}_n{ns}"
    for k, v in out.items():
        np.testing.assert_allclose(np.asarray(v, dtype=np.float32), g[f"{tag}/{k}"], rtol=2e-5, atol=2e-5)
    with pytest.raises(FileNotFoundError):
        ia.qlearning_dataset_bnn(FakeEnv(15), str(tmp_path / "nope"), device=DEV, dataset=ds)


def make_pt(p, S, A, max_ep, heads, inter):
    import iqlpref_amd as ia
    pref = (p["pref_linear.weight"].shape[0] - 1) // 2
    m = ia.RewardPT(S, A, max_ep, embd_dim=64, pref_attn_embd_dim=pref, num_heads=heads,
                    intermediate_dim=inter, num_layers=1, max_pos=64)
    sd = {k: torch.from_numpy(v) for k, v in p.items()}
    missing = m.load_state_dict(sd, strict=False)
    assert all(k.endswith("causal_bias") for k in missing.missing_keys) and not missing.unexpected_keys
    return m.to(DEV)


@pytest.mark.parametrize("S,A,QL,heads", [(5, 3, 6, 4), (45, 24, 20, 4), (29, 8, 100, 2)])
def test_pt_kernel_vs_oracle(S, A, QL, heads):
    """PT numerics are PARITY UNPINNED (no runnable reference): HIP vs the numpy
    restatement of reward_models/pref_transformer.py."""
    rng = np.random.default_rng(S)
    max_ep = 150
    p = ro.make_pt_params(rng, S, A, max_ep, embd=64, pref=8, inter=256, layers=1)
    m = make_pt(p, S, A, max_ep, heads, 256)
    n_rows = 300
    obs = rng.standard_normal((n_rows, S)).astype(np.float32)
    act = rng.uniform(-1, 1, (n_rows, A)).astype(np.float32)
    lens = np.array([1, 2, QL // 2, QL - 1, QL, QL, QL, 3], dtype=np.int32)
    starts = np.array([0, 5, 17, 40, 0, 100, n_rows - QL, 7], dtype=np.int64)
    got = m.window_values(torch.from_numpy(obs).to(DEV), torch.from_numpy(act).to(DEV),
                          torch.from_numpy(starts).to(DEV), torch.from_numpy(lens).to(DEV), QL).cpu().numpy()
    want = np.zeros(len(lens), np.float32)
    for i, (st, ln) in enumerate(zip(starts, lens)):
        sts = np.zeros((1, QL, S), np.float32); acs = np.zeros((1, QL, A), np.float32)
        ts = np.zeros((1, QL), np.int64); am = np.zeros((1, QL), np.float32)
        sts[0, QL - ln:] = obs[st:st + ln]; acs[0, QL - ln:] = act[st:st + ln]
        ts[0, QL - ln:] = np.arange(ln); am[0, QL - ln:] = 1
        want[i] = ro.pt_value_last(p, sts, acs, ts, am, num_heads=heads)[0]
    # bf16 q.k products: a rounding flip moves a logit by one bf16 ulp
    np.testing.assert_allclose(got, want, rtol=5e-3, atol=5e-3)


@pytest.mark.parametrize("heads,inter,QL", [(8, 512, 12), (1, 256, 20), (16, 256, 7)])
def test_pt_kernel_many_windows_per_workgroup(heads, inter, QL):
    """A persistent work-group walks several windows and parks their last tokens in slots of 8
    (full batches, then a partial one at the end of its queue): every window's value must be the
    one it gets when it is the only window of its work-group, bit for bit, and match the oracle.
    Covers 1 / 8 / 16 heads (the block-masked query rows) and a wider MLP.  PARITY UNPINNED."""
    S, A, max_ep = 7, 3, 64
    rng = np.random.default_rng(heads)
    p = ro.make_pt_params(rng, S, A, max_ep, embd=64, pref=8, inter=inter, layers=1)
    m = make_pt(p, S, A, max_ep, heads, inter)
    n_rows = 4000
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    n_win = cus * 9 + 5  # nine or ten windows per work-group: one full batch of 8 and a partial one
    obs = rng.standard_normal((n_rows, S)).astype(np.float32)
    act = rng.uniform(-1, 1, (n_rows, A)).astype(np.float32)
    lens = rng.integers(1, QL + 1, n_win).astype(np.int32)
    starts = rng.integers(0, n_rows - QL, n_win).astype(np.int64)
    t0 = rng.integers(0, max_ep - QL, n_win).astype(np.int32)
    dv = lambda x: torch.from_numpy(x).to(DEV)
    o, a_, st, ln, tt = dv(obs), dv(act), dv(starts), dv(lens), dv(t0)
    got = m.window_values(o, a_, st, ln, QL, win_t0=tt).cpu().numpy()
    assert np.isfinite(got).all()
    alone = np.concatenate([m.window_values(o, a_, st[i:i + 64], ln[i:i + 64], QL, win_t0=tt[i:i + 64]).cpu().numpy()
                            for i in range(0, n_win, 64)])  # <= 64 windows per call: one per work-group
    np.testing.assert_array_equal(got, alone)
    for i in rng.choice(n_win, 24, replace=False):
        s0, l0 = int(starts[i]), int(lens[i])
        sts = np.zeros((1, QL, S), np.float32); acs = np.zeros((1, QL, A), np.float32)
        ts = np.zeros((1, QL), np.int64); am = np.zeros((1, QL), np.float32)
        sts[0, QL - l0:] = obs[s0:s0 + l0]; acs[0, QL - l0:] = act[s0:s0 + l0]
        ts[0, QL - l0:] = int(t0[i]) + np.arange(l0); am[0, QL - l0:] = 1
        want = ro.pt_value_last(p, sts, acs, ts, am, num_heads=heads)[0]
        np.testing.assert_allclose(got[i], want, rtol=5e-3, atol=5e-3)


@pytest.mark.parametrize("correct", [False, True])
def test_qlearning_dataset_pt_vs_oracle(g, correct):
    import iqlpref_amd as ia
    ds = g5_dataset(g)
    S, A, QL = 4, 2, 5
    rng = np.random.default_rng(9)
    p = ro.make_pt_params(rng, S, A, 20, embd=64, pref=8, inter=256, layers=1)
    m = make_pt(p, S, A, 20, 4, 256)
    out = ia.qlearning_dataset_pt(FakeEnv(15), m, query_length=QL, dataset=ds, correct_window_offsets=correct)
    want = ro.qlearning_dataset_pt(ds, p, 15, QL, num_heads=4, correct_window_offsets=correct)
    for k in want:
        np.testing.assert_allclose(np.asarray(out[k], dtype=np.float32), np.asarray(want[k], dtype=np.float32),
                                   rtol=5e-3, atol=5e-3, err_msg=k)
    if not correct:  # kept rows / terminals exactly as the reference returned them (fake r_model golden)
        np.testing.assert_array_equal(out["observations"], g["g5/pt/ql5/observations"])
        np.testing.assert_array_equal(np.asarray(out["terminals"], np.float32), g["g5/pt/ql5/terminals"])


def test_bnn_posterior_at_its_stated_size(tmp_path):
    """500 posterior weight sets (bnn_n_samples = 500, the reference's default) in the
    on-disk form of ref:905-915 -- chain files holding lists of NUMPY arrays -- through
    qlearning_dataset_bnn, against the oracle's loop (ref:978-1011).  N is kept small here; the
    500 x 1M case is a bench leg (tools/bench_relabel.py)."""
    import warnings
    import iqlpref_amd as ia
    rng = np.random.default_rng(4)
    S_, A_, N, NS = 11, 3, 3001, 520
    ws = [[(rng.standard_normal(sh) * 0.4).astype(np.float32)
           for sh in ((S_ + A_, 32), (32,), (32, 32), (32,), (32, 1), (1,))] for _ in range(NS)]
    for c, part in enumerate((ws[:300], ws[300:])):
        cdir = tmp_path / "sampling_f" / f"chain_{c}" / "sampled_weights"
        os.makedirs(cdir)
        torch.save({"sampled_weights": part}, cdir / "sampled_weights_0000000")
    ds = {"observations": rng.standard_normal((N, S_)).astype(np.float32),
          "actions": rng.uniform(-1, 1, (N, A_)).astype(np.float32),
          "rewards": np.zeros(N, np.float32), "terminals": rng.uniform(size=N) < 0.01,
          "timeouts": np.zeros(N, bool)}
    ds["timeouts"][199::200] = True
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out = ia.qlearning_dataset_bnn(FakeEnv(200), str(tmp_path), alpha=0.95, n_samples=500, device=DEV,
                                       dataset=dict(ds))
    # the reference's subsample of the 520 available sets (ref:929-932)
    pick = sorted(np.random.default_rng(seed=0).choice(NS, size=500, replace=False))
    want, _ = ro.qlearning_dataset_ensemble(dict(ds), [ws[i] for i in pick], 0.95, 200)
    assert out["rewards"].shape == want["rewards"].shape
    for k in want:
        np.testing.assert_allclose(np.asarray(out[k], np.float32), np.asarray(want[k], np.float32),
                                   rtol=2e-5, atol=2e-5, err_msg=k)
