"""GPU parity of the HIP path (through the C-ABI) against the oracle and the
golden vectors captured from the reference.  Run with -m gpu on an MI355X."""
import os

import numpy as np
import pytest
import torch

from oracle import iql_oracle as orc
from oracle import philox
from tests import helpers

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gh():
    from tests import gpu_helpers
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return gpu_helpers


def test_replay_pack_views_and_sample(gh):
    d, hyper, data, nets = helpers.load_traj("traj_antmaze", "fp32")
    buf = gh.make_buffer(hyper, data)
    n = hyper["n_rows"]
    assert buf._size == n and buf._pointer == n
    np.testing.assert_array_equal(buf._states.cpu().numpy(), data["observations"])
    np.testing.assert_array_equal(buf._actions.cpu().numpy(), data["actions"])
    np.testing.assert_array_equal(buf._rewards.cpu().numpy()[:, 0], data["rewards"])
    np.testing.assert_array_equal(buf._next_states.cpu().numpy(), data["next_observations"])
    np.testing.assert_array_equal(buf._dones.cpu().numpy()[:, 0], data["terminals"])
    idx = torch.from_numpy(d["indices"][0]).to(gh.DEV)
    got = buf.sample(len(idx), indices=idx)
    want = orc.gather_batch(data, d["indices"][0])
    for g, w in zip(got, want):
        assert g.shape == w.shape
        np.testing.assert_array_equal(g.cpu().numpy(), w)  # bit exact gather
    with pytest.raises(ValueError):
        buf.load_d4rl_dataset(data)
    with pytest.raises(NotImplementedError):
        buf.add_transition()
    # on-device Philox index stream == oracle/philox.py
    torch.manual_seed(99)
    b1 = buf.sample(64)
    b2 = buf.sample(64)
    for call, b in enumerate((b1, b2)):
        ix = philox.sample_indices(99, call, 64, n)
        np.testing.assert_array_equal(b[0].cpu().numpy(), data["observations"][ix])
        np.testing.assert_array_equal(b[2].cpu().numpy()[:, 0], data["rewards"][ix])


def test_oversize_dataset_rejected(gh):
    import iqlpref_amd as ia
    buf = ia.ReplayBuffer(3, 2, 5, gh.DEV)
    data = {"observations": np.zeros((6, 3), np.float32), "actions": np.zeros((6, 2), np.float32),
            "rewards": np.zeros(6, np.float32), "next_observations": np.zeros((6, 3), np.float32),
            "terminals": np.zeros(6, np.float32)}
    with pytest.raises(ValueError):
        buf.load_d4rl_dataset(data)
    with pytest.raises(ValueError):
        buf.sample(4)  # empty buffer


def _diag(line):
    """IQL_TEST_DIAG=<file>: append measured error figures (used to set the bounds in this file)."""
    path = os.environ.get("IQL_TEST_DIAG")
    if path:
        with open(path, "a") as f:
            f.write(line + "\n")


def _drop_tensor(d, hyper, gh):
    if hyper["dropout"] is None:
        return None
    m = np.unpackbits(d["dropout_keep"], axis=-1)[..., :hyper["hidden"]]
    return torch.from_numpy(np.ascontiguousarray(m)).to(gh.DEV)


TOL = {
    # (loss rtol vs oracle, loss rtol vs reference golden, param atol vs oracle, vs golden)
    "fp32": dict(lo=2e-5, lg=2e-5, po=2e-6, pg=2e-6),
    # bf16: an absolute parameter bound of the size of the K-step movement (K * lr = 3e-3) would pin
    # nothing; what pins the bf16 update is the MOVEMENT final - initial of every tensor against the
    # oracle's / the reference's, as a relative L2 error (BF16_DELTA_*), and the Adam moments below.
    "bf16": dict(lo=5e-3, lg=2e-2),
}
# Bounds = ~5 x the largest figure measured on an MI355X (IQL_TEST_DIAG; gpurun_out/r3v/diag.txt and
# this round's run): movement vs the oracle <= 0.0004, vs the reference's golden <= 0.0016.
BF16_DELTA_ORACLE = 0.005
BF16_DELTA_GOLDEN = 0.01
# three / four hidden layers (the general step, TRAJ_SHAPES) AGAINST THE ORACLE: one bf16 rounding tie in a
# gradient entry of step 1 that falls the other way under numpy's summation order, amplified by Adam's sign-like
# early steps -- the oracle itself leaves the reference's trajectory there (traj_deep3_w96: q_loss 6.5e-6 off at
# step 2, 4.7e-4 at step 8; exact at step 1), while the HIP step stays on it (losses 1.1e-7, movement of every
# tensor 0.0000 relative L2 from the reference's; gpurun_out/deep4_diag.txt).  Bounds = 5 x measured against the
# oracle (movement 0.0125, moments 1.2e-2); against the reference's own arrays the ordinary bounds hold.
BF16_DELTA_DEEP = 0.06
MOMENT_TOL_DEEP = (0.06, 0.06)
# Adam moments, largest error relative to the tensor's largest entry: (fp32, bf16) -- the bf16 figures
# are ~5 x the measured ones; at H = 256 one relu'(z) falling the other way under another summation
# order moves isolated entries by a sample's share (see tests/test_oracle_golden.py), so there the
# bound holds for all but 0.5 % of a tensor's entries and a looser one for every entry.
MOMENT_TOL = {"fp32": (1e-4, 1e-4), "bf16": (5e-3, 1e-2)}


def _delta_rel(final, init, want_final):
    got, want = (final - init).reshape(-1).astype(np.float64), (want_final - init).reshape(-1).astype(np.float64)
    return float(np.linalg.norm(got - want) / (np.linalg.norm(want) + 1e-30))


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
@pytest.mark.parametrize("name", helpers.TRAJ + helpers.TRAJ_BIG + helpers.TRAJ_SHAPES)
def test_trajectory_parity(gh, name, mode):
    """K steps with the reference's injected indices (and dropout masks) vs the oracle AND the
    reference's own trajectory.  TRAJ_BIG = BASELINE configs 1 / 3 and config 5's batch at H = 256."""
    d, hyper, data, nets = helpers.load_traj(name, mode)
    K, B = hyper["k_steps"], hyper["batch"]
    tr = gh.make_trainer(hyper, nets, mode)
    if not os.environ.get("IQLHIP_FORCE_GENERAL"):  # (set: every shape through the general step)
        assert tr.step_kind(B) == ("general" if name in helpers.TRAJ_SHAPES else "tuned")
    buf = gh.make_buffer(hyper, data)
    idx = torch.from_numpy(d["indices"]).to(gh.DEV)
    losses = tr.train_steps(buf, K, B, indices=idx, dropout_keep=_drop_tensor(d, hyper, gh),
                            graph_unroll=0).cpu().numpy()
    o = helpers.make_oracle(hyper, nets, mode)
    want = np.zeros((K, 3))
    margin = np.inf  # closest any pre-activation / advantage of any step came to a kink
    for t in range(K):
        out = o.train(orc.gather_batch(data, d["indices"][t]), helpers.keep_masks(d, hyper, t))
        want[t] = [out["value_loss"], out["q_loss"], out["actor_loss"]]
        margin = min(margin, min(o.last_margin.values()))
    tol = TOL[mode]
    _diag(f"{name} {mode} loss rel vs oracle {np.abs(losses / want - 1).max():.2e} vs golden "
          f"{np.abs(losses / d['losses'] - 1).max():.2e}")
    np.testing.assert_allclose(losses, want, rtol=tol["lo"], err_msg="vs oracle")
    np.testing.assert_allclose(losses, d["losses"], rtol=tol["lg"], err_msg="vs reference golden")
    assert tr.total_it == K
    assert abs(tr.actor_optimizer.param_groups[0]["lr"] - float(d["final_actor_lr"])) < 1e-12
    big = hyper["hidden"] >= 256
    for net, mod, opar in (("qf", tr.qf, o.qf), ("vf", tr.vf, o.vf), ("actor", tr.actor, o.actor),
                           ("q_target", tr.q_target, o.q_target)):
        got = gh.module_params(mod)
        init = nets[{"qf": 0, "vf": 1, "actor": 2, "q_target": 0}[net]]
        for k, v in got.items():
            wantg, gotg = helpers.golden_param(d, f"final/{net}/{k}", v)
            assert wantg is not None
            wantg = wantg.reshape(gotg.shape)
            if mode == "fp32" and not big and name in helpers.TRAJ_SHAPES:
                # as below for H = 256: a gradient entry that is pure summation noise (dropout zeroes whole
                # units) takes sign-like Adam steps -- traj_shallow1_w40_drop: 1 of 2,760 entries of a critic's
                # first layer, 7e-6 from the reference's; the oracle shows the same entry 5e-6 away
                # (test_oracle_golden.py).  All but 0.1 % of a tensor within the bound, none beyond K lr.
                for what, a, b in (("oracle", v, opar[k]), ("golden", gotg, wantg)):
                    diff = np.abs(a - b)
                    assert (diff > tol["po"]).mean() < 1e-3 and diff.max() < K * 3e-4 * 1.01, (net, k, what, diff.max())
                continue
            if mode == "fp32" and not big:
                np.testing.assert_allclose(v, opar[k], atol=tol["po"], rtol=0, err_msg=f"{net}/{k} vs oracle")
                np.testing.assert_allclose(gotg, wantg, atol=tol["pg"], rtol=0, err_msg=f"{net}/{k} vs golden")
                continue
            if mode == "fp32":
                # H = 256: Adam's first steps are sign-like (lr g / (|g| + eps)), so summation-order
                # noise on an eps-sized gradient entry shows up in its parameter (measured: 1 of 65,536
                # entries of a hidden layer, 10 of 4,352 of a first layer, <= 2.8e-6 away): all but 1 %
                # of a tensor within the bound, none beyond 5 x it
                # ... unless a pre-activation came within fp32 noise of a ReLU kink (the oracle reports the
                # margin): two fp32 implementations can then fall on different sides of relu'(z) for ONE
                # (sample, unit) pair in ONE step, which moves that unit's row of gradients by the sample's
                # share and, through the sign-like steps, its parameters by up to lr per step (measured on
                # traj_cheetah_h256: 140 of 65,536 entries of one matrix, <= 3.5e-5)
                cap = 5 * tol["po"] if margin > 1e-5 else K * 3e-4 * 1.01
                for what, a, b in (("oracle", v, opar[k]), ("golden", gotg, wantg)):
                    diff = np.abs(a - b)
                    _diag(f"{name} {net}/{k} fp32 max abs vs {what} {diff.max():.2e} frac {(diff > tol['po']).mean():.1e} "
                          f"(kink margin {margin:.1e})")
                    assert (diff > tol["po"]).mean() < 1e-2 and diff.max() < cap, (net, k, what, diff.max(), margin)
                continue
            rel = _delta_rel(v, np.asarray(init[k]), opar[k])
            _diag(f"{name} {net}/{k} delta rel vs oracle {rel:.4f}")
            deep = name in helpers.TRAJ_SHAPES and hyper["n_hidden"] > 2
            assert rel < (BF16_DELTA_DEEP if deep else BF16_DELTA_ORACLE), \
                f"{net}/{k}: movement differs from the oracle's by {rel:.4f} (rel. L2)"
            # (large tensors are stored as every 37th element: the movement of that sample)
            initg = np.asarray(init[k]).reshape(-1)[::37] if wantg.size != v.size else np.asarray(init[k])
            relg = _delta_rel(gotg, initg.reshape(gotg.shape), wantg)
            _diag(f"{name} {net}/{k} delta rel vs golden {relg:.4f}")
            assert relg < BF16_DELTA_GOLDEN, f"{net}/{k}: movement vs the reference {relg:.4f}"
    # Adam moments (exp_avg = EMA of the gradients: pins the backward pass)
    t1, t2 = MOMENT_TOL[mode]
    if mode == "bf16" and name in helpers.TRAJ_SHAPES and hyper["n_hidden"] > 2:
        t1, t2 = MOMENT_TOL_DEEP
    for which, opt, mod in (("q", tr.q_optimizer, tr.qf), ("v", tr.v_optimizer, tr.vf),
                            ("actor", tr.actor_optimizer, tr.actor)):
        for (pname, p) in mod.named_parameters():
            st = opt.state[p]
            assert float(st["step"]) == K
            m_want, v_want = o.m[which][pname], o.v2[which][pname]
            err = np.abs(st["exp_avg"].cpu().numpy() - m_want) / (np.abs(m_want).max() + 1e-30)
            errv = np.abs(st["exp_avg_sq"].cpu().numpy() - v_want) / (np.abs(v_want).max() + 1e-30)
            _diag(f"{name} {mode} moments {which}/{pname}: exp_avg {err.max():.2e} exp_avg_sq {errv.max():.2e} "
                  f"outliers {(err > t1).mean():.1e} {(errv > t2).mean():.1e}")
            if big and (mode == "bf16" or margin < 1e-5):
                few = max(4, int(5e-3 * err.size))  # (a flipped relu'(z) touches one unit's row and its bias)
                assert (err > t1).sum() <= few and err.max() < 0.25, (which, pname, err.max())
                assert (errv > t2).sum() <= few and errv.max() < 0.25, (which, pname, errv.max())
            else:
                assert err.max() < t1 and errv.max() < t2, (which, pname, err.max(), errv.max())


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_train_batch_equals_fused_and_graph(gh, mode):
    """train(batch) (explicit batch), train_steps eager and hipGraph replay agree bit for bit."""
    d, hyper, data, nets = helpers.load_traj("traj_antmaze", mode)
    K, B = hyper["k_steps"], hyper["batch"]
    idx = torch.from_numpy(d["indices"]).to(gh.DEV)
    buf = gh.make_buffer(hyper, data)
    tr_a = gh.make_trainer(hyper, nets, mode)
    la = tr_a.train_steps(buf, K, B, indices=idx, graph_unroll=0).cpu().numpy()
    tr_g = gh.make_trainer(hyper, nets, mode)
    lg = tr_g.train_steps(buf, K, B, indices=idx, graph_unroll=4).cpu().numpy()
    tr_b = gh.make_trainer(hyper, nets, mode)
    lb = []
    for t in range(K):
        out = tr_b.train(buf.sample(B, indices=idx[t]))
        assert set(out) == {"value_loss", "q_loss", "actor_loss"}
        lb.append([out["value_loss"], out["q_loss"], out["actor_loss"]])
    np.testing.assert_array_equal(la, lg)
    np.testing.assert_array_equal(la, np.asarray(lb, dtype=np.float32))
    for ma, mb in ((tr_a.qf, tr_g.qf), (tr_a.actor, tr_b.actor), (tr_a.q_target, tr_b.q_target)):
        for (k, va), (_, vb) in zip(ma.state_dict().items(), mb.state_dict().items()):
            assert torch.equal(va, vb), k


def test_device_philox_indices_match_oracle(gh):
    d, hyper, data, nets = helpers.load_traj("traj_antmaze", "fp32")
    K, B = 5, hyper["batch"]
    tr = gh.make_trainer(hyper, nets, "fp32", seed=4242)
    buf = gh.make_buffer(hyper, data)
    losses = tr.train_steps(buf, K, B).cpu().numpy()
    o = helpers.make_oracle(hyper, nets, "fp32")
    for t in range(K):
        ix = philox.sample_indices(4242, t, B, hyper["n_rows"])
        out = o.train(orc.gather_batch(data, ix))
        np.testing.assert_allclose(losses[t], [out["value_loss"], out["q_loss"], out["actor_loss"]],
                                   rtol=2e-5)


def test_device_philox_dropout_matches_oracle(gh):
    d, hyper, data, nets = helpers.load_traj("traj_pen_dropout", "fp32")
    K, B, H = 4, hyper["batch"], hyper["hidden"]
    tr = gh.make_trainer(hyper, nets, "fp32", seed=77)
    buf = gh.make_buffer(hyper, data)
    idx = torch.from_numpy(d["indices"][:K]).to(gh.DEV)
    losses = tr.train_steps(buf, K, B, indices=idx).cpu().numpy()
    o = helpers.make_oracle(hyper, nets, "fp32")
    for t in range(K):
        km = [philox.dropout_keep(77, t, 1, B, H, hyper["dropout"]),
              philox.dropout_keep(77, t, 2, B, H, hyper["dropout"])]
        out = o.train(orc.gather_batch(data, d["indices"][t]), km)
        np.testing.assert_allclose(losses[t], [out["value_loss"], out["q_loss"], out["actor_loss"]],
                                   rtol=2e-5)


def test_batch_prefetch_is_invisible(gh, monkeypatch):
    """The next-batch prefetch (idle k_update work-groups) must never serve a stale batch:
    switching buffers, index modes and entry points between calls gives the same bits as a
    trainer created with the prefetch disabled."""
    d, hyper, data, nets = helpers.load_traj("traj_antmaze", "bf16")
    B = hyper["batch"]
    buf_a = gh.make_buffer(hyper, data)
    data_b = {k: np.ascontiguousarray(np.asarray(v)[::-1]) for k, v in data.items()}
    buf_b = gh.make_buffer(hyper, data_b)
    idx = torch.from_numpy(d["indices"]).to(gh.DEV)

    def run(tr):
        out = [tr.train_steps(buf_a, 7, B, graph_unroll=0)]          # philox, eager
        out.append(tr.train_steps(buf_b, 9, B, graph_unroll=4))      # other buffer, graph + tail
        out.append(tr.train_steps(buf_a, 3, B, indices=idx[:3], graph_unroll=0))  # injected indices
        o = tr.train(buf_b.sample(B, indices=idx[3]))                # explicit batch
        out.append(torch.tensor([[o["value_loss"], o["q_loss"], o["actor_loss"]]], device=gh.DEV))
        out.append(tr.train_steps(buf_a, 5, B, graph_unroll=2))      # philox again
        return torch.cat(out).cpu().numpy()

    tr_p = gh.make_trainer(hyper, nets, "bf16", seed=11)
    monkeypatch.setenv("IQLHIP_NO_PREFETCH", "1")
    tr_n = gh.make_trainer(hyper, nets, "bf16", seed=11)
    monkeypatch.delenv("IQLHIP_NO_PREFETCH")
    lp, ln = run(tr_p), run(tr_n)
    np.testing.assert_array_equal(lp, ln)
    for ma, mb in ((tr_p.qf, tr_n.qf), (tr_p.vf, tr_n.vf), (tr_p.actor, tr_n.actor), (tr_p.q_target, tr_n.q_target)):
        for (k, va), (_, vb) in zip(ma.state_dict().items(), mb.state_dict().items()):
            assert torch.equal(va, vb), k
    # and the philox stream itself is still the oracle's
    assert np.isfinite(lp).all()


def test_state_dict_roundtrip_and_keys(gh):
    d, hyper, data, nets = helpers.load_traj("traj_antmaze", "fp32")
    K, B = hyper["k_steps"], hyper["batch"]
    idx = torch.from_numpy(d["indices"]).to(gh.DEV)
    buf = gh.make_buffer(hyper, data)
    tr = gh.make_trainer(hyper, nets, "fp32")
    tr.train_steps(buf, 4, B, indices=idx[:4])
    sd = tr.state_dict()
    assert sorted(sd.keys()) == list(d["state_dict_keys"])  # ref:664-674
    assert list(sd["actor"].keys()) == list(d["actor_keys"])
    assert list(sd["qf"].keys()) == list(d["qf_keys"])
    assert list(sd["vf"].keys()) == list(d["vf_keys"])
    sd = {k: (v if not isinstance(v, dict) else __import__("copy").deepcopy(v)) for k, v in sd.items()}
    rest = tr.train_steps(buf, K - 4, B, indices=idx[4:]).cpu().numpy()
    # resume in a fresh trainer from the checkpoint: same continuation except the
    # target net, which the reference rebuilds from qf on load (ref:679)
    tr2 = gh.make_trainer(hyper, nets, "fp32")
    tr2.load_state_dict(sd)
    assert tr2.total_it == 4
    for k, v in tr2.q_target.state_dict().items():
        assert torch.equal(v, tr2.qf.state_dict()[k])
    rest2 = tr2.train_steps(buf, K - 4, B, indices=idx[4:]).cpu().numpy()
    # q-loss and actor loss of the first resumed step do not involve the target
    np.testing.assert_allclose(rest2[0, 1], rest[0, 1], rtol=1e-6)
    assert tr2.total_it == K
    assert tr2.actor_lr_schedule.state_dict()["last_epoch"] == K


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_per_op_forward_vectors(gh, mode):
    """Module forward (fp32, outside autocast) and trainer.forward (inside) vs G1 vectors."""
    d = np.load(helpers.GOLDEN + "/per_op.npz")
    for tag, (S, A) in (("S17A6", (17, 6)), ("S29A8", (29, 8)), ("S45A24", (45, 24))):
        get = lambda net: {k.split("/", 2)[2]: d[k] for k in d.files if k.startswith(f"{tag}/{net}/")}
        hyper = dict(s_dim=S, a_dim=A, hidden=64, deterministic=False, dropout=None, iql_tau=0.7,
                     beta=3.0, max_steps=100, discount=0.99, tau=0.005)
        s = torch.from_numpy(d[f"{tag}/s"]).to(gh.DEV)
        a = torch.from_numpy(d[f"{tag}/a"]).to(gh.DEV)
        if mode == "fp32":
            q, v, actor = gh.make_nets(hyper, (get("qf"), get("vf"), get("gauss")))
            tol = dict(rtol=1e-5, atol=1e-5)
            q1, q2 = q.both(s, a)
            np.testing.assert_allclose(q1.cpu().numpy(), d[f"{tag}/fp32/q1"], **tol)
            np.testing.assert_allclose(q2.cpu().numpy(), d[f"{tag}/fp32/q2"], **tol)
            np.testing.assert_allclose(q(s, a).cpu().numpy(), d[f"{tag}/fp32/qmin"], **tol)
            np.testing.assert_allclose(v(s).cpu().numpy(), d[f"{tag}/fp32/v"], **tol)
            dist = actor(s)
            np.testing.assert_allclose(dist.mean.cpu().numpy(), d[f"{tag}/fp32/mean"], **tol)
            np.testing.assert_allclose(dist.log_prob(a).sum(-1).cpu().numpy(), d[f"{tag}/fp32/logp"],
                                       rtol=1e-4, atol=1e-4)
            hd = dict(hyper, deterministic=True)
            _, _, det = gh.make_nets(hd, (get("qf"), get("vf"), get("det")))
            np.testing.assert_allclose(det(s).cpu().numpy(), d[f"{tag}/fp32/det"], **tol)
            act = det.act(d[f"{tag}/s"][0], gh.DEV)
            np.testing.assert_allclose(act, np.clip(d[f"{tag}/fp32/det"][0], -1, 1), **tol)
        else:
            tr = gh.make_trainer(hyper, (get("qf"), get("vf"), get("gauss")), "bf16")
            tol = dict(rtol=1e-2, atol=1e-2)
            qq = tr.forward("q", s, a).cpu().numpy()
            np.testing.assert_allclose(qq[:, 0], d[f"{tag}/bf16/q1"], **tol)
            np.testing.assert_allclose(qq[:, 1], d[f"{tag}/bf16/q2"], **tol)
            np.testing.assert_allclose(tr.forward("v", s).cpu().numpy()[:, 0], d[f"{tag}/bf16/v"], **tol)
            np.testing.assert_allclose(tr.forward("actor", s).cpu().numpy(), d[f"{tag}/bf16/mean"], **tol)
            qt = tr.forward("q_target", s, a).cpu().numpy()
            np.testing.assert_allclose(qt.min(1), d[f"{tag}/bf16/qmin"], **tol)


def test_errors(gh):
    import iqlpref_amd as ia
    d, hyper, data, nets = helpers.load_traj("traj_antmaze", "fp32")
    tr = gh.make_trainer(hyper, nets, "fp32")
    buf = gh.make_buffer(hyper, data)
    b = buf.sample(64)
    b[1] = b[1][:, :3]
    with pytest.raises(RuntimeError, match="Actions shape missmatch"):
        tr.train(b)
    with pytest.raises(RuntimeError):
        ia.ReplayBuffer(3, 2, 10, "cpu")  # no CPU path


@pytest.mark.parametrize("S,A,H,B,det,drop,E", [
    (29, 8, 256, 1024, False, None, 2), (45, 24, 256, 256, False, 0.1, 2),
    (17, 6, 128, 48, True, None, 2), (11, 3, 64, 16, False, None, 2),
    # (two parts per backward work-group -- from 512 rows per launch on -- where H = 128 has only two)
    (17, 6, 128, 512, False, None, 2), (11, 3, 64, 512, True, None, 2),
    # BASELINE config 5: E-way critic ensemble (no reference implementation: the oracle's E-way
    # generalisation of ref:595-613 is the checker, pinned to the reference at E = 2 only)
    (29, 8, 256, 1024, False, None, 4), (29, 8, 256, 256, False, None, 3), (17, 6, 128, 48, True, None, 8),
    (11, 3, 64, 32, False, 0.1, 5),
    # round 4: the launch shapes that pick the throughput kernels one seed at a time -- both (E = 3 at batch
    # 1024, deterministic policy), the forward only (E = 8 at batch 256: ten trained nets; pen shapes with
    # dropout at batch 1024: its DROP instantiation with Philox / injected masks)
    (17, 6, 256, 1024, True, None, 3), (29, 8, 256, 256, False, None, 8), (45, 24, 256, 1024, False, 0.1, 2)])
@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_shapes_beyond_the_goldens_vs_oracle(gh, S, A, H, B, det, drop, E, mode):
    """BASELINE configs 3/5 shapes and odd sizes: HIP vs the (reference-pinned) oracle."""
    if mode == "fp32" and (S, A, H, B, det, drop, E) in ((17, 6, 256, 1024, True, None, 3), (29, 8, 256, 256, False, None, 8),
                                                         (45, 24, 256, 1024, False, 0.1, 2)):
        pytest.skip("round-4 cases: they exist for the throughput kernels, which are bf16 only")
    _shape_case(gh, S, A, H, 2, B, det, drop, E, mode)


@pytest.mark.parametrize("S,A,H,NH,B,det,drop,E", [
    (29, 8, 384, 2, 256, False, None, 2),    # the default depth, a width beyond the tuned step's
    (17, 6, 512, 3, 64, True, None, 2),      # wide and deep
    (11, 3, 1000, 1, 32, False, None, 2),    # one hidden layer, near the widest (LDS rows beyond 64 KB in fp32)
    (45, 24, 100, 3, 48, False, 0.1, 2),     # Philox dropout masks behind three hidden layers (streams 1, 2, 10)
    (29, 8, 80, 5, 1024, False, None, 4),    # E = 4 critics, batch 1024
    (17, 6, 24, 6, 16, True, 0.2, 3),        # the deepest, one 16-row slab
    (100, 28, 48, 2, 32, False, None, 2),    # the widest input (S + A = 128)
    (20, 32, 80, 2, 32, False, None, 2),     # the widest output (A = 32: two output tiles)
    (29, 8, 256, 2, 64, False, None, 2)])    # (IQLHIP_FORCE_GENERAL only: the default shape through the general step)
@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_general_step_shapes_vs_oracle(gh, S, A, H, NH, B, det, drop, E, mode):
    """Depths and widths only the general layer-wise step runs (ref:417-449: any n_hidden / hidden_dim): HIP vs the
    oracle, which tests/test_oracle_golden.py pins to reference runs at n_hidden = 1, 3, 4 (TRAJ_SHAPES)."""
    if NH == 2 and H == 256 and not os.environ.get("IQLHIP_FORCE_GENERAL"):
        pytest.skip("the tuned step takes this shape")
    _shape_case(gh, S, A, H, NH, B, det, drop, E, mode, after_a_kink=0.5)


def _shape_case(gh, S, A, H, NH, B, det, drop, E, mode, after_a_kink=None):
    import iqlpref_amd as ia
    rng = np.random.default_rng(B)
    N = 2000
    data = {"observations": rng.standard_normal((N, S)).astype(np.float32),
            "actions": rng.uniform(-1, 1, (N, A)).astype(np.float32),
            "rewards": rng.standard_normal(N).astype(np.float32),
            "next_observations": rng.standard_normal((N, S)).astype(np.float32),
            "terminals": (rng.uniform(size=N) < 0.05).astype(np.float32)}
    torch.manual_seed(B)
    q = ia.TwinQ(S, A, hidden_dim=H, n_hidden=NH) if E == 2 else \
        ia.EnsembleQ(S, A, hidden_dim=H, n_hidden=NH, n_critics=E)
    v = ia.ValueFunction(S, hidden_dim=H, n_hidden=NH)
    actor = (ia.DeterministicPolicy if det else ia.GaussianPolicy)(S, A, 1.0, hidden_dim=H, n_hidden=NH, dropout=drop)
    sd = lambda m: {k: t.detach().numpy().copy() for k, t in m.state_dict().items()}
    hyper = dict(s_dim=S, a_dim=A, hidden=H, n_hidden=NH, deterministic=det, dropout=drop, iql_tau=0.8, beta=3.0,
                 max_steps=1000, discount=0.99, tau=0.005, n_rows=N, n_critics=E)
    nets = (sd(q), sd(v), sd(actor))
    tr = gh.make_trainer(hyper, nets, mode, seed=7, keep_grads=True)
    general = NH != 2 or H not in (64, 128, 256) or bool(os.environ.get("IQLHIP_FORCE_GENERAL"))
    assert tr.step_kind(B) == ("general" if general else "tuned")
    assert len(tr.qf.critics()) == E and tr.forward("q", torch.zeros(2, S, device=gh.DEV),
                                                     torch.zeros(2, A, device=gh.DEV)).shape == (2, E)
    buf = gh.make_buffer(hyper, data)
    K = 3
    o = helpers.make_oracle(hyper, nets, mode)
    kinked = set()
    got = np.zeros((K, 3))
    for t in range(K):
        got[t] = tr.train_steps(buf, 1, B, graph_unroll=0).cpu().numpy()[0]
        km = None
        if drop:
            km = [philox.dropout_keep(7, t, philox.dropout_stream(l), B, H, drop) for l in range(NH)]
        out = o.train(orc.gather_batch(data, philox.sample_indices(7, t, B, N)), km)
        np.testing.assert_allclose(got[t], [out["value_loss"], out["q_loss"], out["actor_loss"]],
                                   rtol=3e-5 if mode == "fp32" else 6e-3)
        # The GRADIENT of every step pins the backward pass (a post-Adam parameter cannot: Adam's
        # first steps are sign-like, update = lr * g / (|g| + eps), so an eps-sized gradient entry
        # turns rounding noise into an O(lr) parameter difference).  Errors are relative to the
        # largest entry of the tensor.  fp32: every entry at summation-order noise -- unless the
        # oracle reports a pre-activation (or an advantage) within rounding of a kink: two fp32
        # implementations can then fall on different sides of relu'(z) for ONE (sample, unit) pair,
        # which moves every upstream gradient by that sample's share of the batch sum (seen at
        # B = 1024: 5e-4 of the maximum everywhere); the bound is then a few samples' share.
        # bf16: the entries agree bit for bit except where a bf16 rounding tie flips (same effect).
        for which, mod in (("q", tr.qf), ("v", tr.vf), ("actor", tr.actor)):
            tight = mode == "fp32" and o.last_margin[which] > 1e-5
            bound = 5e-5 if tight else max(4.0 / B, 2e-2 if mode == "bf16" else 0.0)
            if after_a_kink is not None:
                # The deeper / wider nets of the general step meet a kink in most runs (512 x 3 x 64 pre-activations
                # per evaluation).  A relu'(z) that falls the other way moves its unit's row by the WHOLE weight
                # of that sample's delta (measured 8.7 % of the largest entry at batch 64), and from then on this
                # network's parameters differ from the oracle's by sign-like Adam steps, i.e. the two runs are
                # different trajectories of ONE net (tools/deep_probe.py: the other nets stay bit-identical to the
                # oracle in bf16, 3e-7 in fp32).  So: a net that met a kink is only bounded loosely from that
                # step on; every other net keeps its bound.
                if o.last_margin[which] < 1e-5:
                    kinked.add(which)
                if which in kinked:
                    bound = after_a_kink
            for name, p in mod.named_parameters():
                want = o.last_grads[which][name]
                err = float(np.abs(p.grad.cpu().numpy() - want).max() / (np.abs(want).max() + 1e-30))
                _diag(f"shapes S{S} A{A} H{H} B{B} E{E} {mode} step {t} grad {which}/{name}: max {err:.2e} "
                      f"(margin {o.last_margin[which]:.1e}, bound {bound:.1e})")
                assert err < bound, f"step {t} grad {which}/{name}: {err:.3e} (kink margin {o.last_margin[which]:.1e})"
    for name, mod, opar in (("qf", tr.qf, o.qf), ("actor", tr.actor, o.actor), ("q_target", tr.q_target, o.q_target)):
        for k, t in mod.state_dict().items():
            # With the gradients pinned above, the parameters only need the sign-like-step bound:
            # no element may move further from the oracle than the steps themselves (K * lr).
            # (a net that met a kink -- general-step cases, above -- and the oracle's copy of it are two
            # trajectories: they may step in opposite directions, 2 lr apart per step)
            hit = {"qf": "q", "q_target": "q"}.get(name, name) in kinked
            diff = np.abs(t.cpu().numpy() - opar[k])
            assert diff.max() < (2 if hit else 1) * K * 3e-4 * 1.01 + 2e-6, f"{name}/{k}: max {diff.max():.3e}"
            if mode == "fp32" and not hit:
                # and all but a vanishing fraction agree to fp32 rounding (a small tensor may
                # hold a few such entries: 1 of the 4,352 of a 256 x 17 first layer is 2.3e-4 of it)
                assert (diff > 2e-6).sum() <= max(3, 1e-4 * diff.size), f"{name}/{k}: {(diff > 2e-6).mean():.2e} of elements off"


@pytest.mark.parametrize("mode", ["group", "streams", "split"])
def test_seed_group_matches_separate_runs(gh, mode):
    """Several seeds on one GPU -- one launch sequence with gridDim.y = K ("group"), one stream
    per seed ("streams") or two sub-groups on two streams ("split"), shared buffer -- = the same seeds run alone, bit for bit: losses of every
    step, every parameter, Adam moment and target weight."""
    import iqlpref_amd as ia
    d, hyper, data, nets = helpers.load_traj("traj_antmaze", "bf16")
    B = hyper["batch"]
    buf = gh.make_buffer(hyper, data)
    seeds = (3, 4, 5, 6)
    alone = [gh.make_trainer(hyper, nets, "bf16", seed=s) for s in seeds]
    want = [t.train_steps(buf, 37, B, graph_unroll=4).cpu().numpy() for t in alone]
    group = ia.SeedGroup([gh.make_trainer(hyper, nets, "bf16", seed=s) for s in seeds], chunk=10, mode=mode)
    assert group.mode == mode
    got = group.train_steps(buf, 30, B, return_losses=True, graph_unroll=4)
    got2 = group.train_steps(buf, 7, B, return_losses=True, graph_unroll=0)  # a second call continues the run
    group.synchronize()
    for w, g, g2, ta, tg in zip(want, got, got2, alone, group.trainers):
        np.testing.assert_array_equal(w, np.concatenate([g.cpu().numpy(), g2.cpu().numpy()]))
        assert tg.total_it == ta.total_it == 37
        for ma, mg in ((ta.actor, tg.actor), (ta.qf, tg.qf), (ta.vf, tg.vf), (ta.q_target, tg.q_target)):
            for (k, va), (_, vb) in zip(ma.state_dict().items(), mg.state_dict().items()):
                assert torch.equal(va, vb), k
        assert torch.equal(ta._exp_avg, tg._exp_avg) and torch.equal(ta._exp_avg_sq, tg._exp_avg_sq)
    assert not np.array_equal(want[0], want[1])  # different seeds sample different batches
    # a member keeps working on its own (it shares the group's descriptors), and after the group
    # is dissolved; both continue the same run
    solo = group.trainers[1].train_steps(buf, 3, B, graph_unroll=0).cpu().numpy()
    np.testing.assert_array_equal(solo, alone[1].train_steps(buf, 3, B, graph_unroll=0).cpu().numpy())
    group.close()
    solo = group.trainers[1].train_steps(buf, 5, B, graph_unroll=2).cpu().numpy()
    np.testing.assert_array_equal(solo, alone[1].train_steps(buf, 5, B, graph_unroll=2).cpu().numpy())


def test_general_step_seeds_checkpoint_and_forward(gh, tmp_path):
    """Trainers of a shape only the general step runs (three hidden layers of 96 units, the reference's own
    initial weights: traj_deep3_w96): a SeedGroup steps them one by one on their own streams (group launches
    exist for the tuned step only) = the same seeds run alone, bit for bit; a checkpoint written mid-run and
    loaded into a fresh trainer continues the run bit for bit (ref:664-688, keys of a four-Linear MLP);
    iqlhip_forward on the live weights = the oracle's forward."""
    import iqlpref_amd as ia
    d, hyper, data, nets = helpers.load_traj("traj_deep3_w96", "bf16")
    B = hyper["batch"]
    buf = gh.make_buffer(hyper, data)
    seeds = (3, 4, 5)
    alone = [gh.make_trainer(hyper, nets, "bf16", seed=s) for s in seeds]
    assert alone[0].step_kind(B) == "general"
    want = [t.train_steps(buf, 25, B).cpu().numpy() for t in alone]
    members = [gh.make_trainer(hyper, nets, "bf16", seed=s) for s in seeds]
    with pytest.raises(ValueError, match="tuned step"):
        ia.SeedGroup(members, mode="group")
    group = ia.SeedGroup(members, chunk=10)
    assert group.mode == "streams"
    got = group.train_steps(buf, 25, B, return_losses=True)
    group.synchronize()
    for w, g, ta, tg in zip(want, got, alone, group.trainers):
        np.testing.assert_array_equal(w, g.cpu().numpy())
        assert torch.equal(ta._params, tg._params) and torch.equal(ta._target, tg._target)
        assert torch.equal(ta._exp_avg, tg._exp_avg) and torch.equal(ta._exp_avg_sq, tg._exp_avg_sq)
    assert not np.array_equal(want[0], want[1])
    # checkpoint mid-run -> fresh trainer -> the same continuation
    sd = alone[0].state_dict()
    assert list(sd["vf"].keys()) == [f"v.net.{i}.{w}" for i in (0, 2, 4, 6) for w in ("weight", "bias")]
    path = str(tmp_path / "deep.pt")
    torch.save(sd, path)
    fresh = gh.make_trainer(hyper, nets, "bf16", seed=3)
    fresh.load_state_dict(torch.load(path, weights_only=True))
    assert fresh.total_it == 25
    # (the reference's checkpoint holds no target network, ref:664-674: the continuation is bit-identical
    # once the target is carried over, as tests/test_gpu_long_horizon.py does for the reference's own resume)
    with torch.no_grad():
        fresh._target.copy_(alone[0]._target)
    fresh.sync_weights()
    a = alone[0].train_steps(buf, 10, B).cpu().numpy()
    b = fresh.train_steps(buf, 10, B).cpu().numpy()
    np.testing.assert_array_equal(a, b)
    assert torch.equal(alone[0]._params, fresh._params)
    # forward passes on the live weights against the oracle on the same weights
    tr = alone[1]
    sdn = lambda m: {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    rng = np.random.default_rng(0)
    s_np = rng.standard_normal((37, hyper["s_dim"])).astype(np.float32)
    a_np = rng.uniform(-1, 1, (37, hyper["a_dim"])).astype(np.float32)
    st, at = torch.from_numpy(s_np).to(gh.DEV), torch.from_numpy(a_np).to(gh.DEV)
    qs, _ = orc.critics_all(sdn(tr.qf), s_np, a_np, "bf16")
    np.testing.assert_allclose(tr.forward("q", st, at).cpu().numpy(), np.stack(qs, 1), rtol=2e-2, atol=2e-3)
    qt, _ = orc.critics_all(sdn(tr.q_target), s_np, a_np, "bf16")
    np.testing.assert_allclose(tr.forward("q_target", st, at).cpu().numpy(), np.stack(qt, 1), rtol=2e-2, atol=2e-3)
    vv, _ = orc.value_forward(sdn(tr.vf), s_np, "bf16")
    np.testing.assert_allclose(tr.forward("v", st).cpu().numpy().reshape(-1), vv, rtol=2e-2, atol=2e-3)
    mean, _, _ = orc.policy_forward(sdn(tr.actor), s_np, "bf16")
    np.testing.assert_allclose(tr.forward("actor", st).cpu().numpy(), mean, rtol=2e-2, atol=4e-3)


@pytest.mark.parametrize("mode,n_seeds", [("group", 8), ("group", 2), ("split", 8)])
def test_seed_group_at_headline_shapes(gh, mode, n_seeds):
    """The launch geometries that only large launches take -- two parts of the dZ1 columns per backward
    work-group (from 512 rows per launch), two layer-2 parts and 32 rows per forward work-group
    (from 1024 / 512 rows), the group variant of the update kernel with its copies written from LDS --
    at H = 256, batch 256: every seed of the group = the same seed alone, bit for bit."""
    import iqlpref_amd as ia
    d, hyper, data, nets = helpers.load_traj("traj_antmaze_h256", "bf16")
    B = hyper["batch"]
    assert B * n_seeds >= 512 and hyper["hidden"] == 256
    buf = gh.make_buffer(hyper, data)
    seeds = tuple(range(11, 11 + n_seeds))
    alone = [gh.make_trainer(hyper, nets, "bf16", seed=s) for s in seeds]
    want = [t.train_steps(buf, 12, B, graph_unroll=4).cpu().numpy() for t in alone]
    group = ia.SeedGroup([gh.make_trainer(hyper, nets, "bf16", seed=s) for s in seeds], chunk=8, mode=mode)
    got = group.train_steps(buf, 12, B, return_losses=True, graph_unroll=4)
    group.synchronize()
    for w, g, ta, tg in zip(want, got, alone, group.trainers):
        np.testing.assert_array_equal(w, g.cpu().numpy())
        assert torch.equal(ta._params, tg._params) and torch.equal(ta._target, tg._target)
        assert torch.equal(ta._exp_avg, tg._exp_avg) and torch.equal(ta._exp_avg_sq, tg._exp_avg_sq)
    group.close()


@pytest.mark.parametrize("kind", ["pen_dropout", "cheetah_deterministic"])
def test_seed_group_of_eight_on_the_throughput_kernels(gh, kind):
    """Round 4: launches of 8 seeds at H = 256 / batch 256 run on k_forward_tp / k_backward_tp (224 / 256
    work-groups).  Their dropout instantiation (Philox masks of the actor's two Dropout layers, pen shapes:
    in_dim 69 -> three layer-1 k-steps, 24 outputs -> two layer-3 tiles) and the deterministic-policy branch
    must give every seed the bits of the seed alone, which runs on k_forward / k_backward."""
    import iqlpref_amd as ia
    S, A, det, drop = (45, 24, False, 0.1) if kind == "pen_dropout" else (17, 6, True, None)
    rng = np.random.default_rng(4)
    n = 2000
    data = {"observations": rng.standard_normal((n, S)).astype(np.float32),
            "actions": rng.uniform(-1, 1, (n, A)).astype(np.float32), "rewards": rng.standard_normal(n).astype(np.float32),
            "next_observations": rng.standard_normal((n, S)).astype(np.float32),
            "terminals": (rng.uniform(size=n) < 0.02).astype(np.float32)}
    buf = ia.ReplayBuffer(S, A, n, gh.DEV)
    buf.load_d4rl_dataset(data)

    def make(seed):
        torch.manual_seed(seed)
        q, v = ia.TwinQ(S, A).to(gh.DEV), ia.ValueFunction(S).to(gh.DEV)
        actor = (ia.DeterministicPolicy if det else ia.GaussianPolicy)(S, A, 1.0, dropout=drop).to(gh.DEV)
        return ia.ImplicitQLearning(
            max_action=1.0, actor=actor, actor_optimizer=torch.optim.Adam(actor.parameters(), lr=3e-4), q_network=q,
            q_optimizer=torch.optim.Adam(q.parameters(), lr=3e-4), v_network=v,
            v_optimizer=torch.optim.Adam(v.parameters(), lr=3e-4), iql_tau=0.8, beta=3.0, max_steps=1000,
            device=gh.DEV, precision="bf16", seed=seed)

    seeds = tuple(range(21, 29))
    alone = [make(s) for s in seeds]
    want = [t.train_steps(buf, 9, 256, graph_unroll=0).cpu().numpy() for t in alone]
    group = ia.SeedGroup([make(s) for s in seeds], mode="group")
    got = group.train_steps(buf, 9, 256, return_losses=True, graph_unroll=3)
    group.synchronize()
    for w, g, ta, tg in zip(want, got, alone, group.trainers):
        assert np.isfinite(w).all()
        np.testing.assert_array_equal(w, g.cpu().numpy())
        assert torch.equal(ta._params, tg._params) and torch.equal(ta._target, tg._target)
        assert torch.equal(ta._exp_avg, tg._exp_avg) and torch.equal(ta._exp_avg_sq, tg._exp_avg_sq)
    assert not np.array_equal(want[0], want[1])
    group.close()


def test_seed_group_split_uneven_and_cu_slice_streams(gh):
    """mode="split" with three seeds (sub-groups of two and one) on the CU-slice streams, driven
    with per-seed injected indices: bit-identical to the seeds alone.  The C entry point refuses
    slices that do not exist."""
    import ctypes as C
    import iqlpref_amd as ia
    from iqlpref_amd import _lib
    d, hyper, data, nets = helpers.load_traj("traj_antmaze", "bf16")
    B = hyper["batch"]
    buf = gh.make_buffer(hyper, data)
    seeds = (11, 12, 13)
    alone = [gh.make_trainer(hyper, nets, "bf16", seed=s) for s in seeds]
    want = [t.train_steps(buf, 21, B, graph_unroll=5).cpu().numpy() for t in alone]
    group = ia.SeedGroup([gh.make_trainer(hyper, nets, "bf16", seed=s) for s in seeds], chunk=8, mode="split")
    assert [len(ch) for ch in group._children] == [2, 1]
    got = group.train_steps(buf, 21, B, return_losses=True, graph_unroll=5)
    group.synchronize()
    for w, g_ in zip(want, got):
        np.testing.assert_array_equal(w, g_.cpu().numpy())
    group.close()
    lib = _lib.load()
    st = C.c_void_p()
    for sl, n in ((2, 2), (-1, 2), (0, 0), (0, 100000)):
        assert lib.iqlhip_stream_create_cu_slice(C.byref(st), sl, n) != 0
    assert lib.iqlhip_stream_create_cu_slice(None, 0, 2) != 0


def test_seed_group_injected_indices_and_own_buffers(gh):
    """Group launch with per-seed replay buffers, injected indices and dropout masks (the parity
    inputs of the golden trajectories): every member reproduces its solo trajectory bit for bit."""
    import iqlpref_amd as ia
    d, hyper, data, nets = helpers.load_traj("traj_pen_dropout", "fp32")
    K, B = hyper["k_steps"], hyper["batch"]
    idx = torch.from_numpy(d["indices"]).to(gh.DEV)
    keep = _drop_tensor(d, hyper, gh)
    solo = gh.make_trainer(hyper, nets, "fp32")
    want = solo.train_steps(gh.make_buffer(hyper, data), K, B, indices=idx, dropout_keep=keep,
                            graph_unroll=0).cpu().numpy()
    np.testing.assert_allclose(want, d["losses"], rtol=TOL["fp32"]["lg"])  # = the reference's trajectory
    bufs = [gh.make_buffer(hyper, data) for _ in range(3)]
    group = ia.SeedGroup([gh.make_trainer(hyper, nets, "fp32") for _ in range(3)], mode="group")
    assert group.mode == "group"
    # member 1 samples on the device (no injected inputs): it must differ, the others must not
    got = group.train_steps(bufs, K, B, indices=[idx, None, idx], dropout_keep=[keep, None, keep],
                            return_losses=True, graph_unroll=0)
    np.testing.assert_array_equal(got[0].cpu().numpy(), want)
    np.testing.assert_array_equal(got[2].cpu().numpy(), want)
    assert not np.array_equal(got[1].cpu().numpy(), want)
    with pytest.raises(ValueError):
        ia.SeedGroup([solo, solo])
    other = gh.make_trainer(helpers.load_traj("traj_antmaze", "fp32")[1], helpers.load_traj("traj_antmaze", "fp32")[3], "fp32")
    with pytest.raises(ValueError):
        ia.SeedGroup([solo, other], mode="group")
    assert ia.SeedGroup([solo, other]).mode == "streams"
    assert ia.SeedGroup([gh.make_trainer(hyper, nets, "fp32") for _ in range(2)]).mode == "split"  # the default


def test_trainer_lifecycle_does_not_leak(gh):
    """Create / step / destroy many trainers: workspace, graphs, streams and events are released."""
    import gc
    d, hyper, data, nets = helpers.load_traj("traj_antmaze", "bf16")
    buf = gh.make_buffer(hyper, data)
    torch.cuda.synchronize()
    gc.collect()
    torch.cuda.empty_cache()
    free0, _ = torch.cuda.mem_get_info()
    ref = None
    for i in range(40):
        tr = gh.make_trainer(hyper, nets, "bf16", seed=1)
        out = tr.train_steps(buf, 9, hyper["batch"], graph_unroll=4).cpu().numpy()
        if ref is None:
            ref = out
        np.testing.assert_array_equal(out, ref)  # a fresh trainer always starts from the same state
        del tr
    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 64 << 20, f"{(free0 - free1) >> 20} MiB not returned after 40 trainers"


def test_consecutive_calls_continue_each_other(gh):
    """A call with the same replay view (rows, size, generation), on-device indices and no per-step
    outputs continues the previous one: nothing is re-sent, the batch the last step prefetched is
    used.  Bit-identical to one long call, to calls that do re-send (they return losses), and a
    reloaded / different buffer is never served a stale batch."""
    import iqlpref_amd as ia
    d, hyper, data, nets = helpers.load_traj("traj_antmaze", "bf16")
    B = hyper["batch"]
    buf = gh.make_buffer(hyper, data)
    one = gh.make_trainer(hyper, nets, "bf16", seed=11)
    one.train_steps(buf, 21, B, return_losses=False, graph_unroll=0)
    split = gh.make_trainer(hyper, nets, "bf16", seed=11)
    for n, unroll in ((7, 0), (7, 7), (6, 3), (1, 0)):
        split.train_steps(buf, n, B, return_losses=False, graph_unroll=unroll)
    resend = gh.make_trainer(hyper, nets, "bf16", seed=11)
    for n in (7, 7, 7):
        resend.train_steps(buf, n, B, return_losses=True, graph_unroll=0)
    for other in (split, resend):
        assert torch.equal(one._params, other._params) and torch.equal(one._exp_avg_sq, other._exp_avg_sq)
        assert torch.equal(one._target, other._target)
    # another buffer (other contents, other generation) between two calls: its rows are sampled
    data2 = {k: (np.asarray(v)[::-1].copy() if np.asarray(v).ndim else v) for k, v in data.items()}
    buf2 = gh.make_buffer(hyper, data2)
    assert buf2.view().generation != buf.view().generation
    a, b = gh.make_trainer(hyper, nets, "bf16", seed=11), gh.make_trainer(hyper, nets, "bf16", seed=11)
    a.train_steps(buf, 5, B, return_losses=False, graph_unroll=0)
    a.train_steps(buf2, 5, B, return_losses=False, graph_unroll=0)
    a.train_steps(buf, 5, B, return_losses=False, graph_unroll=0)
    for bf in (buf, buf2, buf):
        b.train_steps(bf, 5, B, return_losses=True, graph_unroll=0)
    assert torch.equal(a._params, b._params)
    c = gh.make_trainer(hyper, nets, "bf16", seed=11)
    c.train_steps(buf, 15, B, return_losses=False, graph_unroll=0)
    assert not torch.equal(a._params, c._params)  # (the second buffer did change the run)
    # a group continues itself the same way
    g1 = ia.SeedGroup([gh.make_trainer(hyper, nets, "bf16", seed=s) for s in (3, 4)], mode="group")
    g2 = ia.SeedGroup([gh.make_trainer(hyper, nets, "bf16", seed=s) for s in (3, 4)], mode="group")
    g1.train_steps(buf, 12, B, graph_unroll=4)
    for n in (4, 4, 3, 1):
        g2.train_steps(buf, n, B, graph_unroll=4)
    for t1, t2 in zip(g1.trainers, g2.trainers):
        assert torch.equal(t1._params, t2._params)
    # ... and a member's own call in between does not leave the group with stale arguments
    g2.trainers[0].train_steps(buf, 2, B, return_losses=False, graph_unroll=0)
    g1.trainers[0].train_steps(buf, 2, B, return_losses=False, graph_unroll=0)
    g1.train_steps(buf, 3, B, graph_unroll=0)
    g2.train_steps(buf, 3, B, graph_unroll=0)
    for t1, t2 in zip(g1.trainers, g2.trainers):
        assert torch.equal(t1._params, t2._params)
