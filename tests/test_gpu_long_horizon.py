"""Long horizons against runs of the reference itself.  -m gpu.

Two kinds of fixture, both written by tests/golden/make_fixtures.py from /root/reference:

  traj_resume_{fp32,bf16}   the reference's checkpoint dict after 990 steps (+ its target net) and
                            its next 10 steps (H = 64): loaded into OUR trainer with
                            load_state_dict, replayed with the same batches.  Teacher-forced, so
                            the tolerances are the 10-step ones: pins Adam's bias corrections at
                            t ~ 1000, the cosine schedule mid-flight (max_steps 2000), Polyak, and
                            that our trainer reads a checkpoint WRITTEN BY THE REFERENCE.
  traj_long_{cheetah,antmaze}_{fp32,bf16}
                            1,000 free-running steps at the BASELINE widths (configs 1 and 2, H = 256,
                            batch 256): every step's losses, strided parameter summaries after 100
                            and 1,000 steps.  IQL training amplifies rounding differences (Adam's
                            sign-like steps): two correct fp32 implementations -- the numpy oracle
                            and torch -- are 2e-4 apart in the losses by step 100 and 5e-2 by step
                            1,000 (measured, see BOUNDS).  The comparison is therefore made per
                            window of steps, with bounds <= 5 x the drift measured for THIS library,
                            and the oracle's own drift from the reference is computed in the same
                            test as the yardstick (a step-1,000 loss that differed by more than a
                            chaotic trajectory explains would fail both).
"""
import os

import numpy as np
import pytest
import torch

from oracle import iql_oracle as orc
from tests import helpers

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gh():
    from tests import gpu_helpers
    return gpu_helpers


def _diag(line):
    path = os.environ.get("IQL_TEST_DIAG")
    if path:
        with open(path, "a") as f:
            f.write(line + "\n")


def _hyper(d):
    h = d["hyper"]
    return dict(s_dim=int(h[0]), a_dim=int(h[1]), hidden=int(h[2]), batch=int(h[3]), n_rows=int(h[4]),
                k_steps=int(h[5]), beta=float(h[6]), iql_tau=float(h[7]), discount=float(h[8]), tau=float(h[9]),
                deterministic=bool(h[10]), dropout=None, max_steps=int(h[12]))


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_resume_from_reference_checkpoint(gh, mode):
    d = np.load(f"{helpers.GOLDEN}/traj_resume_{mode}.npz")
    hyper = _hyper(d)
    data, init_nets = helpers.regen_inputs(d)
    idx = helpers.regen_indices(d)
    nets, target, moments, t0 = helpers.resume_state(d)
    K, B = hyper["k_steps"], hyper["batch"]
    tr = gh.make_trainer(hyper, init_nets, mode)
    buf = gh.make_buffer(hyper, data)

    # ---- the reference's checkpoint dict (ref:664-674), rebuilt from the arrays it was stored as ----
    T = lambda a: torch.from_numpy(np.asarray(a).copy())
    sd = {"total_it": t0}
    for net, arrs in zip(("qf", "vf", "actor"), nets):
        sd[net] = {("_orig_mod." + k if net == "vf" else k): T(v) for k, v in arrs.items()}  # (ref:1523-1528 prefix, one net)
    for group, key, opt in (("q", "q_optimizer", tr.q_optimizer), ("v", "v_optimizer", tr.v_optimizer),
                            ("actor", "actor_optimizer", tr.actor_optimizer)):
        osd = opt.state_dict()
        assert osd["state"] == {}  # fresh optimiser: same shape as the reference's before its first step
        osd["state"] = {i: {"step": torch.tensor(float(t0)), "exp_avg": T(m), "exp_avg_sq": T(v2)}
                        for i, (m, v2) in enumerate(moments[group].values())}
        osd["param_groups"][0]["lr"] = float(d[f"ckpt/{key}/lr"])
        sd[key] = osd
    sch = tr.actor_lr_schedule.state_dict()
    ref_sch = d["ckpt/actor_lr_schedule"]  # T_max, eta_min, base_lr, last_epoch, _step_count, _last_lr
    sch.update(T_max=int(ref_sch[0]), eta_min=float(ref_sch[1]), base_lrs=[float(ref_sch[2])],
               last_epoch=int(ref_sch[3]), _step_count=int(ref_sch[4]), _last_lr=[float(ref_sch[5])])
    sd["actor_lr_schedule"] = sch
    tr.load_state_dict(sd)
    assert tr.total_it == t0
    # the reference does not checkpoint its target net (ref:676-688 rebuilds it as a copy of qf); the
    # run this fixture continues kept its own: hand it over
    for name, p in tr.q_target.state_dict().items():
        p.copy_(T(target[name]).to(p.device))
    tr.sync_weights()

    n = K - t0
    losses = tr.train_steps(buf, n, B, indices=torch.from_numpy(idx[t0:]).to(gh.DEV), graph_unroll=0).cpu().numpy()
    rel = np.abs(losses / d["losses"][t0:] - 1).max()
    _diag(f"resume {mode}: loss rel vs reference {rel:.2e}")
    np.testing.assert_allclose(losses, d["losses"][t0:], rtol=2e-5 if mode == "fp32" else 5e-3)
    assert tr.total_it == K
    assert abs(tr.actor_optimizer.param_groups[0]["lr"] - float(d["final_actor_lr"])) < 1e-12
    worst = 0.0
    for net, mod in (("qf", tr.qf), ("vf", tr.vf), ("actor", tr.actor), ("q_target", tr.q_target)):
        start = dict(zip(("qf", "vf", "actor"), nets)).get(net, target)
        for k, v in gh.module_params(mod).items():
            want = d[f"final/{net}/{k}"]
            if mode == "fp32":
                np.testing.assert_allclose(v, want, atol=2e-6, rtol=0, err_msg=f"{net}/{k}")
            else:  # the 10-step movement against the reference's, as in test_trajectory_parity
                got_d, want_d = (v - start[k]).astype(np.float64), (want - start[k]).astype(np.float64)
                r = float(np.linalg.norm(got_d - want_d) / (np.linalg.norm(want_d) + 1e-30))
                worst = max(worst, r)
                assert r < 0.02, (net, k, r)
    _diag(f"resume {mode}: worst movement rel vs reference {worst:.4f}")
    for opt, mod, key in ((tr.q_optimizer, tr.qf, "q_adam"), (tr.v_optimizer, tr.vf, "v_adam"),
                          (tr.actor_optimizer, tr.actor, "actor_adam")):
        for pname, p in mod.named_parameters():
            st = opt.state[p]
            assert float(st["step"]) == K == float(d[f"final/{key}/{pname}/step"])
            for mom in ("exp_avg", "exp_avg_sq"):
                want = d[f"final/{key}/{pname}/{mom}"]
                err = np.abs(st[mom].cpu().numpy() - want).max() / (np.abs(want).max() + 1e-30)
                assert err < (1e-4 if mode == "fp32" else 1e-2), (key, pname, mom, err)


# Windows of steps and the bound on the largest relative loss difference |ours / reference - 1| inside
# each, per mode: (value_loss, q_loss, actor_loss).  Measured on an MI355X (IQL_TEST_DIAG,
# profiles/r04_long_horizon_diag.txt), larger of the two trajectories:
#   this library   fp32  0-10 4.0e-7 | 10-100 6.1e-7 | 100-300 (4.7e-3, 6.9e-4, 2.6e-3) | 300-1000 (0.12, 0.031, 0.20)
#                  bf16  0-10 (6.6e-4, 1.6e-5, 3.1e-4) | 10-100 (1.2e-2, 2.1e-3, 9.4e-3) | 100-300 (0.46, 0.037, 0.22)
#                        | 300-1000 (0.40, 0.092, 0.89)
#   numpy oracle   fp32  0-10 2.5e-7 | 10-100 (2.0e-4, 2.5e-5, 1.8e-5) | 100-300 (9.1e-3, 1.3e-3, 1.7e-3)
#                        | 300-1000 (0.34, 0.088, 0.73);  bf16: this library's figures to two digits
# Bounds = 5 x the LARGER of the two rows.  The trajectories amplify any difference in summation order:
# a build of this library whose fp32 weight-gradient GEMM summed the batch in two halves (round 4,
# gpurun_out/d3) drifted by 5.2e-4 over steps 10-100, like the oracle -- the 6e-7 of the shipped order is
# a coincidence of orders (the MFMA k-order happening to meet mkldnn's), not a property to pin.  A
# defect (a wrong bias correction, schedule or Polyak step at large t) would show at 1e-2 and more from
# its first step on; test_resume_from_reference_checkpoint pins those without the chaos.
WINDOWS = ((0, 10), (10, 100), (100, 300), (300, 1000))
BOUNDS = {
    "fp32": ((2e-6, 2e-6, 2e-6), (1e-3, 1.3e-4, 9e-5), (4.6e-2, 6.5e-3, 1.3e-2), (1.7, 0.44, 3.7)),
    "bf16": ((3.3e-3, 8e-5, 1.5e-3), (6e-2, 1e-2, 4.7e-2), (2.3, 0.18, 1.1), (2.0, 0.46, 4.5)),
}
# ... and on the mean loss of a window (what train() logs, ref:1537-1544), largest of the three losses:
# measured fp32 1.1e-7 | 3.2e-5 (split-order build) | 1.6e-4 | 5.6e-3, bf16 7.3e-5 | 1.9e-4 | 1.7e-2 | 1.7e-2
MEAN_BOUNDS = {"fp32": (1e-6, 1.6e-4, 8e-4, 2.8e-2), "bf16": (4e-4, 1e-3, 9e-2, 9e-2)}


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
@pytest.mark.parametrize("name", ["traj_long_cheetah", "traj_long_antmaze"])
def test_thousand_step_trajectory_vs_reference(gh, name, mode):
    d, hyper, data, nets = helpers.load_traj(name, mode)
    idx = helpers.regen_indices(d)
    K, B = hyper["k_steps"], hyper["batch"]
    assert K == 1000
    tr = gh.make_trainer(hyper, nets, mode)
    buf = gh.make_buffer(hyper, data)
    dev_idx = torch.from_numpy(idx).to(gh.DEV)
    l100 = tr.train_steps(buf, 100, B, indices=dev_idx[:100], graph_unroll=0).cpu().numpy()
    snap100 = {net: gh.module_params(mod) for net, mod in
               (("qf", tr.qf), ("vf", tr.vf), ("actor", tr.actor), ("q_target", tr.q_target))}
    snap100 = {net: {k: v.copy() for k, v in pd.items()} for net, pd in snap100.items()}
    l1000 = tr.train_steps(buf, 900, B, indices=dev_idx[100:], graph_unroll=0).cpu().numpy()
    losses = np.concatenate([l100, l1000])
    assert np.isfinite(losses).all() and tr.total_it == 1000
    assert abs(tr.actor_optimizer.param_groups[0]["lr"] - float(d["final_actor_lr"])) < 1e-12
    # the yardstick: an independent implementation of the same arithmetic on the same inputs
    o = helpers.make_oracle(hyper, nets, mode)
    want_o = np.zeros((K, 3))
    for t in range(K):
        out = o.train(orc.gather_batch(data, idx[t]))
        want_o[t] = [out["value_loss"], out["q_loss"], out["actor_loss"]]
    ref = d["losses"]
    rel, rel_o = np.abs(losses / ref - 1), np.abs(want_o / ref - 1)
    for (a, b), bound, mb in zip(WINDOWS, BOUNDS[mode], MEAN_BOUNDS[mode]):
        got = rel[a:b].max(axis=0)
        mean_rel = np.abs(losses[a:b].mean(axis=0) / ref[a:b].mean(axis=0) - 1)
        _diag(f"{name} {mode} steps {a}-{b}: max rel loss vs reference {got[0]:.2e} {got[1]:.2e} {got[2]:.2e} "
              f"(oracle vs reference {rel_o[a:b].max(axis=0)}); window mean {mean_rel}")
        assert (got <= np.asarray(bound)).all(), (name, mode, (a, b), got, bound)
        assert (mean_rel <= mb).all(), (name, mode, (a, b), mean_rel, mb)
    # parameters after 100 and 1,000 steps: the movement from the initial weights against the reference's
    # (strided samples).  After 100 steps the runs still move together (measured rel. L2: fp32 < 5e-5 with
    # the shipped summation order, the numpy oracle 0.03; bf16 0.11); after 1,000 the trajectories have
    # separated (0.16-0.29): there the movement's SIZE is compared (|norm ratio - 1| measured fp32 0.027,
    # bf16 0.137), not its direction.
    init = dict(zip(("qf", "vf", "actor"), nets))
    init["q_target"] = nets[0]
    final = {net: gh.module_params(mod) for net, mod in
             (("qf", tr.qf), ("vf", tr.vf), ("actor", tr.actor), ("q_target", tr.q_target))}
    for at, snap in ((100, snap100), (1000, final)):
        worst, worst_n = 0.0, 0.0
        for net, pd in snap.items():
            for k, v in pd.items():
                want, got = helpers.golden_param(d, f"at{at}/{net}/{k}", v)
                assert want is not None, (at, net, k)
                i0 = np.asarray(init[net][k])
                i0 = i0.reshape(-1)[::37] if want.size != v.size else i0
                gd = (got.reshape(-1) - i0.reshape(-1)).astype(np.float64)
                wd = (want.reshape(-1) - i0.reshape(-1)).astype(np.float64)
                if np.linalg.norm(wd) < 1e-12:
                    continue
                worst = max(worst, float(np.linalg.norm(gd - wd) / np.linalg.norm(wd)))
                worst_n = max(worst_n, abs(float(np.linalg.norm(gd) / np.linalg.norm(wd)) - 1))
        _diag(f"{name} {mode} at {at}: movement rel L2 vs reference {worst:.4f}, |norm ratio - 1| {worst_n:.4f}")
        if at == 100:
            assert worst < (0.15 if mode == "fp32" else 0.5), (name, mode, at, worst)
        assert worst_n < (0.13 if mode == "fp32" else 0.65), (name, mode, at, worst_n)
