"""Diagnostic: one general-step case against the oracle, parameter and gradient errors per step."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_helpers as gh  # noqa: E402
import helpers  # noqa: E402
import iqlpref_amd as ia  # noqa: E402
from oracle import iql_oracle as orc, philox  # noqa: E402

S, A, H, NH, B, det, drop, E = [eval(x) for x in sys.argv[1:9]]
mode = sys.argv[9]
rng = np.random.default_rng(B)
N = 2000
data = {"observations": rng.standard_normal((N, S)).astype(np.float32),
        "actions": rng.uniform(-1, 1, (N, A)).astype(np.float32),
        "rewards": rng.standard_normal(N).astype(np.float32),
        "next_observations": rng.standard_normal((N, S)).astype(np.float32),
        "terminals": (rng.uniform(size=N) < 0.05).astype(np.float32)}
torch.manual_seed(B)
q = ia.TwinQ(S, A, hidden_dim=H, n_hidden=NH) if E == 2 else ia.EnsembleQ(S, A, hidden_dim=H, n_hidden=NH, n_critics=E)
v = ia.ValueFunction(S, hidden_dim=H, n_hidden=NH)
actor = (ia.DeterministicPolicy if det else ia.GaussianPolicy)(S, A, 1.0, hidden_dim=H, n_hidden=NH, dropout=drop)
sd = lambda m: {k: t.detach().numpy().copy() for k, t in m.state_dict().items()}
hyper = dict(s_dim=S, a_dim=A, hidden=H, n_hidden=NH, deterministic=det, dropout=drop, iql_tau=0.8, beta=3.0,
             max_steps=1000, discount=0.99, tau=0.005, n_rows=N, n_critics=E)
nets = (sd(q), sd(v), sd(actor))
tr = gh.make_trainer(hyper, nets, mode, seed=7, keep_grads=True)
print("kind", tr.step_kind(B))
buf = gh.make_buffer(hyper, data)
o = helpers.make_oracle(hyper, nets, mode)
for t in range(3):
    got = tr.train_steps(buf, 1, B, graph_unroll=0).cpu().numpy()[0]
    km = [philox.dropout_keep(7, t, philox.dropout_stream(l), B, H, drop) for l in range(NH)] if drop else None
    out = o.train(orc.gather_batch(data, philox.sample_indices(7, t, B, N)), km)
    print(t, "loss rel", got / np.array([out["value_loss"], out["q_loss"], out["actor_loss"]]) - 1, "margins", o.last_margin)
    for which, mod, opar in (("q", tr.qf, o.qf), ("v", tr.vf, o.vf), ("actor", tr.actor, o.actor)):
        for name, p in mod.named_parameters():
            want = o.last_grads[which][name]
            g = p.grad.cpu().numpy()
            err = np.abs(g - want)
            perr = np.abs(p.detach().cpu().numpy() - opar[name])
            print(f"  {t} {which}/{name}: grad err/max {err.max() / (np.abs(want).max() + 1e-30):.2e} "
                  f"n>1e-4max {(err > 1e-4 * np.abs(want).max()).sum()} of {err.size}; param err max {perr.max():.2e} "
                  f"n>2e-6 {(perr > 2e-6).sum()}")
