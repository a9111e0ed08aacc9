"""Writes a small checkpoint of OUR trainer (ImplicitQLearning.state_dict(), the on-disk format
of ref:1558-1562) for the reference-side compatibility fixture:

    python tools/make_checkpoint.py gpurun_out/our_checkpoint.pt     # on the GPU box

tests/golden/make_fixtures.py (in the build container, where the reference is importable) then
loads its ["actor"] into the reference's GaussianPolicy the way
evaluation/d4rl/iql_eval_median.py:252-262 does (strict=False) and records the policy's output
on fixed observations; tests/test_gpu_round2.py compares our actor with that record."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import iqlpref_amd as ia  # noqa: E402

S, A, H, B, N = 17, 6, 64, 64, 4096
dev = "cuda:0"
rng = np.random.default_rng(7)
data = {"observations": rng.standard_normal((N, S)).astype(np.float32),
        "actions": rng.uniform(-1, 1, (N, A)).astype(np.float32),
        "rewards": rng.standard_normal(N).astype(np.float32),
        "next_observations": rng.standard_normal((N, S)).astype(np.float32),
        "terminals": (rng.uniform(size=N) < 0.01).astype(np.float32)}
torch.manual_seed(7)
q, v = ia.TwinQ(S, A, hidden_dim=H).to(dev), ia.ValueFunction(S, hidden_dim=H).to(dev)
actor = ia.GaussianPolicy(S, A, 1.0, hidden_dim=H, dropout=0.1).to(dev)  # dropout: Sequential indices 0,3,6
tr = ia.ImplicitQLearning(
    max_action=1.0, actor=actor, actor_optimizer=torch.optim.Adam(actor.parameters(), lr=3e-4),
    q_network=q, q_optimizer=torch.optim.Adam(q.parameters(), lr=3e-4), v_network=v,
    v_optimizer=torch.optim.Adam(v.parameters(), lr=3e-4), iql_tau=0.8, beta=3.0, max_steps=1000,
    device=dev, precision="bf16", seed=7)
buf = ia.ReplayBuffer(S, A, N, dev)
buf.load_d4rl_dataset(data)
tr.train_steps(buf, 25, B)
ck = tr.state_dict()
cpu = lambda o: ({k: cpu(x) for k, x in o.items()} if isinstance(o, dict) else
                 [cpu(x) for x in o] if isinstance(o, list) else o.detach().cpu().clone() if torch.is_tensor(o) else o)
out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/our_checkpoint.pt"
torch.save(cpu(ck), out)
print("wrote", out, "keys", sorted(ck), "actor keys", sorted(ck["actor"]))
