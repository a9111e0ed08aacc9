"""Throughput of the general layer-wise step: python tools/general_run.py H N_HIDDEN B [E] [steps]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import iqlpref_amd as ia  # noqa: E402
import bench  # noqa: E402

H, NH, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
E = int(sys.argv[4]) if len(sys.argv) > 4 else 2
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 2000
dev = "cuda:0"
S, A = bench.S_DIM, bench.A_DIM
data = bench.synth_dataset(1, 200_000)
buf = ia.ReplayBuffer(S, A, 200_000, dev)
buf.load_d4rl_dataset(data)
torch.manual_seed(1)
q = (ia.TwinQ(S, A, hidden_dim=H, n_hidden=NH) if E == 2 else ia.EnsembleQ(S, A, hidden_dim=H, n_hidden=NH, n_critics=E)).to(dev)
v = ia.ValueFunction(S, hidden_dim=H, n_hidden=NH).to(dev)
actor = ia.GaussianPolicy(S, A, 1.0, hidden_dim=H, n_hidden=NH).to(dev)
mk = lambda m: torch.optim.Adam(m.parameters(), lr=3e-4)
tr = ia.ImplicitQLearning(1.0, actor, mk(actor), q, mk(q), v, mk(v), device=dev, seed=1)
kind = tr.step_kind(B)
tr.train_steps(buf, 200, B, return_losses=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
tr.train_steps(buf, steps, B, return_losses=False)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
lib = tr._lib
import ctypes as C
lib.iqlhip_trainer_set_timing(tr._handle, 1)
tr.train_steps(buf, 100, B, return_losses=False)
ms = (C.c_double * 3)()
n = C.c_int64()
lib.iqlhip_trainer_get_timing(tr._handle, C.byref(ms), C.byref(n))
lib.iqlhip_trainer_set_timing(tr._handle, 0)
print(json.dumps({"kind": kind, "H": H, "n_hidden": NH, "B": B, "E": E, "steps_per_s": steps / dt, "us_per_step": dt / steps * 1e6,
                  "kernel_us_events": {"forward": ms[0] * 1e3, "backward": ms[1] * 1e3, "update": ms[2] * 1e3}}))
