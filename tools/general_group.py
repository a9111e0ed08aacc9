"""Aggregate throughput of K general-step trainers in a SeedGroup (streams mode): python tools/general_group.py H N_HIDDEN B K"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import iqlpref_amd as ia  # noqa: E402
import bench  # noqa: E402

H, NH, B, K = [int(x) for x in sys.argv[1:5]]
dev = "cuda:0"
data = bench.synth_dataset(1, 200_000)
buf = ia.ReplayBuffer(bench.S_DIM, bench.A_DIM, 200_000, dev)
buf.load_d4rl_dataset(data)
trs = [bench.build_trainer(ia, torch, dev, 1 + k, "bf16", hidden_dim=H, n_hidden=NH) for k in range(K)]
g = ia.SeedGroup(trs, chunk=int(os.environ.get("GG_CHUNK", "200")))
g.train_steps(buf, 400, B, return_losses=False)
g.synchronize()
n = 3000
t0 = time.perf_counter()
g.train_steps(buf, n, B, return_losses=False)
g.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"mode": g.mode, "kind": trs[0].step_kind(B), "H": H, "n_hidden": NH, "B": B, "K": K,
                  "steps_per_s_total": K * n / dt, "group_step_us": dt / n * 1e6}))
