"""Diagnostic: where a preference-transformer window's time goes.  Runs the antmaze relabel shape
on the IQL_STAMPS build with phases of k_pt_relabel left out (IQLHIP_PT_SKIP bit mask: 1 last-token
phase, 2 LayerNorms, 4 global loads of the token phase, 8 K/V stores, 16 K/V GEMM); the results
of those runs are wrong by construction, only their times mean something.
    python -m iqlpref_amd.build --stamps && python tools/pt_diag.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["IQLHIP_LIB"] = os.path.join(ROOT, "iqlpref_amd", "libiqlhip_stamps.so")
import iqlpref_amd as ia  # noqa: E402
from oracle import relabel_oracle as ro  # noqa: E402  (parameter generator only)

dev = "cuda:0"
rng = np.random.default_rng(0)
S_, A_, QL, NW = 29, 8, 100, 200_000
p = ro.make_pt_params(rng, S_, A_, 1000, embd=64, pref=64, inter=256, layers=1)
m = ia.RewardPT(S_, A_, 1000, embd_dim=64, pref_attn_embd_dim=64, num_heads=4, intermediate_dim=256,
                num_layers=1, max_pos=256)
m.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()}, strict=False)
m = m.to(dev)
obs = torch.from_numpy(rng.standard_normal((NW + QL, S_)).astype(np.float32)).to(dev)
act = torch.from_numpy(rng.uniform(-1, 1, (NW + QL, A_)).astype(np.float32)).to(dev)
starts = torch.arange(NW, device=dev, dtype=torch.int64)
lens = torch.full((NW,), QL, device=dev, dtype=torch.int32)
for mask in (0, 1, 2, 4, 8, 16, 3, 7, 15, 31, 30):
    os.environ["IQLHIP_PT_SKIP"] = str(mask)
    m.window_values(obs, act, starts, lens, QL)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        m.window_values(obs, act, starts, lens, QL)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(f"skip={mask:2d}: {best * 1e3:7.2f} ms  {best / NW * 256 * 1e6:6.2f} us per window and work-group", flush=True)
