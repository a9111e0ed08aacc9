"""Aggregate steps/s of K seeds as ONE group and as two CU-slice sub-groups (SeedGroup mode "split").
Usage on the GPU box: python tools/split_scan.py [K ...]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import iqlpref_amd as ia  # noqa: E402

dev = "cuda:0"
buf = ia.ReplayBuffer(bench.S_DIM, bench.A_DIM, 200_000, dev)
buf.load_d4rl_dataset(bench.synth_dataset(1, 200_000))
for K in [int(a) for a in sys.argv[1:]] or [8, 16]:
    trs = [bench.build_trainer(ia, torch, dev, 10 + i, "bf16") for i in range(K)]
    for mode in ("group", "split"):
        g = ia.SeedGroup(trs, mode=mode, **({"n_streams": 2} if mode == "split" else {}))
        g.train_steps(buf, 1000, bench.BATCH, graph_unroll=50)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 8000
        g.train_steps(buf, n, bench.BATCH, graph_unroll=50)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"K={K} {mode}: {K * n / dt:.0f} steps/s ({dt / n * 1e6:.2f} us per group step)", flush=True)
        g.close()
    del trs
