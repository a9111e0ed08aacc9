"""Aggregate steps/s and per-kernel launch times of a seed group for K = 1, 2, 4, 8 seeds
(one launch sequence, gridDim.y = K).  Usage on the GPU box: python tools/group_scan.py [K ...]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import iqlpref_amd as ia  # noqa: E402

dev = "cuda:0"
ks = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
buf = ia.ReplayBuffer(bench.S_DIM, bench.A_DIM, 200_000, dev)
buf.load_d4rl_dataset(bench.synth_dataset(1, 200_000))
out = {}
for K in ks:
    trs = [bench.build_trainer(ia, torch, dev, 10 + i, "bf16") for i in range(K)]
    g = ia.SeedGroup(trs, mode="group")
    n = int(os.environ.get("GROUP_SCAN_STEPS", "10000"))  # (short regions for rocprofv3 --pmc passes)
    g.train_steps(buf, min(1000, n), bench.BATCH, graph_unroll=int(os.environ.get('GS_UNROLL', '50')))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.train_steps(buf, n, bench.BATCH, graph_unroll=int(os.environ.get('GS_UNROLL', '50')))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kt = g.kernel_times(buf, bench.BATCH, 200)
    out[K] = {"steps_per_s": K * n / dt, "group_step_us": dt / n * 1e6, "kernel_us_events": kt}
    print(K, json.dumps(out[K]), flush=True)
    g.close()
    del g, trs
