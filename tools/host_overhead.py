"""Diagnostic: where the host time of one short train_steps call goes (the driver's
`--steps 20` blocks are ~0.35 ms: every host microsecond before the graph launch is GPU idle time)."""
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
tr = bench.build_trainer(ia, torch, dev, 1, "bf16")
K = 20
pc = time.perf_counter
med = lambda x: sorted(x)[len(x) // 2] * 1e6
for unroll in (K, 10, 5, 0):  # one graph of K steps, shorter graphs, eager launches
    tr.train_steps(buf, 200, bench.BATCH, return_losses=False, graph_unroll=unroll)
    torch.cuda.synchronize()
    host, block, sync_idle = [], [], []
    for _ in range(1500):
        torch.cuda.synchronize()
        t0 = pc()
        tr.train_steps(buf, K, bench.BATCH, return_losses=False, graph_unroll=unroll)
        t1 = pc()
        torch.cuda.synchronize()
        t2 = pc()
        host.append(t1 - t0), block.append(t2 - t0)
        t3 = pc()
        torch.cuda.synchronize()
        sync_idle.append(pc() - t3)
    print(f"unroll {unroll:2d}: train_steps host time {med(host):.1f} us; block {med(block):.1f} us "
          f"({K / med(block) * 1e6:.0f} steps/s); idle synchronize {med(sync_idle):.1f} us", flush=True)
# pieces of the host time
import ctypes as C
from iqlpref_amd import _lib
n = 20000
t0 = pc()
for _ in range(n):
    tr._refresh_lrs()
print(f"_refresh_lrs {(pc() - t0) / n * 1e6:.2f} us")
t0 = pc()
for _ in range(n):
    buf.view()
print(f"view() {(pc() - t0) / n * 1e6:.2f} us")
t0 = pc()
for _ in range(n):
    _lib.stream_ptr()
print(f"stream_ptr() {(pc() - t0) / n * 1e6:.2f} us")
t0 = pc()
for _ in range(n):
    with torch.cuda.device(dev):
        pass
print(f"torch.cuda.device ctx {(pc() - t0) / n * 1e6:.2f} us")
t0 = pc()
for _ in range(n):
    tr._after_steps(0)
print(f"_after_steps {(pc() - t0) / n * 1e6:.2f} us")
