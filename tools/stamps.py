"""Diagnostic: per-work-group stage timeline of one step (IQL_STAMPS build).
Usage on the GPU box: python tools/stamps.py   (expects iqlpref_amd/libiqlhip_stamps.so)"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iqlpref_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "iqlpref_amd", "libiqlhip_stamps.so")
import iqlpref_amd as ia  # noqa: E402
import bench  # noqa: E402

lib = _lib.load()
lib.iqlhip_trainer_set_debug.restype = C.c_int
lib.iqlhip_trainer_set_debug.argtypes = [C.c_void_p, C.c_void_p]
dev = "cuda:0"
data = bench.synth_dataset(1, 200_000)
buf = ia.ReplayBuffer(bench.S_DIM, bench.A_DIM, 200_000, dev)
buf.load_d4rl_dataset(data)
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
tr = bench.build_trainer(ia, torch, dev, 1, prec)
tr.train_steps(buf, 200, bench.BATCH, return_losses=False, graph_unroll=0)
dbg = torch.zeros((3, 512, 8, 2), dtype=torch.int64, device=dev)
_lib.check(lib.iqlhip_trainer_set_debug(tr._handle, C.c_void_p(dbg.data_ptr())))
tr.train_steps(buf, 1, bench.BATCH, return_losses=False, graph_unroll=0)
torch.cuda.synchronize()
d = dbg.cpu().numpy().astype(np.float64)
names = ["k_forward", "k_backward", "k_update"]
t_first = None
for k in range(3):
    w = d[k, :, :, 0]
    c = d[k, :, :, 1]
    used = w[:, 0] > 0
    nb = int(used.sum())
    if nb == 0:
        continue
    w, c = w[used], c[used]
    t0 = w[:, 0].min()
    if t_first is None:
        t_first = t0
    last = np.where(w > 0, w, 0).max()
    print(f"{names[k]}: {nb} blocks; first start at +{(t0 - t_first) * 10:.0f} ns; kernel span "
          f"{(last - t0) * 10:.0f} ns; block start skew max {(w[:, 0].max() - t0) * 10:.0f} ns")
    for s in range(1, 8):
        ok = w[:, s] > 0
        if not ok.any():
            continue
        prev = s - 1
        while prev > 0 and not (w[ok, prev] > 0).all():
            prev -= 1
        dt = (w[ok, s] - w[ok, prev]) * 10
        dc = c[ok, s] - c[ok, prev]
        clk = np.median(dc[dt > 0] / dt[dt > 0]) if (dt > 0).any() else float("nan")
        print(f"   stage {prev}->{s}: median {np.median(dt):7.0f} ns  max {dt.max():7.0f} ns   "
              f"(~{clk:.2f} GHz shader clock)   n={int(ok.sum())}")

if "--detail" in sys.argv:
    k = 2
    w = d[k, :, :, 0]
    used = w[:, 0] > 0
    t0 = w[used, 0].min()
    for b in np.where(used)[0]:
        row = [(w[b, s] - t0) * 10 if w[b, s] > 0 else -1 for s in range(5)]
        print(f"upd block {b:3d} xcd {b % 8}: " + " ".join(f"{x:7.0f}" for x in row))

# the misc block of k_update is the last block of its grid
w = d[2, :, :, 0]
used = np.where(w[:, 0] > 0)[0]
mb = used.max()
print(f"k_update misc block {mb}: start +{(w[mb,0]-w[used,0].min())*10:.0f} ns, duration {(w[mb,4]-w[mb,0])*10:.0f} ns")

if "--hist" in sys.argv:
    w = d[2, :, :, 0]
    used = np.where((w[:, 0] > 0) & (w[:, 4] > 0))[0]
    t0 = w[used, 0].min()
    dur = (w[used, 4] - w[used, 0]) * 10
    end = (w[used, 4] - t0) * 10
    order = np.argsort(end)
    print("k_update blocks by finish time (block, xcd, start, s0->1, s1->3, s3->4, end):")
    for b in used[order][-24:]:
        s = [(w[b, k] - t0) * 10 if w[b, k] > 0 else -1 for k in range(5)]
        print(f"  {b:3d} x{b%8} start {s[0]:6.0f}  {s[1]-s[0]:6.0f} {s[3]-s[1]:6.0f} {s[4]-s[3]:6.0f}  end {s[4]:6.0f}")

if "--upda" in sys.argv:
    w = d[2, :, :, 0]
    t0 = d[0, :, 0, 0][d[0, :, 0, 0] > 0].min()
    print("update-stamp slots >= 128 (merged launch tiles): blk start s0->1 s1->3 s3->4 end (ns from forward start)")
    for b in range(128, 512):
        if w[b, 0] > 0 and w[b, 4] > 0:
            s_ = [(w[b, k] - t0) * 10 if w[b, k] > 0 else -1 for k in range(5)]
            if b % 6 == 0:
                print(f"  {b:3d} x{b%8} start {s_[0]:6.0f}  {s_[1]-s_[0]:6.0f} {s_[3]-s_[1]:6.0f} {s_[4]-s_[3]:6.0f}  end {s_[4]:6.0f}")

if "--upda2" in sys.argv:
    w = d[2, :, :, 0]
    t0 = d[0, :, 0, 0][d[0, :, 0, 0] > 0].min()
    print("merged-launch tiles: blk start | 0->1 state issue | 1->2 dz3 | 2->5 frag issue | 5->6 gen | 6->3 mma+tile | 3->4 adam")
    for b in range(128, 512, 7):
        if w[b, 0] > 0 and w[b, 4] > 0:
            g = lambda k: (w[b, k] - t0) * 10
            print(f"  {b:3d} x{b%8} start {g(0):6.0f} | {g(1)-g(0):6.0f} {g(2)-g(1):6.0f} {g(5)-g(2):6.0f} {g(6)-g(5):6.0f} {g(3)-g(6):6.0f} {g(4)-g(3):6.0f}")
