"""Diagnostic: per-work-group stage timeline of one step (IQL_STAMPS build).
Usage on the GPU box:  python tools/stamps.py [bf16|fp32] [--detail]
(expects iqlpref_amd/libiqlhip_stamps.so: python -m iqlpref_amd.build --stamps)"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# an explicitly named build is loaded as it is (no source-tag check): the diagnostic library
os.environ["IQLHIP_LIB"] = os.path.join(ROOT, "iqlpref_amd", os.environ.get("STAMP_LIB", "libiqlhip_stamps.so"))
from iqlpref_amd import _lib  # noqa: E402
import iqlpref_amd as ia  # noqa: E402
import bench  # noqa: E402

lib = _lib.load()
lib.iqlhip_trainer_set_debug.restype = C.c_int
lib.iqlhip_trainer_set_debug.argtypes = [C.c_void_p, C.c_void_p]
dev = "cuda:0"
data = bench.synth_dataset(1, 200_000)
buf = ia.ReplayBuffer(bench.S_DIM, bench.A_DIM, 200_000, dev)
buf.load_d4rl_dataset(data)
prec = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "bf16"
# STAMP_E / STAMP_B: critics and batch of the stamped configuration (default: the headline's 2 / 256)
_E, _B = int(os.environ.get("STAMP_E", "2")), int(os.environ.get("STAMP_B", str(bench.BATCH)))
bench.BATCH = _B
tr = bench.build_trainer(ia, torch, dev, 1, prec, n_critics=_E)
# STAMP_GROUP=K: the stamped trainer is member 0 of a seed group of K (one launch sequence,
# gridDim.y = K): the timeline of ITS work-groups inside the group launches
_K = int(os.environ.get("STAMP_GROUP", "1"))
grp = None
dbg = torch.zeros((3, 512, 8, 2), dtype=torch.int64, device=dev)
dbg_all = [dbg]
if _K > 1:
    # the group copies its members' descriptors when it is created: attach the buffers first
    # (every member gets its own: STAMP_ALL=1 prints when each member's work-groups ran)
    members = [tr] + [bench.build_trainer(ia, torch, dev, 1 + i, prec, n_critics=_E) for i in range(1, _K)]
    dbg_all = [dbg] + [torch.zeros_like(dbg) for _ in range(1, _K)]
    for t_, d_ in zip(members, dbg_all):
        t_._ensure_handle(bench.BATCH)
        _lib.check(lib.iqlhip_trainer_set_debug(t_._handle, C.c_void_p(d_.data_ptr())))
    grp = ia.SeedGroup(members, mode="group")
    grp.train_steps(buf, 200, bench.BATCH, graph_unroll=0)
    torch.cuda.synchronize()
    for d_ in dbg_all:
        d_.zero_()
else:
    tr.train_steps(buf, 200, bench.BATCH, return_losses=False, graph_unroll=0)
    _lib.check(lib.iqlhip_trainer_set_debug(tr._handle, C.c_void_p(dbg.data_ptr())))
# STAMP_GRAPH=n: n steps replayed as one hipGraph (the stamps of the LAST step survive): kernel
# starts in steady state instead of after a host-side gap
_g = int(os.environ.get("STAMP_GRAPH", "0"))
if grp is not None:
    grp.train_steps(buf, max(_g, 1), bench.BATCH, graph_unroll=_g)
else:
    tr.train_steps(buf, max(_g, 1), bench.BATCH, return_losses=False, graph_unroll=_g)
torch.cuda.synchronize()
names = ["k_forward", "k_backward", "k_update"]
if _K > 1 and os.environ.get("STAMP_ALL"):
    # when did the work-groups of each member run inside the group launches (one line per kernel
    # and member: first start, last end, median / max work-group duration; times from the first
    # work-group start of that kernel over all members)
    allw = [d_.cpu().numpy().astype(np.float64)[:, :, :, 0] for d_ in dbg_all]
    for k in range(3):
        starts = [w[k][w[k][:, 0] > 0][:, 0].min() for w in allw if (w[k][:, 0] > 0).any()]
        if not starts:
            continue
        t0 = min(starts)
        tend = max(w[k].max() for w in allw)
        print(f"{names[k]}: whole launch span {(tend - t0) * 10:.0f} ns over {_K} members")
        for mi, w in enumerate(allw):
            ww = w[k][w[k][:, 0] > 0]
            dur = (ww.max(axis=1) - ww[:, 0]) * 10
            busy = dur > 200  # (idle / padding work-groups write only their start stamp)
            st = (ww[:, 0] - t0) * 10
            print(f"   member {mi}: {len(ww):4d} blocks  start {st.min():6.0f} .. {st.max():6.0f}  "
                  f"end max {(ww.max() - t0) * 10:6.0f}  duration med {np.median(dur[busy]):6.0f} max {dur.max():6.0f}")
    sys.exit(0)
d = dbg_all[int(os.environ.get("STAMP_MEMBER", "0"))].cpu().numpy().astype(np.float64)  # which member's timeline
t_first = None
for k in range(3):
    w = d[k, :, :, 0]
    used = w[:, 0] > 0
    nb = int(used.sum())
    if nb == 0:
        continue
    w = w[used]
    t0 = w[:, 0].min()
    if t_first is None:
        t_first = t0
    last = w.max()
    print(f"{names[k]}: {nb} blocks; first start at +{(t0 - t_first) * 10:.0f} ns; kernel span "
          f"{(last - t0) * 10:.0f} ns; block start skew max {(w[:, 0].max() - t0) * 10:.0f} ns")
    # group blocks by the set of stamps they wrote (= block kind), then stage deltas per kind
    kinds = {}
    for b in range(w.shape[0]):
        kinds.setdefault(tuple(np.nonzero(w[b] > 0)[0]), []).append(b)
    for slots, blks in sorted(kinds.items(), key=lambda kv: -len(kv[1])):
        ww = w[blks]
        line = f"   kind slots={list(slots)} n={len(blks)}: "
        order = sorted(slots, key=lambda sl: np.median(ww[:, sl]))
        for a_, b_ in zip(order[:-1], order[1:]):
            dt = (ww[:, b_] - ww[:, a_]) * 10
            line += f" {a_}->{b_} med {np.median(dt):6.0f} max {dt.max():6.0f} |"
        # (slots are numbered in the order they were added to the source, not in time: the work-group's end
        # is its LATEST stamp)
        tot = (ww[:, list(slots)].max(axis=1) - ww[:, slots[0]]) * 10
        end = (ww[:, list(slots)].max(axis=1) - t0) * 10
        line += f" total med {np.median(tot):6.0f} max {tot.max():6.0f}; end max {end.max():6.0f}"
        print(line)

if "--detail" in sys.argv:
    kd = int(os.environ.get("STAMP_KERNEL", "2"))
    w = d[kd, :, :, 0]
    used = np.where(w[:, 0] > 0)[0]
    t0 = w[used, 0].min()
    for b in used:
        row = [(w[b, s] - t0) * 10 if w[b, s] > 0 else -1 for s in range(8)]
        print(f"{names[kd]} block {b:3d} xcd {b % 8}: " + " ".join(f"{x:7.0f}" for x in row))
