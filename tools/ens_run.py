"""One (critics, batch) configuration of the step on its own (for rocprofv3 passes and A/B runs):
    python tools/ens_run.py [E=4] [B=1024] [steps=3000] [pen]
prints steps/s, algorithmic bytes per step (iqlhip_step_cost) and the HIP-event time of each kernel.
`pen`: BASELINE configs[2] instead of the antmaze shapes (S 45 / A 24, actor dropout 0.1, 5,000 rows)."""
import ctypes as C
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import iqlpref_amd as ia  # noqa: E402
from iqlpref_amd import _lib  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
n = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
unroll = int(os.environ.get("ENS_UNROLL", "50"))
dev = "cuda:0"
lib = _lib.load()
if "pen" in sys.argv[4:]:
    S_, A_ = bench.PEN["dims"]
    buf = ia.ReplayBuffer(S_, A_, bench.PEN["n_rows"], dev)
    buf.load_d4rl_dataset(bench.synth_dataset_dims(1, bench.PEN["n_rows"], S_, A_))
    tr = bench.build_trainer(ia, torch, dev, 3, "bf16", n_critics=E, dims=bench.PEN["dims"],
                             dropout=bench.PEN["dropout"], hyper=bench.PEN["hyper"])
else:
    n_rows = int(os.environ.get("ENS_ROWS", "200000"))  # (bench.py's legs sample a 1M-row buffer)
    buf = ia.ReplayBuffer(bench.S_DIM, bench.A_DIM, n_rows, dev)
    buf.load_d4rl_dataset(bench.synth_dataset(1, n_rows))
    tr = bench.build_trainer(ia, torch, dev, 3, "bf16", n_critics=E)
tr.train_steps(buf, min(500, n), B, return_losses=False, graph_unroll=unroll)
torch.cuda.synchronize()
t = time.perf_counter()
tr.train_steps(buf, n, B, return_losses=False, graph_unroll=unroll)
torch.cuda.synchronize()
dt = time.perf_counter() - t
out = {"E": E, "B": B, "steps_per_s": n / dt, "us_per_step": dt / n * 1e6}
by, fl = C.c_double(), C.c_double()
_lib.check(lib.iqlhip_step_cost(C.byref(tr._cfg(B)), C.byref(by), C.byref(fl)))
out["bytes_step"], out["flops_step"] = by.value, fl.value
if not os.environ.get("ENS_NO_TIMING"):
    _lib.check(lib.iqlhip_trainer_set_timing(tr._handle, 1))
    tr.train_steps(buf, 200, B, return_losses=False, graph_unroll=0)
    avg = (C.c_double * 3)()
    nl = C.c_int64()
    _lib.check(lib.iqlhip_trainer_get_timing(tr._handle, C.byref(avg), C.byref(nl)))
    out["kernel_us_events"] = {"k_forward": avg[0] * 1e3, "k_backward": avg[1] * 1e3, "k_update": avg[2] * 1e3}
print(json.dumps(out), flush=True)
