"""Per-kernel times of a few (critics, batch) configurations (HIP-event timing mode)."""
import sys, ctypes as C, torch, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, iqlpref_amd as ia
from iqlpref_amd import _lib
dev='cuda:0'
data=bench.synth_dataset(1, 200_000)
buf=ia.ReplayBuffer(bench.S_DIM, bench.A_DIM, 200_000, dev); buf.load_d4rl_dataset(data)
lib=_lib.load()
for E,B in ((2,256),(2,1024),(4,1024),(8,1024)):
    tr=bench.build_trainer(ia, torch, dev, 3, 'bf16', n_critics=E)
    tr.train_steps(buf, 200, B, return_losses=False, graph_unroll=50)
    torch.cuda.synchronize()
    import time
    t=time.perf_counter(); tr.train_steps(buf, 2000, B, return_losses=False, graph_unroll=50); torch.cuda.synchronize(); dt=time.perf_counter()-t
    _lib.check(lib.iqlhip_trainer_set_timing(tr._handle, 1))
    tr.train_steps(buf, 200, B, return_losses=False, graph_unroll=0)
    avg=(C.c_double*3)(); n=C.c_int64()
    _lib.check(lib.iqlhip_trainer_get_timing(tr._handle, C.byref(avg), C.byref(n)))
    print(f"E={E} B={B}: {2000/dt:.0f} steps/s ({dt/2000*1e6:.1f} us/step); events fwd/bwd/upd us = {avg[0]*1e3:.1f} {avg[1]*1e3:.1f} {avg[2]*1e3:.1f}")
