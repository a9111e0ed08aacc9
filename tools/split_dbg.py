"""Diagnostic: does SeedGroup(mode="split") keep its speed when it is created after other groups
came and went?  (CU-slice streams created after a stream had been destroyed once shared a hardware
queue: 91k instead of 190k steps/s; the library now keeps one capture stream per thread and the
slice streams for the life of the process.)  Usage on the GPU box: python tools/split_dbg.py [--group-first]"""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import iqlpref_amd as ia
dev = "cuda:0"
buf = ia.ReplayBuffer(bench.S_DIM, bench.A_DIM, 200_000, dev)
buf.load_d4rl_dataset(bench.synth_dataset(1, 200_000))
def rate(group, k_, n_):
    group.train_steps(buf, 1_000, bench.BATCH, graph_unroll=50)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    group.train_steps(buf, n_, bench.BATCH, graph_unroll=50)
    torch.cuda.synchronize()
    return k_ * n_ / (time.perf_counter() - t1)
trs = [bench.build_trainer(ia, torch, dev, 10 + i, "bf16") for i in range(8)]
if "--group-first" in sys.argv:  # bench.py's order: groups come and go before the first split group
    trs[0].train_steps(buf, 2000, bench.BATCH, return_losses=False, graph_unroll=50)
    for k in (2, 4):
        g = ia.SeedGroup(trs[:k], mode="group")
        print("group", k, rate(g, k, 5000), flush=True)
        g.close()
g = ia.SeedGroup(trs, mode="split", n_streams=2)
print("fresh split 2x4", rate(g, 8, 10000), flush=True)
print("again", rate(g, 8, 10000), flush=True)
g.close()
g = ia.SeedGroup(trs, mode="group")
print("group 1x8", rate(g, 8, 10000), flush=True)
g.close()
g = ia.SeedGroup(trs, mode="split", n_streams=2)
print("split after group", rate(g, 8, 10000), flush=True)
g.close()
# a solo run on the default stream before
trs[0].train_steps(buf, 2000, bench.BATCH, return_losses=False, graph_unroll=50)
torch.cuda.synchronize()
g = ia.SeedGroup(trs, mode="split", n_streams=2)
print("split after solo", rate(g, 8, 10000), flush=True)
g.close()
