"""Experiment: G seed groups of K seeds each, every group on its own HIP stream (the kernels of
different groups may overlap: one group's HBM-bound k_update beside another's latency-bound
k_forward).  Usage on the GPU box: python tools/group_streams.py "GxK" ...   (default 1x4 2x2 4x1 2x4 4x2 1x8)"""
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
# CU_MASK=1: every stream gets its own slice of the CUs (hipExtStreamCreateWithCUMask): the
# sub-groups then never share a CU, only the memory system
CU_MASK = os.environ.get("CU_MASK", "0")
combos = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(1, 4), (2, 2), (4, 1), (2, 4), (4, 2), (1, 8)]


def masked_stream(g, G):
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    words = (n_cu + 31) // 32
    mask = (C.c_uint32 * words)()
    for cu in range(n_cu):
        # "interleave": CU i belongs to stream i % G; "block": contiguous ranges
        if CU_MASK == "interleave":
            owner = cu % G
        elif CU_MASK.startswith("run"):  # runs of r consecutive CUs alternate between the streams
            owner = (cu // int(CU_MASK[3:])) % G
        else:
            owner = cu * G // n_cu
        if owner == g:
            mask[cu // 32] |= 1 << (cu % 32)
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), words, mask)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value, device=dev)

buf = ia.ReplayBuffer(bench.S_DIM, bench.A_DIM, 200_000, dev)
buf.load_d4rl_dataset(bench.synth_dataset(1, 200_000))
for G, K in combos:
    groups, streams = [], []
    for g in range(G):
        trs = [bench.build_trainer(ia, torch, dev, 10 + g * K + i, "bf16") for i in range(K)]
        groups.append(ia.SeedGroup(trs, mode="group") if K > 1 else trs[0])
        streams.append(masked_stream(g, G) if CU_MASK != "0" else torch.cuda.Stream(device=dev))

    def run(n, chunk=500):
        for _ in range(n // chunk):
            for g, s in zip(groups, streams):
                with torch.cuda.stream(s):
                    if K > 1:
                        g.train_steps(buf, chunk, bench.BATCH, graph_unroll=int(os.environ.get('GS_UNROLL', '50')))
                    else:
                        g.train_steps(buf, chunk, bench.BATCH, return_losses=False, graph_unroll=int(os.environ.get('GS_UNROLL', '50')))
    run(1000)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10_000
    run(n)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{G}x{K}", json.dumps({"steps_per_s": G * K * n / dt, "us_per_group_step": dt / n * 1e6}), flush=True)
    for g in groups:
        if K > 1:
            g.close()
    del groups
