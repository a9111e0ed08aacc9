"""Diagnostic: device time of the reward MLP [37,256,256,1] over 1M rows (k_mlp_f32), library named by
IQLHIP_LIB.  Usage on the GPU box: IQLHIP_LIB=... python tools/mlp_scan.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import iqlpref_amd as ia  # noqa: E402
from tools import bench_relabel as br  # noqa: E402

dev = "cuda:0"
rng = np.random.default_rng(0)
N = int(os.environ.get("MLP_ROWS", "1000000"))
ws, bs = br._reward_mlp(rng, dev)
x = torch.from_numpy(rng.standard_normal((N, 37)).astype(np.float32)).to(dev)
ref = None
for rep in range(3):
    td = br._timed_device(lambda: ia.mlp_forward_f32(ws, bs, x, w_in_out=True), reps=8)
    flops = 2.0 * N * (37 * 256 + 256 * 256 + 256)
    print(os.path.basename(os.environ.get("IQLHIP_LIB", "libiqlhip.so")), f"{td * 1e3:.3f} ms",
          f"frac {flops / td / 1e12 / br.F32_PEAK_TFLOPS:.3f}", flush=True)
out = ia.mlp_forward_f32(ws, bs, x, w_in_out=True)
print("checksum", float(out.double().sum()), float(out.abs().double().max()))
