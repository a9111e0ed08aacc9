"""Secondary measurements (not the headline metric): throughput of the relabel kernels
on BASELINE-config shapes, with the oracle timed on a bounded sample beside them.
Usage on the GPU box: python tools/bench_relabel.py"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import iqlpref_amd as ia  # noqa: E402
from iqlpref_amd.relabel import cvar_tail_mean_device  # noqa: E402
from oracle import relabel_oracle as ro  # noqa: E402

DEV = "cuda:0"
rng = np.random.default_rng(0)


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


out = {}
# ---- A13: reward MLP [37,256,256,1] over 1M transitions (ref:719-724) ----
N, D_IN = 1_000_000, 37
ws = [torch.from_numpy(rng.standard_normal(s).astype(np.float32) / np.sqrt(s[0])).to(DEV)
      for s in ((D_IN, 256), (256, 256), (256, 1))]
bs = [torch.zeros(s, device=DEV) for s in (256, 256, 1)]
x = torch.from_numpy(rng.standard_normal((N, D_IN)).astype(np.float32)).to(DEV)
t = timed(lambda: ia.mlp_forward_f32(ws, bs, x, w_in_out=True))
flops = 2.0 * N * (D_IN * 256 + 256 * 256 + 256)
xs = x[:20000].cpu().numpy()
wl = [a for pair in zip([w.cpu().numpy() for w in ws], [b.cpu().numpy() for b in bs]) for a in pair]
t0 = time.perf_counter(); ro.reward_mlp_forward(wl, xs); tc = (time.perf_counter() - t0) * N / 20000
out["mr_relabel_1M"] = {"gpu_ms": t * 1e3, "rows_per_s": N / t, "tflops_f32": flops / t / 1e12,
                        "mfma_f32_frac": flops / t / 157.3e12, "cpu_oracle_ms_extrapolated": tc * 1e3}

# ---- A14: ensemble S=20 and S=100 + CVaR tail (ref:1176-1187) ----
for S in (20, 100):
    preds = torch.empty((S, N), device=DEV)
    def ens():
        for k in range(S):
            preds[k] = ia.mlp_forward_f32(ws, bs, x, w_in_out=True)[:, 0]
        return cvar_tail_mean_device(preds, max(1, int((1 - 0.95) * S)))
    t = timed(ens, reps=1)
    tcv = timed(lambda: cvar_tail_mean_device(preds, max(1, int((1 - 0.95) * S))))
    out[f"ensemble_cvar_S{S}_1M"] = {"gpu_ms_total": t * 1e3, "cvar_kernel_ms": tcv * 1e3,
                                     "cvar_read_GBs": 4.0 * S * N / tcv / 1e9}
    del preds

# ---- A12: preference transformer, antmaze shapes, QL=100, one window per transition ----
S_, A_, QL, NW = 29, 8, 100, 200_000
p = ro.make_pt_params(rng, S_, A_, 1000, embd=64, pref=64, inter=256, layers=1)
m = ia.RewardPT(S_, A_, 1000, embd_dim=64, pref_attn_embd_dim=64, num_heads=4, intermediate_dim=256,
                num_layers=1, max_pos=256)
m.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()}, strict=False)
m = m.to(DEV)
obs = torch.from_numpy(rng.standard_normal((NW + QL, S_)).astype(np.float32)).to(DEV)
act = torch.from_numpy(rng.uniform(-1, 1, (NW + QL, A_)).astype(np.float32)).to(DEV)
starts = torch.arange(NW, device=DEV, dtype=torch.int64)
lens = torch.full((NW,), QL, device=DEV, dtype=torch.int32)
t = timed(lambda: m.window_values(obs, act, starts, lens, QL), reps=2)
nb = 64
sts = np.stack([obs[i:i + QL].cpu().numpy() for i in range(nb)]); acs = np.stack([act[i:i + QL].cpu().numpy() for i in range(nb)])
t0 = time.perf_counter()
ro.pt_value_last(p, sts, acs, np.tile(np.arange(QL), (nb, 1)), np.ones((nb, QL), np.float32))
tc = (time.perf_counter() - t0) / nb
out["pt_relabel_QL100"] = {"windows": NW, "gpu_ms": t * 1e3, "windows_per_s": NW / t,
                           "cpu_oracle_windows_per_s": 1.0 / tc}
print(json.dumps(out, indent=1))
