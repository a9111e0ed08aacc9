"""Secondary measurements (not the headline metric): throughput of the relabel kernels on the
BASELINE-config shapes, each against the roofline that bounds it.  ``leg()`` is what bench.py
embeds as its ``relabel`` object; run as a script it prints the same record (plus a bounded
oracle timing with --cpu).

    python tools/bench_relabel.py [--cpu]

Rooflines (MI355X_MICROARCH.md): exact-fp32 MFMA 157.3 TFLOP/s (reward MLP: 2 * sum(in * out)
flops per row); HBM 8 TB/s (CVaR tail mean: 4 S bytes per column read + 4 written; preference
transformer: the flops of one window on the fp32 vector / matrix units).
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

F32_PEAK_TFLOPS = 157.3
HBM_PEAK_GBS = 8000.0
VALU_PEAK_LANE_OPS = 256 * 4 * 16 * 2.4e9  # CUs x SIMDs x lanes per cycle x clock


def _timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


def _timed_device(fn, reps=4):
    """Device time of one call: HIP events on the launch stream around `reps` back-to-back calls
    (the host's share of a short call -- output allocation, the ctypes hop, the wake-up after the
    synchronisation: ~0.1 ms -- then hides behind the previous call's kernel)."""
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def _reward_mlp(rng, dev, d_in=37):
    ws = [torch.from_numpy(rng.standard_normal(s).astype(np.float32) / np.sqrt(s[0])).to(dev)
          for s in ((d_in, 256), (256, 256), (256, 1))]
    bs = [torch.from_numpy(rng.standard_normal(s).astype(np.float32) * 0.1).to(dev) for s in (256, 256, 1)]
    return ws, bs


def pt_flops_per_window(S, A, QL, E=64, inter=256):
    """fp32 multiply-adds x 2 of one relabel window (reward_models/pref_transformer.py:210-277,
    only what the last action token's value needs): embeddings, K / V projection of all 2 QL
    tokens, one query row of attention, the MLP and the value head of the last token."""
    T = 2 * QL
    emb = QL * (S + A) * E
    kv = T * E * 2 * E
    attn = E * E + 2 * T * E + E * E
    mlp = 2 * E * inter + E
    return 2.0 * (emb + kv + attn + mlp)


def leg(device="cuda:0", n_rows=1_000_000, pt_windows=200_000, cpu=False):
    import iqlpref_amd as ia
    from iqlpref_amd.relabel import cvar_tail_mean_device
    from oracle import relabel_oracle as ro

    rng = np.random.default_rng(0)
    out = {"note": "relabel kernels on BASELINE config shapes (configs[2] PT, configs[4] snapshot-ensemble "
                   "CVaR); one-off costs before training, reported beside `value`, never as it"}
    N, D_IN = n_rows, 37
    ws, bs = _reward_mlp(rng, device)
    x = torch.from_numpy(rng.standard_normal((N, D_IN)).astype(np.float32)).to(device)
    # ---- A13: reward MLP [37,256,256,1] over N transitions (ref:719-724) ----
    t = _timed(lambda: ia.mlp_forward_f32(ws, bs, x, w_in_out=True))
    flops = 2.0 * N * (D_IN * 256 + 256 * 256 + 256)
    out["reward_mlp"] = {"workload": f"MLP [37,256,256,1] exact fp32, {N} transitions", "ms": t * 1e3,
                         "rows_per_s": N / t,
                         "roofline": {"bound": "mfma", "achieved": flops / t / 1e12, "peak": F32_PEAK_TFLOPS,
                                      "unit": "TFLOP/s", "frac": flops / t / 1e12 / F32_PEAK_TFLOPS}}
    # ---- A14: snapshot ensemble S = 20 end to end (forwards + tail mean, ref:1176-1187) ----
    S = 20
    preds = torch.empty((S, N), device=device)

    def ensemble():
        for k in range(S):
            preds[k] = ia.mlp_forward_f32(ws, bs, x, w_in_out=True)[:, 0]
        return cvar_tail_mean_device(preds, max(1, int((1 - 0.95) * S)))
    t = _timed(ensemble, reps=2)
    out["mr_ensemble_S20"] = {"workload": f"{S} snapshots x {N} transitions + CVaR(0.95)", "ms": t * 1e3,
                              "roofline": {"bound": "mfma", "achieved": S * flops / t / 1e12,
                                           "peak": F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                           "frac": S * flops / t / 1e12 / F32_PEAK_TFLOPS}}
    del preds
    # ---- CVaR tail mean alone, S = 20 / 100 / 500 (ref:1003-1011, 1185-1187) ----
    for S in (20, 100, 500):
        preds = torch.from_numpy(rng.standard_normal((S, 1)).astype(np.float32)).to(device) + \
            torch.randn((S, N), device=device)
        n_tail = max(1, int(np.floor((1 - 0.95) * S)))
        t = _timed(lambda: cvar_tail_mean_device(preds, n_tail))
        nbytes = 4.0 * S * N + 4.0 * N
        # the selection itself is compare work: ~13 counting passes x S (compare + count) per column on a
        # bracket that shrinks from [min, max] (cvar.hip; 34 passes as a plain bisection in round 2), or
        # n_tail passes of (compare, select, count, min) for tails of at most 8
        lane_ops = (4.0 * n_tail if n_tail <= 8 else 2.0 * 13) * S * N
        out[f"cvar_S{S}"] = {"workload": f"tail mean of the {n_tail} smallest of {S} x {N}", "ms": t * 1e3,
                             "roofline": {"bound": "hbm", "achieved": nbytes / t / 1e9, "peak": HBM_PEAK_GBS,
                                          "unit": "GB/s", "frac": nbytes / t / 1e9 / HBM_PEAK_GBS,
                                          "note": "exact order statistic (counting passes on a shrinking bracket: value-space "
                                                  "interpolation / value midpoint / key midpoint in turn, ~13 passes at "
                                                  "S = 500; successive distinct minima for n_tail <= 8): bound by the LDS "
                                                  "copy and vector compares, not HBM",
                                          "valu_frac": lane_ops / t / VALU_PEAK_LANE_OPS}}
        del preds
    # ---- f3: BNN posterior, 500 weight sets x N transitions end to end (ref:978-1011) ----
    if n_rows >= 1_000_000:
        S = 500
        sets = [_reward_mlp(rng, device) for _ in range(8)]  # 8 distinct sets cycled: the work is the same
        preds = torch.empty((S, N), device=device)

        def bnn():
            for k in range(S):
                w_, b_ = sets[k % len(sets)]
                preds[k] = ia.mlp_forward_f32(w_, b_, x, w_in_out=True)[:, 0]
            return cvar_tail_mean_device(preds, max(1, int((1 - 0.95) * S)))
        t = _timed(bnn, reps=1)
        out["bnn_S500"] = {"workload": f"{S} posterior samples x {N} transitions + CVaR(0.95), [S, N] fp32 "
                                       f"prediction matrix ({4.0 * S * N / 1e9:.1f} GB) resident in HBM",
                           "ms": t * 1e3,
                           "roofline": {"bound": "mfma", "achieved": S * flops / t / 1e12, "peak": F32_PEAK_TFLOPS,
                                        "unit": "TFLOP/s", "frac": S * flops / t / 1e12 / F32_PEAK_TFLOPS}}
        del preds
    # ---- A12: preference transformer ----
    for tag, S_, A_, QL, NW in (("pt_pen_config3", 45, 24, 100, 5_000),
                                ("pt_antmaze_correct_offsets", 29, 8, 100, pt_windows)):
        p = ro.make_pt_params(rng, S_, A_, 1000, embd=64, pref=64, inter=256, layers=1)
        m = ia.RewardPT(S_, A_, 1000, embd_dim=64, pref_attn_embd_dim=64, num_heads=4, intermediate_dim=256,
                        num_layers=1, max_pos=256)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()}, strict=False)
        m = m.to(device)
        obs = torch.from_numpy(rng.standard_normal((NW + QL, S_)).astype(np.float32)).to(device)
        act = torch.from_numpy(rng.uniform(-1, 1, (NW + QL, A_)).astype(np.float32)).to(device)
        starts = torch.arange(NW, device=device, dtype=torch.int64)
        lens = torch.full((NW,), QL, device=device, dtype=torch.int32)
        t = _timed(lambda: m.window_values(obs, act, starts, lens, QL), reps=2)
        td = _timed_device(lambda: m.window_values(obs, act, starts, lens, QL))
        fl = pt_flops_per_window(S_, A_, QL) * NW
        out[tag] = {"workload": f"{NW} windows, QL={QL}, S={S_} A={A_}, embd 64, 1 block", "ms": t * 1e3,
                    "ms_device": td * 1e3, "windows_per_s": NW / t,
                    "roofline": {"bound": "mfma", "achieved": fl / td / 1e12, "peak": F32_PEAK_TFLOPS,
                                 "unit": "TFLOP/s", "frac": fl / td / 1e12 / F32_PEAK_TFLOPS,
                                 "frac_wall": fl / t / 1e12 / F32_PEAK_TFLOPS,
                                 "note": "fp32 flops the window needs / device time of the call (HIP events, "
                                         "calls back to back), against the fp32 matrix peak; frac_wall: against "
                                         "the host's wall time of ONE call incl. allocation and wake-up"}}
        if cpu and tag == "pt_antmaze_correct_offsets":
            nb = 64
            sts = np.stack([obs[i:i + QL].cpu().numpy() for i in range(nb)])
            acs = np.stack([act[i:i + QL].cpu().numpy() for i in range(nb)])
            t0 = time.perf_counter()
            ro.pt_value_last(p, sts, acs, np.tile(np.arange(QL), (nb, 1)), np.ones((nb, QL), np.float32))
            out[tag]["cpu_oracle_windows_per_s"] = nb / (time.perf_counter() - t0)
    if cpu:
        xs = x[:20000].cpu().numpy()
        wl = [a for pair in zip([w.cpu().numpy() for w in ws], [b.cpu().numpy() for b in bs]) for a in pair]
        t0 = time.perf_counter()
        ro.reward_mlp_forward(wl, xs)
        out["reward_mlp"]["cpu_oracle_ms_extrapolated"] = (time.perf_counter() - t0) * N / 20000 * 1e3
    return out


if __name__ == "__main__":
    print(json.dumps(leg(cpu="--cpu" in sys.argv), indent=1))
