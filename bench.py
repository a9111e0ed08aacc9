#!/usr/bin/env python3
"""Headline benchmark: IQL gradient steps/sec, batch 256, 1M-transition device
replay buffer (BASELINE.json configs[1], antmaze-medium-diverse-v2 shapes).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

One step = ReplayBuffer.sample + ImplicitQLearning.train (V/Q/actor updates,
Polyak, cosine LR) = reference iql.py:1535-1536.  Every rank trains its own seed
on its own synthetic dataset (no data-path collective; the only collective is
the final all-gather of the per-rank metric record).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_BF16_PEAK_TFLOPS = 2500.0
MFMA_F32_PEAK_TFLOPS = 157.3

S_DIM, A_DIM, HIDDEN, BATCH, N_ROWS = 29, 8, 256, 256, 1_000_000
HYPER = dict(beta=10.0, iql_tau=0.9, discount=0.99, tau=0.005, max_steps=1_000_000)


def synth_dataset(seed, n=N_ROWS):
    """obs ~ N(0,1), act ~ U(-1,1), sparse reward - 1."""
    rng = np.random.default_rng(seed)
    return {
        "observations": rng.standard_normal((n, S_DIM), dtype=np.float32),
        "actions": rng.uniform(-1, 1, (n, A_DIM)).astype(np.float32),
        "rewards": (rng.uniform(size=n) < 0.01).astype(np.float32) - 1.0,
        "next_observations": rng.standard_normal((n, S_DIM), dtype=np.float32),
        "terminals": (rng.uniform(size=n) < 1e-3).astype(np.float32),
    }


def build_trainer(ia, torch, device, seed, precision, n_critics=2):
    torch.manual_seed(seed)
    q = (ia.TwinQ(S_DIM, A_DIM) if n_critics == 2 else ia.EnsembleQ(S_DIM, A_DIM, n_critics=n_critics)).to(device)
    v = ia.ValueFunction(S_DIM).to(device)
    actor = ia.GaussianPolicy(S_DIM, A_DIM, 1.0).to(device)
    vo = torch.optim.Adam(v.parameters(), lr=3e-4)
    qo = torch.optim.Adam(q.parameters(), lr=3e-4)
    ao = torch.optim.Adam(actor.parameters(), lr=3e-4)
    return ia.ImplicitQLearning(
        max_action=1.0, actor=actor, actor_optimizer=ao, q_network=q, q_optimizer=qo,
        v_network=v, v_optimizer=vo, device=device, precision=precision, seed=seed, **HYPER)


def cpu_baseline(data, seconds=12.0):
    """The oracle (numpy port of the reference step) on the host cores, fp32 BLAS,
    same shapes, bounded sample.  Checker code timed as a baseline, never shipped."""
    import torch
    from oracle import iql_oracle as orc
    from oracle import philox

    torch.manual_seed(0)
    import iqlpref_amd as ia
    q, v, a = ia.TwinQ(S_DIM, A_DIM), ia.ValueFunction(S_DIM), ia.GaussianPolicy(S_DIM, A_DIM, 1.0)
    sd = lambda m: {k: t.detach().numpy() for k, t in m.state_dict().items()}
    o = orc.IQLOracle(sd(q), sd(v), sd(a), mode="fp32", **HYPER)
    n = data["observations"].shape[0]
    for t in range(3):
        o.train(orc.gather_batch(data, philox.sample_indices(0, t, BATCH, n)))
    t0, steps = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        o.train(orc.gather_batch(data, philox.sample_indices(0, 3 + steps, BATCH, n)))
        steps += 1
    dt = time.perf_counter() - t0
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        cores = os.cpu_count() or 1
    return {"value": steps / dt, "unit": "steps/s", "cores": int(cores), "kind": "port",
            "sample": f"{steps} steps in {dt:.1f} s of oracle/iql_oracle.py (numpy fp32 BLAS), "
                      f"same shapes (S=29 A=8 H=256 B=256, N=1M)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50_000)
    ap.add_argument("--warmup", type=int, default=5_000)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--unroll", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ensemble-q", type=int, default=4,
                    help="extra leg (N=1 only): BASELINE configs[4], an E-critic ensemble at batch 1024 "
                         "(same antmaze shapes); 0 disables.  Reported beside `value`, never as it")
    ap.add_argument("--agents-per-gpu", type=int, default=4,
                    help="extra leg (N=1 only): aggregate steps/s of this many independent seeds sharing "
                         "the GPU on separate streams -- the reference launcher's AGENTS_PER_GPU "
                         "(ensemble_sweeps/launch.sh:12); 0 disables.  `value` is always 1 seed per GPU")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (there is no CPU path)")
    # rehearsal knob (1-GPU box): IQL_BENCH_BACKEND=gloo puts every rank on cuda:0 and runs the
    # collectives over gloo; the driver's multi-GPU runs use the default (one GPU per rank, RCCL)
    backend = os.environ.get("IQL_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    cdev = device if backend == "nccl" else "cpu"  # where collective payloads live
    if world > 1:
        kw = {"device_id": torch.device(device)} if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)

    import iqlpref_amd as ia
    from iqlpref_amd import _lib
    import ctypes as C

    seed = 1 + rank  # independent seed + dataset per rank
    data = synth_dataset(seed)
    buf = ia.ReplayBuffer(S_DIM, A_DIM, N_ROWS, device)
    buf.load_d4rl_dataset(data)
    tr = build_trainer(ia, torch, device, seed, args.precision)

    def run(n):
        done = 0
        while done < n:
            c = min(n - done, 20_000)
            tr.train_steps(buf, c, BATCH, return_losses=False, graph_unroll=args.unroll)
            done += c

    def barrier():
        if world > 1:
            dist.barrier()

    run(args.warmup)
    torch.cuda.synchronize()
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()  # same stream the library launches on (torch's current stream)
    run(args.steps)
    ev1.record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    barrier()
    dev_ms = ev0.elapsed_time(ev1)
    last = tr.train_steps(buf, 1, BATCH, graph_unroll=0).cpu().numpy()[0]
    if not np.isfinite(last).all():
        raise SystemExit(f"non-finite losses after the timed region: {last}")

    # max over ranks of the wall time
    tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        rec = torch.tensor([seed, tr.total_it, *last.tolist()], dtype=torch.float64, device=cdev)
        recs = [torch.zeros_like(rec) for _ in range(world)]
        dist.all_gather(recs, rec)  # RCCL: the path's only collective (metric record)
    dt_max = float(tmax.item())

    if rank == 0:
        cfg = tr._cfg(BATCH)
        bytes_step, flops_step = C.c_double(), C.c_double()
        _lib.check(_lib.load().iqlhip_step_cost(C.byref(cfg), C.byref(bytes_step), C.byref(flops_step)))
        steps_per_s = world * args.steps / dt_max
        step_us_dev = dev_ms * 1e3 / args.steps
        achieved = bytes_step.value / (step_us_dev * 1e-6) / 1e9  # GB/s, device time of this rank
        # per-kernel breakdown (diagnostic pass with HIP events around every launch)
        lib = _lib.load()
        _lib.check(lib.iqlhip_trainer_set_timing(tr._handle, 1))
        tr.train_steps(buf, 300, BATCH, return_losses=False, graph_unroll=0)
        avg = (C.c_double * 3)()
        nl = C.c_int64()
        _lib.check(lib.iqlhip_trainer_get_timing(tr._handle, C.byref(avg), C.byref(nl)))
        _lib.check(lib.iqlhip_trainer_set_timing(tr._handle, 0))
        # algorithmic bytes: the gather term belongs to k_forward, every
        # parameter / moment / target byte is moved by k_update -- the dominant kernel
        gather_bytes = 4.0 * BATCH * (2 * S_DIM + A_DIM + 2)
        upd_bytes = bytes_step.value - gather_bytes
        # HIP events on the launch stream give each kernel's share of a step; the shares are
        # scaled so that the three launches tile the measured device time per step, which is how
        # rocprofv3 --kernel-trace attributes the timeline (dispatch to completion, back to back):
        # profiles/r01_e_kernel_stats.csv is the cross-check
        ev_us = [avg[k] * 1e3 for k in range(3)]
        scale = step_us_dev / sum(ev_us) if sum(ev_us) > 0 else 1.0
        upd_us = ev_us[2] * scale
        achieved = upd_bytes / (upd_us * 1e-6) / 1e9  # GB/s of k_update
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_e_traffic.json")
        if os.path.exists(tpath):  # PMC pass (separate rocprofv3 --pmc runs), bytes per launch of k_update
            with open(tpath) as f:
                traffic = json.load(f).get("k_update_bytes_per_launch")
        out = {
            "metric": "iql_grad_steps_per_sec", "value": steps_per_s, "unit": "steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt_max * 1e3 / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "IQL antmaze-medium-diverse-v2 shapes (S=29 A=8 H=256), "
                                   "1M-transition device replay, batch 256, one seed per GPU",
                       "batch": BATCH, "buffer_rows": N_ROWS, "graph_unroll": args.unroll},
            "roofline": {
                "kernel": "k_update", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "algorithmic_bytes_per_launch": upd_bytes, "launch_us": upd_us,
                "launch_us_events_only": ev_us[2], "launches_timed": int(nl.value),
                "step": {"bytes_per_step": bytes_step.value, "device_us_per_step": step_us_dev,
                         "achieved_gbs": bytes_step.value / (step_us_dev * 1e-6) / 1e9,
                         "frac": bytes_step.value / (step_us_dev * 1e-6) / 1e9 / HBM_PEAK_GBS,
                         "kernel_us": {"k_forward": ev_us[0] * scale, "k_backward": ev_us[1] * scale,
                                       "k_update": ev_us[2] * scale},
                         "kernel_us_events_only": {"k_forward": ev_us[0], "k_backward": ev_us[1],
                                                   "k_update": ev_us[2]},
                         "mfma_tflops": flops_step.value / (step_us_dev * 1e-6) / 1e12,
                         "mfma_peak_tflops": MFMA_BF16_PEAK_TFLOPS if args.precision == "bf16"
                         else MFMA_F32_PEAK_TFLOPS},
            },
        }
        if world == 1 and args.agents_per_gpu > 1:
            # independent seeds on independent streams; same dataset (a sweep varies the seed only)
            A_ = args.agents_per_gpu
            trs = [tr] + [build_trainer(ia, torch, device, seed + 100 + i, args.precision) for i in range(1, A_)]
            group = ia.SeedGroup(trs)
            group.train_steps(buf, 2_000, BATCH, graph_unroll=args.unroll)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            n_multi = 20_000
            group.train_steps(buf, n_multi, BATCH, graph_unroll=args.unroll)
            torch.cuda.synchronize()
            dt_m = time.perf_counter() - t1
            out["agents_per_gpu"] = {"agents": A_, "value": A_ * n_multi / dt_m, "unit": "steps/s",
                                     "steps_per_agent": n_multi,
                                     "note": "aggregate of independent seeds sharing one GPU; not `value`"}
        if world == 1 and args.ensemble_q >= 2:
            E_ = args.ensemble_q
            tre = build_trainer(ia, torch, device, seed + 50, args.precision, n_critics=E_)
            tre.train_steps(buf, 1_000, 1024, return_losses=False, graph_unroll=args.unroll)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            n_e = 5_000
            tre.train_steps(buf, n_e, 1024, return_losses=False, graph_unroll=args.unroll)
            torch.cuda.synchronize()
            dt_e = time.perf_counter() - t1
            out["ensemble_q"] = {"n_critics": E_, "batch": 1024, "value": n_e / dt_e, "unit": "steps/s",
                                 "transitions_per_s": 1024 * n_e / dt_e,
                                 "note": "BASELINE configs[4] (E-way critic ensemble, batch 1024); not `value`"}
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(data)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
