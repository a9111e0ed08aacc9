#!/usr/bin/env python3
"""Headline benchmark: IQL gradient steps/sec, batch 256, 1M-transition device
replay buffer (BASELINE.json configs[1], antmaze-medium-diverse-v2 shapes).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

One step = ReplayBuffer.sample + ImplicitQLearning.train (V/Q/actor updates,
Polyak, cosine LR) = reference iql.py:1535-1536.  Every rank trains its own seed
on its own synthetic dataset (no data-path collective; the only collective is
the all-gather of the per-rank metric record).  Rank 0 prints ONE JSON line.

``--gpus N`` with N > 1 and no WORLD_SIZE in the environment starts the N rank
processes itself (one per GPU, RCCL), before anything touches a GPU -- the
reference's process-per-GPU model (ensemble_sweeps/launch.sh:84-94).

Timing: W untimed warm-up steps, then blocks of EXACTLY K steps -- plain kernel
launches from the library's C loop (the default; hipGraphs of --unroll steps on
request: every graph launch costs ~5 us of device time that back-to-back kernels
do not, and the GPU starts with the first launch instead of behind the host's
work for a 60-node graph: 318-322 us against 322-325 us per 20-step block) --
bracketed by barrier + synchronize on both sides.  The block is repeated until >= 0.25 s of
timed work has accumulated; `ms_per_step` is the median block (max over ranks)
divided by K, `value` = N * K / that block time.
"""
import argparse
import json
import os
import socket
import subprocess
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

# The reference itself (algorithms/offline/iql.py, stub-imported, device="cpu", bf16 autocast,
# eager) timed in the build container during the survey: BASELINE.md section 2.  The Python
# reference cannot travel to the GPU box, so this figure is carried, not re-measured.
REFERENCE_CPU = {"value": 125.3, "unit": "steps/s", "cores": 8, "kind": "reference",
                 "where": "build container (8 vCPU Xeon @ 2.1 GHz, torch 2.10 CPU), not this box",
                 "source": "BASELINE.md section 2 (S=29 A=8 B=256, N=1M, 300 timed steps)"}


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


def build_trainer(ia, torch, device, seed, precision, n_critics=2, dims=None, dropout=None, hyper=None,
                  hidden_dim=256, n_hidden=2):
    S, A = dims or (S_DIM, A_DIM)
    torch.manual_seed(seed)
    kw = dict(hidden_dim=hidden_dim, n_hidden=n_hidden)
    q = (ia.TwinQ(S, A, **kw) if n_critics == 2 else ia.EnsembleQ(S, A, n_critics=n_critics, **kw)).to(device)
    v = ia.ValueFunction(S, **kw).to(device)
    actor = ia.GaussianPolicy(S, A, 1.0, dropout=dropout, **kw).to(device)
    vo = torch.optim.Adam(v.parameters(), lr=3e-4)
    qo = torch.optim.Adam(q.parameters(), lr=3e-4)
    ao = torch.optim.Adam(actor.parameters(), lr=3e-4)
    return ia.ImplicitQLearning(
        max_action=1.0, actor=actor, actor_optimizer=ao, q_network=q, q_optimizer=qo,
        v_network=v, v_optimizer=vo, device=device, precision=precision, seed=seed, **(hyper or HYPER))


# BASELINE configs[2] (pen-human-v1, configs/offline/iql/pen/human_v1.yaml: actor_dropout 0.1, beta 3,
# iql_tau 0.8, batch 256; 25 episodes x 200 transitions): the training step of the config whose
# relabel half is the `pt_pen_config3` entry of the relabel leg
PEN = dict(dims=(45, 24), n_rows=5_000, dropout=0.1,
           hyper=dict(beta=3.0, iql_tau=0.8, discount=0.99, tau=0.005, max_steps=1_000_000))


def synth_dataset_dims(seed, n, S, A):
    rng = np.random.default_rng(seed)
    return {"observations": rng.standard_normal((n, S), dtype=np.float32),
            "actions": rng.uniform(-1, 1, (n, A)).astype(np.float32),
            "rewards": rng.standard_normal(n).astype(np.float32),
            "next_observations": rng.standard_normal((n, S), dtype=np.float32),
            "terminals": (rng.uniform(size=n) < 5e-3).astype(np.float32)}


PROFILE_TAG = os.environ.get("IQL_PROFILE_TAG", "r04")


def profile_kernel_avg(fname, *needles):
    """AverageNs of the first kernel whose name contains every needle, out of a committed rocprofv3
    --kernel-trace --stats summary under profiles/ (None when the file or the kernel is missing)."""
    import csv
    path = os.path.join(ROOT, "profiles", fname)
    if not os.path.exists(path):
        return None
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if all(n in row.get("Name", "") for n in needles):
                name = row["Name"].replace("(anonymous namespace)::", "")
                return {"kernel": name.split("(")[0].replace("void iqlhip::", ""), "calls": int(row["Calls"]),
                        "kernel_avg_ns": float(row["AverageNs"])}
    return None


def profile_build(tag=None):
    """Build tag of the library the committed profile set <tag> was collected on
    (profiles/<tag>_meta.json; sets older than round 4 carry it in <tag>_traffic.json only)."""
    for name in (f"{tag or PROFILE_TAG}_meta.json", f"{tag or PROFILE_TAG}_traffic.json"):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            with open(path) as f:
                return json.load(f).get("build")
    return None


def roofline_block(kernel, bytes_launch, live_us, live_mode, prof_file, prof_needles, prof_mode, build, extra=None):
    """`roofline` of a leg.  `frac` / `achieved` follow from the COMMITTED rocprofv3 summary
    (profiles/<prof_file>: algorithmic bytes per launch / the kernel's average duration there), so a
    reader can recompute them from tracked files; the figure measured live in this run (HIP events on
    the launch stream) stands beside it under `live` with its own launch mode."""
    live = {"launch_us": live_us, "achieved": bytes_launch / (live_us * 1e-6) / 1e9 if live_us else None,
            "launch_mode": live_mode}
    live["frac"] = live["achieved"] / HBM_PEAK_GBS if live["achieved"] else None
    out = {"kernel": kernel, "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "algorithmic_bytes_per_launch": bytes_launch, "live": live}
    # (prof_file names this round's set; until it is committed the last round's file of the same kind
    # is used and labelled: matches_this_build says whether it was collected on the library running now)
    pk, ptag = None, None
    for ptag in (PROFILE_TAG, "r03"):
        cand = prof_file.replace(PROFILE_TAG + "_", ptag + "_", 1)
        pk = profile_kernel_avg(cand, *prof_needles)
        if pk is not None:
            prof_file = cand
            break
    if pk is not None:
        ach = bytes_launch / (pk["kernel_avg_ns"] * 1e-9) / 1e9
        pb = profile_build(ptag)
        out["from_profile"] = dict(pk, file=f"profiles/{prof_file}", achieved=ach, frac=ach / HBM_PEAK_GBS,
                                   launch_mode=prof_mode, build=pb, matches_this_build=(pb == build))
        out["achieved"], out["frac"], out["frac_source"] = ach, ach / HBM_PEAK_GBS, f"profiles/{prof_file}"
    else:
        out["achieved"], out["frac"], out["frac_source"] = live["achieved"], live["frac"], "live (no committed profile)"
    if extra:
        out.update(extra)
    return out


def kernel_times(_lib, tr, buf, batch, n=300):
    """HIP-event time of each of the three kernels of a step (diagnostic pass of the library), us."""
    import ctypes as C
    lib = _lib.load()
    _lib.check(lib.iqlhip_trainer_set_timing(tr._handle, 1))
    tr.train_steps(buf, n, batch, return_losses=False, graph_unroll=0)
    avg, nl = (C.c_double * 3)(), C.c_int64()
    _lib.check(lib.iqlhip_trainer_get_timing(tr._handle, C.byref(avg), C.byref(nl)))
    _lib.check(lib.iqlhip_trainer_set_timing(tr._handle, 0))
    return [avg[k] * 1e3 for k in range(3)], int(nl.value)


def step_leg(ia, torch, _lib, tr, buf, batch, n_steps, unroll, prof_file, prof_needles, label):
    """steps/s of one trainer + its roofline block (dominant kernel k_update, algorithmic bytes of
    iqlhip_step_cost minus the gather term)."""
    import ctypes as C
    tr.train_steps(buf, max(200, n_steps // 5), batch, return_losses=False, graph_unroll=unroll)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    tr.train_steps(buf, n_steps, batch, return_losses=False, graph_unroll=unroll)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t1
    by, fl = C.c_double(), C.c_double()
    cfg = tr._cfg(batch)
    _lib.check(_lib.load().iqlhip_step_cost(C.byref(cfg), C.byref(by), C.byref(fl)))
    ev, _ = kernel_times(_lib, tr, buf, batch, 200)
    step_us = dt / n_steps * 1e6
    scale = step_us / sum(ev) if sum(ev) > 0 else 1.0
    gather = 4.0 * batch * (2 * cfg.state_dim + cfg.action_dim + 2)
    mode = f"hipGraphs of {unroll} steps" if unroll else "plain launches"
    rf = roofline_block(label, by.value - gather, ev[2] * scale, mode + "; HIP-event shares scaled to tile the step",
                        prof_file, prof_needles, "hipGraphs of 50 steps under rocprofv3 --kernel-trace", _lib.build_tag(),
                        extra={"step": {"bytes_per_step": by.value, "us_per_step": step_us,
                                        "achieved_gbs": by.value / (step_us * 1e-6) / 1e9,
                                        "frac": by.value / (step_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                        "kernel_us_events_only": {"k_forward": ev[0], "k_backward": ev[1], "k_update": ev[2]},
                                        "mfma_tflops": fl.value / (step_us * 1e-6) / 1e12,
                                        "mfma_peak_tflops": MFMA_BF16_PEAK_TFLOPS}})
    return {"value": n_steps / dt, "unit": "steps/s", "us_per_step": step_us, "roofline": rf}


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
            "sample": f"{steps} steps in {dt:.1f} s of oracle/iql_oracle.py in mode=\"fp32\" (numpy fp32 BLAS: "
                      f"the faster of its two modes; the GPU leg and the reference's CPU path run bf16 "
                      f"autocast), same shapes (S=29 A=8 H=256 B=256, N=1M)",
            "arithmetic": "fp32",
            "reference_in_container": REFERENCE_CPU}


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n, argv):
    """Start n rank processes of this script (fresh interpreters: nothing here has touched a
    GPU yet) and wait for them.  Rank 0 inherits stdout and prints the JSON line."""
    import torch  # device_count() does not initialise HIP on this image

    backend = os.environ.get("IQL_BENCH_BACKEND", "nccl")
    visible = torch.cuda.device_count()
    if backend == "nccl" and visible < n:
        print(f"bench.py: --gpus {n} but only {visible} GPU(s) are visible; refusing to run fewer "
              f"ranks than asked for", file=sys.stderr)
        return 2
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=e,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0:
                    rc = rc or code
                    for q in pending:  # a dead rank leaves the others in a collective: end them
                        procs[q].terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50_000)
    ap.add_argument("--warmup", type=int, default=5_000)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--unroll", type=int, default=0,
                    help="steps per hipGraph; 0 (the default, and any block of fewer steps) = plain kernel launches "
                         "from the library's C loop, the faster mode for one seed (66.9k against 66.1k steps/s with "
                         "graphs of 50 steps; 20-step blocks 62.6k against 61.7k as one graph); "
                         "negative: ONE graph of min(-unroll, steps) steps (the round-2 behaviour, for A/B)")
    ap.add_argument("--min-timed-s", type=float, default=0.25,
                    help="the K-step block is repeated until this much timed work has accumulated")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sustained", action="store_true",
                    help="skip the long comparison region (rocprofv3 --pmc passes: its 60k queued dispatches "
                         "overrun the profiler's intercepted queue)")
    ap.add_argument("--ensemble-q", type=int, default=4,
                    help="extra leg (N=1 only): BASELINE configs[4], an E-critic ensemble at batch 1024 "
                         "(same antmaze shapes); 0 disables.  Reported beside `value`, never as it")
    ap.add_argument("--agents-per-gpu", type=int, default=8,
                    help="extra leg (N=1 only): aggregate steps/s of this many independent seeds sharing "
                         "the GPU -- the reference launcher's AGENTS_PER_GPU "
                         "(ensemble_sweeps/launch.sh:12); 0 disables.  `value` is always 1 seed per GPU")
    ap.add_argument("--no-relabel", action="store_true", help="skip the reward-relabel leg (N=1 only)")
    ap.add_argument("--no-pen", action="store_true", help="skip the config-3 (pen shapes + dropout) training leg (N=1 only)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))

    # stdout carries the ONE JSON line and nothing else: whatever the package prints on the way (the
    # reference-style "Dataset size" / "Training IQL" banners of the dataset loader and of train()) goes
    # to stderr
    json_out = sys.stdout
    sys.stdout = sys.stderr

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (there is no CPU path)")
    # rehearsal knob (1-GPU box): IQL_BENCH_BACKEND=gloo puts every rank on cuda:0 and runs the
    # collectives over gloo; the driver's multi-GPU runs use the default (one GPU per rank, RCCL)
    backend = os.environ.get("IQL_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local_rank = 0
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but {torch.cuda.device_count()} GPU(s) visible")
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
    K = args.steps
    if args.unroll < 0:
        unroll = max(1, min(-args.unroll, K))
    else:
        unroll = args.unroll if 0 < args.unroll <= K else 0  # 0: plain launches
    chunk = 20_000 if unroll == 0 else max(unroll, 20_000 // unroll * unroll)

    def run(n):
        done = 0
        while done < n:
            c = min(n - done, chunk)
            tr.train_steps(buf, c, BATCH, return_losses=False, graph_unroll=unroll)
            done += c

    def barrier():
        if world > 1:
            dist.barrier()

    run(args.warmup)
    run(K)  # untimed: instantiates the K-step graph the timed blocks replay
    torch.cuda.synchronize()

    # ---- timed blocks of exactly K steps ----
    block_s, block_dev_ms = [], []
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps_cap = 2000
    while True:
        barrier()
        torch.cuda.synchronize()
        ev0.record()  # same stream the library launches on (torch's current stream); idle: stamps at once
        t0 = time.perf_counter()
        run(K)
        ev1.record()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        barrier()
        block_s.append(dt)
        block_dev_ms.append(ev0.elapsed_time(ev1))
        # every rank must leave the loop in the same iteration: rank 0 decides
        stop = torch.tensor([1.0 if (sum(block_s) >= args.min_timed_s or len(block_s) >= reps_cap) else 0.0],
                            dtype=torch.float64, device=cdev)
        if world > 1:
            dist.broadcast(stop, src=0)
        if stop.item() > 0:
            break
    reps = len(block_s)
    last = tr.train_steps(buf, 1, BATCH, graph_unroll=0).cpu().numpy()[0]
    if not np.isfinite(last).all():
        raise SystemExit(f"non-finite losses after the timed region: {last}")

    # max over ranks of every block's wall time, then the median block
    tblk = torch.tensor(block_s, dtype=torch.float64, device=cdev)
    recs = None
    if world > 1:
        dist.all_reduce(tblk, op=dist.ReduceOp.MAX)
        rec = torch.tensor([rank, seed, tr.total_it, *last.tolist()], dtype=torch.float64, device=cdev)
        gathered = [torch.zeros_like(rec) for _ in range(world)]
        dist.all_gather(gathered, rec)  # RCCL: the path's only collective (metric record)
        recs = [dict(zip(("rank", "seed", "total_it", "value_loss", "q_loss", "actor_loss"), g.tolist()))
                for g in gathered]
    blk = np.sort(tblk.cpu().numpy())
    dt_med = float(np.median(blk))
    dev_ms_med = float(np.median(block_dev_ms))

    if rank == 0:
        cfg = tr._cfg(BATCH)
        bytes_step, flops_step = C.c_double(), C.c_double()
        _lib.check(_lib.load().iqlhip_step_cost(C.byref(cfg), C.byref(bytes_step), C.byref(flops_step)))
        steps_per_s = world * K / dt_med
        step_us_dev = dev_ms_med * 1e3 / K
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
        # rocprofv3 --kernel-trace attributes the timeline (dispatch to completion, back to back)
        ev_us = [avg[k] * 1e3 for k in range(3)]
        scale = step_us_dev / sum(ev_us) if sum(ev_us) > 0 else 1.0
        upd_us = ev_us[2] * scale
        traffic, traffic_src = None, None
        tname = os.environ.get("IQL_TRAFFIC_PROFILE", f"{PROFILE_TAG}_traffic.json")
        tpath = os.path.join(ROOT, "profiles", tname)
        if not os.path.exists(tpath):  # (no set of this round yet: the last committed one, labelled as such)
            tname = "r03_traffic.json"
            tpath = os.path.join(ROOT, "profiles", tname)
        if os.path.exists(tpath):  # PMC passes (separate rocprofv3 --pmc runs, tools/profile.sh)
            with open(tpath) as f:
                tj = json.load(f)
            traffic = tj.get("k_update_bytes_per_launch")
            traffic_src = {"file": f"profiles/{tname}", "build": tj.get("build"),
                           "note": "PMC (2*FETCH_SIZE + WRITE_SIZE) of an earlier rocprofv3 run of this "
                                   "command; not collected in this timed run",
                           "matches_this_build": tj.get("build") == _lib.build_tag()}
        live_mode = (f"hipGraphs of {unroll} steps" if unroll else "plain kernel launches from the library's C loop") + \
            "; HIP-event shares of 300 eager steps scaled to tile the measured device time per step"
        roof = roofline_block("k_update", upd_bytes, upd_us, live_mode, f"{PROFILE_TAG}_kernel_stats.csv",
                              ("k_update<true, true>",), "hipGraphs of 50 steps under rocprofv3 --kernel-trace --stats "
                              "(plain launches are paced by the tracer's per-dispatch interception)", _lib.build_tag(),
                              extra={"traffic": traffic, "traffic_source": traffic_src,
                                     "launch_us": upd_us, "launch_us_events_only": ev_us[2], "launches_timed": int(nl.value),
                                     "step": {"bytes_per_step": bytes_step.value, "device_us_per_step": step_us_dev,
                                              "achieved_gbs": bytes_step.value / (step_us_dev * 1e-6) / 1e9,
                                              "frac": bytes_step.value / (step_us_dev * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                              "kernel_us": {"k_forward": ev_us[0] * scale, "k_backward": ev_us[1] * scale,
                                                            "k_update": ev_us[2] * scale},
                                              "kernel_us_events_only": {"k_forward": ev_us[0], "k_backward": ev_us[1],
                                                                        "k_update": ev_us[2]},
                                              "mfma_tflops": flops_step.value / (step_us_dev * 1e-6) / 1e12,
                                              "mfma_peak_tflops": MFMA_BF16_PEAK_TFLOPS if args.precision == "bf16"
                                              else MFMA_F32_PEAK_TFLOPS}})
        out = {
            "metric": "iql_grad_steps_per_sec", "value": steps_per_s, "unit": "steps/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": dt_med * 1e3 / K, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "IQL antmaze-medium-diverse-v2 shapes (S=29 A=8 H=256), "
                                   "1M-transition device replay, batch 256, one seed per GPU",
                       "batch": BATCH, "buffer_rows": N_ROWS, "graph_unroll": unroll,
                       "graph_launches_per_block": K // unroll if unroll else 0,
                       "eager_steps_per_block": K % unroll if unroll else K},
            "timing": {"reps": reps, "timed_steps_total": reps * K,
                       "block_ms": {"median": dt_med * 1e3, "min": float(blk[0]) * 1e3,
                                    "max": float(blk[-1]) * 1e3},
                       "ms_per_step_device": dev_ms_med / K,
                       "note": "each block = K steps between barrier+synchronize pairs; median block, "
                               "max over ranks; *_device = HIP events on the launch stream"},
            "build": _lib.build_tag(),
            "roofline": roof,
        }
        if recs is not None:
            out["ranks"] = recs
        if world == 1 and K < 5_000 and not args.no_sustained:
            # the same path sustained over a long region, for comparison with the K-step blocks
            n_long = 20_000 // unroll * unroll if unroll else 20_000
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            run(n_long)
            torch.cuda.synchronize()
            out["sustained"] = {"steps": n_long, "value": n_long / (time.perf_counter() - t1), "unit": "steps/s",
                                "note": "one timed region of this many steps, same launch mode; not `value`"}
        # (secondary legs: none of them may take the headline record down with it)
        if world == 1 and args.agents_per_gpu > 1:
            try:
                out["agents_per_gpu"] = agents_leg(ia, torch, tr, buf, device, seed, args, unroll, bytes_step.value)
            except Exception as e:
                out["agents_per_gpu"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and args.agents_per_gpu > 1:
            # the same aggregate through the PRODUCT entry point: train(config, seeds_per_gpu=K)
            # (iqlpref_amd/train.py; ensemble_sweeps/launch.sh:12 AGENTS_PER_GPU), log windows of 2500 steps
            try:
                out["train_seeds_per_gpu"] = train_leg(ia, data, device, args)
            except Exception as e:
                out["train_seeds_per_gpu"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and args.ensemble_q >= 2:
            try:
                E_ = args.ensemble_q
                tre = build_trainer(ia, torch, device, seed + 50, args.precision, n_critics=E_)
                leg = step_leg(ia, torch, _lib, tre, buf, 1024, 5_000, 50, f"{PROFILE_TAG}_ens{E_}_kernel_stats.csv",
                               ("k_update<true, false>",), f"k_update (E = {E_} critics, batch 1024)")
                out["ensemble_q"] = dict(leg, n_critics=E_, batch=1024, transitions_per_s=1024 * leg["value"],
                                         note="BASELINE configs[4] (E-way critic ensemble, batch 1024, antmaze "
                                              "shapes); not `value`")
                del tre
            except Exception as e:
                out["ensemble_q"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_pen:
            try:
                S_, A_ = PEN["dims"]
                bufp = ia.ReplayBuffer(S_, A_, PEN["n_rows"], device)
                bufp.load_d4rl_dataset(synth_dataset_dims(seed + 70, PEN["n_rows"], S_, A_))
                trp = build_trainer(ia, torch, device, seed + 70, args.precision, dims=PEN["dims"],
                                    dropout=PEN["dropout"], hyper=PEN["hyper"])
                leg = step_leg(ia, torch, _lib, trp, bufp, BATCH, 20_000, 0, f"{PROFILE_TAG}_pen_kernel_stats.csv",
                               ("k_update<true, true>",), "k_update (pen shapes S=45 A=24, actor dropout 0.1)")
                out["pen_config3"] = dict(leg, batch=BATCH, buffer_rows=PEN["n_rows"],
                                          note="BASELINE configs[2] training step (pen-human-v1 shapes and hyper-"
                                               "parameters, Philox dropout masks); its relabel half is "
                                               "relabel.pt_pen_config3; not `value`")
                del trp, bufp
            except Exception as e:
                out["pen_config3"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_pen:
            # a shape off the tuned step's (three hidden layers instead of two, ref:458-459 n_hidden): the general
            # layer-wise step (csrc/iql_deep.hip), plain launches; its update kernel's share of the HBM roofline
            # from HIP events only (no committed trace)
            try:
                import ctypes as C
                trg = build_trainer(ia, torch, device, seed + 90, args.precision, n_hidden=3)
                assert trg.step_kind(BATCH) == "general"
                trg.train_steps(buf, 500, BATCH, return_losses=False)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                trg.train_steps(buf, 5_000, BATCH, return_losses=False)
                torch.cuda.synchronize()
                dtg = time.perf_counter() - t1
                evg, _ = kernel_times(_lib, trg, buf, BATCH, 200)
                byg, flg = C.c_double(), C.c_double()
                cfgg = trg._cfg(BATCH)
                _lib.check(_lib.load().iqlhip_step_cost(C.byref(cfgg), C.byref(byg), C.byref(flg)))
                gath = 4.0 * BATCH * (2 * cfgg.state_dim + cfgg.action_dim + 2)
                out["general_step"] = {
                    "value": 5_000 / dtg, "unit": "steps/s", "us_per_step": dtg / 5_000 * 1e6, "n_hidden": 3,
                    "hidden_dim": 256, "batch": BATCH, "bytes_per_step": byg.value,
                    "kernel_us_events": {"kd_forward": evg[0], "kd_backward": evg[1], "kd_update": evg[2]},
                    "roofline": roofline_block("kd_update (three hidden layers of 256 units)", byg.value - gath, evg[2],
                                               "plain launches; HIP events around each kernel",
                                               f"{PROFILE_TAG}_general_kernel_stats.csv", ("kd_update<true",),
                                               "plain launches under rocprofv3 --kernel-trace", _lib.build_tag()),
                    "note": "the general layer-wise step on a shape the tuned three-kernel step does not take "
                            "(coverage path); not `value`"}
                tpg = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_general_traffic.json")
                if os.path.exists(tpg):  # PMC bytes of a separate rocprofv3 run (tools/profile.sh)
                    with open(tpg) as f:
                        tjg = json.load(f)
                    out["general_step"]["roofline"]["traffic"] = tjg.get("kd_update_bytes_per_launch")
                    out["general_step"]["roofline"]["traffic_source"] = {
                        "file": f"profiles/{PROFILE_TAG}_general_traffic.json", "build": tjg.get("build"),
                        "matches_this_build": tjg.get("build") == _lib.build_tag()}
                del trg
            except Exception as e:
                out["general_step"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_relabel:
            try:
                from tools import bench_relabel
                out["relabel"] = bench_relabel.leg(device)
            except Exception as e:  # the leg must never take the headline down with it
                out["relabel"] = {"error": f"{type(e).__name__}: {e}"}
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(data)
        print(json.dumps(out), file=json_out, flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def train_leg(ia, data, device, args):
    """K seeds through train(config, seeds_per_gpu=K): steps/s between the first and the last log
    window (setup -- upload, dataset preparation, trainer construction -- is outside the figure)."""
    K_ = args.agents_per_gpu
    log_freq, windows = 2_500, 9
    cfg = ia.TrainConfig(env="antmaze-medium-diverse-v2", max_timesteps=log_freq * windows, log_freq=log_freq,
                         eval_freq=10 ** 9, batch_size=BATCH, normalize_reward=0, normalize=True, seed=500,
                         device=device, buffer_size=N_ROWS, **{k: HYPER[k] for k in ("beta", "iql_tau", "discount", "tau")})
    stamps = []
    t0 = time.perf_counter()
    ia.train(cfg, dataset=data, state_dim=S_DIM, action_dim=A_DIM, max_action=1.0, precision=args.precision,
             logger=lambda d, step: stamps.append((time.perf_counter(), step)) if d.get("seed") == 500 else None,
             evaluate=None, seeds_per_gpu=K_)
    total_s = time.perf_counter() - t0
    dt = stamps[-1][0] - stamps[0][0]
    steps = stamps[-1][1] - stamps[0][1]
    return {"seeds_per_gpu": K_, "value": K_ * steps / dt, "unit": "steps/s", "steps_per_seed_timed": steps,
            "log_freq": log_freq, "wall_s_with_setup": total_s,
            "note": "aggregate steps/s of train(config, seeds_per_gpu=K) between its first and last log window "
                    "(per-seed loss windows read back every log_freq steps); not `value` of the record"}


def agents_leg(ia, torch, tr, buf, device, seed, args, unroll, bytes_step):
    """Independent seeds sharing one GPU (same dataset: a sweep varies the seed only): as ONE seed
    group (one launch sequence, gridDim.y = agents) and as TWO sub-groups on the two halves of the
    compute units (SeedGroup mode "split"); `value` is the faster of the two."""
    A_ = args.agents_per_gpu
    trs = [tr] + [build_trainer(ia, torch, device, seed + 100 + i, args.precision) for i in range(1, A_)]
    u = 50

    def rate(group, k_, n_):
        group.train_steps(buf, 1_000, BATCH, graph_unroll=u)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        group.train_steps(buf, n_, BATCH, graph_unroll=u)
        torch.cuda.synchronize()
        return k_ * n_ / (time.perf_counter() - t1)

    scan = {}
    for k_ in sorted({2, 4, A_}):  # smaller groups first, for the scaling of the aggregate with K
        if k_ >= A_:
            break
        g_ = ia.SeedGroup(trs[:k_], mode="group")
        scan[str(k_)] = rate(g_, k_, 5_000)
        g_.close()
        if k_ >= 4:
            try:
                g_ = ia.SeedGroup(trs[:k_], mode="split", n_streams=2)
                scan["%d (2 CU slices x %d)" % (k_, k_ // 2)] = rate(g_, k_, 5_000)
                g_.close()
            except Exception as e:  # (CU-masked streams unavailable: the one-group figures stand)
                scan["%d (2 CU slices x %d)" % (k_, k_ // 2)] = f"{type(e).__name__}: {e}"
    n_multi = 10_000
    v_split = None
    if A_ >= 4:
        try:
            gs = ia.SeedGroup(trs, mode="split", n_streams=2)
            v_split = rate(gs, A_, n_multi)
            scan["%d (2 CU slices x %d)" % (A_, A_ // 2)] = v_split
            gs.close()
        except Exception as e:
            v_split = None
            scan["%d (2 CU slices x %d)" % (A_, A_ // 2)] = f"{type(e).__name__}: {e}"
    group = ia.SeedGroup(trs, mode="group")
    v_group = rate(group, A_, n_multi)
    scan[str(A_)] = v_group
    v = max(v_group, v_split or 0.0)
    out = {"agents": A_, "value": v, "unit": "steps/s", "steps_per_agent": n_multi,
           "mode": "split (2 CU slices)" if v_split and v_split > v_group else getattr(group, "mode", "streams"),
           "value_one_group": v_group, "value_two_cu_slices": v_split, "steps_per_s_by_agents": scan,
           "roofline_step": {"bound": "hbm", "achieved": v * bytes_step / 1e9, "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": v * bytes_step / 1e9 / HBM_PEAK_GBS,
                             "note": "whole step: aggregate steps/s x algorithmic bytes per step"},
           "note": "aggregate of independent seeds sharing one GPU; each seed bit-identical to running "
                   "alone; not `value` of the record"}
    if getattr(group, "mode", "") == "group":
        # the dominant kernel of the ONE-group launch against the HBM roofline: its launch moves the
        # optimiser state of ALL agents (algorithmic bytes x agents); event shares scaled so that
        # the three launches tile the measured time per group step, as for the solo roofline
        kt = group.kernel_times(buf, BATCH, 200)
        ev = [kt["k_forward"], kt["k_backward"], kt["k_update"]]
        step_us = A_ / v_group * 1e6
        scale = step_us / sum(ev) if sum(ev) > 0 else 1.0
        upd_bytes = A_ * (bytes_step - 4.0 * BATCH * (2 * S_DIM + A_DIM + 2))
        from iqlpref_amd import _lib as _l
        out["roofline"] = roofline_block(
            "k_update (gridDim.y = %d, one group on the whole chip)" % A_, upd_bytes, ev[2] * scale,
            f"hipGraphs of {u} steps; HIP-event shares scaled to tile the group step", f"{PROFILE_TAG}_group{A_}_kernel_stats.csv",
            ("k_update<true, false>",), "hipGraphs of 50 steps under rocprofv3 --kernel-trace --stats (tools/group_scan.py)",
            _l.build_tag(),
            extra={"group_step_us": step_us,
                   "kernel_us": {"k_forward": ev[0] * scale, "k_backward": ev[1] * scale, "k_update": ev[2] * scale},
                   "kernel_us_events_only": {"k_forward": ev[0], "k_backward": ev[1], "k_update": ev[2]}})
    group.close()
    return out


if __name__ == "__main__":
    main()
