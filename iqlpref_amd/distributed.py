"""Multi-GPU execution model: one process per GPU, one independent (seed, dataset)
run per rank -- what the reference does with one W&B agent per GPU
(ensemble_sweeps/launch.sh:84-94).  No parameter or gradient ever crosses ranks;
the only collective is the all-gather of a fixed-size metric record at log / eval
time (RCCL over xGMI under ``backend="nccl"``, gloo in CPU tests).
"""
import os
from typing import Dict, List, Optional

import torch
import torch.distributed as dist

# fixed layout of the per-rank record (float64): latency-bound, 9 x 8 bytes
FIELDS = ("seed", "total_it", "value_loss", "q_loss", "actor_loss", "mean_score",
          "avg_steps_to_goal", "steps_per_sec", "rank")


def device_for_rank(local_rank: int, n_visible: int) -> str:
    """The GPU a rank owns: ``cuda:<LOCAL_RANK>`` (one process per GPU, the reference's
    ``CUDA_VISIBLE_DEVICES=$gpu wandb agent``, ensemble_sweeps/launch.sh:91).  More local ranks
    than visible GPUs is an error, never a silent share of GPU 0."""
    if local_rank < 0 or local_rank >= max(n_visible, 1):
        raise RuntimeError(f"LOCAL_RANK {local_rank} but only {n_visible} GPU(s) visible: "
                           "launch one rank per GPU")
    return f"cuda:{local_rank}"


def rehearsal() -> bool:
    """IQL_DIST_BACKEND=gloo: the multi-rank product path rehearsed on a box with ONE GPU -- every
    rank on cuda:0, the metric all-gather over gloo with its payload on the host.  The default
    (unset, or "nccl") is one GPU per rank and RCCL; nothing else differs between the two."""
    return os.environ.get("IQL_DIST_BACKEND", "nccl").lower() == "gloo"


def local_device() -> Optional[str]:
    """Under a torchrun-style launch (WORLD_SIZE > 1): bind this process to its GPU before
    anything else touches a device and return the indexed device string every later object
    (process group, replay buffer, trainer, collective payloads) must use.  None when the
    process is not part of a multi-rank job or has no GPU (gloo CPU tests)."""
    if int(os.environ.get("WORLD_SIZE", "1")) <= 1 or not torch.cuda.is_available():
        return None
    if rehearsal():
        torch.cuda.set_device(0)
        return "cuda:0"
    dev = device_for_rank(int(os.environ.get("LOCAL_RANK", "0")), torch.cuda.device_count())
    torch.cuda.set_device(torch.device(dev))
    return dev


def init_from_env(backend: Optional[str] = None, device: Optional[str] = None) -> int:
    """torchrun-style rendezvous (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT).
    Returns the rank; a no-op (rank 0) when WORLD_SIZE is 1 or unset.  ``device`` must be an
    indexed device (``cuda:3``) for the RCCL backend; ``local_device()`` supplies it."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 or dist.is_initialized():
        return dist.get_rank() if dist.is_initialized() else 0
    if backend is None:
        backend = "nccl" if (torch.cuda.is_available() and not rehearsal()) else "gloo"
    kw = {}
    if backend == "nccl":
        if device is None or torch.device(device).index is None:
            device = local_device()
        kw["device_id"] = torch.device(device)
    dist.init_process_group(backend, **kw)
    return dist.get_rank()


def rank_seed(base_seed: int, seeds_per_gpu: int = 1) -> int:
    """First seed of this rank's runs: base + rank * seeds_per_gpu (tr_sweeps/*.yaml grid their
    seeds; with K seeds per GPU rank r owns base + r K .. base + r K + K - 1)."""
    return base_seed + (dist.get_rank() if dist.is_initialized() else 0) * int(seeds_per_gpu)


def _payload_device(device: Optional[str]) -> str:
    """Where a collective's payload lives: the rank's GPU under RCCL, the host under gloo."""
    if dist.get_backend() != "nccl":
        return "cpu"
    return device if device is not None else "cuda"


def gather_metrics(record: Dict[str, float], device: Optional[str] = None) -> List[Dict[str, float]]:
    """All-gather one metric record per rank; every rank receives the full list."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    vals = [float(record.get(k, float("nan"))) for k in FIELDS[:-1]] + [float(rank)]
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [dict(zip(FIELDS, vals))]
    dev = _payload_device(device)
    t = torch.tensor(vals, dtype=torch.float64, device=dev)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [dict(zip(FIELDS, o.tolist())) for o in out]


def gather_metric_records(records: List[Dict[str, float]], device: Optional[str] = None) -> List[Dict[str, float]]:
    """``gather_metrics`` for K records per rank (K seeds per GPU, the same K on every rank): ONE
    all-gather of a [K, len(FIELDS)] block; every rank receives the K x world records, rank-major."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    rows = [[float(r.get(k, float("nan"))) for k in FIELDS[:-1]] + [float(rank)] for r in records]
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [dict(zip(FIELDS, row)) for row in rows]
    dev = _payload_device(device)
    t = torch.tensor(rows, dtype=torch.float64, device=dev)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [dict(zip(FIELDS, row)) for o in out for row in o.tolist()]


def summarize(records: List[Dict[str, float]]) -> Dict[str, float]:
    """Across-seed mean / std of the scores, total throughput (rank 0 logs this)."""
    import math
    scores = [r["mean_score"] for r in records if not math.isnan(r["mean_score"])]
    out = {"n_seeds": float(len(records)),
           "steps_per_sec_total": sum(r["steps_per_sec"] for r in records
                                      if not math.isnan(r["steps_per_sec"]))}
    if scores:
        m = sum(scores) / len(scores)
        out["mean_score_mean"] = m
        out["mean_score_std"] = (sum((s - m) ** 2 for s in scores) / len(scores)) ** 0.5
    return out
