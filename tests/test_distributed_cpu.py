"""world_size-2 gloo test of the multi-GPU path's only collective (metric all-gather)
and of the rank -> seed mapping.  CPU only."""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from iqlpref_amd import distributed as D
# rank -> GPU mapping (pure part; there is no GPU here, so local_device() binds nothing)
assert D.device_for_rank(int(os.environ["LOCAL_RANK"]), 8) == "cuda:" + os.environ["LOCAL_RANK"]
assert D.local_device() is None
rank = D.init_from_env(backend="gloo")
assert dist.get_world_size() == 2
seed = D.rank_seed(10)
recs = D.gather_metrics({"seed": seed, "total_it": 100 + rank, "value_loss": 0.5 * (rank + 1),
                         "mean_score": 10.0 * (rank + 1), "steps_per_sec": 1000.0}, device="cpu")
assert [r["seed"] for r in recs] == [10.0, 11.0], recs
assert [r["rank"] for r in recs] == [0.0, 1.0]
assert [r["total_it"] for r in recs] == [100.0, 101.0]
s = D.summarize(recs)
assert s["n_seeds"] == 2 and s["steps_per_sec_total"] == 2000.0 and s["mean_score_mean"] == 15.0
assert abs(s["mean_score_std"] - 5.0) < 1e-12
import math
assert math.isnan(recs[0]["avg_steps_to_goal"])
# K = 2 seeds per GPU: rank r owns seeds base + 2 r, base + 2 r + 1; one all-gather carries K x world records
first = D.rank_seed(10, 2)
assert first == 10 + 2 * rank
recs = D.gather_metric_records([{"seed": first + k, "total_it": 50, "q_loss": float(first + k),
                                 "steps_per_sec": 500.0} for k in range(2)], device="cpu")
assert [r["seed"] for r in recs] == [10.0, 11.0, 12.0, 13.0], recs
assert [r["rank"] for r in recs] == [0.0, 0.0, 1.0, 1.0]
assert [r["q_loss"] for r in recs] == [10.0, 11.0, 12.0, 13.0]
assert D.summarize(recs)["steps_per_sec_total"] == 2000.0 and D.summarize(recs)["n_seeds"] == 4
dist.barrier()
dist.destroy_process_group()
print("ok", rank)
'''


def test_metric_allgather_two_ranks(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=120)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert f"ok {r}" in o


def test_single_process_gather_is_identity():
    from iqlpref_amd import distributed as D
    recs = D.gather_metrics({"seed": 3, "total_it": 7, "q_loss": 1.5})
    assert len(recs) == 1 and recs[0]["seed"] == 3.0 and recs[0]["q_loss"] == 1.5 and recs[0]["rank"] == 0.0
    recs = D.gather_metric_records([{"seed": 3, "q_loss": 1.5}, {"seed": 4, "q_loss": 2.5}])
    assert [r["seed"] for r in recs] == [3.0, 4.0] and recs[1]["q_loss"] == 2.5 and D.rank_seed(3, 8) == 3


def test_rank_to_gpu_mapping():
    """One process per GPU: rank r owns cuda:r; more local ranks than GPUs is an error, never a
    silent share of GPU 0 (train() under torchrun bound every rank to cuda:0)."""
    import pytest
    from iqlpref_amd import distributed as D
    assert [D.device_for_rank(r, 8) for r in range(8)] == [f"cuda:{r}" for r in range(8)]
    with pytest.raises(RuntimeError, match="one rank per GPU"):
        D.device_for_rank(1, 1)
    assert D.local_device() is None  # WORLD_SIZE unset: not a multi-rank job


def test_bench_refuses_fewer_gpus_than_asked(tmp_path):
    """bench.py --gpus 2 from plain `python` starts the ranks itself; with fewer visible GPUs it
    exits non-zero instead of benchmarking one rank (no GPU here: 0 visible)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20",
                        "--warmup", "5"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "only 0 GPU(s) are visible" in p.stderr and p.stdout.strip() == ""
