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
