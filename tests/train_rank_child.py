"""One rank of a multi-rank ``train()`` job (test infrastructure): started by
tests/test_gpu_round2.py as a FRESH interpreter with RANK / WORLD_SIZE / MASTER_* in the
environment, before anything in it has touched a GPU.  Runs the product entry point
iqlpref_amd.train.train() on a small synthetic dataset with the deterministic stand-in vector
environment and writes what its logger received to <out_dir>/rank<r>.json."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_dir, k_seeds = sys.argv[1], int(sys.argv[2])
    import iqlpref_amd as ia
    from tests import fake_envs
    name = "antmaze-medium-diverse-v2"
    S, A = fake_envs.DIMS[name]
    rng = np.random.default_rng(5)
    n = 3000
    data = {"observations": rng.standard_normal((n, S)).astype(np.float32),
            "actions": rng.uniform(-1, 1, (n, A)).astype(np.float32),
            "rewards": (rng.uniform(size=n) < 0.05).astype(np.float32),
            "next_observations": rng.standard_normal((n, S)).astype(np.float32),
            "terminals": (rng.uniform(size=n) < 0.01).astype(np.float32)}
    cfg = ia.TrainConfig(env=name, seed=100, max_timesteps=40, log_freq=10, eval_freq=20, n_episodes=5,
                         batch_size=64, normalize_reward=1, beta=10.0, iql_tau=0.9, device="cuda",
                         buffer_size=10_000)

    def vector_env(env_name, seeds, mean, std):
        def make(seed):
            def thunk():
                e = fake_envs.TransformObservation(fake_envs.FakeGymEnv(env_name), lambda o: (o - mean) / std)
                e.seed(seed)
                return e
            return thunk
        return fake_envs.SyncVectorEnv([make(s) for s in seeds])

    logs = []
    out = ia.train(cfg, dataset=data, state_dim=S, action_dim=A, max_action=1.0, precision="bf16",
                   logger=lambda rec, step: logs.append([int(step), {k: float(v) for k, v in rec.items()}]),
                   vector_env=vector_env, seeds_per_gpu=k_seeds)
    trainers = out if isinstance(out, list) else [out]
    rank = int(os.environ.get("RANK", "0"))
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump({"rank": rank, "device": cfg.device, "seeds": [t._seed for t in trainers],
                   "total_it": [t.total_it for t in trainers], "logs": logs,
                   "param_sum": [float(t._params.double().sum()) for t in trainers]}, f)
    import torch.distributed as dist
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
