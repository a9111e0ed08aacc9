"""``train(config)`` -- the reference's entry point (ref:1393-1570, "ref" =
/root/reference/algorithms/offline/iql.py) on the fused HIP step.

The hot loop runs ``log_freq`` optimisation steps per call into the library and
reads the losses back once per logging window (the reference syncs three times per
step, ref:589,607,633).  gym / d4rl / wandb are optional: pass ``env`` and
``dataset`` to run without them; metrics go to ``logger`` (a callable) or wandb
when it is importable.
"""
import os
import time
from dataclasses import asdict
from pathlib import Path
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
import torch

from . import distributed as D
from .iql import (DeterministicPolicy, EnsembleQ, GaussianPolicy, ImplicitQLearning, ReplayBuffer, TrainConfig, TwinQ,
                  ValueFunction, compute_mean_std, normalize_states, set_seed)
from .relabel import (load_mlp_reward_model, load_pt_reward_model, modify_reward, qlearning_dataset_bnn,
                      qlearning_dataset_mr, qlearning_dataset_mr_ensemble, qlearning_dataset_pt)


def wrap_env(env, state_mean=0.0, state_std=1.0, reward_scale: float = 1.0):
    """ref:140-161"""
    import gym

    def normalize_state(state):
        return (state - state_mean) / state_std

    def scale_reward(reward):
        return reward_scale * reward

    env = gym.wrappers.TransformObservation(env, normalize_state)
    if reward_scale != 1.0:
        env = gym.wrappers.TransformReward(env, scale_reward)
    return env


@torch.inference_mode()
def eval_actor(env_name: str, actor, max_action: float, state_mean, state_std, device: str,
               n_episodes: int, seed: int, n_envs: int = 25, make_env: Optional[Callable] = None
               ) -> Tuple[np.ndarray, List[int]]:
    """ref:265-341: ``n_envs`` parallel episodes, batched actor inference on the GPU
    (one iqlhip_mlp_forward per env step for all envs)."""
    import gym
    from functools import partial

    is_antmaze = "antmaze" in env_name.lower()
    n_envs = min(n_envs, n_episodes)

    def _make(i):
        e = make_env(env_name) if make_env else gym.make(env_name)
        e = wrap_env(e, state_mean=state_mean, state_std=state_std)
        e.seed(seed + i)
        return e

    vec_env = gym.vector.AsyncVectorEnv([partial(_make, i) for i in range(n_envs)])
    actor.eval()
    obs = vec_env.reset()
    ep_rewards = np.zeros(n_envs, dtype=np.float64)
    ep_steps = np.zeros(n_envs, dtype=np.int64)
    completed: List[float] = []
    steps_to_goal: List[int] = []
    try:
        while len(completed) < n_episodes:
            states_t = torch.tensor(obs, dtype=torch.float32, device=device)
            out = actor(states_t)
            mean = out.mean if isinstance(out, torch.distributions.Distribution) else out
            actions = torch.clamp(max_action * mean, -max_action, max_action).cpu().numpy()
            obs, rewards, dones, _ = vec_env.step(actions)
            ep_rewards += rewards
            ep_steps += 1
            for i in range(n_envs):
                if dones[i]:
                    if is_antmaze and ep_rewards[i] > 0.5:
                        steps_to_goal.append(int(ep_steps[i]))
                    completed.append(float(ep_rewards[i]))
                    ep_rewards[i] = 0.0
                    ep_steps[i] = 0
                    if len(completed) >= n_episodes:
                        break
    finally:
        vec_env.close()
    actor.train()
    return np.asarray(completed[:n_episodes]), steps_to_goal


def build_dataset(config: TrainConfig, env, dataset=None) -> Dict[str, np.ndarray]:
    """ref:1402-1433: pick the relabel path from the config."""
    if config.reward_model_path:
        path = os.path.expanduser(config.reward_model_path)
        if config.bnn_reward_model:
            return qlearning_dataset_bnn(env, path, alpha=config.bnn_alpha, n_samples=config.bnn_n_samples,
                                         device=config.device, dataset=dataset)
        if config.mr_ensemble:
            return qlearning_dataset_mr_ensemble(env, path, alpha=config.mr_alpha, burn_in=config.mr_burn_in,
                                                 device=config.device, dataset=dataset)
        if config.query_length > 1:
            model = load_pt_reward_model(path, device=config.device)
            return qlearning_dataset_pt(env, model, config.query_length, dataset=dataset)
        model = load_mlp_reward_model(path, device=config.device)
        return qlearning_dataset_mr(env, model, dataset=dataset)
    if dataset is not None:
        return dataset
    import d4rl
    return d4rl.qlearning_dataset(env)


def train(config: TrainConfig, env=None, dataset=None, *, state_dim: Optional[int] = None,
          action_dim: Optional[int] = None, max_action: Optional[float] = None,
          logger: Optional[Callable[[Dict[str, float], int], None]] = None,
          evaluate: Optional[Callable] = None, precision: str = "bf16",
          raw_dataset=None) -> ImplicitQLearning:
    """ref:1393-1570.  ``dataset``: an already-built qlearning dataset (skips d4rl);
    ``raw_dataset``: an env.get_dataset()-style dict handed to the relabel functions;
    ``evaluate(actor, step) -> (scores, steps_to_goal)`` replaces eval_actor when gym is
    not installed (None: evaluation is skipped)."""
    # one process per GPU: under torchrun this rank owns cuda:<LOCAL_RANK>, and everything
    # below (process group, buffer, trainer, metric all-gather) lives there
    bound = D.local_device()
    if bound is not None:
        config.device = bound
    rank = D.init_from_env(device=config.device)
    if env is None and (state_dim is None or action_dim is None):
        import gym
        env = gym.make(config.env)
    if state_dim is None:
        state_dim = env.observation_space.shape[0]
        action_dim = env.action_space.shape[0]
    dataset = build_dataset(config, env, dataset if dataset is not None else raw_dataset) \
        if (config.reward_model_path or dataset is None) else dataset

    if config.normalize_reward:
        modify_reward(dataset, config.env, config.normalize_reward)
    if config.normalize:
        state_mean, state_std = compute_mean_std(dataset["observations"], eps=1e-3)
    else:
        state_mean, state_std = 0, 1
    dataset["observations"] = normalize_states(dataset["observations"], state_mean, state_std)
    dataset["next_observations"] = normalize_states(dataset["next_observations"], state_mean, state_std)
    replay_buffer = ReplayBuffer(state_dim, action_dim, config.buffer_size, config.device)
    replay_buffer.load_d4rl_dataset(dataset)
    if max_action is None:
        max_action = float(env.action_space.high[0])

    if config.checkpoints_path is not None:
        print(f"Checkpoints path: {config.checkpoints_path}")
        os.makedirs(config.checkpoints_path, exist_ok=True)
        import yaml
        with open(os.path.join(config.checkpoints_path, "config.yaml"), "w") as f:
            yaml.safe_dump(asdict(config), f)

    seed = D.rank_seed(config.seed)  # one independent seed per rank / GPU
    set_seed(seed, None)
    if config.n_critics == 2:
        q_network = TwinQ(state_dim, action_dim).to(config.device)
    else:
        q_network = EnsembleQ(state_dim, action_dim, n_critics=config.n_critics).to(config.device)
    v_network = ValueFunction(state_dim).to(config.device)
    pol = DeterministicPolicy if config.iql_deterministic else GaussianPolicy
    actor = pol(state_dim, action_dim, max_action, dropout=config.actor_dropout).to(config.device)
    v_optimizer = torch.optim.Adam(v_network.parameters(), lr=config.vf_lr)
    q_optimizer = torch.optim.Adam(q_network.parameters(), lr=config.qf_lr)
    actor_optimizer = torch.optim.Adam(actor.parameters(), lr=config.actor_lr)
    print("---------------------------------------")
    print(f"Training IQL, Env: {config.env}, Seed: {seed}")
    print("---------------------------------------")
    trainer = ImplicitQLearning(
        max_action=max_action, actor=actor, actor_optimizer=actor_optimizer, q_network=q_network,
        q_optimizer=q_optimizer, v_network=v_network, v_optimizer=v_optimizer, discount=config.discount,
        tau=config.tau, device=config.device, beta=config.beta, iql_tau=config.iql_tau,
        max_steps=config.max_timesteps, precision=precision, seed=seed)
    if config.load_model != "":
        trainer.load_state_dict(torch.load(Path(config.load_model), weights_only=True))

    if logger is None:
        try:
            import wandb
            wandb.init(config=asdict(config), project=config.project, group=config.group, name=config.name)
            logger = lambda d, step: wandb.log(d, step=step)
        except ImportError:
            logger = lambda d, step: print(f"[{step}] " + " ".join(f"{k}={v:.5g}" for k, v in d.items()))

    total = int(config.max_timesteps)
    t = 0
    t_window = time.perf_counter()
    while t < total:
        # run to the next logging / evaluation boundary in one library call (ref:1533-1544)
        nxt = min(total, (t // config.log_freq + 1) * config.log_freq,
                  (t // config.eval_freq + 1) * config.eval_freq)
        losses = trainer.train_steps(replay_buffer, nxt - t, config.batch_size)
        window = losses if t % config.log_freq == 0 else torch.cat([window, losses])
        t = nxt
        if t % config.log_freq == 0:
            mean = window.mean(dim=0).tolist()  # the only host sync of the window
            now = time.perf_counter()
            rec = {"value_loss": mean[0], "q_loss": mean[1], "actor_loss": mean[2]}
            logger(dict(rec), trainer.total_it)
            last = dict(rec, steps_per_sec=window.shape[0] / max(now - t_window, 1e-9))
            t_window = now
        if t % config.eval_freq == 0:
            eval_log: Dict[str, float] = {}
            if evaluate is not None:
                scores, steps_to_goal = evaluate(actor, t)
            elif env is not None and hasattr(env, "spec"):
                scores, steps_to_goal = eval_actor(config.env, actor, max_action, state_mean, state_std,
                                                   config.device, config.n_episodes, config.seed)
            else:
                scores, steps_to_goal = None, []
            if scores is not None:
                eval_log["mean_score"] = float(np.mean(scores))
                if "antmaze" in config.env.lower():
                    eval_log["avg_steps_to_goal"] = float(np.mean(steps_to_goal)) if steps_to_goal else -1.0
                logger(dict(eval_log), trainer.total_it)
            if config.checkpoints_path is not None:
                torch.save(trainer.state_dict(), os.path.join(config.checkpoints_path, f"checkpoint_{t - 1}.pt"))
            # end-of-eval metric exchange across the per-GPU seeds (the path's only collective)
            recs = D.gather_metrics(dict(last if t >= config.log_freq else {}, seed=seed,
                                         total_it=trainer.total_it, **eval_log), device=config.device)
            if rank == 0 and len(recs) > 1:
                logger(D.summarize(recs), trainer.total_it)
    return trainer


def main(argv=None):
    """CLI with the reference's pyrallis surface: --config_path file.yaml --field value ..."""
    import argparse
    from .iql import load_config
    ap = argparse.ArgumentParser()
    ap.add_argument("--config_path", default=None)
    args, rest = ap.parse_known_args(argv)
    over = {}
    it = iter(rest)
    for tok in it:
        if tok.startswith("--"):
            k = tok[2:]
            if "=" in k:
                k, v = k.split("=", 1)
            else:
                v = next(it)
            over[k] = v
    train(load_config(args.config_path, **over))


if __name__ == "__main__":
    main()
