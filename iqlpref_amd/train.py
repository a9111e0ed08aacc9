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
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import distributed as D
from . import prep
from .iql import (DeterministicPolicy, EnsembleQ, GaussianPolicy, ImplicitQLearning, ReplayBuffer, TrainConfig, TwinQ,
                  ValueFunction, compute_mean_std, normalize_states, set_seed)
from .relabel import (load_mlp_reward_model, load_pt_reward_model, modify_reward, qlearning_dataset_bnn,
                      qlearning_dataset_mr, qlearning_dataset_mr_ensemble, qlearning_dataset_pt)


class _NormalizedEnv:
    """What the reference builds from gym.wrappers.TransformObservation / TransformReward
    (ref:140-161), without importing gym: observations are z-scored with the dataset
    statistics on the way out of reset() / step(), rewards are scaled.  Everything else is the
    wrapped environment's."""

    def __init__(self, env, state_mean, state_std, reward_scale):
        self.env = env
        self._mean, self._std, self._scale = state_mean, state_std, reward_scale

    def __getattr__(self, name):
        return getattr(self.env, name)

    def _obs(self, obs):
        return (obs - self._mean) / self._std  # the epsilon is already part of state_std

    def reset(self, **kw):
        out = self.env.reset(**kw)
        return (self._obs(out[0]),) + tuple(out[1:]) if isinstance(out, tuple) else self._obs(out)

    def step(self, action):
        obs, reward, *rest = self.env.step(action)
        return (self._obs(obs), reward * self._scale if self._scale != 1.0 else reward, *rest)


def wrap_env(env, state_mean=0.0, state_std=1.0, reward_scale: float = 1.0):
    """ref:140-161: normalised observations (and scaled rewards) around ``env``."""
    return _NormalizedEnv(env, state_mean, state_std, reward_scale)


def _gym_vector_env(env_name: str, seeds: Sequence[int], state_mean, state_std):
    """Default evaluation back end: one gym environment per seed in its own subprocess
    (ref:289-295 AsyncVectorEnv).  gym / d4rl are imported only here."""
    import gym

    def make(seed):
        def thunk():
            env = wrap_env(gym.make(env_name), state_mean=state_mean, state_std=state_std)
            env.seed(seed)
            return env
        return thunk

    return gym.vector.AsyncVectorEnv([make(s) for s in seeds])


def policy_actions(actor, obs, max_action: float, device) -> np.ndarray:
    """Greedy actions for a batch of observations [n, S]: ONE exact-fp32 MLP forward on the GPU
    for all environments (iqlhip_mlp_forward), tanh mean scaled and clamped to the action box
    (ref:306-319; the Gaussian policy is evaluated at its mean, ref:309)."""
    states = torch.as_tensor(np.asarray(obs), dtype=torch.float32, device=device)
    mean = actor.net(states)
    return torch.clamp(max_action * mean, -max_action, max_action).cpu().numpy()


class EpisodeLedger:
    """Episode accounting of ref:296-333 for ``n_envs`` environments stepped in lock step:
    running return and length per environment, returns of finished episodes in the order
    (step, environment index), lengths of the successful antmaze episodes (return > 0.5:
    the goal pays a sparse 1)."""

    def __init__(self, n_envs: int, n_episodes: int, goal_env: bool):
        self.ret = np.zeros(n_envs, dtype=np.float64)
        self.length = np.zeros(n_envs, dtype=np.int64)
        self.scores: List[float] = []
        self.steps_to_goal: List[int] = []
        self._want, self._goal_env = n_episodes, goal_env

    @property
    def full(self) -> bool:
        return len(self.scores) >= self._want

    def record(self, rewards, dones) -> None:
        self.ret += np.asarray(rewards, dtype=np.float64)
        self.length += 1
        ended = np.flatnonzero(np.asarray(dones))[:self._want - len(self.scores)]
        self.scores.extend(self.ret[ended].tolist())
        if self._goal_env:
            self.steps_to_goal.extend(int(n) for n in self.length[ended][self.ret[ended] > 0.5])
        self.ret[ended] = 0.0
        self.length[ended] = 0


@torch.inference_mode()
def eval_actor(env_name: str, actor, max_action: float, state_mean, state_std, device: str,
               n_episodes: int, seed: int, n_envs: int = 25, *,
               vector_env: Optional[Callable] = None) -> Tuple[np.ndarray, List[int]]:
    """ref:265-341.  ``n_envs`` environments run in lock step until ``n_episodes`` episodes have
    finished; every step is one batched actor forward on the GPU.  Returns (episode returns
    [n_episodes], steps-to-goal of the successful antmaze episodes).

    ``vector_env(env_name, seeds, state_mean, state_std)`` builds the environments (default:
    gym's AsyncVectorEnv over ``gym.make``); it must offer ``reset() -> obs [n, S]``,
    ``step(actions [n, A]) -> (obs, rewards, dones, infos)`` with auto-reset, and ``close()``."""
    n_envs = min(n_envs, n_episodes)
    envs = (vector_env or _gym_vector_env)(env_name, [seed + i for i in range(n_envs)], state_mean, state_std)
    ledger = EpisodeLedger(n_envs, n_episodes, goal_env="antmaze" in env_name.lower())
    actor.eval()
    try:
        obs = envs.reset()
        while not ledger.full:
            obs, rewards, dones, _ = envs.step(policy_actions(actor, obs, max_action, device))
            ledger.record(rewards, dones)
    finally:
        envs.close()
        actor.train()  # ref:340: the reference always hands the actor back in train mode
    return np.asarray(ledger.scores[:n_episodes]), ledger.steps_to_goal


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


def seed_configs(config: TrainConfig, seeds: Sequence[int]) -> List[TrainConfig]:
    """The config each of this rank's seeds trains under.  iql_eval.py:143-146 ties the reward model
    to the RUN's seed (``reward_model_path = f"{root}_{seed}"``): with ``reward_model_root`` set, seed s
    gets its own copy of ``config`` pointing at ``{root}_{s}`` -- for every rank and slot, also with one
    seed per GPU (under torchrun rank r > 0 trains ``config.seed + r``, not the seed
    ``TrainConfig.__post_init__`` saw).  Without it all seeds share ``config``."""
    if not config.reward_model_root:
        return [config] * len(seeds)
    import copy
    out = []
    for s in seeds:
        c = copy.copy(config)
        c.seed = int(s)
        c.reward_model_path = f"{config.reward_model_root}_{int(s)}"
        out.append(c)
    return out


def _prepare_replay(config: TrainConfig, dataset, state_dim, action_dim, host_prep: bool):
    """ref:1435-1456: reward normalisation, state statistics, z-scoring, one device buffer."""
    replay_buffer = ReplayBuffer(state_dim, action_dim, config.buffer_size, config.device)
    if host_prep:  # ref:1435-1456 as written: numpy on the host, then one upload
        if config.normalize_reward:
            modify_reward(dataset, config.env, config.normalize_reward)
        if config.normalize:
            state_mean, state_std = compute_mean_std(dataset["observations"], eps=1e-3)
        else:
            state_mean, state_std = 0, 1
        dataset["observations"] = normalize_states(dataset["observations"], state_mean, state_std)
        dataset["next_observations"] = normalize_states(dataset["next_observations"], state_mean, state_std)
        replay_buffer.load_d4rl_dataset(dataset)
    else:  # the same on the device: one upload, the z-scoring fused into the buffer load.  The state
        # statistics are numpy's (ref:132-135, float32 row sums): with them every normalised state --
        # and with that the whole trajectory -- is the reference's to the last bit; stats="device"
        # (double accumulation, ~1e-6 away) stays available through iqlpref_amd.prep
        state_mean, state_std = prep.prepare_replay(
            dataset, replay_buffer, env_name=config.env, normalize_reward=config.normalize_reward,
            normalize=config.normalize, eps=1e-3, stats="host")
    return replay_buffer, state_mean, state_std


def _build_trainer(config: TrainConfig, seed: int, state_dim, action_dim, max_action, precision) -> ImplicitQLearning:
    """ref:1467-1520 for one seed: seeded initial weights, three Adam optimisers, the trainer."""
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
    return trainer


def train(config: TrainConfig, env=None, dataset=None, *, state_dim: Optional[int] = None,
          action_dim: Optional[int] = None, max_action: Optional[float] = None,
          logger: Optional[Callable[[Dict[str, float], int], None]] = None,
          evaluate: Optional[Callable] = None, precision: str = "bf16",
          raw_dataset=None, host_prep: bool = False, seeds_per_gpu: int = 1,
          vector_env: Optional[Callable] = None,
          index_stream: Optional[Callable[[int, int, int], torch.Tensor]] = None):
    """ref:1393-1570.  ``dataset``: an already-built qlearning dataset (skips d4rl);
    ``raw_dataset``: an env.get_dataset()-style dict handed to the relabel functions;
    ``evaluate(actor, step) -> (scores, steps_to_goal)`` replaces eval_actor when gym is
    not installed (None: evaluation is skipped); ``host_prep``: run the whole dataset preparation of
    ref:1435-1456 in numpy on the host instead of the device kernels of iqlpref_amd.prep (both give
    the reference's values to the last bit: the default path takes the state statistics from numpy
    and does the reward normalisation, z-scoring and packing on the device).

    ``seeds_per_gpu`` = K > 1: K independent seeds of this config trained side by side on this
    rank's GPU -- the reference launcher's AGENTS_PER_GPU (ensemble_sweeps/launch.sh:12,84-94: K
    agents per GPU, each its own run).  Seed k of rank r is ``config.seed + r K + k``; every seed
    has its own initial weights, Philox stream, optimiser state, logging window, evaluation seed
    (ref:1547-1556: ``seed=config.seed`` of ITS run) and ``checkpoint_{t}.pt`` files (under
    ``<checkpoints_path>/seed_<seed>/``), and is bit-identical to ``train()`` of that seed alone;
    the steps of all K run as one ``SeedGroup`` (iqlpref_amd.multi).  With ``reward_model_root``
    (iql_eval.py:143-146: the reward model is tied to the seed) every seed trains on its own
    relabelled dataset, otherwise all share one device buffer.  Records of the loggers carry a
    ``seed`` entry; the metric all-gather carries K x world records.  Returns the list of K
    trainers (the single trainer when K = 1).

    ``vector_env``: the environment factory handed to ``eval_actor`` (default: gym's AsyncVectorEnv);
    given one, evaluation runs through ``eval_actor`` even without a gym ``env``.
    ``index_stream(slot, first_step, n) -> int64 [n, batch_size]`` (tests): the replay rows of steps
    ``first_step .. first_step + n - 1`` of seed slot ``slot`` instead of the on-device Philox
    stream -- how a run of the reference's own ``train()`` (whose sampler draws from torch's global
    CPU generator, ref:212-214) is replayed step for step."""
    # one process per GPU: under torchrun this rank owns cuda:<LOCAL_RANK>, and everything
    # below (process group, buffer, trainer, metric all-gather) lives there
    bound = D.local_device()
    if bound is not None:
        config.device = bound
    rank = D.init_from_env(device=config.device)
    K = int(seeds_per_gpu)
    if K < 1:
        raise ValueError("seeds_per_gpu must be >= 1")
    if env is None and (state_dim is None or action_dim is None):
        import gym
        env = gym.make(config.env)
    if state_dim is None:
        state_dim = env.observation_space.shape[0]
        action_dim = env.action_space.shape[0]
    seeds = [D.rank_seed(config.seed, K) + k for k in range(K)]  # one independent seed per rank / GPU and slot

    # ---- datasets: one per seed when the reward model is tied to the seed, else one for all ----
    source = dataset if dataset is not None else raw_dataset
    # iql_eval.py:143-146 ties the reward model to the RUN's seed: rank r / slot k trains seed
    # seeds[k] on the dataset relabelled by f"{root}_{seeds[k]}" -- also with one seed per GPU (under
    # torchrun rank r > 0 trains config.seed + r, not the seed TrainConfig.__post_init__ saw)
    per_seed_data = bool(config.reward_model_root)
    cfgs = seed_configs(config, seeds)
    buffers, stats = [], []
    for k in range(K if per_seed_data else 1):
        cfg_k = cfgs[k]
        ds = build_dataset(cfg_k, env, source) if (cfg_k.reward_model_path or dataset is None) else dataset
        if per_seed_data and ds is source:
            ds = {key: np.array(val) for key, val in ds.items()}
        buf, state_mean, state_std = _prepare_replay(config, ds, state_dim, action_dim, host_prep)
        buffers.append(buf), stats.append((state_mean, state_std))
    if not per_seed_data:
        buffers, stats = buffers * K, stats * K
    if max_action is None:
        max_action = float(env.action_space.high[0])

    ckpt_dirs = [None] * K
    if config.checkpoints_path is not None:
        print(f"Checkpoints path: {config.checkpoints_path}")
        os.makedirs(config.checkpoints_path, exist_ok=True)
        import yaml
        with open(os.path.join(config.checkpoints_path, "config.yaml"), "w") as f:
            yaml.safe_dump(asdict(config), f)
        for k in range(K):
            ckpt_dirs[k] = config.checkpoints_path if K == 1 else os.path.join(config.checkpoints_path, f"seed_{seeds[k]}")
            os.makedirs(ckpt_dirs[k], exist_ok=True)

    if K > 1 and config.load_model != "":
        import warnings
        warnings.warn(f"seeds_per_gpu={K} with load_model set: all {K} seeds start from the SAME checkpoint "
                      f"({config.load_model}) and differ only in their sample streams", stacklevel=2)
    trainers = [_build_trainer(config, seeds[k], state_dim, action_dim, max_action, precision) for k in range(K)]
    group = None
    if K > 1:
        from .multi import SeedGroup
        group = SeedGroup(trainers)

    if logger is None:
        try:
            import wandb
            wandb.init(config=asdict(config), project=config.project, group=config.group, name=config.name)
            if K == 1:
                logger = lambda d, step: wandb.log(d, step=step)
            else:  # one process, K runs: the records of seed s go under "seed<s>/"
                logger = lambda d, step: wandb.log({(f"seed{int(d['seed'])}/{n}" if "seed" in d else n): v
                                                    for n, v in d.items() if n != "seed"}, step=step)
        except ImportError:
            logger = lambda d, step: print(f"[{step}] " + " ".join(f"{n}={v:.5g}" for n, v in d.items()))
    tag = (lambda rec, k: rec) if K == 1 else (lambda rec, k: dict(rec, seed=seeds[k]))

    total = int(config.max_timesteps)
    t = 0
    t_window = time.perf_counter()
    windows: List[Optional[torch.Tensor]] = [None] * K
    last: List[Dict[str, float]] = [{} for _ in range(K)]
    while t < total:
        # run to the next logging / evaluation boundary in one library call (ref:1533-1544)
        nxt = min(total, (t // config.log_freq + 1) * config.log_freq,
                  (t // config.eval_freq + 1) * config.eval_freq)
        idx = None
        if index_stream is not None:
            idx = [index_stream(k, t, nxt - t).to(device=config.device, dtype=torch.int64) for k in range(K)]
        if group is None:
            losses = [trainers[0].train_steps(buffers[0], nxt - t, config.batch_size,
                                              indices=None if idx is None else idx[0])]
        else:
            losses = group.train_steps(buffers, nxt - t, config.batch_size, return_losses=True, indices=idx)
        for k in range(K):
            windows[k] = losses[k] if t % config.log_freq == 0 else torch.cat([windows[k], losses[k]])
        t = nxt
        if t % config.log_freq == 0:
            means = [w.mean(dim=0).tolist() for w in windows]  # the only host syncs of the window
            now = time.perf_counter()
            for k, mean in enumerate(means):
                rec = {"value_loss": mean[0], "q_loss": mean[1], "actor_loss": mean[2]}
                logger(tag(dict(rec), k), trainers[k].total_it)
                last[k] = dict(rec, steps_per_sec=windows[k].shape[0] / max(now - t_window, 1e-9))
            t_window = now
        if t % config.eval_freq == 0:
            if group is not None:
                group.synchronize()
            records = []
            for k, trainer in enumerate(trainers):
                eval_log: Dict[str, float] = {}
                actor = trainer.actor
                if evaluate is not None:
                    scores, steps_to_goal = evaluate(actor, t)
                elif vector_env is not None or (env is not None and hasattr(env, "spec")):
                    # the evaluation seed of a run is the run's own seed (ref:1547-1556)
                    scores, steps_to_goal = eval_actor(config.env, actor, max_action, stats[k][0], stats[k][1],
                                                       config.device, config.n_episodes, seeds[k],
                                                       vector_env=vector_env)
                else:
                    scores, steps_to_goal = None, []
                if scores is not None:
                    eval_log["mean_score"] = float(np.mean(scores))
                    if "antmaze" in config.env.lower():
                        eval_log["avg_steps_to_goal"] = float(np.mean(steps_to_goal)) if steps_to_goal else -1.0
                    logger(tag(dict(eval_log), k), trainer.total_it)
                if ckpt_dirs[k] is not None:
                    torch.save(trainer.state_dict(), os.path.join(ckpt_dirs[k], f"checkpoint_{t - 1}.pt"))
                records.append(dict(last[k] if t >= config.log_freq else {}, seed=seeds[k],
                                    total_it=trainer.total_it, **eval_log))
            # end-of-eval metric exchange across the per-GPU seeds (the path's only collective)
            recs = D.gather_metric_records(records, device=config.device)
            if rank == 0 and len(recs) > 1:
                logger(D.summarize(recs), trainers[0].total_it)
    if group is not None:
        group.synchronize()
        group.close()
        return trainers
    return trainers[0]


def main(argv=None):
    """CLI with the reference's pyrallis surface: --config_path file.yaml --field value ..."""
    import argparse
    from .iql import load_config
    ap = argparse.ArgumentParser()
    ap.add_argument("--config_path", default=None)
    ap.add_argument("--seeds_per_gpu", type=int, default=int(os.environ.get("AGENTS_PER_GPU", "1")),
                    help="independent seeds trained side by side on each GPU "
                         "(ensemble_sweeps/launch.sh:12 AGENTS_PER_GPU)")
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
    train(load_config(args.config_path, **over), seeds_per_gpu=args.seeds_per_gpu)


if __name__ == "__main__":
    main()
