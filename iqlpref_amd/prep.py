"""Dataset preparation on the device.

Between the relabel and the first optimisation step the reference runs Python loops over
the N transitions ("ref:" = /root/reference/algorithms/offline/iql.py): the keep mask and
episode-step counter (ref:701-716), the per-episode return range (ref:344-360), the reward
normalisation (ref:363-401), the state statistics and z-scoring (ref:132-139, 1438-1448) and
the buffer load (ref:193-209).  Here they are kernels of libiqlhip.so (csrc/prep.hip) over
device tensors; ``prepare_replay`` chains them so that the dataset crosses PCIe once and the
normalised states are written straight into the packed replay rows.

Parity: masks, counters and trajectory lengths are exact; episode returns (and with them every
reward normalisation) are bit-identical to the reference (double sums in transition order,
numpy's in-place float32 arithmetic); state mean / std are accumulated in double in a fixed
order and differ from numpy's float32 row-order sums at the 1e-6 level (pass
``stats="host"`` to use numpy's values instead).
"""
import ctypes as C
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr

_RANGE_ENVS = ("halfcheetah", "hopper", "walker2d")


def _flags(x, dev) -> torch.Tensor:
    """bool / 0-1 array (numpy or tensor) -> contiguous uint8 device tensor."""
    t = torch.as_tensor(np.asarray(x).reshape(-1).astype(np.uint8) if not torch.is_tensor(x) else x.reshape(-1))
    return t.to(device=dev, dtype=torch.uint8).contiguous()


def keep_mask_and_steps(terminals, timeouts, max_episode_steps: int, terminate_on_end: bool = False,
                        device="cuda") -> Tuple[torch.Tensor, torch.Tensor]:
    """ref:701-716 on the device: (keep[N-1] bool, ep_steps[N-1] int64) tensors."""
    lib = _lib.load()
    dev = _lib.require_gpu(device)
    term = _flags(terminals, dev)
    tmo = None if timeouts is None else _flags(timeouts, dev)
    n = term.shape[0]
    keep = torch.empty(max(n - 1, 0), dtype=torch.uint8, device=dev)
    steps = torch.empty(max(n - 1, 0), dtype=torch.int64, device=dev)
    if n > 1:
        with torch.cuda.device(dev):
            check(lib.iqlhip_prep_keep_mask(ptr(term), ptr(tmo), n, int(max_episode_steps),
                                            int(bool(terminate_on_end)), ptr(keep), ptr(steps), stream_ptr()))
    return keep.bool(), steps


def return_reward_range(rewards: torch.Tensor, terminals, max_episode_steps: int):
    """ref:344-360 on the device: (min_ret, max_ret, trj_lens float64 tensor [N])."""
    lib = _lib.load()
    dev = _lib.require_gpu(rewards.device)
    rew = rewards.reshape(-1).to(torch.float32).contiguous()
    term = _flags(terminals, dev)
    n = rew.shape[0]
    trj = torch.empty(n, dtype=torch.float64, device=dev)
    lo, hi = C.c_double(), C.c_double()
    with torch.cuda.device(dev):
        try:
            check(lib.iqlhip_prep_reward_range(ptr(rew), ptr(term), n, int(max_episode_steps), ptr(trj),
                                               C.byref(lo), C.byref(hi), stream_ptr()))
        except ValueError as e:  # the reference fails on min([]) here
            raise AssertionError("dataset holds no complete episode") from e
    return lo.value, hi.value, trj


def reward_ops(env_name: str, normalize_reward: int):
    """Which steps of ref:363-401 apply: (needs_range, sub_first, scale, sub_one), or None when
    the reference leaves the rewards of this environment untouched."""
    if any(s in env_name for s in _RANGE_ENVS):
        return True, 0, 1, 0
    if "antmaze" not in env_name:
        return None
    if normalize_reward == 1:
        return False, 0, 0, 1
    if normalize_reward in (2, 3):
        sub_first = 0
    elif normalize_reward in (4, 5):
        sub_first = 1
    else:
        sub_first = 2
    return True, sub_first, 1, 0 if normalize_reward in (2, 4, 6) else 1


def modify_reward(rewards: torch.Tensor, terminals, env_name: str, normalize_reward: int,
                  max_episode_steps: int = 1000) -> None:
    """ref:363-401 in place on a float32 device tensor ``rewards`` [N]."""
    ops = reward_ops(env_name, normalize_reward)
    if ops is None:
        return
    if rewards.dtype != torch.float32 or not rewards.is_contiguous():
        raise ValueError("rewards must be a contiguous float32 device tensor (modified in place)")
    lib = _lib.load()
    dev = _lib.require_gpu(rewards.device)
    needs_range, sub_first, scale, sub_one = ops
    lo = hi = 0.0
    trj = None
    if needs_range:
        lo, hi, trj = return_reward_range(rewards, terminals, max_episode_steps)
    with torch.cuda.device(dev):
        check(lib.iqlhip_prep_modify_reward(ptr(rewards), rewards.numel(), ptr(trj), sub_first, scale, sub_one,
                                            lo, hi, int(max_episode_steps), stream_ptr()))


def state_stats(observations: torch.Tensor, eps: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """ref:132-135 on the device: (mean[S], std[S] + eps) float32 tensors."""
    lib = _lib.load()
    dev = _lib.require_gpu(observations.device)
    obs = observations.to(torch.float32).contiguous()
    n, S = obs.shape
    mean = torch.empty(S, dtype=torch.float32, device=dev)
    std = torch.empty(S, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(lib.iqlhip_prep_state_stats(ptr(obs), n, S, float(eps), ptr(mean), ptr(std), stream_ptr()))
    return mean, std


def prepare_replay(dataset: Dict[str, np.ndarray], replay_buffer, *, env_name: str = "",
                   normalize_reward: int = 0, normalize: bool = True, eps: float = 1e-3,
                   max_episode_steps: int = 1000, stats: str = "device"):
    """ref:1435-1456 in one pass on the device: upload the five arrays once, apply
    ``modify_reward``, compute the state statistics, and write the z-scored transitions straight
    into ``replay_buffer``'s packed rows.  Returns (state_mean, state_std) as numpy arrays
    (0 / 1 when ``normalize`` is false) for the evaluation environments.  ``dataset`` itself is
    left untouched (the reference rewrites it in place)."""
    if stats not in ("device", "host"):
        raise ValueError("stats must be 'device' or 'host'")
    dev = replay_buffer._dev
    up = lambda a, dt=torch.float32: torch.as_tensor(np.ascontiguousarray(a)).to(device=dev, dtype=dt)
    obs, act = up(dataset["observations"]), up(dataset["actions"])
    nxt = up(dataset["next_observations"])
    rew = up(np.asarray(dataset["rewards"]).reshape(-1)).contiguous()
    term = _flags(dataset["terminals"], dev)
    if normalize_reward:
        modify_reward(rew, term, env_name, normalize_reward, max_episode_steps)
    if normalize:
        if stats == "device":
            mean, std = state_stats(obs, eps)
        else:
            m, s = dataset["observations"].mean(0), dataset["observations"].std(0) + eps
            mean, std = up(m), up(s)
    else:
        mean = std = None
    replay_buffer.load_device_arrays(obs, act, rew, nxt, term.to(torch.float32), mean, std)
    if mean is None:
        return 0, 1
    if stats == "host":
        return m, s  # numpy's own arrays (dtype included), as ref:1438-1448 hands them to wrap_env
    return mean.cpu().numpy(), std.cpu().numpy()
