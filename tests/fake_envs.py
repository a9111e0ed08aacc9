"""Deterministic stand-in environments (test infrastructure, our own code).

``tests/golden/make_fixtures.py`` lets the REFERENCE's evaluation loops (offline/iql.py:265-341
``eval_actor``; custom_offline/iql.py:559-579 ``evaluate``) drive these environments and records
what they saw and returned; the GPU tests let OUR loops drive fresh copies built from the same
seeds and compare.  Everything an environment does is a function of (name, seed, actions), so the
two runs meet the same episodes as long as the actions agree to rounding.

Two API flavours, as the two reference files use them:
  FakeGymEnv        gym < 0.26: ``seed(s)``, ``reset() -> obs``, ``step(a) -> (obs, r, done, info)``
  FakeGymnasiumEnv  gymnasium:  ``reset(seed=s) -> (obs, info)``, ``step(a) -> (obs, r, term, trunc, info)``
"""
import zlib

import numpy as np

DIMS = {"antmaze-medium-diverse-v2": (29, 8), "halfcheetah-medium-v2": (17, 6), "pen-human-v1": (45, 24)}


class _Dynamics:
    """state' = 0.8 state + 0.2 tanh(M a) + noise; episodes of 5..13 steps.  Goal environments
    ("antmaze" in the name) pay a sparse 1 on the last step of a successful episode (decided by
    the seed, not by the actions); the others pay a smooth function of action and state."""

    def __init__(self, name):
        self.name = name
        self.S, self.A = DIMS[name]
        self.goal = "antmaze" in name.lower()
        m_rng = np.random.default_rng(zlib.crc32(name.encode()))
        self.M = m_rng.standard_normal((self.S, self.A)) / np.sqrt(self.A)
        self.rng = np.random.default_rng(0)
        self.t = 0

    def _seed(self, seed):
        self.rng = np.random.default_rng(int(seed))

    def _reset(self):
        self.t = 0
        self.horizon = int(self.rng.integers(5, 14))
        self.success = bool(self.rng.uniform() < 0.6)
        self.state = self.rng.standard_normal(self.S)
        return self.state.copy()

    def _step(self, action):
        a = np.asarray(action, dtype=np.float64).reshape(self.A)
        self.t += 1
        self.state = 0.8 * self.state + 0.2 * np.tanh(self.M @ a) + 0.05 * self.rng.standard_normal(self.S)
        done = self.t >= self.horizon
        if self.goal:
            reward = 1.0 if (done and self.success) else 0.0
        else:
            reward = float(0.1 * a.sum() + 0.01 * self.state[0])
        return self.state.copy(), reward, done


class _Box:
    """The fields of a gym Box space the training entry points read (ref:1400-1401, 1458, 232-233)."""

    def __init__(self, n, high=1.0):
        self.shape, self.high, self.low = (n,), np.full(n, high), np.full(n, -high)
        self.seeded = None

    def seed(self, seed):
        self.seeded = seed


class FakeGymEnv(_Dynamics):
    def __init__(self, name):
        super().__init__(name)
        self.observation_space, self.action_space = _Box(self.S, np.inf), _Box(self.A, 1.0)

    def seed(self, seed):
        self._seed(seed)

    def reset(self):
        return self._reset()

    def step(self, action):
        obs, r, done = self._step(action)
        return obs, r, done, {}

    def close(self):
        pass


class FakeGymnasiumEnv(_Dynamics):
    def reset(self, seed=None):
        if seed is not None:
            self._seed(seed)
        return self._reset(), {}

    def step(self, action):
        obs, r, done = self._step(action)
        return obs, r, done and self.success, done and not self.success, {}


class TransformObservation:
    """What gym.wrappers.TransformObservation does for these loops: f on every observation."""

    def __init__(self, env, f, *_unused):
        self.env, self.f = env, f

    def __getattr__(self, name):
        return getattr(self.env, name)

    def reset(self, **kw):
        out = self.env.reset(**kw)
        return (self.f(out[0]),) + tuple(out[1:]) if isinstance(out, tuple) else self.f(out)

    def step(self, action):
        obs, *rest = self.env.step(action)
        return (self.f(obs), *rest)


class TransformReward(TransformObservation):
    def reset(self, **kw):
        return self.env.reset(**kw)

    def step(self, action):
        obs, r, *rest = self.env.step(action)
        return (obs, self.f(r), *rest)


class SyncVectorEnv:
    """gym.vector.AsyncVectorEnv's contract in one process: ``env_fns`` are called once each;
    ``reset()`` -> obs [n, S]; ``step(actions [n, A])`` -> (obs, rewards, dones, infos) where an
    environment whose episode ended has already been reset (its row of obs is the first
    observation of the next episode)."""

    def __init__(self, env_fns):
        self.envs = [fn() for fn in env_fns]
        self.closed = False
        self.actions_seen = []

    def reset(self):
        return np.stack([e.reset() for e in self.envs])

    def step(self, actions):
        self.actions_seen.append(np.asarray(actions).copy())
        obs, rew, done = [], [], []
        for e, a in zip(self.envs, actions):
            o, r, d, _ = e.step(a)
            if d:
                o = e.reset()
            obs.append(o), rew.append(r), done.append(d)
        return np.stack(obs), np.asarray(rew, dtype=np.float64), np.asarray(done, dtype=bool), [{}] * len(self.envs)

    def close(self):
        self.closed = True


# --------------------------------------------------------------------------- #
# stand-in reward models for the custom_offline relabel loop (cref:158-225): plain functions of
# their inputs, causal over the window (position i depends on positions <= i only, as the
# preference transformer's value head does), so the reference's loop and ours can be compared
# on what they feed the model and what they do with its output.
# --------------------------------------------------------------------------- #
def fake_pt_values(sts, acts, ts, am):
    """[1, L, S], [1, L, A], [1, L], [1, L] -> [1, L] per-position values."""
    # (inputs through float32 first, as a JAX model without x64 takes them)
    sts, acts = np.asarray(sts, np.float32).astype(np.float64), np.asarray(acts, np.float32).astype(np.float64)
    w = np.arange(1, sts.shape[1] + 1, dtype=np.float64)
    v = (sts.sum(-1) + 2.0 * acts.sum(-1) + 0.01 * np.asarray(ts, np.float64)) * np.asarray(am, np.float64) * w
    return v.cumsum(1)


def fake_markov_reward(obs, act):
    obs, act = np.asarray(obs, np.float32).astype(np.float64), np.asarray(act, np.float32).astype(np.float64)
    return (obs.sum(-1) + 2.0 * act.sum(-1)).astype(np.float32)


class Episode:
    """The fields of a minari EpisodeData the loop reads."""

    def __init__(self, observations, actions, terminations):
        self.observations, self.actions, self.terminations = observations, actions, terminations


def make_episodes(seed, S, A, lengths):
    rng = np.random.default_rng(seed)
    eps = []
    for L in lengths:
        term = np.zeros(L, dtype=bool)
        term[-1] = rng.uniform() < 0.5
        eps.append(Episode(rng.standard_normal((L + 1, S)), rng.uniform(-1, 1, (L, A)), term))
    return eps
