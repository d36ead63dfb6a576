"""The few pieces of gym 0.18 the env facade needs (gym itself is optional and absent from the
build image): ``Env`` base, ``spaces.Box`` and ``seeding.np_random`` (see terrain.np_random)."""
import numpy as np

from .terrain import np_random

try:  # pragma: no cover - real gym is used when present
    import gym as _gym
    Env = _gym.Env
except Exception:  # noqa: BLE001
    class Env:
        metadata = {}
        reward_range = (-float("inf"), float("inf"))
        action_space = None
        observation_space = None

        def close(self):
            pass


class Box:
    def __init__(self, low, high, dtype=np.float32):
        self.low = np.asarray(low, dtype=dtype)
        self.high = np.asarray(high, dtype=dtype)
        self.shape = self.low.shape
        self.dtype = np.dtype(dtype)
        self._rng = np.random.RandomState()

    def sample(self):
        lo = np.where(np.isfinite(self.low), self.low, -1.0)
        hi = np.where(np.isfinite(self.high), self.high, 1.0)
        return self._rng.uniform(lo, hi).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and np.all(x >= self.low) and np.all(x <= self.high)


class TimeLimit:
    """gym.wrappers.TimeLimit as gym 0.18 applies it for ``max_episode_steps`` (the reference registers
    ``Modular2DLocomotion-v0`` with 240 * 20 = 4800 steps, gym_rem2D/__init__.py:5-7): after that many steps ``done``
    becomes True and ``info['TimeLimit.truncated']`` says whether the env itself had finished.  The reference's env
    returns the int 0 as info (Modular2DEnv.py:653); a dict replaces it only on the truncating step, as gym does."""

    def __init__(self, env, max_episode_steps):
        self.env = env
        self._max_episode_steps = int(max_episode_steps)
        self._elapsed_steps = None

    def __getattr__(self, name):
        return getattr(self.env, name)

    @property
    def unwrapped(self):
        return getattr(self.env, "unwrapped", self.env)

    def step(self, action):
        assert self._elapsed_steps is not None, "Cannot call env.step() before calling reset()"
        observation, reward, done, info = self.env.step(action)
        self._elapsed_steps += 1
        if self._elapsed_steps >= self._max_episode_steps:
            info = dict(info) if isinstance(info, dict) else {}
            info["TimeLimit.truncated"] = not done
            done = True
        return observation, reward, done, info

    def reset(self, **kwargs):
        self._elapsed_steps = 0
        return self.env.reset(**kwargs)


_REGISTRY = {}


def register(id, entry_point, max_episode_steps=None, kwargs=None):
    """gym.envs.registration.register (the subset the reference uses)."""
    _REGISTRY[id] = (entry_point, max_episode_steps, dict(kwargs or {}))


def make(id, **kwargs):
    """gym.make: REM2D_main.getEnv (:57-67) does ``gym.make('Modular2DLocomotion-v0')``."""
    if id not in _REGISTRY:
        raise KeyError("No registered env with id: %s (registered: %s)" % (id, sorted(_REGISTRY)))
    entry_point, max_steps, kw = _REGISTRY[id]
    kw = dict(kw, **kwargs)
    if isinstance(entry_point, str):
        import importlib
        mod, _, name = entry_point.partition(":")
        entry_point = getattr(importlib.import_module(mod), name)
    env = entry_point(**kw)
    return TimeLimit(env, max_steps) if max_steps else env


# gym_rem2D/__init__.py:5-7
register(id="Modular2DLocomotion-v0", entry_point="gym_rem2d_amd.env:Modular2D", max_episode_steps=240 * 20)

__all__ = ["Env", "Box", "np_random", "TimeLimit", "register", "make"]
