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


__all__ = ["Env", "Box", "np_random"]
