"""Minimal stand-ins for ``gym.spaces`` (gym is not installed in this image).

Only what the reference's env surface touches: ``Box(low, high, dtype)`` with ``low/high/shape/dtype/
sample/contains`` (hsr/mujoco_env.py:44-56, hsr/env.py:152-154,161-165, rl_utils/argparse.py:57-59)."""
from __future__ import annotations

import numpy as np


class Space:
    def sample(self):
        raise NotImplementedError

    def seed(self, seed=None):
        self.np_random = np.random.default_rng(seed)
        return [seed]


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        low = np.asarray(low, dtype=np.float64)
        high = np.asarray(high, dtype=np.float64)
        if shape is not None:
            low = np.broadcast_to(low, shape).copy()
            high = np.broadcast_to(high, shape).copy()
        assert low.shape == high.shape
        self.low, self.high = low.astype(dtype), high.astype(dtype)
        self.shape, self.dtype = low.shape, np.dtype(dtype)
        self.np_random = np.random.default_rng()

    def sample(self, n=None, rng=None):
        """One sample (shape) or n samples ([n, *shape]); unbounded sides sample a standard normal."""
        rng = rng or self.np_random
        shp = self.shape if n is None else (n,) + self.shape
        lo = np.where(np.isfinite(self.low), self.low, -1.0)
        hi = np.where(np.isfinite(self.high), self.high, 1.0)
        return rng.uniform(lo, hi, shp).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape[-len(self.shape):] == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def __repr__(self):
        return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"


def space_to_size(space) -> int:
    """rl_utils/gym.py:80-90 for the only case the driver touches (hsr/control.py:49,70)."""
    if isinstance(space, Box):
        return int(np.prod(space.shape))
    raise NotImplementedError(type(space))
