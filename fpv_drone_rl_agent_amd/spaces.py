"""Box space: gymnasium's if it is installed, otherwise a structural stand-in with the same
attributes (low, high, shape, dtype, sample, contains) — gymnasium is a dependency of the
reference (simulation/hover.py:2-3) but is not importable in this image."""
from __future__ import annotations

import numpy as np

try:  # pragma: no cover - depends on the environment
    from gymnasium.spaces import Box  # type: ignore
    HAVE_GYMNASIUM = True
except Exception:  # ImportError or a broken install
    HAVE_GYMNASIUM = False

    class Box:  # type: ignore[no-redef]
        def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
            low, high = np.asarray(low, dtype=dtype), np.asarray(high, dtype=dtype)
            if shape is None:
                shape = low.shape
            self.low = np.broadcast_to(low, shape).astype(dtype)
            self.high = np.broadcast_to(high, shape).astype(dtype)
            self.shape, self.dtype = tuple(shape), np.dtype(dtype)
            self._rng = np.random.default_rng(seed)

        def sample(self):
            lo = np.where(np.isfinite(self.low), self.low, -1.0)
            hi = np.where(np.isfinite(self.high), self.high, 1.0)
            return self._rng.uniform(lo, hi).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

        def __repr__(self):
            return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"


def hover_action_space():
    """simulation/hover.py:59-61"""
    return Box(low=-np.ones(4), high=np.ones(4), dtype=np.float64)


def hover_observation_space():
    """simulation/hover.py:67-70"""
    return Box(low=-np.inf * np.ones(20), high=np.inf * np.ones(20), dtype=np.float64)
