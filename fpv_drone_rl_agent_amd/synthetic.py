"""Seeded synthetic drone states for benchmarks (BASELINE.md §3 / SURVEY.md §8d).  numpy only:
this generates INPUT DATA on the host; no environment arithmetic happens here."""
from __future__ import annotations

import numpy as np


def synthetic_states(n: int, seed: int = 0) -> np.ndarray:
    """Canonical [n,36] float64 states: position uniform in the ball |p| < 2.5 m with z in [0.3, 2.5];
    roll, pitch ~ U(-0.5, 0.5), yaw ~ U(-pi, pi); body velocity U(-1,1)^3; body rates U(-2,2)^3; motor
    throttle U(0.3, 0.7)^4; step_count ~ U{0..400} so that episode ends are staggered."""
    rng = np.random.Generator(np.random.Philox(seed))
    st = np.zeros((n, 36))
    pos = np.zeros((n, 3))
    todo = np.arange(n)
    while todo.size:
        c = rng.uniform([-2.5, -2.5, 0.3], [2.5, 2.5, 2.5], size=(todo.size, 3))
        ok = np.linalg.norm(c, axis=1) < 2.5
        pos[todo[ok]] = c[ok]
        todo = todo[~ok]
    st[:, 0:3] = pos
    eul = np.stack([rng.uniform(-0.5, 0.5, n), rng.uniform(-0.5, 0.5, n), rng.uniform(-np.pi, np.pi, n)], 1)
    sr, cr = np.sin(eul[:, 0] / 2), np.cos(eul[:, 0] / 2)
    sp, cp = np.sin(eul[:, 1] / 2), np.cos(eul[:, 1] / 2)
    sy, cy = np.sin(eul[:, 2] / 2), np.cos(eul[:, 2] / 2)
    x = st[:, 3] = sr * cp * cy - cr * sp * sy
    y = st[:, 4] = cr * sp * cy + sr * cp * sy
    z = st[:, 5] = cr * cp * sy - sr * sp * cy
    w = st[:, 6] = cr * cp * cy + sr * sp * sy
    vb = rng.uniform(-1, 1, (n, 3))
    R = np.empty((n, 3, 3))
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - w * z); R[:, 0, 2] = 2 * (x * z + w * y)
    R[:, 1, 0] = 2 * (x * y + w * z); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - w * x)
    R[:, 2, 0] = 2 * (x * z - w * y); R[:, 2, 1] = 2 * (y * z + w * x); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    st[:, 7:10] = np.einsum("nij,nj->ni", R, vb)
    st[:, 10:13] = rng.uniform(-2, 2, (n, 3))
    st[:, 13:17] = rng.uniform(0.3, 0.7, (n, 4))
    st[:, 23:26] = eul
    st[:, 34] = rng.integers(0, 401, n).astype(np.float64)
    return st
