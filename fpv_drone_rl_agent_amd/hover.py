"""QuadXHoverEnv — the reference's single-environment surface (simulation/hover.py:10-365) as an
N = 1 view of the batched MI355X engine.

Same constructor keywords, spaces, return arity / types / info keys.  Differences that follow from
the engine (docs/SPEC.md): physics is the build's written spec (PyFlyt parity unpinned), camera
features come from the reference's pixel rules restated without a renderer (`detector="pixel"`, the default) or from the
continuous projection of the panel (`detector="analytic"`), and `reset()` is a constant-state overwrite instead of rebuilding a
simulator (the reference spends ~0.5 s there).
"""
from __future__ import annotations

import numpy as np
import torch

from . import spaces
from .engine import FpvEngine

try:  # pragma: no cover
    import gymnasium as _gym  # type: ignore
    _EnvBase = _gym.Env
except Exception:
    _EnvBase = object


class QuadXHoverEnv(_EnvBase):
    metadata = {"render_modes": []}

    def __init__(self, flight_mode: int = 0, agent_hz: int = 40, render: bool = False, *, device=0,
                 noise_ratio: float = 0.02, seed: int = 0, detector: str = "pixel"):
        if _EnvBase is not object:
            super().__init__()
        if flight_mode != 0:
            raise ValueError("only flight_mode=0 (body rates + thrust) exists on this path (hover.py:92)")
        if agent_hz != 40:
            raise ValueError("agent_hz is fixed at 40 (env_step_ratio 6, hover.py:24) in the HIP kernel")
        self.flight_mode, self.agent_hz, self.render = flight_mode, agent_hz, render  # hover.py:19-21
        self.physics_hz = 240.0
        self.env_step_ratio = int(self.physics_hz / self.agent_hz)
        self.agent_dt = 1.0 / self.agent_hz
        self.max_steps, self.flight_dome_size, self.floor_threshold = 400, 3.0, 0.1  # hover.py:35-38
        self.target_center = np.array([0.0, 0.0], dtype=np.float32)
        self.target_area, self.target_ratio = 0.013, 1.53                        # hover.py:49-51
        self.action_space = spaces.hover_action_space()
        self.observation_space = spaces.hover_observation_space()
        self._engine = FpvEngine(1, device, render=render, noise_ratio=noise_ratio, seed=seed, ledger_capacity=4,
                                 detector=detector)      # "pixel": the 128 x 128 rules of hover.py:157-222 (SPEC §6b)
        self._act = torch.zeros((1, 4), dtype=torch.float32, device=self._engine.device)
        self.info = {"out_of_bounds": False, "collision": False, "env_complete": False, "on_floor": False}
        self.termination = self.truncation = False
        self.step_count = 0
        self.state = np.zeros(20)

    def reset(self, seed=None, options=None):
        # hover.py:72-114; `seed` only seeds gymnasium's RNG in the reference and `options` is ignored
        if _EnvBase is not object:
            super().reset(seed=seed)
        obs = self._engine.reset()
        self._engine.ledger_clear()
        self.step_count, self.termination, self.truncation = 0, False, False
        self.info = {"out_of_bounds": False, "collision": False, "env_complete": False, "on_floor": False}
        self.state = obs[0].double().cpu().numpy()
        return self.state, self.info

    def step(self, action):
        a = np.asarray(action, dtype=np.float32).reshape(1, 4)                   # hover.py:335
        self._act.copy_(torch.from_numpy(a))
        obs, rew, term, trunc = self._engine.step(self._act)
        self.termination, self.truncation = bool(term[0].item()), bool(trunc[0].item())
        if self.termination or self.truncation:
            led = self._engine.ledger_read()
            self._engine.ledger_clear()
            bits = int(led["info"][-1]) & 0xFF
            self.info["out_of_bounds"] = self.info["out_of_bounds"] or bool(bits & 4)   # hover.py:280
            self.info["on_floor"] = self.info["on_floor"] or bool(bits & 8)             # hover.py:289
            self.state = led["obs"][-1].astype(np.float64)          # the terminal observation
        else:
            self.state = obs[0].double().cpu().numpy()
        self.step_count += 1                                                      # hover.py:356
        self.reward = float(rew[0].item())
        return self.state, self.reward, self.termination, self.truncation, self.info

    def close(self):                                                              # hover.py:363-365
        if getattr(self, "_engine", None) is not None:
            self._engine.close()
            self._engine = None


def make_env():
    """simulation/train_hover.py:36-38"""
    return QuadXHoverEnv()
