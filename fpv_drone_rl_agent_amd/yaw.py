"""DroneEnv — the yaw-only task of simulation/yaw.py:25-149 as an N = 1 view of the batched engine.

The reference file cannot run at HEAD (its parent class `hover.DroneEnv` does not exist; reward, sphere
detector and angular-velocity helper are missing — SURVEY.md D3).  The surviving surface is kept: 1-D action
Box(-1, 1, (1,), float32) (:37), 12-D observation Box(-1, 1, (12,), float32) (:39-45), `reset(seed, options)
-> (obs, {})` (:76-100), `step(action) -> (obs, reward, terminated, truncated, info)` (:102-149).  What is
missing in the reference is defined in docs/SPEC.md §10 and labelled [BUILD] there.
"""
from __future__ import annotations

import numpy as np
import torch

from .engine import FpvEngine
from .spaces import Box

try:  # pragma: no cover - gymnasium is not in the build image; the reference's class is a gymnasium.Env (yaw.py:25 via hover)
    import gymnasium as _gym  # type: ignore
    _EnvBase = _gym.Env
except Exception:
    _EnvBase = object


class DroneEnv(_EnvBase):
    metadata = {"render_modes": []}

    def __init__(self, render: bool = False, *, device=0, noise_ratio: float = 0.02, seed: int = 0,
                 domain_rand: bool = False):
        if _EnvBase is not object:
            super().__init__()
        self.render = render
        self.action_space = Box(low=-1.0, high=1.0, shape=(1,), dtype=np.float32)                    # yaw.py:37
        self.observation_space = Box(low=-np.ones(12, np.float32), high=np.ones(12, np.float32), shape=(12,),
                                     dtype=np.float32)                                                # yaw.py:41-45
        self.hardcoded_roll, self.hardcoded_pitch, self.hardcoded_throttle = 0.0, 0.0, -1.0          # yaw.py:48-50
        self.max_steps, self.flight_dome_size = 400, 3.0
        self._engine = FpvEngine(1, device, task="yaw", noise_ratio=noise_ratio, seed=seed, domain_rand=domain_rand,
                                 ledger_capacity=4)
        self._act = torch.zeros((1, 1), dtype=torch.float32, device=self._engine.device)
        self.step_count, self.termination, self.truncation, self.info = 0, False, False, {}

    def reset(self, seed=None, options=None):
        if _EnvBase is not object:
            super().reset(seed=seed)                # yaw.py:77: seeds gymnasium's RNG only
        obs = self._engine.reset()
        self._engine.ledger_clear()
        self.step_count, self.termination, self.truncation, self.info = 0, False, False, {}
        return obs[0].cpu().numpy(), {}                                                               # yaw.py:99-100

    def step(self, action):
        self._act[0, 0] = float(np.asarray(action).reshape(-1)[0])                                   # yaw.py:105
        obs, rew, term, trunc = self._engine.step(self._act)
        self.termination, self.truncation = bool(term[0].item()), bool(trunc[0].item())
        self.info = {}                                                                                # yaw.py:136
        if self.termination or self.truncation:
            led = self._engine.ledger_read()
            self._engine.ledger_clear()
            if int(led["info"][-1]) & 4:
                self.info["out_of_bounds"] = True                                                     # yaw.py:142
            o = led["obs"][-1].astype(np.float32)
        else:
            o = obs[0].cpu().numpy()
        self.step_count += 1
        return o, float(rew[0].item()), self.termination, self.truncation, self.info

    def close(self):
        if getattr(self, "_engine", None) is not None:
            self._engine.close()
            self._engine = None
