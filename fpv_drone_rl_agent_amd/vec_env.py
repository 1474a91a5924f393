"""SB3-VecEnv-shaped batched facade and an on-device VecNormalize.

Stands where `make_vec_env(make_env, n_envs=16, vec_env_cls=SubprocVecEnv)` and
`VecNormalize(env, norm_obs=True, norm_reward=True)` stand in simulation/train_hover.py:42-43,
and `DummyVecEnv` / `VecNormalize.load` in simulation/test_hover.py:7-10.

Two data modes:
  * host mode (default): numpy in, numpy out — drop-in under stock SB3 (BASELINE cfg "policy on
    CPU"); one H2D of actions and one D2H of obs/reward/flags (+ the done rows) per step.
  * device mode (`device_mode=True`): torch device tensors in and out, nothing leaves HBM.
SB3 is not installed in this image: the class subclasses SB3's VecEnv when importable and is
otherwise a structural stand-in with the same method names, shapes and dtypes.
"""
from __future__ import annotations

import time

import numpy as np
import torch

from . import ops, spaces
from .engine import FpvEngine

try:  # pragma: no cover
    from stable_baselines3.common.vec_env import VecEnv as _VecEnvBase  # type: ignore
    HAVE_SB3 = True
except Exception:
    _VecEnvBase = object
    HAVE_SB3 = False


class FpvVecEnv(_VecEnvBase):
    def __init__(self, n_envs: int = 16, *, device=0, render: bool = False, noise_ratio: float = 0.02,
                 seed: int = 0, id_offset: int = 0, device_mode: bool = False):
        self.engine = FpvEngine(n_envs, device, render=render, noise_ratio=noise_ratio, seed=seed,
                                id_offset=id_offset)
        obs_space, act_space = spaces.hover_observation_space(), spaces.hover_action_space()
        if HAVE_SB3:
            super().__init__(n_envs, obs_space, act_space)
        else:
            self.num_envs, self.observation_space, self.action_space = n_envs, obs_space, act_space
        self.device_mode = bool(device_mode)
        self.render_mode = None
        self._actions = torch.zeros((n_envs, 4), dtype=torch.float32, device=self.engine.device)
        self._t_start = time.time()
        self._empty_infos = [{} for _ in range(n_envs)]
        self._pending = False

    # -- VecEnv protocol ---------------------------------------------------------------
    def reset(self):
        obs = self.engine.reset()
        self.engine.ledger_clear()
        return obs if self.device_mode else obs.cpu().numpy()

    def step_async(self, actions) -> None:
        if self.device_mode:
            self._actions.copy_(actions)
        else:
            a = np.asarray(actions, dtype=np.float32).reshape(self.num_envs, 4)
            self._actions.copy_(torch.from_numpy(a), non_blocking=False)
        self.engine.ledger_clear()
        self.engine.step(self._actions)
        self._pending = True

    def step_wait(self):
        if not self._pending:
            raise RuntimeError("step_wait() without step_async()")
        self._pending = False
        e = self.engine
        if self.device_mode:
            dones = torch.bitwise_or(e.terminated, e.truncated).bool()
            return e.obs, e.reward, dones, LedgerInfos(e)
        obs = e.obs.cpu().numpy()
        rewards = e.reward.cpu().numpy()
        term, trunc = e.terminated.cpu().numpy().astype(bool), e.truncated.cpu().numpy().astype(bool)
        dones = term | trunc
        infos = self._empty_infos
        if dones.any():
            infos = list(self._empty_infos)
            led = e.ledger_read()
            now = time.time() - self._t_start
            for k in range(len(led["idx"])):
                i, bits = int(led["idx"][k]), int(led["info"][k])
                infos[i] = {
                    "terminal_observation": led["obs"][k],
                    "TimeLimit.truncated": bool(bits & 2) and not bool(bits & 1),
                    "episode": {"r": float(led["ep_return"][k]), "l": bits >> 8, "t": round(now, 6)},
                    "out_of_bounds": bool(bits & 4), "on_floor": bool(bits & 8),
                    "collision": False, "env_complete": False,
                }
        return obs, rewards, dones, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self) -> None:
        if getattr(self, "engine", None) is not None:
            self.engine.close()
            self.engine = None

    # -- the rest of SB3's abstract surface ---------------------------------------------
    _ATTRS = {"flight_mode": 0, "agent_hz": 40, "max_steps": 400, "flight_dome_size": 3.0,
              "floor_threshold": 0.1, "target_area": 0.013, "target_ratio": 1.53, "render_mode": None}

    def _indices(self, indices):
        if indices is None:
            return list(range(self.num_envs))
        return [indices] if isinstance(indices, int) else list(indices)

    def get_attr(self, attr_name, indices=None):
        if attr_name not in self._ATTRS:
            raise AttributeError(attr_name)
        return [self._ATTRS[attr_name] for _ in self._indices(indices)]

    def set_attr(self, attr_name, value, indices=None):
        raise AttributeError("per-environment attributes are constants of the HIP kernel")

    def env_method(self, method_name, *args, indices=None, **kwargs):
        raise AttributeError(f"no per-environment method {method_name!r} on the batched engine")

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False for _ in self._indices(indices)]

    def seed(self, seed=None):
        return [None] * self.num_envs  # hover.py:72-73: seeding never reaches the simulator

    def get_images(self):
        return [None] * self.num_envs


class LedgerInfos:
    """Device-mode stand-in for the `infos` list: exposes the done ledger as device tensors."""

    def __init__(self, engine: FpvEngine):
        self.tile_count, self.rows_per_tile = engine.led_count, engine.ledger_rows_per_tile
        self.idx, self.info = engine.led_idx, engine.led_info
        self.ep_return, self.terminal_observation = engine.led_ret, engine.led_obs
        self.rows = engine.ledger_rows  # () -> (total, flat row indices); synchronises


def make_vec_env(env_fn=None, n_envs: int = 16, vec_env_cls=None, **kw) -> FpvVecEnv:
    """simulation/train_hover.py:42 — env_fn / vec_env_cls are accepted and ignored: the batch is one engine."""
    return FpvVecEnv(n_envs, **kw)


class VecNormalize:
    """VecNormalize(venv, norm_obs=True, norm_reward=True) — simulation/train_hover.py:43; `.training`
    and `.norm_reward` switches and `save` / `load` as used by train_hover.py:27,63 and test_hover.py:8-10.
    Statistics live on the device (fp64) and are updated by fpv_normalize / fpv_normalize_reward."""

    def __init__(self, venv: FpvVecEnv, training: bool = True, norm_obs: bool = True, norm_reward: bool = True,
                 clip_obs: float = 10.0, clip_reward: float = 10.0, gamma: float = 0.99, epsilon: float = 1e-8):
        self.venv = venv
        self.num_envs = venv.num_envs
        self.observation_space, self.action_space = venv.observation_space, venv.action_space
        self.training, self.norm_obs, self.norm_reward = training, norm_obs, norm_reward
        self.clip_obs, self.clip_reward, self.gamma, self.epsilon = clip_obs, clip_reward, gamma, epsilon
        dev = venv.engine.device
        self.obs_rms = ops.RunningNorm(20, dev, clip=clip_obs, eps=epsilon)
        self.ret_rms = ops.RewardNorm(venv.num_envs, dev, gamma=gamma, clip=clip_reward, eps=epsilon)
        self._obs_n = torch.zeros((venv.num_envs, 20), dtype=torch.float32, device=dev)
        self._rew_n = torch.zeros((venv.num_envs,), dtype=torch.float32, device=dev)
        self._done = torch.zeros((venv.num_envs,), dtype=torch.uint8, device=dev)
        self.old_obs = None

    def _norm_obs_dev(self, raw: torch.Tensor) -> torch.Tensor:
        if not self.norm_obs:
            return raw
        return self.obs_rms(raw, out=self._obs_n, update=self.training)

    def normalize_obs(self, obs):
        """Normalise without touching the statistics (numpy or device tensor)."""
        if not self.norm_obs:
            return obs
        host = not isinstance(obs, torch.Tensor)
        x = torch.as_tensor(np.asarray(obs, np.float32) if host else obs, device=self.venv.engine.device)
        y = self.obs_rms(x.contiguous().float(), update=False)
        return y.cpu().numpy() if host else y

    def reset(self):
        self.venv.reset()
        e = self.venv.engine
        self.ret_rms.ret.zero_()
        out = self._norm_obs_dev(e.obs)
        return out if self.venv.device_mode else out.cpu().numpy()

    def step_async(self, actions):
        self.venv.step_async(actions)

    def step_wait(self):
        obs, rewards, dones, infos = self.venv.step_wait()
        e = self.venv.engine
        obs_n = self._norm_obs_dev(e.obs)
        torch.bitwise_or(e.terminated, e.truncated, out=self._done)
        if self.norm_reward:
            # SB3 updates the return statistics only while training, but always clears finished returns
            rew_n = self.ret_rms(e.reward, self._done, out=self._rew_n, update=self.training)
        else:
            rew_n = e.reward
        if self.venv.device_mode:
            return obs_n, rew_n, dones, infos
        if self.norm_obs and dones.any():
            idx = [i for i in np.nonzero(dones)[0] if "terminal_observation" in infos[i]]
            if idx:                                   # one normalisation launch for all terminal observations
                tn = self.normalize_obs(np.stack([infos[i]["terminal_observation"] for i in idx]))
                for k, i in enumerate(idx):
                    infos[i]["terminal_observation"] = tn[k]
        return obs_n.cpu().numpy(), rew_n.cpu().numpy(), dones, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        self.venv.close()

    # -- persistence (train_hover.py:27,63; test_hover.py:8) --------------------------------
    def save(self, path: str, format=None) -> None:
        """Statistics + switches as an npz archive at exactly `path`.  SB3 pickles the wrapper object; without SB3 in the
        image that format is not reproducible, so a `.pkl` path is refused unless format="npz" says the
        archive is wanted under that name."""
        from .checkpoint import check_stats_save_path
        check_stats_save_path(path, format)
        d = self.obs_rms.dim
        rms, rr = self.obs_rms.rms.cpu().numpy(), self.ret_rms.rms.cpu().numpy()
        with open(path, "wb") as f:
            np.savez(f, obs_mean=rms[:d], obs_var=rms[d:2 * d], obs_count=rms[2 * d], ret_mean=rr[0], ret_var=rr[1],
                     ret_count=rr[2], clip_obs=self.clip_obs, clip_reward=self.clip_reward, gamma=self.gamma,
                     epsilon=self.epsilon, norm_obs=self.norm_obs, norm_reward=self.norm_reward)

    @staticmethod
    def load(load_path: str, venv: FpvVecEnv) -> "VecNormalize":
        from .checkpoint import open_stats_archive
        z = open_stats_archive(load_path)       # raises a clear error for a real pickle (a file written by SB3)
        vn = VecNormalize(venv, norm_obs=bool(z["norm_obs"]), norm_reward=bool(z["norm_reward"]),
                          clip_obs=float(z["clip_obs"]), clip_reward=float(z["clip_reward"]), gamma=float(z["gamma"]),
                          epsilon=float(z["epsilon"]))
        d = vn.obs_rms.dim
        rms = np.concatenate([z["obs_mean"], z["obs_var"], [z["obs_count"]]])
        vn.obs_rms.rms.copy_(torch.from_numpy(rms))
        vn.ret_rms.rms.copy_(torch.tensor([float(z["ret_mean"]), float(z["ret_var"]), float(z["ret_count"])],
                                          dtype=torch.float64))
        return vn
