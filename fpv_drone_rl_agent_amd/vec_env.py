"""SB3-VecEnv-shaped batched facade and an on-device VecNormalize.

Stands where `make_vec_env(make_env, n_envs=16, vec_env_cls=SubprocVecEnv)` and
`VecNormalize(env, norm_obs=True, norm_reward=True)` stand in simulation/train_hover.py:42-43,
and `DummyVecEnv` / `VecNormalize.load` in simulation/test_hover.py:7-10.

Two data modes:
  * host mode (default): numpy in, numpy out — drop-in under stock SB3 (BASELINE cfg "policy on
    CPU").  Per step ONE H2D of the actions from a pinned staging buffer and ONE D2H into pinned memory of a single
    device block that the step's kernels fill: flags, observations, rewards (normalised by VecNormalize's kernels when the
    wrapper is there) and the done ledger packed densely (fpv_ledger_pack: terminal observations, episode returns / lengths,
    hover.py:102's flags).  Everything is queued in step_async; the per-step `infos` dicts are made while the device works.
  * device mode (`device_mode=True`): torch device tensors in and out, nothing leaves HBM.
SB3 is not installed in this image: the class subclasses SB3's VecEnv when importable and is
otherwise a structural stand-in with the same method names, shapes and dtypes.
"""
from __future__ import annotations

import ctypes as C
import time
from itertools import repeat

import numpy as np
import torch

from . import _lib as L
from . import ops, spaces
from .engine import FpvEngine

# hover.py:102 — the info dict a live env returns on every step (hover.py:358); a finished one adds SB3's keys
_LIVE_INFO = {"out_of_bounds": False, "collision": False, "env_complete": False, "on_floor": False}


class _HostBlock:
    """What a host-mode step hands to the host, as one device allocation and its pinned mirror:
    [terminated N u8 | truncated N u8 | pad | obs N x O f32 | reward N f32 | pad | ledger header 8 u32 | ledger rows].
    The copy that follows every step takes the block up to `first` ledger rows (a step ends ~N / 31 episodes); the steps that
    end more (a truncation wave after a common reset) fetch the rest with a second copy."""

    def __init__(self, n: int, obs_words: int, device):
        al = lambda x: (x + 15) // 16 * 16
        self.n, self.W = n, 4 + obs_words
        self.first = min(n, max(64, n // 8))
        o_obs = al(2 * n)
        o_rew = o_obs + 4 * n * obs_words
        o_led = al(o_rew + 4 * n)
        self.o_led, self.main_bytes = o_led, o_led + 4 * (8 + self.first * self.W)
        total = o_led + 4 * (8 + n * self.W)
        self.dev = torch.zeros(total, dtype=torch.uint8, device=device)
        self.host = torch.zeros(total, dtype=torch.uint8).pin_memory()
        d = self.dev
        self.term, self.trunc = d[0:n], d[n:2 * n]
        self.obs = d[o_obs:o_rew].view(torch.float32).view(n, obs_words)
        self.rew = d[o_rew:o_rew + 4 * n].view(torch.float32)
        self.led = d[o_led:]
        h = self.host.numpy()
        self.h_term, self.h_trunc = h[0:n], h[n:2 * n]
        self.h_obs = h[o_obs:o_rew].view(np.float32).reshape(n, obs_words)
        self.h_rew = h[o_rew:o_rew + 4 * n].view(np.float32)
        self.h_hdr = h[o_led:o_led + 32].view(np.uint32)
        self.h_rows = h[o_led + 32:].view(np.uint32).reshape(n, self.W)
        self.act_host = torch.zeros((n, 4), dtype=torch.float32).pin_memory()
        self.event = torch.cuda.Event()

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
        self._pending = False
        self._normalizer = None        # the VecNormalize wrapped around this env, host mode: its kernels write into the block
        self._infos = None
        self._blk = None
        if not self.device_mode:
            e = self.engine
            with torch.cuda.device(e.device):
                self._blk = _HostBlock(n_envs, e.obs_words, e.device)
            e.terminated, e.truncated = self._blk.term, self._blk.trunc     # the engine's flags live in the block
            self._raw_obs, self._raw_rew = e.obs, e.reward                  # (raw outputs when a normaliser stands behind)

    # -- VecEnv protocol ---------------------------------------------------------------
    def reset(self):
        obs = self.engine.reset()
        self.engine.ledger_clear()
        return obs if self.device_mode else obs.cpu().numpy()

    def step_async(self, actions) -> None:
        e = self.engine
        if self.device_mode:
            self._actions.copy_(actions)
            e.ledger_clear()
            e.step(self._actions)
            self._pending = True
            return
        b, nz = self._blk, self._normalizer
        b.act_host.numpy()[...] = np.asarray(actions, dtype=np.float32).reshape(self.num_envs, 4)
        self._actions.copy_(b.act_host, non_blocking=True)
        e.ledger_clear()
        rms = None
        # engine.obs / engine.reward name the step's RAW outputs: the block itself, or tensors of their own under a normaliser
        e.obs, e.reward = (b.obs, b.rew) if nz is None else (self._raw_obs, self._raw_rew)
        e.step(self._actions)
        if nz is not None:
            rms = nz._enqueue(e.obs, e.reward, b)                 # VecNormalize's kernels: block <- normalised obs / rewards
        L.check(e.lib.fpv_ledger_pack(e._h, C.c_void_p(rms.data_ptr()) if rms is not None else None,
                                      nz.clip_obs if nz is not None else 0.0, nz.epsilon if nz is not None else 0.0,
                                      C.c_void_p(b.led.data_ptr()), self.num_envs,
                                      C.c_void_p(torch.cuda.current_stream(e.device).cuda_stream)), "fpv_ledger_pack")
        b.host[:b.main_bytes].copy_(b.dev[:b.main_bytes], non_blocking=True)
        b.event.record(torch.cuda.current_stream(e.device))
        # hover.py:102,358: every env returns its own info dict on every step — made here, while the device works
        self._infos = list(map(dict.copy, repeat(_LIVE_INFO, self.num_envs)))
        self._pending = True

    def step_wait(self):
        if not self._pending:
            raise RuntimeError("step_wait() without step_async()")
        self._pending = False
        e = self.engine
        if self.device_mode:
            dones = torch.bitwise_or(e.terminated, e.truncated).bool()
            return e.obs, e.reward, dones, LedgerInfos(e)
        b, infos = self._blk, self._infos
        self._infos = None
        b.event.synchronize()
        obs, rewards = b.h_obs.copy(), b.h_rew.copy()              # (the pinned block is overwritten by the next step)
        dones = (b.h_term | b.h_trunc).astype(bool)
        appended, packed, held = int(b.h_hdr[0]), int(b.h_hdr[1]), int(b.h_hdr[3])
        if appended > held:
            raise RuntimeError(f"done ledger overflow: {appended} episode ends, {held} rows (raise ledger_capacity)")
        if packed:
            if packed > b.first:                                   # a wave of episode ends: the rows behind the first copy
                lo, hi = b.o_led + 32 + 4 * b.first * b.W, b.o_led + 32 + 4 * packed * b.W
                b.host[lo:hi].copy_(b.dev[lo:hi])
            rows = b.h_rows[:packed].copy()
            bits = rows[:, 1]
            tobs = np.ascontiguousarray(rows[:, 4:]).view(np.float32)
            now = round(time.time() - self._t_start, 6)
            for i, bt, r, to in zip(rows[:, 0].tolist(), bits.tolist(), rows[:, 3].view(np.float32).tolist(), tobs):
                infos[i] = {
                    "terminal_observation": to,
                    "TimeLimit.truncated": (bt & 3) == 2,
                    "episode": {"r": r, "l": bt >> 8, "t": now},
                    "out_of_bounds": (bt & 4) != 0, "on_floor": (bt & 8) != 0,
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

    def _enqueue(self, raw_obs, raw_rew, blk):
        """Host mode, called by the env's step_async behind the env-step kernel: SB3's VecNormalize.step_wait as kernels on the
        stream — statistics update (while training), normalised observations and rewards straight into the block the host will
        receive.  Returns the statistics the terminal observations are to be normalised with (None: not at all)."""
        if self.norm_obs:
            self.obs_rms(raw_obs, out=blk.obs, update=self.training)
        else:
            blk.obs.copy_(raw_obs)
        torch.bitwise_or(blk.term, blk.trunc, out=self._done)
        if self.norm_reward:
            # SB3 updates the return statistics only while training, but always clears finished returns
            self.ret_rms(raw_rew, self._done, out=blk.rew, update=self.training)
        else:
            blk.rew.copy_(raw_rew)
        return self.obs_rms.rms if self.norm_obs else None

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
        if self.venv.device_mode:
            self.venv.step_async(actions)
            return
        self.venv._normalizer = self              # the wrapper that steps is the one whose kernels fill the block
        try:
            self.venv.step_async(actions)
        finally:
            self.venv._normalizer = None          # (a step of the bare env returns raw values again)

    def step_wait(self):
        if not self.venv.device_mode:
            return self.venv.step_wait()          # already normalised: the env's step_async queued this wrapper's kernels
        obs, rewards, dones, infos = self.venv.step_wait()
        e = self.venv.engine
        obs_n = self._norm_obs_dev(e.obs)
        torch.bitwise_or(e.terminated, e.truncated, out=self._done)
        if self.norm_reward:
            # SB3 updates the return statistics only while training, but always clears finished returns
            rew_n = self.ret_rms(e.reward, self._done, out=self._rew_n, update=self.training)
        else:
            rew_n = e.reward
        return obs_n, rew_n, dones, infos

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
