"""FpvEngine — thin owner of one fpv_env handle over torch-allocated device buffers.

torch is used here only for device memory and stream handles; every computation is a call
into libfpv_env.so through the C ABI of include/fpv_env.h.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class FpvEngine:
    """N parallel hover drones on one GPU.  Mirrors QuadXHoverEnv's constructor arguments
    (simulation/hover.py:11-16) plus the vectorisation / sharding knobs of SURVEY.md §8e."""

    def __init__(self, num_envs: int, device: int | str | torch.device = 0, *, render: bool = False,
                 noise_ratio: float = 0.02, seed: int = 0, id_offset: int = 0, precision: str = "f32",
                 ledger_capacity: int = 0, task: str = "hover", domain_rand: bool = False, detector: str | None = None):
        if not torch.cuda.is_available():
            raise RuntimeError("FpvEngine needs a HIP device: the MI355X extension has no CPU fallback")
        self.lib = L.load()
        dev = torch.device(device if not isinstance(device, int) else f"cuda:{device}")
        if dev.type != "cuda":
            raise ValueError("FpvEngine runs on a HIP device only")
        self.device = dev
        self.num_envs = int(num_envs)
        self.id_offset = int(id_offset)
        self.dtype = torch.float64 if precision == "f64" else torch.float32
        self._f64 = precision == "f64"
        if task not in ("hover", "yaw"):
            raise ValueError("task must be 'hover' (simulation/hover.py) or 'yaw' (simulation/yaw.py)")
        self.task = task
        if detector is None:         # hover: the reference's own pixel rules (hover.py:157-222); the yaw task has no pixel detector
            detector = "pixel" if task == "hover" else "analytic"
        if detector not in ("analytic", "pixel"):
            raise ValueError("detector must be 'pixel' (the 128 x 128 rules of hover.py:157-222, docs/SPEC.md §6b; the default) "
                             "or 'analytic' (the continuous projection of SPEC §6)")
        self.detector = detector
        self.noise_ratio = float(noise_ratio)
        cfg = L.FpvConfig(self.num_envs, dev.index or 0, L.FPV_F64 if self._f64 else L.FPV_F32,
                          0 if render else 1, L.TASK_YAW if task == "yaw" else L.TASK_HOVER, float(noise_ratio),
                          int(seed), int(id_offset), int(ledger_capacity), 1 if domain_rand else 0,
                          L.DETECTOR_PIXEL if detector == "pixel" else L.DETECTOR_ANALYTIC)
        h = C.c_void_p()
        L.check(self.lib.fpv_create(C.byref(cfg), C.byref(h)), "fpv_create")
        self._h = h
        sw, ow, aw = C.c_int32(), C.c_int32(), C.c_int32()
        L.check(self.lib.fpv_dims(self._h, C.byref(sw), C.byref(ow), C.byref(aw)), "fpv_dims")
        self.state_words, self.obs_words, self.act_words = sw.value, ow.value, aw.value
        n = self.num_envs
        with torch.cuda.device(dev):
            self.obs = torch.empty((n, self.obs_words), dtype=self.dtype, device=dev)
            self.reward = torch.empty((n,), dtype=self.dtype, device=dev)
            self.terminated = torch.empty((n,), dtype=torch.uint8, device=dev)
            self.truncated = torch.empty((n,), dtype=torch.uint8, device=dev)
            # caller-owned done ledger (fpv_ledger_bind): partitioned per 64-drone tile, see fpv_env.h
            tiles = (n + 63) // 64
            want = int(ledger_capacity) if ledger_capacity > 0 else 2 * n
            rows = (want + tiles - 1) // tiles
            cap = rows * tiles
            self.led_count = torch.zeros((tiles,), dtype=torch.int32, device=dev)
            self.led_idx = torch.zeros((cap,), dtype=torch.int32, device=dev)
            self.led_info = torch.zeros((cap,), dtype=torch.int32, device=dev)
            self.led_tag = torch.zeros((cap,), dtype=torch.int32, device=dev)
            self.led_ret = torch.zeros((cap,), dtype=self.dtype, device=dev)
            self.led_obs = torch.zeros((cap, self.obs_words), dtype=self.dtype, device=dev)
            self._led_row = torch.arange(rows, dtype=torch.int32, device=dev).unsqueeze(0)
        self.ledger_capacity, self.ledger_rows_per_tile, self.tiles = cap, rows, tiles
        led = L.FpvLedger(cap, rows, self.led_count.data_ptr(), self.led_idx.data_ptr(), self.led_info.data_ptr(),
                          self.led_tag.data_ptr(), self.led_ret.data_ptr(), self.led_obs.data_ptr())
        L.check(self.lib.fpv_ledger_bind(self._h, C.byref(led)), "fpv_ledger_bind")
        tp = C.c_void_p()
        L.check(self.lib.fpv_tick_ptr(self._h, C.byref(tp)), "fpv_tick_ptr")
        self.tick_ptr = tp.value  # device address of the agent-tick counter (keys in-kernel RNG streams)

    # -- lifecycle -----------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            torch.cuda.synchronize(self.device)
            self.lib.fpv_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- reference-shaped operations -----------------------------------------------
    def reset(self, mask: torch.Tensor | None = None, obs: torch.Tensor | None = None) -> torch.Tensor:
        obs = self.obs if obs is None else obs
        if mask is not None:
            mask = mask.to(device=self.device, dtype=torch.uint8).contiguous()
        L.check(self.lib.fpv_reset(self._h, _ptr(mask), _ptr(obs), _stream(self.device)), "fpv_reset")
        return obs

    def step(self, actions: torch.Tensor, obs=None, reward=None, terminated=None, truncated=None):
        """actions [N,4] on device, same dtype as the engine.  Returns (obs, reward, terminated, truncated)
        — the engine's own buffers unless output tensors are given."""
        if actions.dtype != self.dtype or actions.device != self.device or not actions.is_contiguous():
            raise ValueError("actions must be a contiguous device tensor of the engine's dtype")
        if actions.shape != (self.num_envs, self.act_words):
            raise ValueError(f"actions must have shape ({self.num_envs}, {self.act_words})")
        obs = self.obs if obs is None else obs
        reward = self.reward if reward is None else reward
        terminated = self.terminated if terminated is None else terminated
        truncated = self.truncated if truncated is None else truncated
        fn = self.lib.fpv_step_f64 if self._f64 else self.lib.fpv_step
        L.check(fn(self._h, _ptr(actions), _ptr(obs), _ptr(reward), _ptr(terminated), _ptr(truncated),
                   _stream(self.device)), "fpv_step")
        return obs, reward, terminated, truncated

    @property
    def partial_rows(self) -> int:
        """Rows of `partials` that step_fused writes (one per 64-drone tile, or per four tiles): fpv_partial_rows."""
        r = C.c_int64()
        L.check(self.lib.fpv_partial_rows(self._h, C.byref(r)), "fpv_partial_rows")
        return int(r.value)

    def step_fused(self, actions, *, done=None, ret=None, partials=None, gamma: float = 0.99, obs=None, reward=None,
                   obs_rms=None, ret_rms=None, snapshot=None):
        """fpv_step_fused: the env step plus VecNormalize's per-step bookkeeping (done mask, discounted-return
        accumulator, moment sums in the first `partial_rows` rows of partials [tiles, 2 * obs_words + 4]) in the same
        kernel.  obs_rms / ret_rms (float64 {mean, var, count}) are updated in place with this step's sums inside the
        same call — by the launch's last workgroup where the launch shape allows, else by the merge kernel — and
        snapshot receives the statistics as they stand afterwards.  fp32 engines only."""
        if self._f64:
            raise ValueError("step_fused runs on fp32 engines")
        if partials is not None and (partials.dtype != torch.float64 or not partials.is_contiguous()
                                     or partials.shape[0] < self.partial_rows
                                     or partials.shape[1] != 2 * self.obs_words + 4):
            raise ValueError(f"partials must be a contiguous float64 [>= partial_rows, {2 * self.obs_words + 4}] device tensor")
        if actions.dtype != self.dtype or actions.device != self.device or not actions.is_contiguous() \
                or actions.shape != (self.num_envs, self.act_words):
            raise ValueError(f"actions must be a contiguous [num_envs, {self.act_words}] device tensor of the engine's dtype")
        obs = self.obs if obs is None else obs
        reward = self.reward if reward is None else reward
        for name, t, words in (("obs_rms", obs_rms, 2 * self.obs_words + 1), ("ret_rms", ret_rms, 3), ("snapshot", snapshot, 44)):
            if t is not None and (t.dtype != torch.float64 or t.device != self.device or not t.is_contiguous() or t.numel() < words):
                raise ValueError(f"{name} must be a contiguous float64 device tensor of at least {words} words")
        if (obs_rms is not None or ret_rms is not None) and partials is None:
            raise ValueError("obs_rms / ret_rms (the in-call statistics update) need partials")
        ex = L.FpvStepExtras(done.data_ptr() if done is not None else None, ret.data_ptr() if ret is not None else None,
                             partials.data_ptr() if partials is not None else None, float(gamma), 0,
                             obs_rms.data_ptr() if obs_rms is not None else None,
                             ret_rms.data_ptr() if ret_rms is not None else None,
                             snapshot.data_ptr() if snapshot is not None else None)
        L.check(self.lib.fpv_step_fused(self._h, _ptr(actions), _ptr(obs), _ptr(reward), _ptr(self.terminated),
                                        _ptr(self.truncated), C.byref(ex), _stream(self.device)), "fpv_step_fused")
        return obs, reward, self.terminated, self.truncated

    # -- state blobs ----------------------------------------------------------------
    def get_state(self) -> torch.Tensor:
        out = torch.empty((self.num_envs, self.state_words), dtype=self.dtype, device=self.device)
        L.check(self.lib.fpv_get_state(self._h, _ptr(out), _stream(self.device)), "fpv_get_state")
        return out

    def set_state(self, state: torch.Tensor) -> None:
        state = state.to(device=self.device, dtype=self.dtype).contiguous()
        if state.shape != (self.num_envs, self.state_words):
            raise ValueError(f"state must be [num_envs, {self.state_words}]")
        L.check(self.lib.fpv_set_state(self._h, _ptr(state), _stream(self.device)), "fpv_set_state")
        torch.cuda.current_stream(self.device).synchronize()  # `state` may be a temporary

    def set_tick(self, tick: int) -> None:
        L.check(self.lib.fpv_set_tick(self._h, C.c_uint64(tick), _stream(self.device)), "fpv_set_tick")

    def get_tick(self) -> int:
        t = C.c_uint64()
        L.check(self.lib.fpv_get_tick(self._h, C.byref(t), _stream(self.device)), "fpv_get_tick")
        return int(t.value)

    # -- done ledger ------------------------------------------------------------------
    def ledger_clear(self) -> None:
        L.check(self.lib.fpv_ledger_clear(self._h, _stream(self.device)), "fpv_ledger_clear")

    def rollout_begin(self, starts_dst=None, starts_src=None, scan_counters=None, scan_sums=None) -> None:
        """fpv_rollout_begin: ledger_clear(), the episode-start carry (starts_dst <- starts_src, uint8 [num_envs]) and the zeroing
        of the words this rollout's ledger scan will accumulate into (int32[4], float64[2]) as one kernel."""
        L.check(self.lib.fpv_rollout_begin(self._h, _ptr(starts_dst), _ptr(starts_src), _ptr(scan_counters), _ptr(scan_sums),
                                           _stream(self.device)), "fpv_rollout_begin")

    def ledger_rows(self):
        """Device-side compaction of the per-tile ledger: (total_appended, flat row indices [k]) — one
        synchronising nonzero over a [tiles, rows_per_tile] mask."""
        counts = self.led_count
        valid = self._led_row < counts.clamp(max=self.ledger_rows_per_tile).unsqueeze(1)
        rows = torch.nonzero(valid.view(-1), as_tuple=False).squeeze(1)
        return counts.sum(), rows

    def ledger_read(self):
        """Synchronising read of the ledger: dict of host numpy arrays sorted by (tag, drone)."""
        import numpy as np
        total, rows = self.ledger_rows()
        idx = self.led_idx[rows].cpu().numpy()
        info = self.led_info[rows].cpu().numpy().view(np.uint32)
        tag = self.led_tag[rows].cpu().numpy().view(np.uint32)
        ret = self.led_ret[rows].cpu().numpy()
        obs = self.led_obs[rows].cpu().numpy()
        order = np.lexsort((idx, tag))
        return {"count": int(total.item()), "idx": idx[order], "info": info[order], "tag": tag[order],
                "ep_return": ret[order], "obs": obs[order]}
