"""Device ops of the PPO rollout path — thin ctypes calls into libfpv_env.so.

GAE: SB3 RolloutBuffer.compute_returns_and_advantage; RunningNorm: SB3 VecNormalize's
RunningMeanStd + clip (simulation/train_hover.py:43).  No torch arithmetic here.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _need(t, dtype, name):
    if t.dtype != dtype or not t.is_cuda or not t.is_contiguous():
        raise ValueError(f"{name} must be a contiguous device tensor of dtype {dtype}")


def gae(rewards, values, episode_starts, last_values, last_dones, gamma=0.99, lam=0.95,
        advantages=None, returns=None, stats=None, mean_std=None):
    """rewards/values [T,N] f32, episode_starts [T,N] u8, last_values [N] f32, last_dones [N] u8.
    Returns (advantages, returns); stats (optional f64[3]) gets {count, sum A, sum A^2}, mean_std (optional f64[2], needs
    stats) {mean, population std} of this shard's advantages in the same launch."""
    T, n = rewards.shape
    for t, dt, nm in ((rewards, torch.float32, "rewards"), (values, torch.float32, "values"),
                      (episode_starts, torch.uint8, "episode_starts"), (last_values, torch.float32, "last_values"),
                      (last_dones, torch.uint8, "last_dones")):
        _need(t, dt, nm)
    if advantages is None:
        advantages = torch.empty_like(rewards)
    if returns is None:
        returns = torch.empty_like(rewards)
    if stats is not None:
        _need(stats, torch.float64, "stats")
    if mean_std is not None:
        _need(mean_std, torch.float64, "mean_std")
    lib = L.load()
    L.check(lib.fpv_gae(T, n, _ptr(rewards), _ptr(values), _ptr(episode_starts), _ptr(last_values),
                        _ptr(last_dones), float(gamma), float(lam), _ptr(advantages), _ptr(returns), _ptr(stats),
                        _ptr(mean_std), _stream(rewards.device)), "fpv_gae")
    return advantages, returns


class RunningNorm:
    """Running mean/var/count of a [N,d] stream with SB3 RunningMeanStd semantics (epsilon-count
    start, parallel-variance merge) and VecNormalize's clip((x-mu)/sqrt(var+eps), +-clip)."""

    def __init__(self, dim: int, device, clip: float = 10.0, eps: float = 1e-8, count_eps: float = 1e-4):
        self.dim, self.clip, self.eps = int(dim), float(clip), float(eps)
        self.rms = torch.zeros(2 * dim + 1, dtype=torch.float64, device=device)
        self.rms[dim:2 * dim] = 1.0
        self.rms[2 * dim] = count_eps
        self.device = torch.device(device)

    @property
    def mean(self):
        return self.rms[: self.dim]

    @property
    def var(self):
        return self.rms[self.dim: 2 * self.dim]

    @property
    def count(self):
        return self.rms[2 * self.dim]

    def __call__(self, x, out=None, update=True):
        _need(x, torch.float32, "x")
        if x.shape[-1] != self.dim:
            raise ValueError("last dimension must equal dim")
        n = x.numel() // self.dim
        if out is None:
            out = torch.empty_like(x)
        L.check(L.load().fpv_normalize(n, self.dim, _ptr(x), _ptr(out), _ptr(self.rms), 1 if update else 0,
                                       self.clip, self.eps, _stream(x.device)), "fpv_normalize")
        return out

    def update_only(self, x):
        _need(x, torch.float32, "x")
        n = x.numel() // self.dim
        L.check(L.load().fpv_normalize(n, self.dim, _ptr(x), C.c_void_p(0), _ptr(self.rms), 1, self.clip, self.eps,
                                       _stream(x.device)), "fpv_normalize")


class RewardNorm:
    """VecNormalize reward path: discounted-return accumulator + running variance of it."""

    def __init__(self, num_envs: int, device, gamma: float = 0.99, clip: float = 10.0, eps: float = 1e-8,
                 count_eps: float = 1e-4):
        self.gamma, self.clip, self.eps = float(gamma), float(clip), float(eps)
        self.ret = torch.zeros(num_envs, dtype=torch.float32, device=device)
        self.rms = torch.tensor([0.0, 1.0, count_eps], dtype=torch.float64, device=device)

    def __call__(self, r, done, out=None, update=True):
        _need(r, torch.float32, "r")
        _need(done, torch.uint8, "done")
        if out is None:
            out = torch.empty_like(r)
        L.check(L.load().fpv_normalize_reward(r.numel(), _ptr(r), _ptr(done), _ptr(self.ret), _ptr(out),
                                              _ptr(self.rms), 1 if update else 0, self.gamma, self.clip, self.eps,
                                              _stream(r.device)), "fpv_normalize_reward")
        return out


def moments_merge(partials, batch_count, obs_rms=None, ret_rms=None, snapshot=None, obs_rms_in=None, ret_rms_in=None):
    """Fold the per-tile sums written by FpvEngine.step_fused into the running statistics (fpv_moments_merge_to).
    The observation width follows from the row width (2 O + 4).  obs_rms_in / ret_rms_in: read the statistics from
    these buffers instead and write obs_rms / ret_rms."""
    _need(partials, torch.float64, "partials")
    o = (partials.shape[1] - 4) // 2
    L.check(L.load().fpv_moments_merge_to(partials.shape[0], o, _ptr(partials), float(batch_count),
                                          _ptr((obs_rms_in if obs_rms_in is not None else obs_rms) if obs_rms is not None else None),
                                          _ptr((ret_rms_in if ret_rms_in is not None else ret_rms) if ret_rms is not None else None),
                                          _ptr(obs_rms), _ptr(ret_rms), _ptr(snapshot), _stream(partials.device)),
            "fpv_moments_merge_to")


def advantage_stats(gathered, out):
    """gathered [world, >=3] f64 rows {count, sum, sum_sq} -> out[2] = {mean, population std} (fpv_advantage_stats)."""
    _need(gathered, torch.float64, "gathered")
    L.check(L.load().fpv_advantage_stats(_ptr(gathered), gathered.shape[0], gathered.shape[1], _ptr(out),
                                         _stream(gathered.device)), "fpv_advantage_stats")


def rank_merge(gathered, obs_words, *, base_obs_rms=None, base_ret_rms=None, obs_rms=None, ret_rms=None):
    """fpv_rank_merge: gathered [world, >= 3 + 2 O + 1 + 3] f64 rows {adv stats, obs_rms, ret_rms} after a rollout every rank began
    from the base statistics -> obs_rms / ret_rms = statistics of the union of all ranks' samples (merged in rank order)."""
    _need(gathered, torch.float64, "gathered")
    for t in (base_obs_rms, base_ret_rms, obs_rms, ret_rms):
        if t is not None:
            _need(t, torch.float64, "statistics")
    L.check(L.load().fpv_rank_merge(_ptr(gathered), gathered.shape[0], gathered.shape[1], int(obs_words), _ptr(base_obs_rms),
                                    _ptr(base_ret_rms), _ptr(obs_rms), _ptr(ret_rms), _stream(gathered.device)), "fpv_rank_merge")


def ledger_scan(engine, T, *, snapshots, clip_obs, eps, tobs, t_idx, d_idx, counters, sums, prezeroed=False, status_out=None,
                finalize=None):
    """fpv_ledger_scan on the engine's done ledger: episode statistics + the dense list of truncated-not-terminated
    episode ends of the last T agent steps (see include/fpv_env.h).  No host synchronisation.
    prezeroed: counters / sums were zeroed by this rollout's engine.rollout_begin(scan_counters=, scan_sums=).
    status_out: >= 32 bytes that receive {counters[4], sums[2]} at the end of the launch — a device tensor or a PINNED host tensor
    (the kernel writes it directly; no copy is queued).
    finalize = {raw [T, n] f32, snapshots [T, >= 43] f64, clip, eps, out [T, n] f32}: reward_finalize(...) in the same launch
    (fpv_ledger_scan_finalize)."""
    if status_out is not None:
        if status_out.numel() * status_out.element_size() < 32 or not status_out.is_contiguous() or status_out.data_ptr() % 8:
            raise ValueError("status_out: >= 32 contiguous bytes, 8-byte aligned")
        if not status_out.is_cuda and not status_out.is_pinned():
            raise ValueError("status_out on the host must be pinned memory (the device writes it)")
    if finalize is not None:
        raw, fs, out = finalize["raw"], finalize["snapshots"], finalize["out"]
        _need(raw, torch.float32, "raw")
        _need(fs, torch.float64, "snapshots")
        _need(out, torch.float32, "out")
        if raw.shape != (int(T), engine.num_envs) or out.shape != raw.shape or fs.shape[0] < int(T) or fs.shape[1] < 43:
            raise ValueError("finalize: raw / out are [T, num_envs], snapshots [T, >= 43]")
        L.check(L.load().fpv_ledger_scan_finalize(engine._h, _ptr(snapshots), snapshots.shape[1] if snapshots is not None else 0, int(T),
                                                  float(clip_obs), float(eps), _ptr(tobs), _ptr(t_idx), _ptr(d_idx), tobs.shape[0],
                                                  _ptr(counters), _ptr(sums), 1 if prezeroed else 0, _ptr(status_out), _ptr(raw),
                                                  _ptr(fs), fs.shape[1], float(finalize["clip"]), float(finalize["eps"]), _ptr(out),
                                                  _stream(tobs.device)), "fpv_ledger_scan_finalize")
        return
    L.check(L.load().fpv_ledger_scan(engine._h, _ptr(snapshots), snapshots.shape[1] if snapshots is not None else 0, int(T),
                                     float(clip_obs), float(eps), _ptr(tobs), _ptr(t_idx), _ptr(d_idx), tobs.shape[0],
                                     _ptr(counters), _ptr(sums), 1 if prezeroed else 0, _ptr(status_out), _stream(tobs.device)),
            "fpv_ledger_scan")


def ledger_scatter(counters, t_idx, d_idx, values, gamma, rewards):
    """rewards[t_idx[k], d_idx[k]] += gamma * values[k] for k < counters[0] (fpv_ledger_scatter)."""
    L.check(L.load().fpv_ledger_scatter(_ptr(counters), _ptr(t_idx), _ptr(d_idx), _ptr(values), float(gamma), _ptr(rewards),
                                        rewards.shape[1], t_idx.shape[0], _stream(rewards.device)), "fpv_ledger_scatter")


def reward_finalize(raw, snapshots, clip=10.0, eps=1e-8, out=None):
    """out[t] = clip(raw[t] / sqrt(ret_var_t + eps)) with ret_var_t = snapshots[t, 42] (fpv_reward_finalize)."""
    _need(raw, torch.float32, "raw")
    _need(snapshots, torch.float64, "snapshots")
    T, n = raw.shape
    if out is None:
        out = torch.empty_like(raw)
    L.check(L.load().fpv_reward_finalize(T, n, _ptr(raw), _ptr(snapshots), float(clip), float(eps), _ptr(out),
                                         _stream(raw.device)), "fpv_reward_finalize")
    return out


def noop(device) -> None:
    """Empty kernel through the library's launch path (timing calibration)."""
    L.check(L.load().fpv_noop(_stream(torch.device(device))), "fpv_noop")


def shuffle_indices(out: torch.Tensor, seed: int, counter: int) -> torch.Tensor:
    """out[i] = the i-th row of a PPO epoch's order: a permutation of 0..n-1 chosen by (seed, counter) (fpv_shuffle_indices: a keyed
    bijection, one launch without memory reads — where torch.randperm sorts 2 x 10^6 random keys)."""
    if out.dtype != torch.int64 or not out.is_contiguous() or not out.is_cuda or out.dim() != 1:
        raise ValueError("shuffle_indices needs a contiguous 1-D int64 device tensor")
    L.check(L.load().fpv_shuffle_indices(_ptr(out), out.numel(), C.c_uint64(seed & (2 ** 64 - 1)), C.c_uint64(counter & (2 ** 64 - 1)),
                                         _stream(out.device)), "fpv_shuffle_indices")
    return out


def ppo_kl_gate(epoch_stats: torch.Tensor, limit: float, gate: torch.Tensor) -> None:
    """fpv_ppo_kl_gate behind an epoch's minibatches: gate = int32[2] {stopped, epochs that counted}; epoch_stats = fp32 [n_minibatches, 4]
    (column 2: approx_kl).  An epoch that started unstopped counts, and stops the ones behind it when its mean approx_kl > limit."""
    if epoch_stats.dtype != torch.float32 or not epoch_stats.is_contiguous() or epoch_stats.dim() != 2 or epoch_stats.shape[1] != 4:
        raise ValueError("ppo_kl_gate: epoch_stats is a contiguous fp32 [n_minibatches, 4] device tensor")
    if gate.dtype != torch.int32 or gate.numel() != 2 or not gate.is_cuda:
        raise ValueError("ppo_kl_gate: gate is an int32[2] device tensor")
    L.check(L.load().fpv_ppo_kl_gate(_ptr(epoch_stats), epoch_stats.shape[0], float(limit), _ptr(gate), _stream(gate.device)),
            "fpv_ppo_kl_gate")


class FusedAdam:
    """clip_grad_norm_ + torch.optim.Adam.step() for ONE flat fp32 parameter tensor as a single launch
    (fpv_adam_clip_step).  state_dict() has torch.optim.Adam's shape ({"state": {0: {step, exp_avg, exp_avg_sq}},
    "param_groups": [...]}), so checkpoints written by the torch optimiser and by this one are interchangeable."""

    def __init__(self, param: torch.Tensor, lr: float = 3e-4, betas=(0.9, 0.999), eps: float = 1e-5,
                 max_grad_norm: float = 0.5):
        if param.dtype != torch.float32 or not param.is_contiguous() or not param.is_cuda or param.dim() != 1:
            raise ValueError("FusedAdam needs one flat contiguous fp32 device tensor")
        self.param, self.lr, self.betas, self.eps, self.max_grad_norm = param, float(lr), tuple(betas), float(eps), float(max_grad_norm)
        self._step2 = torch.zeros(2, dtype=torch.int64, device=param.device)   # {step, kernel-internal arrival counter}
        self.state = {0: {"step": self._step2[0],
                          "exp_avg": torch.zeros_like(param.data), "exp_avg_sq": torch.zeros_like(param.data)}}
        self.grad_norm = torch.zeros((), dtype=torch.float32, device=param.device)
        self.lib = L.load()

    def step(self, grad: torch.Tensor | None = None, gate: torch.Tensor | None = None) -> None:
        """gate: optional int32 device tensor (fpv_adam_clip_step_gated): a launch that finds gate[0] != 0 changes nothing."""
        g = self.param.grad if grad is None else grad
        if g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.numel() != self.param.numel():
            raise ValueError("FusedAdam.step: gradient must be a contiguous fp32 tensor of the parameter's size")
        st = self.state[0]
        if gate is not None:
            if gate.dtype != torch.int32 or not gate.is_cuda:
                raise ValueError("FusedAdam.step: gate is an int32 device tensor")
            L.check(self.lib.fpv_adam_clip_step_gated(_ptr(self.param.data), _ptr(g), _ptr(st["exp_avg"]), _ptr(st["exp_avg_sq"]),
                                                      _ptr(st["step"]), _ptr(self.grad_norm), self.param.numel(), self.lr,
                                                      self.betas[0], self.betas[1], self.eps, self.max_grad_norm, _ptr(gate),
                                                      _stream(self.param.device)), "fpv_adam_clip_step_gated")
            return
        L.check(self.lib.fpv_adam_clip_step(_ptr(self.param.data), _ptr(g), _ptr(st["exp_avg"]), _ptr(st["exp_avg_sq"]),
                                            _ptr(st["step"]), _ptr(self.grad_norm), self.param.numel(), self.lr,
                                            self.betas[0], self.betas[1], self.eps, self.max_grad_norm,
                                            _stream(self.param.device)), "fpv_adam_clip_step")

    def state_dict(self) -> dict:
        st = self.state[0]
        return {"state": {0: {"step": st["step"].clone(), "exp_avg": st["exp_avg"].clone(),
                              "exp_avg_sq": st["exp_avg_sq"].clone()}},
                "param_groups": [{"lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": 0, "amsgrad": False,
                                  "params": [0]}]}

    def load_state_dict(self, sd: dict) -> None:
        src = sd["state"].get(0) if sd.get("state") else None
        st = self.state[0]
        if src:
            st["step"].copy_(torch.as_tensor(src["step"]).to(torch.int64))
            st["exp_avg"].copy_(src["exp_avg"])
            st["exp_avg_sq"].copy_(src["exp_avg_sq"])
        else:
            for t in st.values():
                t.zero_()
        if sd.get("param_groups"):
            g = sd["param_groups"][0]
            self.lr, self.betas, self.eps = float(g["lr"]), tuple(g["betas"]), float(g["eps"])
