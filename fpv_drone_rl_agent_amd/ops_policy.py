"""FusedPolicy — the rollout-time forward of policy.ActorCritic as ONE HIP kernel (fpv_policy_forward):
VecNormalize.normalize_obs -> pi and V MLPs on exact-fp32 MFMA -> Gaussian sample -> log-prob -> Box clip.
The torch module keeps owning the parameters; call load() after every optimiser step."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L
from .policy import ActorCritic


def _ptr(t):
    """device address of a tensor, a raw integer address, or None"""
    if t is None:
        return None
    return t if isinstance(t, int) else t.data_ptr()


class FusedPolicy:
    def __init__(self, module: ActorCritic, device=None, seed: int = 0, id_offset: int = 0, gemm: str = "f32"):
        if module.hidden != 128 or module.obs_dim % 4 or not (4 <= module.obs_dim <= 24) or not (1 <= module.act_dim <= 4):
            raise ValueError("the fused kernel needs hidden 128 (train_hover.py:57), obs_dim in {4..24 step 4}, act_dim 1..4")
        self.obs_dim, self.act_dim = module.obs_dim, module.act_dim
        self.module = module
        p = next(module.parameters())
        if not p.is_cuda:
            raise RuntimeError("FusedPolicy needs the module on a HIP device: there is no CPU fallback")
        self.device = p.device if device is None else torch.device(device)
        self.seed, self.id_offset = int(seed), int(id_offset)
        self.lib = L.load()
        h = C.c_void_p()
        L.check(self.lib.fpv_policy_create(self.device.index or 0, C.byref(h)), "fpv_policy_create")
        self._h = h
        L.check(self.lib.fpv_policy_set_dims(self._h, self.obs_dim, self.act_dim), "fpv_policy_set_dims")
        self.set_gemm(gemm)
        self.load()

    def set_gemm(self, gemm: str) -> None:
        """'f32': exact fp32 MFMA (default).  'bf16x6': 6-term bf16 split on bf16 MFMA, fp32-level error.
        'f32_w16': the fp32 arithmetic on 16x16x4 tiles with 16-wave workgroups (rounding-level differences to 'f32')."""
        modes = {"f32": 0, "bf16x6": 1, "f32_w16": 2}
        if gemm not in modes:
            raise ValueError("gemm must be 'f32', 'bf16x6' or 'f32_w16'")
        if gemm == "f32_w16":
            L.require_experiments("policy_gemm='f32_w16' (the 16-wave policy kernel)")
        if gemm == "bf16x6" and (self.obs_dim, self.act_dim) not in ((20, 4), (12, 1)):
            raise ValueError("the bf16 split kernels are built for the hover task's widths (20 -> 4) and the yaw task's (12 -> 1)")
        L.check(self.lib.fpv_policy_set_gemm(self._h, modes[gemm]), "fpv_policy_set_gemm")
        self.gemm = gemm

    def _params(self):
        m = self.module
        ps = [m.pi[0].weight, m.pi[0].bias, m.pi[2].weight, m.pi[2].bias, m.action_net.weight, m.action_net.bias,
              m.log_std, m.vf[0].weight, m.vf[0].bias, m.vf[2].weight, m.vf[2].bias, m.value_net.weight,
              m.value_net.bias]
        for t in ps:
            if t.dtype != torch.float32 or not t.is_contiguous() or t.device != self.device:
                raise ValueError("policy parameters must be contiguous fp32 tensors on the policy's device")
        return ps

    def load(self) -> None:
        """Repack the module's current parameters into the kernel's MFMA-ordered weight image."""
        ps = self._params()
        arr = (C.c_void_p * 13)(*[t.data_ptr() for t in ps])
        L.check(self.lib.fpv_policy_load(self._h, arr, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)),
                "fpv_policy_load")

    def forward(self, obs, *, rms=None, eps_in=None, tick=None, obs_n=None, action=None, clipped=None, value=None,
                logp=None, eps_out=None, clip_obs: float = 10.0, eps_norm: float = 1e-8, deterministic: bool = False,
                nets: str = "both", narrow: bool = False, merge: dict | None = None, n_dev=None, scatter: dict | None = None):
        """obs [n,20] fp32 raw observations; rms = fp64[41] VecNormalize statistics or None.  Output tensors
        are written in place when given; value is always produced.  Returns dict of the outputs.
        merge = {partials [rows,44] f64, count, rms_out, ret_rms, ret_rms_out, snapshot}: fold the previous
        fpv_step_fused's tile sums into the statistics in this launch (rms / ret_rms are read, *_out written).
        n_dev: int32 device scalar; only the first min(n, n_dev) rows are evaluated.
        scatter = {t_idx int32[n], d_idx int32[n], out fp32 [T][stride], scale}: out[t_idx[k], d_idx[k]] += scale * value[k] for the
        evaluated rows, in this launch (ops.ledger_scatter without its own)."""
        n = obs.shape[0]
        if narrow:
            L.require_experiments("narrow=True (256-thread policy launches)")
        if obs.dtype != torch.float32 or not obs.is_contiguous() or obs.shape[1] != self.obs_dim:
            raise ValueError(f"obs must be a contiguous [n,{self.obs_dim}] fp32 device tensor")
        mask = {"both": 3, "pi": 1, "v": 2}[nets]
        if value is None and mask & 2:
            value = torch.empty((n,), dtype=torch.float32, device=obs.device)
        io = L.FpvPolicyIo(n, _ptr(obs), _ptr(rms), _ptr(eps_in), _ptr(tick), _ptr(obs_n), _ptr(action), _ptr(clipped),
                           _ptr(value), _ptr(logp), _ptr(eps_out), self.seed, self.id_offset, float(clip_obs),
                           float(eps_norm), 1 if deterministic else 0, mask, 1 if narrow else 0, 0)
        io.n_dev = _ptr(n_dev)
        if scatter is not None:
            out = scatter["out"]
            if out.dtype != torch.float32 or out.dim() != 2 or not out.is_contiguous() or not mask & 2:
                raise ValueError("scatter: out is a contiguous fp32 [T][n] table and the V net must be evaluated")
            if scatter["t_idx"].dtype != torch.int32 or scatter["d_idx"].dtype != torch.int32 or min(scatter["t_idx"].numel(), scatter["d_idx"].numel()) < n:
                raise ValueError("scatter: t_idx / d_idx are int32 with one entry per row")
            io.scatter_t, io.scatter_d, io.scatter_out = _ptr(scatter["t_idx"]), _ptr(scatter["d_idx"]), _ptr(out)
            io.scatter_stride, io.scatter_scale = out.shape[1], float(scatter["scale"])
        if merge is not None:
            io.merge_partials, io.merge_rows = _ptr(merge["partials"]), merge["partials"].shape[0]
            io.merge_count = float(merge["count"])
            io.rms_out, io.ret_rms = _ptr(merge.get("rms_out")), _ptr(merge.get("ret_rms"))
            io.ret_rms_out, io.merge_snapshot = _ptr(merge.get("ret_rms_out")), _ptr(merge.get("snapshot"))
        L.check(self.lib.fpv_policy_forward(self._h, C.byref(io),
                                            C.c_void_p(torch.cuda.current_stream(obs.device).cuda_stream)),
                "fpv_policy_forward")
        return {"obs_n": obs_n, "action": action, "clipped": clipped, "value": value, "logp": logp, "eps": eps_out}

    def rollout_step(self, engine, *, rms=None, obs_n=None, action=None, clipped=None, value=None, logp=None,
                     clip_obs: float = 10.0, eps_norm: float = 1e-8, merge: dict | None = None, done=None, ret=None,
                     partials=None, gamma: float = 0.99, reward=None, nets: str = "both"):
        """fpv_rollout_step: forward() on the engine's raw observations and the env step of the same drones in ONE launch
        (fpv_rollstep.hip); engine.obs then holds the next observations.  Arguments as for forward() and
        FpvEngine.step_fused(); `partials` receives num_envs / 256 rows and must be a different table from
        merge["partials"].  Bit-identical to forward() followed by engine.step_fused(clipped, ...).
        nets="pi" (gemm "bf16x6"): the critic deferred — no value is written; forward(obs_n rows, nets="v") gives the same bits later."""
        n = engine.num_envs
        if nets not in ("both", "pi"):
            raise ValueError("fpv_rollout_step serves nets 'both' or 'pi'")
        if value is None and nets == "both":
            value = torch.empty((n,), dtype=torch.float32, device=engine.device)
        reward = engine.reward if reward is None else reward
        if partials is not None and (partials.dtype != torch.float64 or not partials.is_contiguous()
                                     or partials.shape[0] < n // 256 or partials.shape[1] != 2 * self.obs_dim + 4):
            raise ValueError(f"partials must be a contiguous float64 [>= num_envs / 256, {2 * self.obs_dim + 4}] device tensor")
        io = L.FpvPolicyIo(n, _ptr(engine.obs), _ptr(rms), None, None, _ptr(obs_n), _ptr(action), _ptr(clipped),
                           _ptr(value if nets == "both" else None), _ptr(logp), None, self.seed, self.id_offset, float(clip_obs),
                           float(eps_norm), 0, 3 if nets == "both" else 1, 0, 0)
        if merge is not None:
            io.merge_partials, io.merge_rows = _ptr(merge["partials"]), merge["partials"].shape[0]
            io.merge_count = float(merge["count"])
            io.rms_out, io.ret_rms = _ptr(merge.get("rms_out")), _ptr(merge.get("ret_rms"))
            io.ret_rms_out, io.merge_snapshot = _ptr(merge.get("ret_rms_out")), _ptr(merge.get("snapshot"))
        ex = L.FpvStepExtras(_ptr(done), _ptr(ret), _ptr(partials), float(gamma), 0, None, None, None)
        L.check(self.lib.fpv_rollout_step(engine._h, self._h, C.byref(io), _ptr(reward), _ptr(engine.terminated),
                                          _ptr(engine.truncated), C.byref(ex),
                                          C.c_void_p(torch.cuda.current_stream(engine.device).cuda_stream)),
                "fpv_rollout_step")
        return {"obs_n": obs_n, "action": action, "clipped": clipped, "value": value, "logp": logp, "reward": reward}

    def rollout_steps(self, engine, T: int, *, obs_n=None, action=None, value, logp=None, reward, done=None, ret=None,
                      gamma: float = 0.99, clip_obs: float = 10.0, eps_norm: float = 1e-8, rms=None, ret_rms=None,
                      rms_out=(None, None), ret_rms_out=(None, None), snapshots=None, partials, status):
        """fpv_rollout_steps: T rollout steps in ONE persistent launch (fpv_rollstep.hip).  Time-major [T, n, ...] buffers;
        partials = float64 [2, n / 256, 44]; status = uint32 device word (nonzero afterwards: a grid wait timed out)."""
        n = engine.num_envs
        for name, t, shape in (("value", value, (T, n)), ("reward", reward, (T, n)), ("obs_n", obs_n, (T, n, 20)),
                               ("action", action, (T, n, 4)), ("logp", logp, (T, n)), ("done", done, (T, n))):
            if t is not None and (not t.is_contiguous() or tuple(t.shape[:len(shape)]) != shape):
                raise ValueError(f"{name} must be a contiguous tensor of shape {shape}")
        if partials.dtype != torch.float64 or not partials.is_contiguous() or tuple(partials.shape) != (2, n // 256, 44):
            raise ValueError("partials must be a contiguous float64 [2, num_envs / 256, 44] device tensor")
        io = L.FpvRolloutIo()
        io.T = int(T)
        io.obs, io.obs_n, io.action, io.value, io.logp = _ptr(engine.obs), _ptr(obs_n), _ptr(action), _ptr(value), _ptr(logp)
        io.reward, io.done, io.terminated, io.truncated = _ptr(reward), _ptr(done), _ptr(engine.terminated), _ptr(engine.truncated)
        io.ret, io.gamma, io.clip_obs, io.eps_norm = _ptr(ret), float(gamma), float(clip_obs), float(eps_norm)
        io.seed, io.id_offset = self.seed, self.id_offset
        io.rms, io.ret_rms = _ptr(rms), _ptr(ret_rms)
        for k in range(2):
            io.rms_out[k], io.ret_rms_out[k] = _ptr(rms_out[k]), _ptr(ret_rms_out[k])
            io.partials[k] = partials[k].data_ptr()
        io.snapshots = _ptr(snapshots)
        io.snapshot_stride = snapshots.stride(0) if snapshots is not None else 0
        io.status = _ptr(status)
        L.check(self.lib.fpv_rollout_steps(engine._h, self._h, C.byref(io),
                                           C.c_void_p(torch.cuda.current_stream(engine.device).cuda_stream)),
                "fpv_rollout_steps")

    def _grads(self):
        gs = []
        for t in self._params():
            if t.grad is None:
                t.grad = torch.zeros_like(t)
            if not t.grad.is_contiguous() or t.grad.dtype != torch.float32:
                raise ValueError("parameter gradients must be contiguous fp32 tensors")
            gs.append(t.grad)
        return gs

    def ppo_workspace(self, rows: int) -> torch.Tensor:
        """Scratch tensor for ppo_grad on `rows` transitions (allocate once, outside any graph capture)."""
        b = C.c_int64()
        L.check(self.lib.fpv_ppo_workspace_bytes(self._h, int(rows), C.byref(b)), "fpv_ppo_workspace_bytes")
        return torch.empty((b.value + 3) // 4, dtype=torch.float32, device=self.device)

    def ppo_grad(self, obs, actions, old_logp, advantages, returns, *, index=None, rows=None, clip_range: float = 0.2,
                 vf_coef: float = 0.5, ent_coef: float = 0.0, stats=None, workspace=None, storage: str = "f32",
                 path: str | None = None):
        """Gradient of SB3's PPO minibatch loss w.r.t. every parameter of the module, written into the parameters'
        .grad tensors by fpv_ppo_grad (forward, loss derivatives and backward as hand-written MFMA kernels; no torch
        autograd involved).  index: optional int64 row indices into the five arrays.  storage: what the two-kernel paths write between
        their kernels — "f32" words or "bf16" (half the HBM traffic, gradients within ~1e-3 of their max-norm; two-kernel path
        only); a storage format, not an arithmetic.  path: "one_kernel" (FPV_PPO_ONE_KERNEL: forward, backward and weight gradients
        of a row in one kernel, nothing stored, the three 128 x 128 products per row as 6-term bf16 splits with fp32 accumulation —
        fp32-level, golden error 4.5e-7, not exact fp32-input MFMA; the default with fp32 storage), "two_kernels" (the activations / deltas pass through memory;
        the default with bf16 storage) or "two_kernels_bf16x6" (as before, the weight-gradient products on the bf16 MFMA with every
        fp32 operand cut into three bf16 pieces, six piece products per product: fp32-level error; "two_kernels_bf16x3": three piece
        products, weight gradients to ~5e-6 of their max-norm — on record, not for use).
        Returns stats[4] = (policy-gradient loss, value loss, approx KL, clip fraction)."""
        if index is not None:
            if index.dtype != torch.int64 or not index.is_contiguous():
                raise ValueError("index must be a contiguous int64 device tensor")
            m = index.numel()
        else:
            m = obs.shape[0] if rows is None else int(rows)
        for name, t, w in (("obs", obs, self.obs_dim), ("actions", actions, self.act_dim)):
            if t.dtype != torch.float32 or not t.is_contiguous() or t.dim() != 2 or t.shape[1] != w:
                raise ValueError(f"{name} must be a contiguous [*,{w}] fp32 device tensor")
        for name, t in (("old_logp", old_logp), ("advantages", advantages), ("returns", returns)):
            if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != obs.shape[0]:
                raise ValueError(f"{name} must be a contiguous fp32 vector with one entry per obs row")
        if stats is None:
            stats = torch.empty(4, dtype=torch.float32, device=self.device)
        if workspace is None:
            workspace = self.ppo_workspace(m)
        io = L.FpvPpoIo()
        io.rows, io.index = m, _ptr(index)
        io.obs, io.actions, io.old_logp = _ptr(obs), _ptr(actions), _ptr(old_logp)
        io.advantages, io.returns = _ptr(advantages), _ptr(returns)
        io.clip_range, io.vf_coef, io.ent_coef = float(clip_range), float(vf_coef), float(ent_coef)
        for k, (w, g) in enumerate(zip(self._params(), self._grads())):
            io.weights[k], io.grads[k] = w.data_ptr(), g.data_ptr()
        io.stats, io.workspace, io.workspace_bytes = _ptr(stats), _ptr(workspace), workspace.numel() * 4
        io.storage = {"f32": 0, "bf16": 1}[storage]
        if path is None:
            path = "one_kernel" if storage == "f32" else "two_kernels"
        io.path = {"one_kernel": 0, "two_kernels": 1, "two_kernels_bf16x6": 2, "two_kernels_bf16x3": 3}[path]
        L.check(self.lib.fpv_ppo_grad(self._h, C.byref(io), C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)),
                "fpv_ppo_grad")
        return stats

    def close(self):
        if getattr(self, "_h", None):
            self.lib.fpv_policy_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
