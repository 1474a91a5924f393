"""On-device PPO rollout collector: observations, actions, rewards and values never leave HBM.

Semantics follow SB3's OnPolicyAlgorithm.collect_rollouts over VecNormalize(VecEnv) — the
stack simulation/train_hover.py:42-60 builds — as restated in SURVEY.md §3.5:

    obs_t (normalised) -> actor-critic -> a_t ~ N(mu, sigma), v_t, log-prob
    env.step(clip(a_t, -1, 1)) -> r_t, done_t            (fpv_step, auto-reset, done ledger)
    VecNormalize: obs running moments + clip, reward / sqrt(var(discounted return)) + clip
    truncated & not terminated: r_t += gamma * V(normalised terminal observation)
    GAE backward scan per env (fpv_gae) -> advantages, returns, {count, sum, sum^2}
    ONE all-gather of those statistics per update -> global advantage mean / std

Every per-step operation is asynchronous device work on one stream; the T-step loop is
captured once into a HIP graph and replayed, so the host issues one launch per rollout.

Two step implementations:
  * fused, persistent: ONE launch per ROLLOUT — fpv_rollout_steps keeps weights in LDS and drone state in registers across
    the n_steps, with a grid-wide barrier + statistics merge between steps (fpv_rollstep.hip).  Same bits.
  * fused, one_launch (default where served): ONE kernel per step — fpv_rollout_step = the two kernels below in one
    launch, same bits (fpv_rollstep.hip).
  * fused: TWO kernels per step — fpv_policy_forward (RunningMeanStd.update of both statistics
    with the previous step's moment sums in its prologue -> normalise -> pi, V on fp32 MFMA -> sample ->
    log-prob -> clip, writing straight into the rollout buffer) and fpv_step_fused (env step + done mask +
    discounted return + moment sums).  Reward scaling is applied once per rollout (fpv_reward_finalize)
    with each step's own statistics snapshot, which is arithmetically what VecNormalize does step by step.
  * unfused (`fused=False`): the same semantics through torch ops + the stand-alone normaliser
    kernels; kept as the readable reference of the fused path (tests compare both to the oracle).
  * an fp64 engine (`FpvEngine(precision="f64")`): the configuration in the reference's own arithmetic type — its spaces are
    float64 (simulation/hover.py:61, :70) and numpy / pybullet compute in double, while SB3's torch policy is float32.  The
    policy kernel stays fp32; the env step is fpv_step_f64; observation, action and reward are converted at the seam; the
    normaliser runs as its stand-alone kernels and the timeout bootstrap through the torch statement (host synchronisations:
    this shape exists to be right, not to be fast).
"""
from __future__ import annotations

import os
import struct
import sys

import torch

from . import dist as D
from . import ops
from .engine import FpvEngine
from .ops_policy import FusedPolicy
from .policy import ActorCritic


def _hip_runtime():
    """The HIP runtime this process already runs on (torch ships its own copy: a second dlopen by name could load another one)."""
    import ctypes
    path = None
    with open("/proc/self/maps") as maps:
        for line in maps:
            if "libamdhip64" in line:
                path = line.split()[-1]
                break
    return ctypes.CDLL(path or "libamdhip64.so")


def _new_capture_stream(device):
    """A stream of the collector's own to capture on.  torch.cuda.Stream() hands out one of 32 pooled streams per device in turn,
    and torch.cuda.graph's default capture stream is one of them: a capture that fails leaves its stream "invalidated" for good
    (hipStreamEndCapture does not bring it back on this stack), and sooner or later somebody is handed that stream again.  A stream
    created here is nobody else's; after a failed capture it is simply abandoned."""
    import ctypes
    try:
        handle = ctypes.c_void_p()
        with torch.cuda.device(device):
            rc = _hip_runtime().hipStreamCreateWithFlags(ctypes.byref(handle), ctypes.c_uint(1))       # hipStreamNonBlocking
        if rc == 0 and handle.value:
            return torch.cuda.ExternalStream(handle.value, device=device)
    except (OSError, AttributeError):
        pass
    return torch.cuda.Stream(device)


def _end_broken_capture(stream) -> None:
    """A failed torch.cuda.graph capture leaves its stream capturing: end it (the call returns the error once more)."""
    import ctypes
    try:
        hip = _hip_runtime()
    except OSError:
        return
    graph = ctypes.c_void_p()
    hip.hipStreamEndCapture(ctypes.c_void_p(stream.cuda_stream), ctypes.byref(graph))
    if graph.value:
        hip.hipGraphDestroy(graph)
    hip.hipGetLastError()


class _Batch(dict):
    """collect()'s result.  "adv_mean" / "adv_std" come out behind the collective: reading them orders the current stream after
    it when it ran on another stream (FPV_POST_STREAM=side); everything else is ready in main-stream order."""

    def __init__(self, col, items):
        super().__init__(items)
        self._col = col

    def __getitem__(self, key):
        if key in ("adv_mean", "adv_std"):
            self._col.wait_stats()
        return super().__getitem__(key)


class RolloutCollector:
    def __init__(self, engine: FpvEngine, policy: ActorCritic, n_steps: int = 32, gamma: float = 0.99,
                 lam: float = 0.95, norm_obs: bool = True, norm_reward: bool = True, use_graph: bool = True,
                 clip_obs: float = 10.0, clip_reward: float = 10.0, fused: bool = True, policy_seed: int = 0,
                 policy_gemm: str | None = None, overlap_critic: bool = False, merge_in_policy: bool = True,
                 fused_step: bool = True, merge_in_step: bool = False, one_launch: bool | None = None,
                 persistent: bool | None = None, deferred_critic: bool | None = None):
        self._f64_env = engine.dtype == torch.float64
        if self._f64_env:                                  # fp32 policy kernel + fpv_step_f64 + stand-alone normaliser kernels
            if not fused:
                raise ValueError("an fp64 engine is collected through the fused policy kernel (fused=True)")
            fused_step, use_graph, one_launch, persistent, overlap_critic, merge_in_step = False, False, False, None, False, False
        from . import _lib as _L
        for flag, what in ((persistent, "persistent=True (the whole rollout as one launch)"),
                           (overlap_critic, "overlap_critic=True (the critic beside the env step)")):
            if flag:
                _L.require_experiments(what)       # (merge_in_step works in either library: in-launch there, a merge kernel here)
        self.engine, self.policy = engine, policy
        self.T, self.N = int(n_steps), engine.num_envs
        self.gamma, self.lam = float(gamma), float(lam)
        self.norm_obs, self.norm_reward = bool(norm_obs), bool(norm_reward)
        self.use_graph = bool(use_graph)
        dev = self.device = engine.device
        T, N = self.T, self.N
        f32, u8 = torch.float32, torch.uint8
        O, A = engine.obs_words, engine.act_words       # 20 / 4 hover (hover.py:59-70), 12 / 1 yaw (yaw.py:37-45)
        self.O, self.A = O, A
        if (policy.obs_dim, policy.act_dim) != (O, A):
            raise ValueError(f"policy is {policy.obs_dim}->{policy.act_dim} but the task needs {O}->{A}")
        fused_policy = bool(fused)                       # fpv_policy_forward and fpv_step_fused take both tasks' widths
        fused = bool(fused) and bool(fused_step)         # fused_step=False: stand-alone env-step + normaliser kernels
        self.obs_norm = ops.RunningNorm(O, dev, clip=clip_obs)
        self.rew_norm = ops.RewardNorm(N, dev, gamma=gamma, clip=clip_reward)
        self.buf_obs = torch.zeros((T, N, O), dtype=f32, device=dev)
        self.buf_act = torch.zeros((T, N, A), dtype=f32, device=dev)
        self.buf_rew = torch.zeros((T, N), dtype=f32, device=dev)
        self._starts = torch.zeros((T + 1, N), dtype=u8, device=dev)   # row t: episode_start before step t
        self._starts[T].fill_(1)
        self.buf_start = self._starts[:T]
        self.buf_val = torch.zeros((T, N), dtype=f32, device=dev)
        self.buf_logp = torch.zeros((T, N), dtype=f32, device=dev)
        self.adv = torch.zeros((T, N), dtype=f32, device=dev)
        self.ret = torch.zeros((T, N), dtype=f32, device=dev)
        self.cur_obs = torch.zeros((N, O), dtype=f32, device=dev)        # normalised s_t
        self.episode_start = self._starts[T]                                # carried into the next rollout
        self.done = torch.zeros((N,), dtype=u8, device=dev)
        self.clipped = torch.zeros((N, A), dtype=f32, device=dev)
        self.last_val = torch.zeros((N,), dtype=f32, device=dev)
        self.snap = torch.zeros((T, max(44, 2 * O + 4)), dtype=torch.float64, device=dev)  # {obs_rms, ret_rms} after step t
        if self._f64_env:
            self._obs32 = torch.zeros((N, O), dtype=f32, device=dev)       # the env's fp64 outputs as the fp32 policy side sees them
            self._rew32 = torch.zeros((N,), dtype=f32, device=dev)
            self._clip64 = torch.zeros((N, A), dtype=torch.float64, device=dev)
            self.env_obs = torch.zeros((T, N, O), dtype=torch.float64, device=dev)   # raw fp64 env outputs of the rollout, step by step
            self.env_rew = torch.zeros((T, N), dtype=torch.float64, device=dev)
        # policy_gemm=None, the default (round 5): the hidden 128 x 128 products of pi and V as 6-term bf16 splits with
        # fp32 accumulation ("bf16x6": every row at fp32 rounding level against an fp64 evaluation, tests/test_policy_gpu.py) wherever
        # that kernel exists — both tasks' widths on the fused path; "f32" (exact fp32-input MFMA) stays selectable and is what the
        # experiment shapes (persistent, overlap_critic) and any other widths run on
        if policy_gemm is None:
            policy_gemm = "bf16x6" if (fused and fused_policy and (O, A) in ((20, 4), (12, 1)) and not persistent and not overlap_critic) else "f32"
        self.policy_gemm = policy_gemm
        self.fused = bool(fused)
        self._one_launch = self._one_launch_ok = False
        self._persistent = self._persistent_ok = False
        self.fused_policy = fused_policy
        if self.fused_policy:
            self.fp = FusedPolicy(policy, seed=policy_seed, id_offset=engine.id_offset,
                                  gemm=policy_gemm if self.fused else "f32")
        if self.fused:
            self._partials_full = torch.zeros((engine.tiles, 2 * O + 4), dtype=torch.float64, device=dev)
            self.partials = self._partials_full[:engine.partial_rows]       # the rows fpv_step_fused writes
            self.raw_rew = torch.zeros((T, N), dtype=f32, device=dev)
            # overlap_critic: the env step needs pi only, so the critic pass (MFMA-bound) can run on a second stream
            # beside the env-step kernel (VALU-bound).  Measured on MI355X (profiles/r01_overlap_experiment.txt): alone
            # V takes 32 us and the env step 17 us, co-resident 59 us and 41-52 us — the two kernels fight over VALU
            # issue slots and the instruction cache — so the serial schedule is faster and this stays off by default.
            self.overlap_critic = bool(overlap_critic) and policy_gemm == "f32"
            self._vstream = torch.cuda.Stream(dev) if self.overlap_critic else None
            # VecNormalize's statistics update inside the next policy launch (fpv_policy_io.merge_*): up to 1 024 tiles
            # Where VecNormalize's statistics update runs.  Default: in the next policy launch's prologue (every policy
            # workgroup folds the step's <= 1 024 rows in the same order).  merge_in_step=True: inside fpv_step_fused (its
            # last workgroup folds the rows — write-through rows + ticket — or the merge kernel behind it for launch
            # shapes without that reducer); measured at 65 536 drones the policy launch then drops 58.0 -> 55.9 us but the
            # env step grows 16.1 -> 23.0 us (the reducer's ticket + eight dependent sc1 round trips sit on the step's
            # tail), 72.5 -> 77.1 us per rollout step, so it is not the default.  Both off: the stand-alone merge kernel.
            self._merge_in_step = bool(merge_in_step) and (self.norm_obs or self.norm_reward)
            self._merge_in_policy = (not self._merge_in_step and bool(merge_in_policy) and engine.partial_rows <= 1024
                                     and (self.norm_obs or self.norm_reward)
                                     and not self.overlap_critic)   # (the table is <= 1 024 rows by construction)
            self._obs_rms_alt = torch.zeros_like(self.obs_norm.rms)
            self._ret_rms_alt = torch.zeros_like(self.rew_norm.rms)
            # one_launch: policy forward and env step of a rollout step as ONE kernel (fpv_rollout_step, fpv_rollstep.hip) —
            # bit-identical results.  Served for both tasks with motor noise, drone counts that are multiples of 256 up
            # to 256 x 1 024, the exact-fp32 policy, and the statistics merge in the launch prologue (or no normalisation).
            norm = self.norm_obs or self.norm_reward
            # (either arithmetic: rollout_step_kernel<TASK> / rollout_step_split_kernel<TASK>)
            self._one_launch_gemms = ("f32", "bf16x6")
            can = (engine.noise_ratio > 0.0 and N % 256 == 0 and N // 256 <= 1024
                   and policy_gemm in self._one_launch_gemms and not self.overlap_critic and not self._merge_in_step
                   and (self._merge_in_policy or not norm))
            if one_launch and not can:
                raise ValueError("one_launch needs motor noise, num_envs a multiple of 256 (<= 262 144), policy_gemm 'f32' or 'bf16x6' "
                                 "and the default statistics-merge placement")
            # Default: where every workgroup finds a CU of its own (num_envs <= 256 x CUs).  Beyond that the second round of
            # workgroups stages the weight image again, while the two-kernel shape's policy kernel keeps it in LDS across its
            # chunks: measured 128.2 vs 126.3 us per step at 131 072 drones, 257.2 vs 234.9 at 262 144.
            cus = torch.cuda.get_device_properties(dev).multi_processor_count
            self._one_launch = self._one_launch_ok = (can and N // 256 <= cus) if one_launch is None else bool(one_launch)
            # persistent: the WHOLE rollout of n_steps as one launch (fpv_rollout_steps): weights stay in LDS, drone state in
            # registers, a grid-wide barrier + statistics merge between steps.  Same bits again.  Needs every workgroup resident:
            # at most one per CU, i.e. num_envs <= 256 x CUs.
            can_p = can and N // 256 <= cus and engine.task == "hover" and engine.detector == "analytic" and policy_gemm == "f32"
            if persistent and not can_p:
                raise ValueError("persistent needs what one_launch needs, num_envs <= 256 x the device's CU count and "
                                 "detector='analytic'")
            self._persistent = self._persistent_ok = False if persistent is None else bool(persistent)
            # deferred_critic (round 5; measured, NOT the default — DESIGN §5b): V(s_t) is an input of GAE only, so the one-launch step
            # can evaluate pi alone (fpv_policy_io.nets = 1) and the tail run V once over all T x N stored, normalised rows — the same
            # kernel arithmetic on the same inputs, the same bits (tests/test_rollstep_gpu.py) — with V's 121 KB image staged once per
            # workgroup and rollout instead of once per workgroup and step.  The step kernel drops 57 -> 43 us, but the batched pass
            # costs 12.3-12.7 us per 65 536 rows: 0.7-1.4 us per step slower overall at 65 536 drones.  FPV_DEFER_CRITIC=1 selects it.
            env_defer = os.environ.get("FPV_DEFER_CRITIC")
            if deferred_critic is None and env_defer is not None:
                deferred_critic = env_defer not in ("0", "")
            self._defer_opt = bool(deferred_critic)
            if deferred_critic and not (self._one_launch and policy_gemm == "bf16x6"):
                raise ValueError("deferred_critic needs the one-launch step on policy_gemm 'bf16x6'")
            if self._one_launch or self._persistent:    # step t's rows are read while step t + 1's are written: two tables
                self._rs_partials = torch.zeros((2, N // 256, 2 * O + 4), dtype=torch.float64, device=dev)
        # What the host / the collective read after a rollout — [0:8] ledger status words (fused policy), [8:11] {count, sum A,
        # sum A^2} of the advantages — exists twice, and so does the captured graph: rollouts alternate between the two, so the
        # host (and a side stream, if asked for) reads rollout r's words while rollout r + 1 already runs, with no copy and no wait between them
        # (a slot is written again two rollouts later).  self._slot selects the pair in use (see the properties below).
        self._post = [torch.zeros((16,), dtype=torch.float64, device=dev) for _ in range(2)]
        self._slot = 0
        if self.fused_policy:
            # sync-free timeout bootstrap (fpv_ledger_scan / fpv_ledger_scatter): a drone is truncated at most once per
            # 401 steps, so the dense list never needs more than N * ceil(T / 401) rows
            cap = N * ((T + 400) // 401)
            self._tobs = torch.zeros((cap, O), dtype=f32, device=dev)
            self._t_idx = torch.zeros((cap,), dtype=torch.int32, device=dev)
            self._d_idx = torch.zeros((cap,), dtype=torch.int32, device=dev)
            self._tv = torch.zeros((cap,), dtype=f32, device=dev)
            # per slot: counters[4] as int32 in words [0:2], sums in [2:4], fpv_rollout_steps' timeout word in [4]
            self._led_counters_k = [t[0:2].view(torch.int32) for t in self._post]
            self._rs_status_k = [t[4:5].view(torch.int32)[0:1] for t in self._post]
            self._led_sums_k = [t[2:4] for t in self._post]
            self._status_host = [torch.zeros((8,), dtype=torch.float64).pin_memory() for _ in range(2)]
            self._status_np = [t.numpy() for t in self._status_host]      # the same pinned words
            self._status_event = [torch.cuda.Event() for _ in range(2)]
            self._status_pending = [False, False]
        # FPV_POST_STREAM=side: the collective and the advantage statistics on a side stream (the arrangement of rounds 1-2;
        # measured slower than following the rollout on its own stream: see collect())
        self._side = torch.cuda.Stream(dev)
        self._stats_k = [t[8:11] for t in self._post]
        self._adv_ms = [torch.zeros((2,), dtype=torch.float64, device=dev) for _ in range(2)]
        self._stats_ready = [torch.cuda.Event() for _ in range(2)]
        self._stats_done = [torch.cuda.Event() for _ in range(2)]
        self._done_stream = [None, None]                 # the stream each _stats_done was last recorded on
        self._capture_mode = None                        # None: "global" with one rank, "thread_local" with several (_capture)
        self.capture_mode_used = None                    # the mode the captured graphs were taken in (None: eager launches so far)
        self._stats_last = 0
        # several ranks: VecNormalize's statistics ride in the same payload ({adv stats[3], obs_rms[2 O + 1], ret_rms[3]}); every
        # rank starts a rollout from the statistics of the union of all ranks' samples so far (fpv_rank_merge) — kept here as the base
        self._base_obs_rms = torch.zeros_like(self.obs_norm.rms)
        self._base_ret_rms = torch.zeros_like(self.rew_norm.rms)
        self._rollouts = 0
        self._episodes_host = {"count": 0, "ep_len_mean": float("nan"), "ep_rew_mean": float("nan"), "truncated": 0}
        self._graphs = [None, None]
        self._scan_prezeroed = False
        self._tick0 = 0
        self._started = False

    # ------------------------------------------------------------------ the slot in use (alternates per rollout)
    @property
    def stats(self):
        return self._stats_k[self._slot]

    @property
    def _led_counters(self):
        return self._led_counters_k[self._slot]

    @property
    def _led_sums(self):
        return self._led_sums_k[self._slot]

    @property
    def _rs_status(self):
        return self._rs_status_k[self._slot]

    @property
    def _graph(self):
        return self._graphs[self._slot]

    @_graph.setter
    def _graph(self, g):
        if g is None:
            self._graphs = [None, None]      # dropping the capture drops both slots' graphs
        else:
            self._graphs[self._slot] = g

    @property
    def deferred_critic(self) -> bool:
        """True when the rollout's steps evaluate pi only and V runs once over the stored rows in the tail."""
        return bool(self.fused and self._one_launch and self.policy_gemm == "bf16x6" and self._defer_opt)

    # ------------------------------------------------------------------ one vectorised step
    def _normalise_obs(self, raw, out, update=True):
        if self.norm_obs:
            self.obs_norm(raw, out=out, update=update)
        else:
            out.copy_(raw)

    def _env_obs(self):
        """The env's current observation as the fp32 policy side reads it."""
        return self._obs32 if self._f64_env else self.engine.obs

    def _step(self, t: int) -> None:
        if self.fused:
            self._step_fused(t)
        elif self.fused_policy:
            self._step_semifused(t)
        else:
            self._step_torch(t)

    def _step_semifused(self, t: int) -> None:
        """Fused policy kernel + the stand-alone env-step / normaliser kernels (`fused_step=False`: the reference of the
        in-kernel VecNormalize bookkeeping)."""
        eng = self.engine
        obs, reward = self._env_obs(), (self._rew32 if self._f64_env else eng.reward)
        self.fp.forward(obs, rms=self.obs_norm.rms if self.norm_obs else None, tick=eng.tick_ptr,
                        obs_n=self.buf_obs[t], action=self.buf_act[t], clipped=self.clipped, value=self.buf_val[t],
                        logp=self.buf_logp[t], clip_obs=self.obs_norm.clip, eps_norm=self.obs_norm.eps)
        if self._f64_env:                                  # the seam: fp32 action in, fp64 observation / reward out
            self._clip64.copy_(self.clipped)
            eng.step(self._clip64)
            self.env_obs[t].copy_(eng.obs)
            self.env_rew[t].copy_(eng.reward)
            self._obs32.copy_(eng.obs)
            self._rew32.copy_(eng.reward)
        else:
            eng.step(self.clipped)
        torch.bitwise_or(eng.terminated, eng.truncated, out=self._starts[t + 1])
        if self.norm_obs:
            self.obs_norm.update_only(obs)
        self.snap[t, :2 * self.O + 1].copy_(self.obs_norm.rms)
        if self.norm_reward:
            self.rew_norm(reward, self._starts[t + 1], out=self.buf_rew[t])
        else:
            self.buf_rew[t].copy_(reward)

    def _step_fused(self, t: int) -> None:
        eng = self.engine
        rms = self.obs_norm.rms if self.norm_obs else None
        if self._one_launch:
            (o_in, r_in), (o_out, r_out) = self._stat_bufs(t - 1), self._stat_bufs(t)
            merge = None
            if self._merge_in_policy and t > 0:
                merge = {"partials": self._rs_partials[(t - 1) % 2], "count": self.N, "rms_out": o_out, "ret_rms": r_in,
                         "ret_rms_out": r_out, "snapshot": self.snap[t - 1]}
            self.fp.rollout_step(eng, rms=self._stat_bufs(0)[0] if t == 0 else o_in, obs_n=self.buf_obs[t], action=self.buf_act[t],
                                 value=self.buf_val[t], logp=self.buf_logp[t], clip_obs=self.obs_norm.clip,
                                 eps_norm=self.obs_norm.eps, merge=merge, done=self._starts[t + 1], ret=self.rew_norm.ret,
                                 partials=self._rs_partials[t % 2], gamma=self.gamma, reward=self.raw_rew[t],
                                 nets="pi" if self.deferred_critic else "both")
            return
        if self.overlap_critic:
            self.fp.forward(eng.obs, rms=rms, tick=eng.tick_ptr, obs_n=self.buf_obs[t], action=self.buf_act[t],
                            clipped=self.clipped, logp=self.buf_logp[t], clip_obs=self.obs_norm.clip,
                            eps_norm=self.obs_norm.eps, nets="pi")
            main = torch.cuda.current_stream(self.device)
            self._vstream.wait_stream(main)
            with torch.cuda.stream(self._vstream):     # V(s_t) from the normalised row the pi pass just wrote
                self.fp.forward(self.buf_obs[t], value=self.buf_val[t], nets="v", narrow=True, deterministic=True)
        elif self._merge_in_policy:
            # the statistics update of step t - 1 rides in this policy launch's prologue (no merge launch on the
            # critical path); the statistics ping-pong between two buffers because workgroup 0 publishes the merged
            # values while the other workgroups still read the previous ones
            (o_in, r_in), (o_out, r_out) = self._stat_bufs(t - 1), self._stat_bufs(t)
            merge = None if t == 0 else {"partials": self.partials, "count": self.N, "rms_out": o_out, "ret_rms": r_in,
                                         "ret_rms_out": r_out, "snapshot": self.snap[t - 1]}
            self.fp.forward(eng.obs, rms=self._stat_bufs(0)[0] if t == 0 else o_in, tick=eng.tick_ptr, obs_n=self.buf_obs[t],
                            action=self.buf_act[t], clipped=self.clipped, value=self.buf_val[t], logp=self.buf_logp[t],
                            clip_obs=self.obs_norm.clip, eps_norm=self.obs_norm.eps, merge=merge)
        else:
            self.fp.forward(eng.obs, rms=rms, tick=eng.tick_ptr, obs_n=self.buf_obs[t], action=self.buf_act[t],
                            clipped=self.clipped, value=self.buf_val[t], logp=self.buf_logp[t],
                            clip_obs=self.obs_norm.clip, eps_norm=self.obs_norm.eps)
        if self._merge_in_step:
            eng.step_fused(self.clipped, done=self._starts[t + 1], ret=self.rew_norm.ret, partials=self.partials,
                           gamma=self.gamma, reward=self.raw_rew[t], obs_rms=self.obs_norm.rms if self.norm_obs else None,
                           ret_rms=self.rew_norm.rms if self.norm_reward else None, snapshot=self.snap[t])
            return
        eng.step_fused(self.clipped, done=self._starts[t + 1], ret=self.rew_norm.ret, partials=self.partials,
                       gamma=self.gamma, reward=self.raw_rew[t])
        if not self._merge_in_policy:
            ops.moments_merge(self.partials, self.N, obs_rms=self.obs_norm.rms if self.norm_obs else None,
                              ret_rms=self.rew_norm.rms if self.norm_reward else None, snapshot=self.snap[t])

    def _stat_bufs(self, merges_done: int):
        """(obs statistics, return statistics) as they stand after `merges_done` in-launch merges of this rollout:
        the canonical tensors after an even number, the alternates after an odd one (None where normalisation is off)."""
        k = max(merges_done, 0) % 2
        return ((self.obs_norm.rms, self._obs_rms_alt)[k] if self.norm_obs else None,
                (self.rew_norm.rms, self._ret_rms_alt)[k] if self.norm_reward else None)

    def _merge_tail(self, t_last: int) -> None:
        """Fold the tile sums of step t_last (the last one executed) and leave the result in the canonical tensors."""
        if not (self.fused and self._merge_in_policy):
            return
        o_in, r_in = self._stat_bufs(t_last)
        ops.moments_merge(self._rs_partials[t_last % 2] if (self._one_launch or self._persistent) else self.partials, self.N, obs_rms=self.obs_norm.rms if self.norm_obs else None,
                          ret_rms=self.rew_norm.rms if self.norm_reward else None, snapshot=self.snap[t_last],
                          obs_rms_in=o_in, ret_rms_in=r_in)

    def _step_torch(self, t: int) -> None:
        eng, pol = self.engine, self.policy
        self.buf_obs[t].copy_(self.cur_obs)
        eps = torch.randn((self.N, self.A), dtype=torch.float32, device=self.device)
        action, value, logp = pol.act(self.cur_obs, eps)
        self.buf_act[t].copy_(action)
        self.buf_val[t].copy_(value)
        self.buf_logp[t].copy_(logp)
        torch.clamp(action, -1.0, 1.0, out=self.clipped)       # SB3 clips to the Box before env.step
        eng.step(self.clipped)
        torch.bitwise_or(eng.terminated, eng.truncated, out=self.done)
        self._normalise_obs(eng.obs, self.cur_obs)
        self.snap[t, :2 * self.O + 1].copy_(self.obs_norm.rms)
        if self.norm_reward:
            self.rew_norm(eng.reward, self.done, out=self.buf_rew[t])
        else:
            self.buf_rew[t].copy_(eng.reward)
        self._starts[t + 1].copy_(self.done)

    def _all_steps(self) -> None:
        self._scan_prezeroed = False
        if self._f64_env:
            self.engine.ledger_clear()
            self._starts[0].copy_(self._starts[self.T])
            for t in range(self.T):
                self._step(t)
            return
        if self.fused_policy and self.N % 16 == 0:
            # ledger clear + episode-start carry + zeroing of the words the tail's ledger scan accumulates into: one kernel
            self.engine.rollout_begin(self._starts[0], self._starts[self.T], self._led_counters, self._led_sums)
            self._scan_prezeroed = True
        else:
            if self.fused_policy:
                self.engine.ledger_clear()
            self._starts[0].copy_(self._starts[self.T])
        if self._persistent:
            self._rollout_persistent()
        else:
            for t in range(self.T):
                self._step(t)
        # the last step's statistics update: inside the V(s_T) launch of the tail when the ping-pong parity lets its output land
        # in the canonical tensors (even n_steps), else as its own merge kernel
        fold = self.fused and self._merge_in_policy and self.fused_policy and self.T % 2 == 0 and not self.overlap_critic
        if not fold:
            self._merge_tail(self.T - 1)
        if self.fused and self.overlap_critic:
            torch.cuda.current_stream(self.device).wait_stream(self._vstream)   # join the critic passes
        if self.fused_policy:
            self._tail(merge_step=self.T - 1 if fold else None)

    def _rollout_persistent(self) -> None:
        """All T steps as one persistent launch (fpv_rollout_steps); leaves what T calls of _step_fused leave."""
        o0, r0 = self._stat_bufs(0)
        self.fp.rollout_steps(self.engine, self.T, obs_n=self.buf_obs, action=self.buf_act, value=self.buf_val,
                              logp=self.buf_logp, reward=self.raw_rew, done=self._starts[1:], ret=self.rew_norm.ret,
                              gamma=self.gamma, clip_obs=self.obs_norm.clip, eps_norm=self.obs_norm.eps, rms=o0, ret_rms=r0,
                              rms_out=(self.obs_norm.rms, self._obs_rms_alt) if self.norm_obs else (None, None),
                              ret_rms_out=(self.rew_norm.rms, self._ret_rms_alt) if self.norm_reward else (None, None),
                              snapshots=self.snap, partials=self._rs_partials, status=self._rs_status)

    def _tail(self, merge_step: int | None = None) -> None:
        """Everything between the last env step and the collective, without a host round trip (so it is part of the
        captured graph): V(s_T), deferred reward scaling, timeout bootstrap from the done ledger, GAE + its moments.
        merge_step: fold that step's moment sums into the statistics in the V(s_T) launch's prologue (what _merge_tail does)."""
        eng = self.engine
        if merge_step is None:
            self.fp.forward(eng.obs, rms=self.obs_norm.rms if self.norm_obs else None, obs_n=self.cur_obs, value=self.last_val,
                            deterministic=True, nets="v", clip_obs=self.obs_norm.clip, eps_norm=self.obs_norm.eps)
        else:
            o_in, r_in = self._stat_bufs(merge_step)             # odd number of merges so far: the alternates
            rows = self._rs_partials[merge_step % 2] if (self._one_launch or self._persistent) else self.partials
            merge = {"partials": rows, "count": self.N, "rms_out": self.obs_norm.rms if self.norm_obs else None, "ret_rms": r_in,
                     "ret_rms_out": self.rew_norm.rms if self.norm_reward else None, "snapshot": self.snap[merge_step]}
            self.fp.forward(eng.obs, rms=o_in, obs_n=self.cur_obs, value=self.last_val, deterministic=True, nets="v",
                            clip_obs=self.obs_norm.clip, eps_norm=self.obs_norm.eps, merge=merge)
        if self.deferred_critic:      # V(s_0 .. s_{T-1}) from the stored normalised rows: one launch, V's image staged once per workgroup
            self.fp.forward(self.buf_obs.view(self.T * self.N, self.O), value=self.buf_val.view(self.T * self.N),
                            deterministic=True, nets="v")
        # (V(s_T) beside the chain below on a second captured stream was measured: 70.7 vs 69.3 us per step with one 20-step
        # rollout per timed region — each fork / join edge of the graph costs 4-12 us, more than the overlap returns)
        # the deferred reward scaling rides in the ledger scan's launch (fpv_ledger_scan_finalize: the two are independent of each
        # other and of V(s_T); as launches of their own they stood one behind the other on the tail)
        fin = None
        if self.fused:
            if self.norm_reward:
                fin = {"raw": self.raw_rew, "snapshots": self.snap, "clip": self.rew_norm.clip, "eps": self.rew_norm.eps, "out": self.buf_rew}
            else:
                self.buf_rew.copy_(self.raw_rew)
        ops.ledger_scan(eng, self.T, snapshots=self.snap if self.norm_obs else None, clip_obs=self.obs_norm.clip,
                        eps=self.obs_norm.eps, tobs=self._tobs, t_idx=self._t_idx, d_idx=self._d_idx,
                        counters=self._led_counters, sums=self._led_sums, prezeroed=self._scan_prezeroed,
                        status_out=None if self._persistent else self._status_host[self._slot],   # straight into pinned host memory
                        finalize=fin)
        # V(terminal observations) and r_t += gamma * V in one launch (fpv_policy_io.scatter_*; ops.ledger_scatter is that as a launch)
        self.fp.forward(self._tobs, value=self._tv, deterministic=True, nets="v", n_dev=self._led_counters[0:1],
                        scatter={"t_idx": self._t_idx, "d_idx": self._d_idx, "out": self.buf_rew, "scale": self.gamma})
        # one rank: the advantage mean / std come out of the GAE launch itself (with several ranks: after the all-gather)
        ops.gae(self.buf_rew, self.buf_val, self.buf_start, self.last_val, self.episode_start, self.gamma,
                self.lam, advantages=self.adv, returns=self.ret, stats=self.stats,
                mean_std=self._adv_ms[self._slot] if not D.several_ranks() else None)

    # ------------------------------------------------------------------ public API
    def start(self) -> None:
        """VecNormalize.reset(): reset every drone, fold the first observation into the statistics."""
        self.engine.reset()
        if self._f64_env:
            self._obs32.copy_(self.engine.obs)
        multi = D.several_ranks() and self.norm_obs
        if multi:
            self._base_obs_rms.copy_(self.obs_norm.rms)
        self._normalise_obs(self._env_obs(), self.cur_obs)
        if multi:       # every rank's reset observations count: start from the statistics over all ranks' drones
            gathered = D.allgather_payload(torch.cat([self.obs_norm.rms.new_zeros(3), self.obs_norm.rms, self.rew_norm.rms]))
            ops.rank_merge(gathered, self.O, base_obs_rms=self._base_obs_rms, obs_rms=self.obs_norm.rms)
            self._normalise_obs(self._env_obs(), self.cur_obs, update=False)
        self.rew_norm.ret.zero_()
        self._starts[self.T].fill_(1)
        self._tick0 = self.engine.get_tick()
        self._started = True

    def _capture(self) -> None:
        """Warm up once, then capture BOTH slots' graphs (the slots differ only in which post-rollout words the tail writes).
        Capturing the second slot lazily — at the start of rollout 1 — would put its eager warm-up step into the middle of
        training: an env step whose transitions are never returned, whose episode ends the next rollout_begin wipes from the
        ledger, and which lands inside a timed region of bench.py.  Here every discarded step happens before the first
        rollout."""
        side = torch.cuda.Stream(self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):       # warm-up outside capture: lazy allocations, rocBLAS handles
            if self._persistent:
                self._all_steps()           # the persistent launch has no single-step form: one whole rollout
            else:
                self._step(0)
                self._merge_tail(0)
                if self.fused_policy:
                    self._scan_prezeroed = False   # (no rollout_begin ahead of this warm-up tail)
                    self._tail()            # first-call allocations of the tail (reduction scratch) happen here
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        self._tick0 += self.T if self._persistent else 1
        keep = self._slot
        multi = D.several_ranks()

        home = torch.cuda.current_stream(self.device)

        def capture_both(mode):
            stream = _new_capture_stream(self.device)
            graphs = []
            try:
                for slot in (0, 1):
                    self._slot = slot
                    graphs.append(torch.cuda.CUDAGraph())
                    with torch.cuda.graph(graphs[-1], stream=stream, capture_error_mode=mode):
                        self._all_steps()
            except Exception:
                # torch 2.10's CUDAGraph destructor aborts the process ("The graph should be registered to the state") when the
                # capture it belongs to failed: such an object is never released
                import ctypes
                for g in graphs:
                    ctypes.pythonapi.Py_IncRef(ctypes.py_object(g))
                torch.cuda.set_stream(home)             # (its context manager does not leave the capture stream when capture_end raises)
                _end_broken_capture(stream)
                raise
            return graphs

        # several ranks: the collective library's watchdog thread polls events while this thread captures; thread-local capture
        # mode keeps another thread's calls from invalidating the capture (nothing collective is captured here)
        mode = self._capture_mode or os.environ.get("FPV_CAPTURE_MODE") or ("thread_local" if multi else "global")
        # A capture in global mode is invalidated by ANY implicitly synchronising runtime call of the process while it is open —
        # hipFree / hipHostFree / hipGraphExecDestroy, an event query of an allocator — whichever object makes it.  The occurrence on
        # record (round 4) came when a collector was captured right after others had been dropped: their finalisers (pinned status
        # words, captured graphs, cached blocks with pending stream-use events) are exactly such calls, and they run whenever the
        # garbage collector decides to — inside the capture window if it so happens.  So the window is fenced: collect the garbage
        # NOW, drain the device, hand the allocator's cached blocks back (its event bookkeeping ends with them), and keep the
        # collector off until the capture has ended.  If a capture still fails it is taken again IN THE SAME MODE on a fresh private
        # stream (nothing has executed: a capture only records); relaxed mode — which would also let a genuinely unsafe call of ours
        # through — is the last resort and is reported (capture_mode_used, bench.py prints it); then eager launches (several ranks)
        # or the error (one rank).  The process keeps running either way — never re-exec.
        import gc

        def fenced(m):
            gc.collect()
            torch.cuda.synchronize(self.device)
            torch.cuda.empty_cache()
            was = gc.isenabled()
            gc.disable()
            try:
                return capture_both(m)
            finally:
                if was:
                    gc.enable()

        self.capture_mode_used, errors = None, []
        for attempt in (mode, mode, "relaxed"):
            try:
                self._graphs = fenced(attempt)
                self.capture_mode_used = attempt
                break
            except Exception as exc:
                errors.append(exc)
                self._graphs = [None, None]
                torch.cuda.synchronize(self.device)
                print(f"fpv rollout (rank {D.rank()}): HIP-graph capture failed in {attempt!r} mode ({exc!r})" +
                      ("; capturing again" if len(errors) < 3 else ""), file=sys.stderr, flush=True)
        if self.capture_mode_used is None:
            if not multi:
                raise errors[-1] from errors[0]
            self.use_graph = False
            print(f"fpv rollout (rank {D.rank()}): no capture succeeded; collecting with eager launches", file=sys.stderr, flush=True)
        self._slot = keep

    def set_policy_gemm(self, gemm: str) -> None:
        """'f32' (exact fp32 MFMA) or 'bf16x6' (fp32-faithful 6-term bf16 split); drops the captured graph."""
        if not self.fused:
            raise ValueError("policy_gemm applies to the fused path")
        self.fp.set_gemm(gemm)
        self.policy_gemm = gemm
        self._one_launch = self._one_launch_ok and gemm in self._one_launch_gemms   # fpv_rollout_step: exact fp32, and the split for hover
        self._persistent = self._persistent_ok and gemm == "f32"
        self.overlap_critic = self._vstream is not None and gemm == "f32"
        self._graph = None

    def sync_policy(self) -> None:
        """Repack the torch parameters into the fused kernel's weight image (after every optimiser step)."""
        if self.fused_policy:
            self.fp.load()

    def collect(self) -> dict:
        """Collect T steps from every drone, compute GAE and the global advantage statistics."""
        if not self._started:
            self.start()
        eng = self.engine
        k = self._slot = self._rollouts % 2
        late_error = None
        main = torch.cuda.current_stream(self.device)
        if self._done_stream[k] is not None and self._done_stream[k] != main:
            main.wait_event(self._stats_done[k])            # this slot's words were last read two rollouts ago: no stall
        multi = D.several_ranks() and (self.norm_obs or self.norm_reward)
        if multi:
            # the previous rollout's merged statistics (written behind its collective): this rollout starts from them, and they
            # are the base its own samples are measured against (taken before the graphs' warm-up step, whose samples the
            # statistics absorb like any other)
            if self._done_stream[self._stats_last] is not None and self._done_stream[self._stats_last] != main:
                main.wait_event(self._stats_done[self._stats_last])
            self._base_obs_rms.copy_(self.obs_norm.rms)
            self._base_ret_rms.copy_(self.rew_norm.rms)
        if self.use_graph and self._graph is None:
            self._capture()
        if self._f64_env:
            self._all_steps()
            self.fp.forward(self._obs32, rms=self.obs_norm.rms if self.norm_obs else None, obs_n=self.cur_obs, value=self.last_val,
                            deterministic=True, nets="v", clip_obs=self.obs_norm.clip, eps_norm=self.obs_norm.eps)
            self._bootstrap_timeouts()
            ops.gae(self.buf_rew, self.buf_val, self.buf_start, self.last_val, self.episode_start, self.gamma,
                    self.lam, advantages=self.adv, returns=self.ret, stats=self.stats)
        elif self.fused_policy:
            # What this slot's status words still hold is two rollouts old (no stall).  They are taken off the pinned words before
            # the launch — the rollout about to start overwrites them at its end — and turned into numbers / exceptions after it:
            # host work ahead of the launch is time the device sits idle when a caller synchronises after every rollout.
            raw = self._fetch_status(k)
            if self._persistent and self._rollouts:
                # the persistent launch wants the whole device: let the previous rollout's collective finish first
                main.wait_event(self._stats_done[self._stats_last])
            if self.use_graph:
                self._graph.replay()
            else:
                self._all_steps()
            try:
                self._parse_status(raw)
            except RuntimeError as exc:         # raised at the end of this call, once the collector's own bookkeeping is consistent
                late_error = exc
        else:
            eng.ledger_clear()
            self._all_steps()
            with torch.no_grad():
                self.last_val.copy_(self.policy.value(self.cur_obs))
            self._bootstrap_timeouts()
            ops.gae(self.buf_rew, self.buf_val, self.buf_start, self.last_val, self.episode_start, self.gamma,
                    self.lam, advantages=self.adv, returns=self.ret, stats=self.stats)
        self._rollouts += 1
        # What follows the rollout reads this slot's words where the rollout left them, on the rollout's own stream — with several
        # ranks too (round 3).  A side stream for the collective keeps the main stream free until somebody reads the statistics,
        # but the next rollout needs the merged VecNormalize statistics and the first minibatch the advantages' mean / std, and
        # each of the two hand-overs between streams costs 10-25 us of event latency: measured at 65 536 drones, one rank,
        # 65.6 vs 67.0 us per step with one 20-step rollout per timed region; in the N > 1 arrangement rehearsed with one rank
        # over RCCL (FPV_REHEARSE_RANKS=1) 66.7 vs 68.5 us per step in steady state, 69.5 vs 71.4 with one 20-step rollout per
        # region.  FPV_POST_STREAM=side restores the side stream.
        # Every event recorded between two graph launches is a packet the second launch queues behind: on one stream the rollout
        # needs exactly one (the host waits on it for the status words), where it used to record three.
        post = self._side if os.environ.get("FPV_POST_STREAM", "") == "side" else main
        with torch.cuda.stream(post):
            if post is not main:
                self._stats_ready[k].record(main)           # the rollout is complete on the main stream here
                post.wait_event(self._stats_ready[k])
            if self.fused_policy and not self._f64_env:
                # the ledger scan has written the episode statistics / overflow flags into the pinned host words itself: a copy
                # node here stood ~20 us between this rollout's last kernel and the next one's first.  (The persistent launch of
                # the experiments library has one more status word, on the device: it keeps the copy.)
                if self._persistent:
                    self._status_host[k].copy_(self._post[k][0:8], non_blocking=True)
                self._status_event[k].record(post)          # (the one event left costs ~0.2 us per step at 32-step rollouts)
                self._status_pending[k] = True
            if self._f64_env or not (self.fused_policy and not D.several_ranks()):     # (one rank, fused tail: fpv_gae has written mean / std already)
                payload = self._stats_k[k]
                if multi:                                           # VecNormalize's statistics ride in the same payload
                    payload = torch.cat([self._stats_k[k], self.obs_norm.rms, self.rew_norm.rms])
                gathered = D.allgather_payload(payload)             # the one collective per update (a view at one rank)
                ops.advantage_stats(gathered, self._adv_ms[k])
                if multi:                                           # every rank: statistics of the union of all ranks' samples
                    ops.rank_merge(gathered, self.O, base_obs_rms=self._base_obs_rms if self.norm_obs else None,
                                   base_ret_rms=self._base_ret_rms if self.norm_reward else None,
                                   obs_rms=self.obs_norm.rms if self.norm_obs else None,
                                   ret_rms=self.rew_norm.rms if self.norm_reward else None)
            if post is main and self.fused_policy and not self._f64_env:
                self._stats_done[k] = self._status_event[k]     # recorded just above, at this point of this stream
            else:
                self._stats_done[k].record(post)
        self._done_stream[k] = post
        self._stats_last = k
        self._tick0 += self.T
        if late_error is not None:              # a flag of the rollout two back (ledger overflow, ...): its data was invalid
            raise late_error
        return _Batch(self, {"obs": self.buf_obs, "actions": self.buf_act, "values": self.buf_val, "log_probs": self.buf_logp,
                             "advantages": self.adv, "returns": self.ret, "adv_mean": self._adv_ms[k][0],
                             "adv_std": self._adv_ms[k][1]})

    def wait_stats(self) -> None:
        """Order the current stream after the collective + advantage statistics of the last collect()."""
        cur = torch.cuda.current_stream(self.device)
        if self._done_stream[self._stats_last] is not None and self._done_stream[self._stats_last] != cur:
            cur.wait_event(self._stats_done[self._stats_last])      # (the same stream is ordered already)

    @property
    def adv_mean(self):
        self.wait_stats()
        return self._adv_ms[self._stats_last][0]

    @property
    def adv_std(self):
        self.wait_stats()
        return self._adv_ms[self._stats_last][1]

    def _fetch_status(self, slot: int):
        """Wait for the rollout that last used this slot and take its status words (None when they were taken already)."""
        if not self._status_pending[slot]:
            return None
        self._status_event[slot].synchronize()
        self._status_pending[slot] = False
        return self._status_np[slot].tobytes()

    def _check_status(self, slot: int) -> None:
        """Wait for the status words written at the end of an earlier rollout and turn its flags into exceptions."""
        self._parse_status(self._fetch_status(slot))

    def _parse_status(self, raw) -> None:
        if raw is None:
            return
        k, valid, episodes, flags, sum_ret, sum_len, timed_out = struct.unpack_from("<4i2di", raw)
        if timed_out != 0:
            raise RuntimeError("persistent rollout: a grid-wide wait timed out (not every workgroup was resident, or the device "
                               "was shared); this rollout is invalid — use RolloutCollector(persistent=False)")
        if flags & 1:
            raise RuntimeError(f"done ledger overflow: {episodes} episode ends but only {valid} rows fit "
                               f"({self.engine.ledger_rows_per_tile} per 64-drone tile); create the engine with a larger "
                               "ledger_capacity or collect shorter rollouts")
        if flags & 2:
            raise RuntimeError("done-ledger tag outside this rollout: tick bookkeeping is out of sync")
        if flags & 4:
            raise RuntimeError("more truncated episodes in one rollout than drones x ceil(T / 401): ledger inconsistent")
        self._episodes_host = {"count": episodes, "truncated": k,
                               "ep_rew_mean": sum_ret / valid if valid else float("nan"),
                               "ep_len_mean": sum_len / valid if valid else float("nan")}

    @property
    def episodes(self) -> dict:
        """Monitor-style statistics of the last collected rollout: {count, ep_len_mean, ep_rew_mean, truncated}.
        Waits for that rollout to finish (the only host synchronisation of the collector)."""
        if self.fused_policy and not self._f64_env and self._rollouts:
            self._check_status((self._rollouts - 1) % 2)
        return self._episodes_host

    def _bootstrap_timeouts(self) -> None:
        """r_t += gamma * V(terminal obs) for truncated-but-not-terminated episode ends; also harvests
        the Monitor-style episode statistics from the done ledger.  The torch reference of fpv_ledger_scan /
        fpv_ledger_scatter (several host synchronisations); used by the unfused collector only."""
        eng = self.engine
        total, all_rows = eng.ledger_rows()
        if int(total.item()) != all_rows.numel():
            raise RuntimeError(f"done ledger overflow: {int(total.item())} episode ends but {all_rows.numel()} rows fit "
                               f"({eng.ledger_rows_per_tile} per 64-drone tile); create the engine with a larger "
                               "ledger_capacity or collect shorter rollouts")
        k = all_rows.numel()
        self._episodes_host = {"count": int(total.item()), "truncated": 0,
                               "ep_rew_mean": float(eng.led_ret[all_rows].double().mean()) if k else float("nan"),
                               "ep_len_mean": float((eng.led_info[all_rows] >> 8).double().mean()) if k else float("nan")}
        if k == 0:
            return
        info = eng.led_info[all_rows]
        rows = all_rows[((info & 2) != 0) & ((info & 1) == 0)]
        self._episodes_host["truncated"] = int(rows.numel())
        if rows.numel() == 0:
            return
        t_idx = (eng.led_tag[rows].to(torch.int64) - (self._tick0 & 0xFFFFFFFF)) & 0xFFFFFFFF
        if int(t_idx.max()) >= self.T:
            raise RuntimeError("done-ledger tag outside this rollout: tick bookkeeping is out of sync")
        d_idx = eng.led_idx[rows].to(torch.int64)
        tobs = eng.led_obs[rows]
        if self.norm_obs:
            rms = self.snap[t_idx]                          # statistics as they were at that step
            mean, var = rms[:, :self.O], rms[:, self.O:2 * self.O]
            tobs = torch.clamp((tobs.double() - mean) / torch.sqrt(var + self.obs_norm.eps), -self.obs_norm.clip,
                               self.obs_norm.clip).float()
        with torch.no_grad():
            v = self.policy.value(tobs.float())
        self.buf_rew.index_put_((t_idx, d_idx), self.gamma * v, accumulate=True)
