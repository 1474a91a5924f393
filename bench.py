#!/usr/bin/env python3
"""bench.py — env-steps/sec of the on-device PPO rollout at 65 536 drones per MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched by
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`, one rank per GPU
over RCCL.  A "step" is one vectorised pass of the hot path over this rank's drones: policy MLP
forward + Gaussian sample, the fused HIP env-step (12 physics sub-steps, obs, reward, done,
auto-reset), VecNormalize of obs and reward, rollout-buffer writes — ONE kernel per step
(fpv_rollout_step; `--two-launch` runs it as the policy kernel + the env-step kernel); every n_steps steps the rollout
is closed with the value bootstrap, GAE (HIP) and the ONE all-gather of advantage statistics.
K is timed exactly (barrier + synchronize on both sides, MAX over ranks); value = drones on all
ranks x K / time.  Inputs are synthetic (random-init policy, seeded random-init drone states,
BASELINE.md §3) and resident in HBM before the timed region.

The K-step timed region is repeated `--repeats` times (default 5, more when K is small so that about
1 024 steps are timed in all), each bracketed by barrier + synchronize and reduced with MAX over
ranks; the line reports the MEDIAN repeat (all repeats are listed under `repeats_ms_per_step`).

Adds to the JSON line: `roofline` for the kernel that dominates the timed region by time (the rollout-step
kernel: actor-critic forward on fp32 MFMA + the env step of the same drones; its `phases` come from
in-kernel clock stamps) and `roofline_env_step` for the HBM-bound dynamics (the env-step phase of that
kernel, and the stand-alone env-step kernels), from device timestamps of the launches inside the
replayed graph with a HIP-event back-to-back figure beside them; `cpu_baseline` (the C oracle on this box's host cores: all cores and one thread; rank 0,
N = 1 only); `extra.value_f64_env` (the fp64 engine's env step alone).  oracle/ is used here ONLY for
the cpu_baseline leg.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES_PER_STEP = 382  # 35 words r+w, 16 B action, 80 B obs, 4 B reward, 2 B flags
HBM_PEAK_GBPS = 8000.0    # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense f32-input MFMA (= vector fp32) peak
MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (~2.5 PFLOP/s; the 2:1-sparsity figure is not used)
# what the 6-term bf16 split of the policy issues per drone on the bf16 pipe: both nets, layer 1 (K = 20 padded to 32) and layer 2,
# six piece products each (csrc/fpv_policy_split.h mfma6)
SPLIT_BF16_FLOP_PER_ROW = 2 * 6 * 2 * (128 * 32 + 128 * 128)  # 491 520
POLICY_FLOP_PER_ROW = 2 * (2 * (20 * 128 + 128 * 128) + 128 * 4 + 128)  # 77 056: both MLPs + heads
# fp32 operations of one hover env step per drone, counted (profiles/r02_step_instruction_mix.json, per 64-drone wave: 529 add +
# 692 mul + 831 fma (x 2) + 120 transcendental); SURVEY.md §8(d) had estimated 5.2 k
ENV_FLOP_PER_STEP = 529 + 692 + 2 * 831 + 120  # 3 003


def synthetic_states(n: int, seed: int):
    """Seeded airborne random-init drone states, canonical [n,36] float32 numpy."""
    from fpv_drone_rl_agent_amd.synthetic import synthetic_states as gen
    return gen(n, seed=seed).astype("float32")


def cpu_baseline(n: int, seed: int, budget_s: float = 9.0, detector: str = "pixel"):
    """The C oracle (kind "port") on this box's host cores: OpenMP over drones on ALL cores, and one thread.
    Bounded samples (~budget_s of wall time each on the all-cores leg, a 4 096-drone slice for the single thread)."""
    import numpy as np
    from oracle import oracle as orc
    orc.build()
    host = os.cpu_count() or 1
    usable = host
    try:
        usable = min(usable, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:                                                        # cgroup v2 CPU quota of this container, if any
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            usable = max(1, min(usable, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass

    def sample(m, threads, budget):
        st = orc.synthetic_states(m, seed=seed)
        act = np.random.Generator(np.random.Philox(seed + 1)).uniform(-1, 1, (m, 4))
        cfg = orc.make_cfg(noise_ratio=0.02, seed=seed, detector=detector)
        orc.step_batch(cfg, st, act, 0, nthreads=threads)       # warms the thread pool
        steps, t0 = 0, time.perf_counter()
        while True:
            orc.step_batch(cfg, st, act, 1 + steps, nthreads=threads)
            steps += 1
            dt = time.perf_counter() - t0
            if dt >= budget or steps >= 100000:
                return m * steps / dt, steps, dt

    # "all cores": every usable core; where the box shares its cores (more threads than this container is really given)
    # fewer threads can be faster, so 64 and 16 are tried too and the best count is the one reported and stated
    tries = sorted({usable, min(usable, 64), min(usable, 16)}, reverse=True)
    probe = {c: round(sample(n, c, 1.5)[0]) for c in tries} if len(tries) > 1 else None
    cores = max(probe, key=probe.get) if probe else usable
    v_all, k_all, t_all = sample(n, cores, budget_s)
    v_one, k_one, t_one = sample(4096, 1, budget_s)
    return {"value": v_all, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"oracle/fpv_oracle.c env-step only (no policy, detector={detector}), {n} drones x {k_all} steps, OpenMP {cores} threads "
                      f"(host reports {host} cores, {usable} usable" + (f"; thread counts probed {probe}" if probe else "") + f"), {t_all:.1f} s",
            "single_thread": {"value": v_one, "unit": "env-steps/s", "cores": 1,
                              "sample": f"4096 drones x {k_one} steps, 1 thread, {t_one:.1f} s"}}


def host_mode_rate(n: int = 4096, steps: int = 200):
    """BASELINE configs[1]: 4 096 hover drones, HIP dynamics kernel, policy on the CPU — FpvVecEnv's host (numpy) mode under
    VecNormalize, the way stock SB3 would drive it (simulation/train_hover.py:42-43): per step one torch-CPU forward of
    MlpPolicy [128, 128] on the normalised observations, one H2D of the actions, the env-step kernel, one D2H of observations /
    rewards / flags (+ the finished episodes' ledger rows).  PCIe-inclusive by construction: not the headline."""
    import numpy as np
    import torch
    from fpv_drone_rl_agent_amd.policy import ActorCritic
    from fpv_drone_rl_agent_amd.vec_env import FpvVecEnv, VecNormalize
    env = VecNormalize(FpvVecEnv(n, noise_ratio=0.02, seed=0))
    torch.manual_seed(0)
    pol = ActorCritic()                                          # on the CPU
    threads0 = torch.get_num_threads()
    try:                                                         # torch sizes its pool by the host's cores, not by what this process may use
        torch.set_num_threads(max(1, min(threads0, len(os.sched_getaffinity(0)), 16)))
    except (AttributeError, OSError):
        pass
    obs = env.reset()
    t_pol = 0.0

    def one(obs):
        nonlocal t_pol
        t0 = time.perf_counter()
        with torch.no_grad():
            o = torch.from_numpy(np.ascontiguousarray(obs, dtype=np.float32))
            a, _, _ = pol.act(o, torch.randn(n, 4))
        act = a.clamp(-1.0, 1.0).numpy()
        t_pol += time.perf_counter() - t0
        return env.step(act)[0]

    for _ in range(20):
        obs = one(obs)
    t_pol = 0.0
    t0 = time.perf_counter()
    for _ in range(steps):
        obs = one(obs)
    dt = time.perf_counter() - t0
    env.close()
    used = torch.get_num_threads()
    torch.set_num_threads(threads0)
    return {"value": n * steps / dt, "unit": "env-steps/s", "drones": n, "us_per_step": round(dt / steps * 1e6, 1),
            "us_per_step_policy_on_cpu": round(t_pol / steps * 1e6, 1), "cpu_threads_torch": used,
            "workload": "BASELINE.json configs[1]: 4 096 hover drones, HIP env-step kernel, torch-CPU MlpPolicy, numpy in / out through "
                        "FpvVecEnv + VecNormalize (PCIe-inclusive)"}


def plumbing_only(args) -> None:
    """bench.py's launch contract without a GPU: what `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`
    must get right before any kernel runs — env-var rendezvous, WORLD_SIZE == --gpus, the rollout length that makes exactly K
    steps, barrier-bracketed timing reduced with MAX over ranks, the advantage-statistics all-gather, one JSON line from rank 0."""
    import torch
    import torch.distributed as td
    from fpv_drone_rl_agent_amd import dist as D
    rank, world, local = D.init_from_env("gloo")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    K = args.steps
    T = max(d for d in range(1, min(args.n_steps, K) + 1) if K % d == 0)
    if world > 1:
        td.barrier()
    t0 = time.perf_counter()
    payload = torch.tensor([float(args.drones), 1.0 + rank, 2.0 + rank], dtype=torch.float64)
    gathered = D.allgather_payload(payload)
    mean, std = D.merge_advantage_stats(gathered)
    if world > 1:
        td.barrier()
    elapsed = time.perf_counter() - t0 + 0.001 * rank            # ranks disagree on purpose: the line must carry the MAX
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64)
        td.all_reduce(tmax, op=td.ReduceOp.MAX)
        elapsed = float(tmax.item())
    if rank == 0:
        print(json.dumps({"metric": "env_steps_per_sec", "value": None, "unit": "env-steps/s", "n_gpus": world, "steps": K,
                          "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f32", "data": "none (plumbing only)",
                          "config": {"workload": "plumbing only", "drones_per_gpu": args.drones, "total_drones": args.drones * world,
                                     "n_steps": T},
                          "plumbing": {"gathered_rows": int(gathered.shape[0]), "adv_mean": float(mean), "max_elapsed_s": elapsed}}),
              flush=True)
    if world > 1:
        td.destroy_process_group()


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher around it (no RANK / WORLD_SIZE in the environment): this process makes NO
    GPU call — it starts the N ranks as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port <free> bench.py <same arguments>` (the driver's own launch line), hands rank 0's one JSON line on to stdout and
    returns the launcher's exit code.  The reference's counterpart is make_vec_env(..., n_envs=16, vec_env_cls=SubprocVecEnv)
    (simulation/train_hover.py:41-42): worker processes started by the script itself."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this driver stack
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    for ln in proc.stdout.splitlines():
        if ln not in lines:
            print(ln, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    elif proc.returncode == 0:
        print("bench.py: the ranks ended without a result line", file=sys.stderr)
        return 4
    return proc.returncode


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1024)
    ap.add_argument("--warmup", type=int, default=128)
    ap.add_argument("--drones", type=int, default=65536, help="drones per GPU (BASELINE.json configs[2])")
    ap.add_argument("--n-steps", type=int, default=32, help="rollout length between GAE passes")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--task", choices=["hover", "yaw"], default="hover",
                    help="yaw = BASELINE configs[4] (yaw-setpoint task with domain randomisation); not the headline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-split-extra", action="store_true",
                    help="skip the extra measurements with the other policy arithmetic (exact fp32-input MFMA) and launch shape")
    ap.add_argument("--policy-gemm", choices=["bf16x6", "f32"], default=None,
                    help="policy arithmetic of the headline: bf16x6 (the default since round 5: the 128-wide products as 6-term bf16 "
                         "splits, fp32 accumulate, every row at fp32 rounding level) or f32 (exact fp32-input MFMA, rounds 1-4's headline)")
    ap.add_argument("--no-host-mode", action="store_true", help="skip the extra BASELINE configs[1] (host numpy mode) measurement")
    ap.add_argument("--kernel-launches", type=int, default=200)
    ap.add_argument("--plumbing-only", action="store_true",
                    help="no GPU work: rendezvous, argument / rank checks, the barrier + MAX-over-ranks timing and the one "
                         "all-gather on the gloo backend, then the JSON line with value = null (tests/test_dist_gloo.py)")
    ap.add_argument("--repeats", type=int, default=0,
                    help="the K-step timed region is run this many times and the median is reported; default: 5, or as many as "
                         "it takes to time about 1 024 steps in all when K is small (a 20-step region is 1.4 ms: the first few "
                         "run on a chip that is still raising its clock)")
    ap.add_argument("--detector", choices=["analytic", "pixel"], default="pixel",
                    help="hover task camera features: the reference's 128x128 pixel rules (SPEC §6b, hover.py:157-222; the default and "
                         "the headline) or the continuous projection (§6)")
    ap.add_argument("--no-f64-rollout", action="store_true",
                    help="skip the extra rollout measurement on an fp64 engine (the reference's arithmetic type); runs with --steps >= 256 only")
    ap.add_argument("--two-launch", action="store_true",
                    help="policy forward and env step as two kernels per rollout step (the default is fpv_rollout_step: one)")
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))

    import torch
    import torch.distributed as td
    from fpv_drone_rl_agent_amd import dist as D
    from fpv_drone_rl_agent_amd import ops
    from fpv_drone_rl_agent_amd.engine import FpvEngine
    from fpv_drone_rl_agent_amd.policy import ActorCritic
    from fpv_drone_rl_agent_amd.rollout import RolloutCollector

    if args.plumbing_only:
        return plumbing_only(args)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    # FPV_DIST_BACKEND=gloo rehearses several ranks on a one-GPU box; the driver's runs use RCCL
    backend = os.environ.get("FPV_DIST_BACKEND", "nccl")
    try:
        rank, world, local = D.init_from_env(backend)
    except Exception as exc:        # a failed RCCL bring-up must end this rank loudly, not hang the launcher
        print(f"bench.py: torch.distributed init over {backend} failed on RANK={os.environ.get('RANK')}: {exc!r}",
              file=sys.stderr, flush=True)
        raise SystemExit(3)
    if backend != "nccl":
        local = local % torch.cuda.device_count()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    n = args.drones
    K, W = args.steps, args.warmup
    # rollout length: the largest divisor of K not above --n-steps, so that exactly K steps are timed
    T = max(d for d in range(1, min(args.n_steps, K) + 1) if K % d == 0)

    torch.manual_seed(0)
    yaw = args.task == "yaw"
    policy = (ActorCritic(obs_dim=12, act_dim=1) if yaw else ActorCritic()).to(dev)
    eng = FpvEngine(n, dev, noise_ratio=0.02, seed=0, id_offset=rank * n, task=args.task, domain_rand=yaw,
                    ledger_capacity=n * max(2, T // 8), detector="analytic" if yaw else args.detector)
    col = RolloutCollector(eng, policy, n_steps=T, use_graph=not args.no_graph, one_launch=False if args.two_launch else None,
                           policy_gemm=args.policy_gemm)
    gemm = col.policy_gemm                                   # "bf16x6" unless --policy-gemm f32
    col.start()
    if not yaw:
        eng.set_state(torch.from_numpy(synthetic_states(n, seed=rank)))  # staggered step counts => staggered dones
    else:
        st = eng.get_state()                                             # same staggering for the yaw episodes
        st[:, 34] = torch.randint(0, 401, (n,), device=dev, generator=torch.Generator(device=dev).manual_seed(rank)).float()
        eng.set_state(st)

    def barrier():
        if world > 1:
            td.barrier()
        torch.cuda.synchronize(dev)

    def timed_region():
        """EXACTLY K steps between barrier + synchronize on both sides; MAX over ranks."""
        barrier()
        t0 = time.perf_counter()
        for _ in range(K // T):
            col.collect()
        barrier()
        el = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([el], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            td.all_reduce(tmax, op=td.ReduceOp.MAX)
            el = float(tmax.item())
        return el

    for _ in range(max(1, math.ceil(W / T))):
        col.collect()
    n_rep = args.repeats if args.repeats > 0 else min(65, max(5, math.ceil(1024 / K)))
    repeats = sorted(timed_region() for _ in range(n_rep))
    elapsed = repeats[len(repeats) // 2] if len(repeats) % 2 else 0.5 * (repeats[len(repeats) // 2 - 1] + repeats[len(repeats) // 2])
    episodes = int(col.episodes["count"])

    # ---- extras (not the headline): the same K steps with the OTHER policy arithmetic — exact fp32-input MFMA when the headline runs
    # the 6-term bf16 split (the default), the split when the headline was asked to be exact — as one kernel per step where that is
    # served, and as two kernels per step ----
    other_gemm = "f32" if gemm == "bf16x6" else "bf16x6"
    other_value = other_value_two = None
    if not yaw and not args.no_split_extra:
        def other_rate():
            for _ in range(2):
                col.collect()
            r2 = sorted(timed_region() for _ in range(max(1, min(3, n_rep))))
            return n * world * K / r2[len(r2) // 2]
        was_one = col._one_launch
        col.set_policy_gemm(other_gemm)
        if col._one_launch:
            other_value = other_rate()
            col._one_launch = False
            col._graph = None
        other_value_two = other_rate()
        col.set_policy_gemm(gemm)
        col._one_launch = was_one
        col._graph = None

    # ---- extra (not the headline): the same rollout with the env in the reference's own arithmetic type (float64 spaces,
    # simulation/hover.py:61, :70): fp32 policy kernel + fpv_step_f64 + stand-alone normaliser kernels, no graph, host-synchronising
    # timeout bootstrap (rollout.py).  A few rollouts only: it exists to be right, not to be fast. ----
    f64_rollout_value = None
    if not yaw and not args.no_f64_rollout and K >= 256:       # (a short driver-shaped run, --steps 20, skips it: four host-synchronising fp64 rollouts)
        eng64 = FpvEngine(n, dev, noise_ratio=0.02, seed=0, id_offset=rank * n, ledger_capacity=n * max(2, T // 8), detector=args.detector,
                          precision="f64")
        col64 = RolloutCollector(eng64, policy, n_steps=T)
        col64.start()
        eng64.set_state(torch.from_numpy(synthetic_states(n, seed=rank)).double())
        col64.collect()
        barrier()
        t0 = time.perf_counter()
        reps64 = 3
        for _ in range(reps64):
            col64.collect()
        barrier()
        el64 = time.perf_counter() - t0
        if world > 1:
            tm = torch.tensor([el64], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            td.all_reduce(tm, op=td.ReduceOp.MAX)
            el64 = float(tm.item())
        f64_rollout_value = n * world * T * reps64 / el64
        eng64.close()
        del col64, eng64

    # ---- rooflines: kernel-only passes in this process, HIP events on the launch stream (events cannot bracket a
    # node inside the replayed graph).  `roofline` is the kernel that dominates the timed region by time (the fused
    # actor-critic forward, MFMA-bound); `roofline_env_step` is the HBM-bound dynamics kernel BASELINE's "% HBM
    # roofline" refers to, as it runs in the rollout (with the VecNormalize bookkeeping fused in) and stand-alone.
    roofline = roofline_env = roofline_env_plain = roofline_env_fused = out_hbm_whole = None
    if True:   # every rank runs the same passes (collect() contains the collective); only rank 0 reports them
        L = args.kernel_launches
        act = torch.rand((n, eng.act_words), device=dev) * 2 - 1

        def timed(fn):
            for _ in range(20):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(dev)
            e0.record()
            for _ in range(L):
                fn()
            e1.record()
            torch.cuda.synchronize(dev)
            return e0.elapsed_time(e1) * 1e3 / L

        def traffic_of(kernel):
            # HBM bytes per launch from the PMC passes recorded under profiles/ (not collectable in-process)
            try:
                rec = json.load(open(os.path.join(ROOT, "profiles", "kernel_traffic.json")))
                if int(rec["drones"]) == n and kernel in rec["kernels"]:
                    return int(rec["kernels"][kernel]["hbm_bytes_per_launch"])
            except (OSError, ValueError, KeyError):
                pass
            return None

        def hbm(kernel, us, alg_bytes):
            achieved = alg_bytes * n / us / 1e3
            return {"bound": "hbm", "kernel": kernel, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic_of(kernel), "us_per_launch": round(us, 2),
                    "launches": L, "algorithmic_bytes_per_launch": alg_bytes * n}

        def graph_kernel_times(rollouts=4):
            """Average device-side duration of every kernel inside the replayed rollout graph — the timed region's own
            launches — from the HIP activity tracer (torch.profiler / roctracer timestamps).  HIP events cannot bracket a
            node inside a replayed graph; {} if the tracer is unavailable (e.g. another profiler owns it)."""
            under_profiler = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or \
                "rocprof" in os.environ.get("LD_PRELOAD", "")
            if under_profiler:       # rocprofv3 owns the activity tracer: its own trace is the measurement then
                return {}
            try:
                from torch.profiler import ProfilerActivity, profile
                with profile(activities=[ProfilerActivity.CUDA]) as prof:
                    for _ in range(rollouts):
                        col.collect()
                    torch.cuda.synchronize(dev)
                acc = {}
                for e in prof.events():
                    if e.device_type == torch.autograd.DeviceType.CUDA:
                        acc.setdefault(e.name, []).append(getattr(e, "device_time", None) or e.cuda_time)
                return acc
            except Exception:
                return {}

        def kernel_phases():
            """Median in-kernel phase durations (us) of fpv::rollout_step_kernel from its diagnostic clock stamps
            (fpv_debug_rollout_stamps): one extra captured rollout with the stamp buffer attached."""
            try:
                import ctypes as C
                from fpv_drone_rl_agent_amd import _lib as LIB
                import numpy as np
                grid = n // 256
                stamps = torch.zeros((grid, 16), dtype=torch.int64, device=dev)
                LIB.load().fpv_debug_rollout_stamps(eng._h, C.c_void_p(stamps.data_ptr()))
                col._graph = None                     # the stamp pointer is a kernel argument: capture again
                col.collect()
                torch.cuda.synchronize(dev)
                LIB.load().fpv_debug_rollout_stamps(eng._h, None)
                col._graph = None
                col.collect()
                torch.cuda.synchronize(dev)
                st = stamps.cpu().numpy().astype(np.float64) / 100.0
                names = ["staging + statistics merge", "policy (normalise, pi, V, sample)", "env step + bookkeeping"]
                out = {nm: round(float(np.median(st[:, k + 1] - st[:, k])), 2) for k, nm in enumerate(names)}
                out["first workgroup start -> last workgroup end"] = round(float(st[:, 3].max() - st[:, 0].min()), 2)
                return out
            except Exception:
                return None

        def pick(times, needle, at_least=0.0):
            """mean duration of the launches of the kernel whose name contains `needle`; at_least drops the short launches
            of the same kernel (the rollout tail runs the policy kernel V-only and over the short done-ledger list)"""
            for k, v in times.items():
                if needle in k.replace(", ", ","):
                    v = [x for x in v if x >= at_least]
                    return float(sum(v) / len(v)) if v else None
            return None

        if yaw and not (col.fused and col._one_launch):
            us = timed(lambda: eng.step(act))
            roofline = hbm("fpv::yaw_step_kernel<float,true>", us, 458)   # 44 words r+w, 4 B action, 48 B obs, 4 B reward, 2 B flags
        else:
            us_plain = timed(lambda: eng.step(act))
            env_bytes = 458 if yaw else ALG_BYTES_PER_STEP
            # launch_step's dispatch (csrc/fpv_step.hip): pixel-detector handles of 8 192 < n <= 65 536 drones step through
            # step_quad_kernel (physics + helper wave per tile, the phase 2 of the rollout-step kernel as a launch of its own)
            quad = not yaw and args.detector == "pixel" and 8192 < n <= 65536 and n % 256 == 0
            k_plain = "fpv::step_quad_kernel" if quad else "fpv::step_kernel<float,true,false>"
            k_fused = "fpv::step_quad_kernel" if quad else "fpv::step_kernel<float,true,true>"
            roofline_env_plain = hbm("fpv::yaw_step_kernel<float,true>" if yaw else k_plain, us_plain, env_bytes)
            if not yaw:
                roofline_env_plain["detector"] = args.detector
                if args.detector == "pixel" and world == 1:
                    # the same stand-alone kernel with the analytic detector (the configuration of rounds 1-2 and of the N sweep in
                    # profiles/r02_step_sweep.json): the pixel rules cost this kernel a serial row walk per drone, which the
                    # rollout-step kernel spreads over two waves
                    ea = FpvEngine(n, dev, noise_ratio=0.02, seed=0, detector="analytic")
                    ea.set_state(torch.from_numpy(synthetic_states(n, seed=0)))
                    roofline_env_plain["same_kernel_with_the_analytic_detector"] = hbm("fpv::step_kernel<float,true,false>", timed(lambda: ea.step(act)), env_bytes)
                    ea.close()
            pol_flop = 2 * (2 * (12 * 128 + 128 * 128) + 128 * 1 + 128) if yaw else POLICY_FLOP_PER_ROW
            env_flop = 0 if yaw else ENV_FLOP_PER_STEP      # (the yaw env step's operations have not been counted: left out)
            if col.fused and col._one_launch:
                # ---- the rollout-step kernel (policy + env step in one launch)
                ev_fused = timed(lambda: eng.step_fused(col.clipped, done=col._starts[1], ret=col.rew_norm.ret,
                                                        partials=col.partials, gamma=col.gamma, reward=col.raw_rew[0]))
                # the same launch as inside the rollout, statistics merge of the previous step's rows in its prologue included
                # (two row tables and two statistics buffers alternate, as in the collector)
                o_alt, r_alt = col._obs_rms_alt, col._ret_rms_alt
                flip = [0]

                def one_step():
                    k = flip[0] = flip[0] ^ 1
                    merge = {"partials": col._rs_partials[k ^ 1], "count": n, "rms_out": (col.obs_norm.rms, o_alt)[k],
                             "ret_rms": (col.rew_norm.rms, r_alt)[k ^ 1], "ret_rms_out": (col.rew_norm.rms, r_alt)[k], "snapshot": col.snap[0]}
                    col.fp.rollout_step(eng, rms=(col.obs_norm.rms, o_alt)[k ^ 1], obs_n=col.buf_obs[0], action=col.buf_act[0],
                                        value=col.buf_val[0], logp=col.buf_logp[0], done=col._starts[1], ret=col.rew_norm.ret,
                                        partials=col._rs_partials[k], gamma=col.gamma, reward=col.raw_rew[0], merge=merge,
                                        clip_obs=col.obs_norm.clip, eps_norm=col.obs_norm.eps)
                ev_one = timed(one_step)
                in_graph = graph_kernel_times()
                split = gemm == "bf16x6"
                k0_name = "fpv::rollout_step_split_kernel" if split else "fpv::rollout_step_kernel<%d>" % (1 if yaw else 0)
                us_traced = pick(in_graph, "rollout_step_split_kernel" if split else "rollout_step_kernel")
                # `us_per_launch` = HIP events on the launch stream around back-to-back launches of the kernel (what rocprofv3's
                # per-kernel average agrees with, profiles/); the activity tracer's timestamps of the launches inside the replayed
                # graph stand beside it (the tracer stretches the replay it traces)
                us_one = ev_one
                how = "HIP events on the launch stream around %d back-to-back launches (statistics-merge prologue included)" % L
                # us_per_launch is what was MEASURED for the kernel (nothing is capped or substituted); the timed region's own
                # time per step stands beside it (one launch per step on one stream: the kernel cannot really be longer than
                # that — the activity tracer stretches the replay it traces — and `frac_upper_bound_from_step_cadence` is the
                # fraction that figure would give, published separately and named for what it is)
                cadence = elapsed / K * 1e6
                phases = kernel_phases() if world == 1 else None      # (extra captures + collects: kept off the multi-rank runs)
                tf = pol_flop * n / us_one / 1e6                         # MFMA work only: what `bound: mfma` prices
                tf_all = (pol_flop + env_flop) * n / us_one / 1e6
                roofline = {"bound": "mfma", "kernel": k0_name, "achieved": round(tf, 1),
                            "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / MFMA_F32_PEAK_TFLOPS, 4),
                            "traffic": None if yaw else traffic_of(k0_name.split("<")[0]), "us_per_launch": round(us_one, 2),
                            "launches": L, "algorithmic_flop_per_launch": pol_flop * n,
                            "algorithmic_flop_per_drone": {"policy (MFMA)": pol_flop, "env step (VALU, counted, not in frac)": env_flop or None},
                            "frac_fp32_pipe_incl_env_step_valu_flops": round(tf_all / MFMA_F32_PEAK_TFLOPS, 4),
                            "dtype": "f32 (policy products: 6-term bf16 split, fp32 accumulate)" if split else "f32",
                            "measured": how,
                            "us_per_launch_traced_inside_the_replayed_graph": round(us_traced, 2) if us_traced else None,
                            "us_per_step_timed_region": round(cadence, 2),
                            "frac_upper_bound_from_step_cadence": round(pol_flop * n / min(us_one, cadence) / 1e6 / MFMA_F32_PEAK_TFLOPS, 4)}
                if split:
                    bf = SPLIT_BF16_FLOP_PER_ROW * n / us_one / 1e6
                    roofline["as_issued_on_the_bf16_pipe"] = {
                        "flop_per_drone": SPLIT_BF16_FLOP_PER_ROW, "achieved": round(bf, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(bf / MFMA_BF16_PEAK_TFLOPS, 4),
                        "note": "the six bf16 piece products per fp32 product that the kernel really issues, over the WHOLE kernel; the "
                                "matrix pipe is no longer what bounds this kernel — vector issue is (the cut of the activations into "
                                "pieces, tanh, heads, sampling, and the env step: DESIGN §4)"}
                roofline["note"] = (("`achieved` / `frac` price the ALGORITHMIC work — the policy's 77 056 fp32 multiply-add flops per drone, "
                                     "delivered at fp32 rounding level — against the dense fp32-input MFMA peak, the rate an exact-fp32 matrix "
                                     "formulation is bound by (it is the figure rounds 1-4 reported for rollout_step_kernel<0>, so the two are "
                                     "comparable); the products themselves run as 6-term bf16 splits on the bf16 pipe, priced in "
                                     "`as_issued_on_the_bf16_pipe`.  " if split else "") +
                                    "one kernel per rollout step: actor-critic forward (the matrix work `frac` prices: MFMA flops only over the "
                                    "WHOLE kernel's measured duration) AND the env step of the same drones (vector work on the same fp32 pipe; "
                                    "`frac_fp32_pipe_incl_env_step_valu_flops` adds its counted flops); peak = dense fp32-input MFMA rate (155 TF "
                                    "measured with resident operands at a 2.38 GHz in-kernel clock, profiles/r02_mfma_peak.txt); vector "
                                    "instructions do not co-execute with fp32-input MFMA (profiles/r02_mfma_valu_mix.txt), so tanh / heads / "
                                    "sampling add to the product time (profiles/r02_policy_phase_stamps.md)")
                roofline["traffic_source"] = "profiles/kernel_traffic.json (rocprofv3 --pmc passes of this command; not collectable in-process)"
                # the same kernel on the HBM roofline: every byte it must move (SURVEY §8d full-rollout figure minus the GAE pass that
                # runs once per rollout: 20-word observation in, rollout-buffer row out, state r + w, next observation, reward,
                # flags, return accumulator) over the whole kernel's measured duration
                k0_bytes = 563 if not yaw else None
                if k0_bytes:
                    out_hbm_whole = hbm(k0_name + " (whole kernel)", us_one, k0_bytes)
                    out_hbm_whole.update(traffic=traffic_of(k0_name.split("<")[0]), measured=how,
                                         note="563 B per drone and step over the WHOLE kernel's duration: the kernel is bound by instruction "
                                              "issue (policy products and vector work), not by HBM; this is the honest HBM figure of the rollout step")
                else:
                    out_hbm_whole = None
                if phases:
                    roofline["phases_us"] = phases
                    pol = phases.get("policy (normalise, pi, V, sample)")
                    if pol:
                        roofline["frac_of_peak_within_policy_phase"] = round(pol_flop * n / pol / 1e6 / MFMA_F32_PEAK_TFLOPS, 4)
                    envp = phases.get("env step + bookkeeping")
                    if envp:
                        # NOT a bandwidth measurement: the drone state was requested ~45 us earlier, under the matrix work, so the
                        # phase's bytes are not moved inside the phase — reported as a duration only
                        roofline_env = {"kernel": "env-step phase of fpv::rollout_step_kernel", "exposed_phase_us": envp,
                                        "measured": "in-kernel s_memrealtime stamps, median over workgroups",
                                        "note": "duration of the phase only; no bandwidth fraction is claimed for it (its state loads were "
                                                "issued under the policy phase) — see roofline_hbm_whole_kernel and "
                                                "roofline_env_step_unfused for HBM figures"}
                roofline_env_fused = hbm("fpv::yaw_step_kernel<float,true,true>" if yaw else k_fused, ev_fused, env_bytes)
                roofline_env_fused.update(measured="HIP events around back-to-back launches (the stand-alone fpv_step_fused kernel; "
                                                   "not launched by the default rollout)")
            elif col.fused:
                io = dict(rms=col.obs_norm.rms, tick=eng.tick_ptr, obs_n=col.buf_obs[0], action=col.buf_act[0],
                          clipped=col.clipped, value=col.buf_val[0], logp=col.buf_logp[0])
                ev_pol = timed(lambda: col.fp.forward(eng.obs, **io))
                ev_fused = timed(lambda: eng.step_fused(col.clipped, done=col._starts[1], ret=col.rew_norm.ret,
                                                        partials=col.partials, gamma=col.gamma, reward=col.raw_rew[0]))
                in_graph = graph_kernel_times()
                pol_sym = "policy_split_forward_kernel" if gemm == "bf16x6" else "policy_forward_kernel"
                us_pol = pick(in_graph, pol_sym, at_least=0.75 * ev_pol) or ev_pol
                us_pol_all = pick(in_graph, pol_sym)     # what a per-symbol profiler average shows
                us_fused = pick(in_graph, "step_quad_kernel" if quad else "step_kernel<float,true,true") or ev_fused
                if not (0.6 * ev_pol < us_pol < 1.6 * ev_pol and 0.6 * ev_fused < us_fused < 1.6 * ev_fused):
                    in_graph, us_pol, us_fused = {}, ev_pol, ev_fused     # tracer output implausible: trust the events
                how = "device timestamps of the launches inside the replayed rollout graph" if in_graph else \
                      "HIP events around back-to-back launches"
                roofline_env = hbm(k_fused, us_fused, ALG_BYTES_PER_STEP)
                roofline_env.update(measured=how, us_per_launch_hip_events_back_to_back=round(ev_fused, 2))
                tf = POLICY_FLOP_PER_ROW * n / us_pol / 1e6
                roofline = {"bound": "mfma", "kernel": "fpv::" + pol_sym, "achieved": round(tf, 1),
                            "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / MFMA_F32_PEAK_TFLOPS, 4),
                            "traffic": traffic_of("fpv::" + pol_sym), "us_per_launch": round(us_pol, 2),
                            "launches": L, "algorithmic_flop_per_launch": POLICY_FLOP_PER_ROW * n,
                            "dtype": "f32 (policy products: 6-term bf16 split, fp32 accumulate)" if gemm == "bf16x6" else "f32",
                            "measured": how, "us_per_launch_hip_events_back_to_back": round(ev_pol, 2)}
                roofline["note"] = ("peak = dense fp32-input MFMA rate (155 TF measured with resident random operands, 2.38 GHz in-kernel "
                                    "clock: profiles/r02_mfma_peak.txt); vector instructions do not co-execute with fp32-input MFMA "
                                    "(profiles/r02_mfma_valu_mix.txt), so the kernel's tanh / head / sampling work adds to its "
                                    "product time: phase table in profiles/r02_policy_phase_stamps.md")
                roofline["traffic_source"] = "profiles/kernel_traffic.json (rocprofv3 --pmc passes of this command; not collectable in-process)"
                if in_graph and us_pol_all:
                    # rocprofv3 --stats averages over ALL launches of the symbol: per rollout of n_steps that is n_steps
                    # full launches plus the tail's V-only launch and the short done-ledger launch
                    roofline["us_per_launch_all_launches_of_the_symbol"] = round(us_pol_all, 2)
            else:
                roofline = roofline_env_plain

    # ---- extra: the fp64 handle's env step alone (the reference computes in float64; this build's product is fp32) ----
    f64_value = None
    if not yaw and world == 1:
        e64 = FpvEngine(n, dev, noise_ratio=0.02, seed=0, precision="f64", detector=args.detector)
        e64.set_state(torch.from_numpy(synthetic_states(n, seed=0)).double())
        a64 = torch.rand((n, 4), device=dev, dtype=torch.float64) * 2 - 1
        us64 = timed(lambda: e64.step(a64))
        f64_value = n / us64 * 1e6
        e64.close()

    if rank == 0:
        out = {
            "metric": "env_steps_per_sec", "value": n * world * K / elapsed, "unit": "env-steps/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": elapsed / K * 1e3,
            "repeats": len(repeats), "repeats_ms_per_step": [round(r / K * 1e3, 5) for r in repeats],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (policy products: 6-term bf16 split, fp32 accumulate)" if gemm == "bf16x6" else "f32",
            "data": "synthetic (random-init policy, seeded random-init drone states, motor noise on)",
            "config": {"workload": ("BASELINE.json configs[4]: yaw-setpoint task with domain-randomised mass/drag, on-device "
                                    "PPO rollout (torch policy 12->1, HIP env-step, VecNormalize) + GAE + all-gather" if yaw else
                                    "BASELINE.json configs[2]: 65 536 hover drones per GPU, on-device PPO rollout "
                                    + ("(one kernel per step: VecNormalize statistics merge, MFMA policy forward + sample, then the "
                                       "env step + VecNormalize bookkeeping of the same drones; " if col._one_launch else
                                       "(two kernels per step: MFMA policy forward + sample with the VecNormalize statistics merge in "
                                       "its prologue, HIP env-step + VecNormalize bookkeeping; ") +
                                    "done-ledger timeout bootstrap, GAE) + advantage-stat all-gather every n_steps"),
                       "drones_per_gpu": n, "total_drones": n * world, "n_steps": T,
                       "hip_graph": not args.no_graph, "hip_graph_capture_mode": col.capture_mode_used, "kernels_per_step": 1 if col._one_launch else 2, "detector": "analytic" if yaw else args.detector, "policy": "MlpPolicy pi/vf [128,128] tanh" + ("" if yaw else " (39 049 params), " + ("hidden products as 6-term bf16 splits on the bf16 MFMA, fp32 accumulate (fp32 rounding level on every row)" if gemm == "bf16x6" else "exact fp32-input MFMA")), "policy_gemm": gemm,
                       "episodes_in_last_rollout": episodes, **({"rehearsal": "FPV_REHEARSE_RANKS=1: the collector's N > 1 arrangement (the real all-gather over a single-rank RCCL group behind every rollout, advantage statistics from the gathered payload, rank merge of the VecNormalize statistics) with ONE rank"} if (world == 1 and D.several_ranks()) else {})},
            "roofline": roofline,
            "library": __import__("fpv_drone_rl_agent_amd")._lib.library_id(),      # which build produced this line (path, sha256[:16])
        }
        if roofline_env is not None:
            out["env_step_phase" if "exposed_phase_us" in roofline_env else "roofline_env_step"] = roofline_env
            out["roofline_env_step_unfused"] = roofline_env_plain
        if out_hbm_whole is not None:
            out["roofline_hbm_whole_kernel"] = out_hbm_whole
        if roofline_env_fused is not None:
            out["roofline_env_step_standalone_fused"] = roofline_env_fused
        if other_value is not None or other_value_two is not None:
            key = "exact_fp32_policy" if other_gemm == "f32" else "policy_gemm_bf16x6"
            out["extra"] = {f"value_one_kernel_{key}": other_value, f"value_two_kernels_{key}": other_value_two,
                            "note": ("same workload with the policy's products on the exact fp32-input MFMA (rounds 1-4's headline "
                                     "arithmetic; rollout_step_kernel<0> / policy_forward_kernel)" if other_gemm == "f32" else
                                     "same workload with the policy's hidden products as 6-term bf16 splits (rollout_step_split_kernel)") +
                                    ": one kernel per step, and two kernels per step; DESIGN §5b has the decision and its evidence"}
        if f64_value is not None:
            out.setdefault("extra", {})["value_f64_env"] = f64_value
            out["extra"]["value_f64_env_note"] = ("env-steps/s of fpv_step on an FPV_F64 handle (env step only, no policy, HIP events over "
                                                  f"{args.kernel_launches} launches): the reference's arithmetic type")
        if f64_rollout_value is not None:
            out.setdefault("extra", {})["value_rollout_f64_env"] = f64_rollout_value
            out["extra"]["value_rollout_f64_env_note"] = ("env-steps/s of the SAME rollout with the env in the reference's arithmetic type (float64 "
                                                          "spaces, simulation/hover.py:61, :70): fp32 policy kernel + fpv_step_f64 + stand-alone "
                                                          "normaliser kernels, conversions at the seam, no HIP graph, host-synchronising timeout "
                                                          "bootstrap; 3 rollouts, wall clock (tests/test_facades_gpu.py holds its env outputs to the "
                                                          "oracle at 1e-9)")
        if world == 1 and not yaw and not args.no_host_mode:
            out.setdefault("extra", {})["cfg1_host_mode"] = host_mode_rate()
        if world == 1 and not args.no_cpu_baseline and not yaw:
            out["cpu_baseline"] = cpu_baseline(n, seed=0, detector=args.detector)
        print(json.dumps(out), flush=True)
    barrier()
    eng.close()
    if world > 1:
        td.destroy_process_group()


if __name__ == "__main__":
    main()
