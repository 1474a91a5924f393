#!/usr/bin/env python3
"""bench.py — env-steps/sec of the on-device PPO rollout at 65 536 drones per MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched by
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`, one rank per GPU
over RCCL.  A "step" is one vectorised pass of the hot path over this rank's drones: policy MLP
forward + Gaussian sample, the fused HIP env-step (12 physics sub-steps, obs, reward, done,
auto-reset), VecNormalize of obs and reward, rollout-buffer writes; every n_steps steps the rollout
is closed with the value bootstrap, GAE (HIP) and the ONE all-gather of advantage statistics.
K is timed exactly (barrier + synchronize on both sides, MAX over ranks); value = drones on all
ranks x K / time.  Inputs are synthetic (random-init policy, seeded random-init drone states,
BASELINE.md §3) and resident in HBM before the timed region.

Adds to the JSON line: `roofline` for the dominant kernel (the env-step kernel) from a kernel-only
pass timed with HIP events on the launch stream, and `cpu_baseline` (the C oracle on host cores,
rank 0, N = 1 only).  oracle/ is used here ONLY for the cpu_baseline leg.
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
POLICY_FLOP_PER_ROW = 2 * (2 * (20 * 128 + 128 * 128) + 128 * 4 + 128)  # 77 056: both MLPs + heads


def synthetic_states(n: int, seed: int):
    """Seeded airborne random-init drone states, canonical [n,36] float32 numpy."""
    from fpv_drone_rl_agent_amd.synthetic import synthetic_states as gen
    return gen(n, seed=seed).astype("float32")


def cpu_baseline(n: int, seed: int, budget_s: float = 12.0):
    """The C oracle (kind "port") on this box's host cores, OpenMP over drones, bounded sample."""
    import numpy as np
    from oracle import oracle as orc
    orc.build()
    cores = os.cpu_count() or 1
    threads = max(1, min(cores, 64))
    st = orc.synthetic_states(n, seed=seed)
    rng = np.random.Generator(np.random.Philox(seed + 1))
    act = rng.uniform(-1, 1, (n, 4))
    cfg = orc.make_cfg(noise_ratio=0.02, seed=seed)
    orc.step_batch(cfg, st, act, 0, nthreads=threads)           # warms the thread pool
    steps, t0 = 0, time.perf_counter()
    while True:                                                 # bounded sample: ~budget_s of CPU work
        orc.step_batch(cfg, st, act, 1 + steps, nthreads=threads)
        steps += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or steps >= 100000:
            break
    return {"value": n * steps / dt, "unit": "env-steps/s", "cores": threads, "kind": "port",
            "sample": f"oracle/fpv_oracle.c env-step only (no policy), {n} drones x {steps} steps, "
                      f"OpenMP {threads} threads of {cores} host cores, {dt:.1f} s"}


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
    ap.add_argument("--no-split-extra", action="store_true", help="skip the extra bf16x6-policy measurement")
    ap.add_argument("--kernel-launches", type=int, default=200)
    args = ap.parse_args()

    import torch
    import torch.distributed as td
    from fpv_drone_rl_agent_amd import dist as D
    from fpv_drone_rl_agent_amd import ops
    from fpv_drone_rl_agent_amd.engine import FpvEngine
    from fpv_drone_rl_agent_amd.policy import ActorCritic
    from fpv_drone_rl_agent_amd.rollout import RolloutCollector

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    # FPV_DIST_BACKEND=gloo rehearses several ranks on a one-GPU box; the driver's runs use RCCL
    backend = os.environ.get("FPV_DIST_BACKEND", "nccl")
    rank, world, local = D.init_from_env(backend)
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
                    ledger_capacity=n * max(2, T // 8))
    col = RolloutCollector(eng, policy, n_steps=T, use_graph=not args.no_graph)
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

    for _ in range(max(1, math.ceil(W / T))):
        col.collect()
    barrier()
    t0 = time.perf_counter()
    for _ in range(K // T):
        col.collect()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        td.all_reduce(tmax, op=td.ReduceOp.MAX)
        elapsed = float(tmax.item())
    episodes = int(col.episodes["count"])

    # ---- extra (not the headline): the same K steps with the policy GEMMs on the 6-term bf16 split ----
    split_value = None
    if not yaw and not args.no_split_extra:
        col.set_policy_gemm("bf16x6")
        for _ in range(2):
            col.collect()
        barrier()
        t0 = time.perf_counter()
        for _ in range(K // T):
            col.collect()
        barrier()
        el2 = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([el2], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            td.all_reduce(tmax, op=td.ReduceOp.MAX)
            el2 = float(tmax.item())
        split_value = n * world * K / el2
        col.set_policy_gemm("f32")

    # ---- rooflines: kernel-only passes in this process, HIP events on the launch stream (events cannot bracket a
    # node inside the replayed graph).  `roofline` is the kernel that dominates the timed region by time (the fused
    # actor-critic forward, MFMA-bound); `roofline_env_step` is the HBM-bound dynamics kernel BASELINE's "% HBM
    # roofline" refers to, as it runs in the rollout (with the VecNormalize bookkeeping fused in) and stand-alone.
    roofline = roofline_env = roofline_env_plain = None
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

        def pick(times, needle, at_least=0.0):
            """mean duration of the launches of the kernel whose name contains `needle`; at_least drops the short launches
            of the same kernel (the rollout tail runs the policy kernel V-only and over the short done-ledger list)"""
            for k, v in times.items():
                if needle in k.replace(", ", ","):
                    v = [x for x in v if x >= at_least]
                    return float(sum(v) / len(v)) if v else None
            return None

        if yaw:
            us = timed(lambda: eng.step(act))
            roofline = hbm("fpv::yaw_step_kernel<float,true>", us, 458)   # 44 words r+w, 4 B action, 48 B obs, 4 B reward, 2 B flags
        else:
            us_plain = timed(lambda: eng.step(act))
            roofline_env_plain = hbm("fpv::step_kernel<float,true,false>", us_plain, ALG_BYTES_PER_STEP)
            if col.fused:
                io = dict(rms=col.obs_norm.rms, tick=eng.tick_ptr, obs_n=col.buf_obs[0], action=col.buf_act[0],
                          clipped=col.clipped, value=col.buf_val[0], logp=col.buf_logp[0])
                ev_pol = timed(lambda: col.fp.forward(eng.obs, **io))
                ev_fused = timed(lambda: eng.step_fused(col.clipped, done=col._starts[1], ret=col.rew_norm.ret,
                                                        partials=col.partials, gamma=col.gamma, reward=col.raw_rew[0]))
                in_graph = graph_kernel_times()
                us_pol = pick(in_graph, "policy_forward_kernel", at_least=0.75 * ev_pol) or ev_pol
                us_pol_all = pick(in_graph, "policy_forward_kernel")     # what a per-symbol profiler average shows
                us_fused = pick(in_graph, "step_kernel<float,true,true") or ev_fused
                if not (0.6 * ev_pol < us_pol < 1.6 * ev_pol and 0.6 * ev_fused < us_fused < 1.6 * ev_fused):
                    in_graph, us_pol, us_fused = {}, ev_pol, ev_fused     # tracer output implausible: trust the events
                how = "device timestamps of the launches inside the replayed rollout graph" if in_graph else \
                      "HIP events around back-to-back launches"
                roofline_env = hbm("fpv::step_kernel<float,true,true>", us_fused, ALG_BYTES_PER_STEP)
                roofline_env.update(measured=how, us_per_launch_hip_events_back_to_back=round(ev_fused, 2))
                tf = POLICY_FLOP_PER_ROW * n / us_pol / 1e6
                roofline = {"bound": "mfma", "kernel": "fpv::policy_forward_kernel", "achieved": round(tf, 1),
                            "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / MFMA_F32_PEAK_TFLOPS, 4),
                            "traffic": traffic_of("fpv::policy_forward_kernel"), "us_per_launch": round(us_pol, 2),
                            "launches": L, "algorithmic_flop_per_launch": POLICY_FLOP_PER_ROW * n, "dtype": "f32",
                            "measured": how, "us_per_launch_hip_events_back_to_back": round(ev_pol, 2)}
                roofline["note"] = ("peak is the nominal dense fp32 MFMA rate; on operands with real mantissas the instruction "
                                    "itself sustains 115-127 TFLOP/s on this chip (tools/mfma_peak, profiles/r01_mfma_ceiling.txt)")
                if in_graph and us_pol_all:
                    # rocprofv3 --stats averages over ALL launches of the symbol: per rollout of n_steps that is n_steps
                    # full launches plus the tail's V-only launch and the short done-ledger launch
                    roofline["us_per_launch_all_launches_of_the_symbol"] = round(us_pol_all, 2)
            else:
                roofline = roofline_env_plain

    if rank == 0:
        out = {
            "metric": "env_steps_per_sec", "value": n * world * K / elapsed, "unit": "env-steps/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (random-init policy, seeded random-init drone states, motor noise on)",
            "config": {"workload": ("BASELINE.json configs[4]: yaw-setpoint task with domain-randomised mass/drag, on-device "
                                    "PPO rollout (torch policy 12->1, HIP env-step, VecNormalize) + GAE + all-gather" if yaw else
                                    "BASELINE.json configs[2]: 65 536 hover drones per GPU, on-device PPO rollout "
                                    "(two kernels per step: fp32-MFMA policy forward + sample with the VecNormalize statistics merge in its "
                                    "prologue, HIP env-step + VecNormalize bookkeeping; done-ledger timeout bootstrap, GAE) "
                                    "+ advantage-stat all-gather every n_steps"),
                       "drones_per_gpu": n, "total_drones": n * world, "n_steps": T,
                       "hip_graph": not args.no_graph, "policy": "MlpPolicy pi/vf [128,128] tanh" + ("" if yaw else " (39 049 params), exact-fp32 MFMA"),
                       "episodes_in_last_rollout": episodes},
            "roofline": roofline,
        }
        if roofline_env is not None:
            out["roofline_env_step"] = roofline_env
            out["roofline_env_step_unfused"] = roofline_env_plain
        if split_value is not None:
            out["extra"] = {"value_with_policy_gemm_bf16x6": split_value,
                            "note": "same workload with the policy's hidden GEMMs on a 6-term bf16 split (fp32-level error, "
                                    "tests/test_policy_gpu.py); the headline value uses exact-fp32 MFMA"}
        if world == 1 and not args.no_cpu_baseline and not yaw:
            out["cpu_baseline"] = cpu_baseline(n, seed=0)
        print(json.dumps(out), flush=True)
    barrier()
    eng.close()
    if world > 1:
        td.destroy_process_group()


if __name__ == "__main__":
    main()
