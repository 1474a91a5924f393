"""Per-step time of the rollout at 65 536 drones by launch shape: two kernels per step, one kernel per step
(fpv_rollout_step), one launch per rollout (fpv_rollout_steps).  HIP-graph replay, median of 7 timed blocks."""
import json
import os
import sys
import time

import numpy as np
import torch

os.environ.setdefault("FPV_EXPERIMENTS", "1")     # the one-launch-per-rollout shape lives in libfpv_env_experiments.so
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synthetic_states  # noqa: E402
from fpv_drone_rl_agent_amd.engine import FpvEngine  # noqa: E402
from fpv_drone_rl_agent_amd.policy import ActorCritic  # noqa: E402
from fpv_drone_rl_agent_amd.rollout import RolloutCollector  # noqa: E402

n, T = int(os.environ.get("DRONES", 65536)), int(os.environ.get("NSTEPS", 32))
shapes = {"two_kernels_per_step": dict(one_launch=False, persistent=False),
          "one_kernel_per_step": dict(one_launch=True, persistent=False),
          "one_launch_per_rollout": dict(one_launch=True, persistent=True)}
for name, kw in shapes.items():
    if kw["persistent"] and n // 256 > torch.cuda.get_device_properties(0).multi_processor_count:
        continue                      # the persistent launch needs one workgroup per CU at most
    torch.manual_seed(0)
    eng = FpvEngine(n, 0, noise_ratio=0.02, seed=0, ledger_capacity=n * max(2, T // 8))
    col = RolloutCollector(eng, ActorCritic().cuda(), n_steps=T, **kw)
    col.start()
    eng.set_state(torch.from_numpy(synthetic_states(n, seed=0)))
    for _ in range(4):
        col.collect()
    ts = []
    for _ in range(7):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            col.collect()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / (8 * T) * 1e6)
    _ = col.episodes
    print(json.dumps({"shape": name, "drones": n, "n_steps": T, "us_per_step_median": round(float(np.median(ts)), 2),
                      "us_per_step_all": [round(x, 2) for x in sorted(ts)], "env_steps_per_s": round(n / np.median(ts) * 1e6)}))
    eng.close()
