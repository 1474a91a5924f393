"""BASELINE configs[1]: 4 096 drones, HIP dynamics kernel, policy on the CPU — the SB3 host (numpy) mode of
FpvVecEnv, i.e. one H2D of actions and one D2H of obs/reward/flags per step.  PCIe-inclusive rate."""
import json, sys, time
import numpy as np
sys.path.insert(0, ".")
from fpv_drone_rl_agent_amd.vec_env import FpvVecEnv, VecNormalize

for n in (16, 4096, 65536):
    env = VecNormalize(FpvVecEnv(n))
    obs = env.reset()
    rng = np.random.default_rng(0)
    acts = rng.uniform(-1, 1, (64, n, 4)).astype(np.float32)
    for t in range(20):
        env.step(acts[t % 64])
    t0 = time.perf_counter()
    K = 300
    for t in range(K):
        obs, r, d, info = env.step(acts[t % 64])
    dt = time.perf_counter() - t0
    print(json.dumps({"mode": "host numpy + VecNormalize", "n": n, "us_per_step": round(dt / K * 1e6, 1), "steps_per_s": round(n * K / dt)}), flush=True)
    env.close()
