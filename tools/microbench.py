"""Kernel-only timing of fpv_step at several N (HIP events on the launch stream)."""
import sys, json
import numpy as np, torch
sys.path.insert(0, ".")
from fpv_drone_rl_agent_amd.engine import FpvEngine

def run(n, steps=200, warm=20, noise=0.02, fused=False):
    eng = FpvEngine(n, 0, noise_ratio=noise)
    rng = np.random.default_rng(0)
    st = np.zeros((n, 36), np.float32)
    st[:, 0:3] = rng.uniform(-1, 1, (n, 3)); st[:, 2] += 1.5
    st[:, 6] = 1.0; st[:, 13:17] = 0.5; st[:, 34] = rng.integers(0, 400, n)
    eng.set_state(torch.from_numpy(st))
    act = torch.rand(n, 4, device="cuda") * 2 - 1
    ret = torch.zeros(n, device='cuda'); done = torch.zeros(n, dtype=torch.uint8, device='cuda')
    part = torch.zeros((eng.tiles, 44), dtype=torch.float64, device='cuda')
    step = (lambda a: eng.step_fused(a, done=done, ret=ret, partials=part)) if fused else eng.step
    for _ in range(warm): step(act)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps): step(act)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / steps
    eng.close()
    return {"fused": fused, "n": n, "us_per_step": round(us, 2), "steps_per_s": round(n / us * 1e6), "alg_GBps": round(382 * n / us / 1e3, 1)}

if __name__ == "__main__":
    for n in [4096, 65536, 524288, 4194304]:
        print(json.dumps(run(n)), flush=True)
    for n in [65536, 4194304]:
        print(json.dumps(run(n, fused=True)), flush=True)
