"""Fixed cost of ONE rollout timed on its own (the driver's `--steps 20` shape): [synchronize, collect(), synchronize]
against the steady-state time of back-to-back rollouts, and the floor of an (almost) empty graph replay."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synthetic_states  # noqa: E402
from fpv_drone_rl_agent_amd import _lib as L  # noqa: E402
from fpv_drone_rl_agent_amd.engine import FpvEngine  # noqa: E402
from fpv_drone_rl_agent_amd.policy import ActorCritic  # noqa: E402
from fpv_drone_rl_agent_amd.rollout import RolloutCollector  # noqa: E402

n, T = int(os.environ.get("DRONES", 65536)), int(os.environ.get("NSTEPS", 20))
torch.manual_seed(0)
eng = FpvEngine(n, 0, noise_ratio=0.02, seed=0, ledger_capacity=n * max(2, T // 8))
col = RolloutCollector(eng, ActorCritic().cuda(), n_steps=T)
col.start()
eng.set_state(torch.from_numpy(synthetic_states(n, seed=0)))
for _ in range(4):
    col.collect()


def region(fn, reps=31):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e6)
    return float(np.median(ts))


single = region(col.collect)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(32):
    col.collect()
torch.cuda.synchronize()
steady = (time.perf_counter() - t0) / 32 * 1e6
graph_only = region(col._graph.replay)
# floor: a graph of one empty kernel
lib = L.load()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    lib.fpv_noop(None)
torch.cuda.synchronize()
with torch.cuda.graph(g):
    import ctypes as C
    lib.fpv_noop(C.c_void_p(torch.cuda.current_stream().cuda_stream))
floor = region(g.replay)
print(json.dumps({"drones": n, "n_steps": T, "single_rollout_region_us": round(single, 1), "steady_state_rollout_us": round(steady, 1),
                  "graph_replay_only_region_us": round(graph_only, 1), "empty_graph_region_us": round(floor, 1),
                  "per_step_single": round(single / T, 2), "per_step_steady": round(steady / T, 2)}))
