"""Where a step of the rollout kernels of fpv_rollstep.hip spends its time: s_memrealtime stamps per workgroup
(fpv_debug_rollout_stamps), medians over workgroups (and steps) in microseconds.
SHAPE=step (default): fpv_rollout_step, one launch per step.  SHAPE=rollout: fpv_rollout_steps, the persistent launch."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

if os.environ.get("SHAPE", "step") == "rollout":
    os.environ.setdefault("FPV_EXPERIMENTS", "1")  # the persistent launch lives in libfpv_env_experiments.so
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synthetic_states  # noqa: E402
from fpv_drone_rl_agent_amd import _lib as L  # noqa: E402
from fpv_drone_rl_agent_amd.engine import FpvEngine  # noqa: E402
from fpv_drone_rl_agent_amd.policy import ActorCritic  # noqa: E402
from fpv_drone_rl_agent_amd.rollout import RolloutCollector  # noqa: E402

n, T = int(os.environ.get("DRONES", 65536)), int(os.environ.get("NSTEPS", 32))
torch.manual_seed(0)
eng = FpvEngine(n, 0, noise_ratio=0.02, seed=0, ledger_capacity=n * max(2, T // 8), detector=os.environ.get("DETECTOR", "pixel"))
persistent = os.environ.get("SHAPE", "step") == "rollout"
col = RolloutCollector(eng, ActorCritic().cuda(), n_steps=T, persistent=persistent, one_launch=True)
col.start()
eng.set_state(torch.from_numpy(synthetic_states(n, seed=0)))
for _ in range(4):
    col.collect()
grid = n // 256
stamps = torch.zeros((grid, T, 8) if persistent else (grid, 16), dtype=torch.int64, device="cuda")
L.load().fpv_debug_rollout_stamps(eng._h, C.c_void_p(stamps.data_ptr()))
col._graph = None            # the stamp pointer is a kernel argument: capture again
col.collect()
col.collect()
torch.cuda.synchronize()
L.load().fpv_debug_rollout_stamps(eng._h, None)
s = stamps.cpu().numpy().astype(np.float64) / 100.0          # microseconds
out = {"drones": n, "n_steps": T, "shape": "one launch per rollout" if persistent else "one launch per step"}
if persistent:
    names = ["policy", "env step", "hand-over + restage", "grid barrier + reduction", "statistics update"]
    for k, nm in enumerate(names):
        dt = (s[:, :T - 1, k + 1] - s[:, :T - 1, k]).ravel()
        out[nm] = {"median_us": round(float(np.median(dt)), 2), "p10": round(float(np.percentile(dt, 10)), 2),
                   "p90": round(float(np.percentile(dt, 90)), 2)}
    step = (s[:, 1:, 0] - s[:, :-1, 0]).ravel()
    out["step"] = {"median_us": round(float(np.median(step)), 2)}
    arrive = s[:, :T - 1, 3]
    out["arrival_skew_us"] = round(float(np.median(arrive.max(axis=0) - arrive.min(axis=0))), 2)
else:
    # (the last launch of the last rollout)
    names = ["staging + statistics merge", "policy (normalise, pi, V, sample)", "env step + bookkeeping"]
    for k, nm in enumerate(names):
        dt = s[:, k + 1] - s[:, k]
        out[nm] = {"median_us": round(float(np.median(dt)), 2), "p10": round(float(np.percentile(dt, 10)), 2),
                   "p90": round(float(np.percentile(dt, 90)), 2)}
    # inside the env step, tile-0 physics wave: stamps 4..7 = sub-step pairs done, observation / reward done, barrier 6, barrier 7
    tn = ["exchange + 6 control ticks + 12 physics sub-steps", "euler + camera features + reward", "ledger / reset / staging -> barrier 6",
          "helper: tile store + moment sums -> barrier 7"]
    edges = [s[:, 2], s[:, 4], s[:, 5], s[:, 6], s[:, 7]]
    out["env step, physics wave"] = {nm: round(float(np.median(edges[k + 1] - edges[k])), 2) for k, nm in enumerate(tn)}
    if eng.detector == "pixel":
        # pixel detector (round 4): wave 0 [8] corners projected + border pre-check done (before barrier P), [10] P passed, [11] its
        # lane groups' blobs done, [12] barrier R passed; wave 4 (tile 0's helper) [14] attitude converted, [15] its groups' blobs done
        med = lambda a, b: round(float(np.median(s[:, b] - s[:, a])), 2)
        q = lambda a, b, p: round(float(np.percentile(s[:, b] - s[:, a], p)), 2)
        out["pixel detector, wave 0"] = {"barrier P0 + corner projection + border pre-check": med(4, 8), "barrier P": med(8, 10),
                                         "its lane groups' blobs (spans, contour, arcLength, approxPolyDP)": med(10, 11),
                                         "the same, p90 / max over the workgroups": [q(10, 11, 90), q(10, 11, 100)],
                                         "wait for the workgroup (barrier R)": med(11, 12),
                                         "P passed -> R passed, median / p90 / max": [med(10, 12), q(10, 12, 90), q(10, 12, 100)],
                                         "observation + reward": med(12, 5)}
        out["pixel detector, wave 4"] = {"pairs done -> attitude converted": med(4, 14), "its lane groups' blobs (from P)": med(10, 15),
                                         "the same, p90 / max over the workgroups": [q(10, 15, 90), q(10, 15, 100)]}
    out["workgroup"] = {"median_us": round(float(np.median(s[:, 3] - s[:, 0])), 2)}
    out["launch_first_start_to_last_end_us"] = round(float(s[:, 3].max() - s[:, 0].min()), 2)
    out["start_skew_us"] = round(float(s[:, 0].max() - s[:, 0].min()), 2)
print(json.dumps(out))
