"""How far an fp32 episode drifts from the fp64 one (the reference computes in float64: simulation/hover.py:61, :70).

Two engines — FPV_F32 (the product's default) and FPV_F64 — from reset, the same seed (the same motor-noise draws: Philox is keyed by
drone id and tick, not by precision), the same fixed action sequence (fp32 values, held in fp64 for the second engine), 4 096 drones,
804 agent steps (two full episodes of 402 calls for a drone that never crashes).  Reported: per step the drones whose done masks
differ, the first step at which a drone's masks differ, and per drone the first episode's length and return in both precisions.
Two action sequences: "hover" — a proportional altitude / levelling law evaluated on the fp64 engine's state, its fp32-rounded output
given to BOTH engines (so both see one sequence; the episodes run to the 402-call truncation) — and "random" (uniform actions: the
drones leave the dome or hit the floor within tens of steps).  Writes one JSON record (profiles/r04_fp32_vs_fp64_trajectory.json)."""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from fpv_drone_rl_agent_amd.engine import FpvEngine

N, STEPS = 4096, 804


def run(kind, detector):
    g = torch.Generator(device="cpu").manual_seed(7)
    acts = (torch.rand((STEPS, N, 4), generator=g) * 2 - 1).float()
    jitter = (torch.randn((STEPS, N, 4), generator=g) * 0.01).float().cuda()

    def hover_law(st):
        """st: canonical fp64 state [N, 36] (docs/SPEC.md §2).  Hold 1 m, level, zero yaw rate."""
        z, vz = st[:, 2], st[:, 9]
        qx, qy, qz, qw = st[:, 3], st[:, 4], st[:, 5], st[:, 6]
        roll = torch.atan2(2 * (qy * qz + qw * qx), qw * qw - qx * qx - qy * qy + qz * qz)
        pitch = torch.asin((-2 * (qx * qz - qw * qy)).clamp(-1, 1))
        thr = 0.4952 + 0.15 * (1.0 - z) - 0.12 * vz                 # (hover throttle 0.4952)
        a = torch.stack([-4.0 * roll / 30.0, -4.0 * pitch / 30.0, 0.0 * z, 2.0 * thr - 1.0], dim=1)
        return a.clamp(-1, 1).float()
    e32 = FpvEngine(N, 0, noise_ratio=0.02, seed=3, detector=detector, ledger_capacity=64 * N)
    e64 = FpvEngine(N, 0, noise_ratio=0.02, seed=3, detector=detector, precision="f64", ledger_capacity=64 * N)
    e32.reset(); e64.reset()
    differ_per_step, first_diff = [], np.full(N, -1)
    obs_gap = []
    for t in range(STEPS):
        a = (hover_law(e64.get_state()) + jitter[t]).clamp(-1, 1) if kind == "hover" else acts[t].cuda()
        o32, r32, te32, tr32 = e32.step(a)
        o64, r64, te64, tr64 = e64.step(a.double())
        d32 = (te32 | tr32).cpu().numpy().astype(bool)
        d64 = (te64 | tr64).cpu().numpy().astype(bool)
        dif = d32 != d64
        differ_per_step.append(int(dif.sum()))
        first_diff[(first_diff < 0) & dif] = t
        if t in (0, 9, 49, 99, 199, 399):
            same = first_diff < 0
            gap = (o32.double() - o64).abs()[torch.from_numpy(same).cuda()]
            obs_gap.append({"step": t + 1, "drones_still_in_step": int(same.sum()),
                            "max_abs_obs_difference": float(gap.max()) if gap.numel() else None,
                            "median_of_row_max": float(gap.max(dim=1).values.median()) if gap.numel() else None})
    l32, l64 = e32.ledger_read(), e64.ledger_read()

    def first_episode(led):
        length, ret = np.full(N, -1), np.full(N, np.nan)
        for k in range(len(led["idx"])):                       # (sorted by tag, i.e. by time)
            i = led["idx"][k]
            if length[i] < 0:
                length[i] = led["info"][k] >> 8
                ret[i] = led["ep_return"][k]
        return length, ret

    len32, ret32 = first_episode(l32)
    len64, ret64 = first_episode(l64)
    both = (len32 >= 0) & (len64 >= 0)
    dl, dr = np.abs(len32 - len64)[both], np.abs(ret32 - ret64)[both]
    q = lambda x: {p: float(np.percentile(x, p)) for p in (50, 90, 99, 100)} if len(x) else None
    e32.close(); e64.close()
    return {"actions": kind, "detector": detector, "drones": N, "steps": STEPS,
            "episodes_in_the_ledger": {"f32": int(l32["count"]), "f64": int(l64["count"])},
            "steps_with_any_done_mask_difference": int(np.count_nonzero(differ_per_step)),
            "drone_steps_with_a_done_mask_difference": int(np.sum(differ_per_step)), "of_drone_steps": N * STEPS,
            "drones_whose_masks_never_differ": int((first_diff < 0).sum()),
            "first_step_with_a_difference_quantiles": q(first_diff[first_diff >= 0] + 1),
            "first_episode_length": {"f32_mean": float(len32[both].mean()), "f64_mean": float(len64[both].mean()), "abs_difference_quantiles": q(dl),
                                     "drones_with_equal_length": int((dl == 0).sum()), "of": int(both.sum())},
            "first_episode_return": {"f32_mean": float(ret32[both].mean()), "f64_mean": float(ret64[both].mean()), "abs_difference_quantiles": q(dr)},
            "observation_gap_while_the_masks_still_agree": obs_gap}


out = {"what": "fp32 engine vs fp64 engine from reset: same seed (same motor noise), same fp32-valued actions, auto-reset on; a drone counts "
               "as 'in step' until its done masks first differ (after that the two engines hold different episodes of it)",
       "runs": [run("hover", "pixel"), run("random", "pixel"), run("hover", "analytic")]}
print(json.dumps(out, indent=1))
