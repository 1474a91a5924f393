"""What one GPU can rehearse of the multi-GPU path: a single-rank RCCL process group (communicator set-up, the watchdog thread,
a real ncclAllGather of the collector's payload on a side stream) alive WHILE the collector captures and replays its HIP graphs —
in both capture modes.  The N > 1 job itself is the driver's."""
import json
import os
import sys

import torch
import torch.distributed as td

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29531")
torch.cuda.set_device(0)
td.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from fpv_drone_rl_agent_amd import dist as D  # noqa: E402
from fpv_drone_rl_agent_amd.engine import FpvEngine  # noqa: E402
from fpv_drone_rl_agent_amd.policy import ActorCritic  # noqa: E402
from fpv_drone_rl_agent_amd.rollout import RolloutCollector  # noqa: E402

out = {}
x = torch.arange(47, dtype=torch.float64, device="cuda")
g = torch.empty(47, dtype=torch.float64, device="cuda")
td.all_gather_into_tensor(g, x)                       # brings the communicator (and its watchdog) up
torch.cuda.synchronize()
out["allgather_ok"] = bool(torch.equal(g, x))
side = torch.cuda.Stream()
for mode in ("global", "thread_local"):
    eng = FpvEngine(4096, 0, seed=1)
    col = RolloutCollector(eng, ActorCritic().cuda(), n_steps=8)
    col.start()
    col._capture_mode = mode
    ok = True
    try:
        for _ in range(6):
            b = col.collect()
            with torch.cuda.stream(side):             # a collective on a side stream between two graph launches, as with N > 1
                side.wait_stream(torch.cuda.current_stream())
                td.all_gather_into_tensor(g, torch.cat([col.stats, col.obs_norm.rms, col.rew_norm.rms]))
        torch.cuda.synchronize()
        ok = bool(torch.isfinite(b["advantages"]).all()) and col._graphs[0] is not None and col._graphs[1] is not None
    except Exception as exc:  # noqa: BLE001
        ok = repr(exc)
    out[f"capture_{mode}"] = ok
    out[f"episodes_{mode}"] = col.episodes["count"] if ok is True else None
    eng.close()
# ---- the N > 1 arrangement of the collector itself with G = 1 (FPV_REHEARSE_RANKS=1): side stream, the REAL collective on the
# payload, advantage statistics from the gathered row, base snapshots and rank merge of the VecNormalize statistics — against the
# one-rank arrangement from the same seeds.  One rank's batch merged into the base is the same statistics up to fp64 rounding.
def run(rehearse):
    os.environ["FPV_REHEARSE_RANKS"] = "1" if rehearse else "0"
    torch.manual_seed(3)
    eng = FpvEngine(16384, 0, seed=2)
    col = RolloutCollector(eng, ActorCritic(log_std_init=-1.0).cuda(), n_steps=16, policy_seed=5)
    col.start()
    res = []
    for _ in range(4):
        b = col.collect()
        col.wait_stats()
        torch.cuda.synchronize()
        res.append({"adv": b["advantages"].clone(), "obs": b["obs"].clone(), "mean": float(b["adv_mean"]), "std": float(b["adv_std"]),
                    "obs_rms": col.obs_norm.rms.clone(), "ret_rms": col.rew_norm.rms.clone(), "episodes": col.episodes["count"]})
    several = D.several_ranks()
    eng.close()
    return res, several


try:
    (a, sa), (b, sb) = run(True), run(False)
    out["rehearsed_path_is_the_multi_rank_one"] = bool(sa and not sb)
    out["episodes_equal"] = all(x["episodes"] == y["episodes"] and x["episodes"] > 0 for x, y in zip(a, b))
    out["advantages_max_abs_diff"] = max(float((x["adv"] - y["adv"]).abs().max()) for x, y in zip(a, b))
    out["obs_max_abs_diff"] = max(float((x["obs"] - y["obs"]).abs().max()) for x, y in zip(a, b))
    out["adv_mean_std_max_rel_diff"] = max(max(abs(x["mean"] - y["mean"]) / max(1e-12, abs(y["mean"])), abs(x["std"] - y["std"]) / y["std"])
                                           for x, y in zip(a, b))
    out["statistics_max_rel_diff"] = max(float(((x[k] - y[k]).abs() / y[k].abs().clamp_min(1e-6)).max()) for x, y in zip(a, b)
                                         for k in ("obs_rms", "ret_rms"))
except Exception as exc:  # noqa: BLE001
    out["multi_rank_path_with_one_rank"] = repr(exc)
td.barrier()
td.destroy_process_group()
print(json.dumps(out))
