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
td.barrier()
td.destroy_process_group()
print(json.dumps(out))
