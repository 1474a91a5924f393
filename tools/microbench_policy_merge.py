"""Back-to-back timing of fpv_policy_forward with and without the in-launch statistics merge (HIP events)."""
import os, sys, json
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from fpv_drone_rl_agent_amd.policy import ActorCritic
from fpv_drone_rl_agent_amd.ops_policy import FusedPolicy
n = 65536
torch.manual_seed(0)
m = ActorCritic().cuda(); fp = FusedPolicy(m)
obs = torch.randn(n, 20, device="cuda")
bufs = dict(obs_n=torch.empty(n, 20, device="cuda"), action=torch.empty(n, 4, device="cuda"), clipped=torch.empty(n, 4, device="cuda"),
            value=torch.empty(n, device="cuda"), logp=torch.empty(n, device="cuda"))
rms = torch.cat([torch.zeros(20), torch.ones(20), torch.ones(1)]).double().cuda()
rms_b = rms.clone(); ret = torch.tensor([0.0, 1.0, 1.0], dtype=torch.float64, device="cuda"); ret_b = ret.clone()
snap = torch.zeros(44, dtype=torch.float64, device="cuda")
def timed(fn, it=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / it
out = {"plain_us": round(timed(lambda: fp.forward(obs, rms=rms, **bufs)), 2)}
for rows in (256, 512, 1024):
    part = torch.rand(rows, 44, dtype=torch.float64, device="cuda")
    mg = {"partials": part, "count": n, "rms_out": rms_b, "ret_rms": ret, "ret_rms_out": ret_b, "snapshot": snap}
    out[f"merge_{rows}_rows_us"] = round(timed(lambda: fp.forward(obs, rms=rms, merge=mg, **bufs)), 2)
print(json.dumps(out))
