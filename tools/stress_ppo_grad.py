"""Repeat fpv_ppo_grad on identical inputs and compare every gradient bitwise with the first launch: the kernels use no
atomics, so any difference is a hardware/scheduling hazard (cf. the ds_bpermute case in fpv_policy_split.hip).
usage: python tools/stress_ppo_grad.py [launches] [rows]"""
import os, sys, json
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from fpv_drone_rl_agent_amd.ops_policy import FusedPolicy
from fpv_drone_rl_agent_amd.policy import ActorCritic
launches = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
dev = torch.device("cuda:0")
torch.manual_seed(1)
m = ActorCritic(log_std_init=-1.0).to(dev)
with torch.no_grad():
    m.action_net.weight.mul_(30.0)
flat, fgrad = m.flatten_parameters()
fp = FusedPolicy(m)
total = 4 * rows
obs = torch.randn(total, 20, device=dev); act = torch.randn(total, 4, device=dev) * 0.3
lp = torch.randn(total, device=dev) - 2; adv = torch.randn(total, device=dev); ret = torch.randn(total, device=dev)
ws = fp.ppo_workspace(rows); stats = torch.zeros(4, device=dev)
bad = 0
for seed in range(4):
    idx = torch.randperm(total, device=dev, generator=torch.Generator(device=dev).manual_seed(seed))[:rows].contiguous()
    fp.ppo_grad(obs, act, lp, adv, ret, index=idx, stats=stats, workspace=ws)
    ref, ref_stats = fgrad.clone(), stats.clone()
    for k in range(launches // 4):
        fp.ppo_grad(obs, act, lp, adv, ret, index=idx, stats=stats, workspace=ws)
        if not (torch.equal(fgrad, ref) and torch.equal(stats, ref_stats)):
            bad += 1
print(json.dumps({"launches": launches, "rows": rows, "launches_differing_from_the_first": bad}))
