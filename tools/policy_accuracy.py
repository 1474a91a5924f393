"""Worst error of the fused actor-critic forward (exact-fp32 MFMA kernel) against an fp64 evaluation of the same weights:
action means and values over 65 536 random rows, for the initial weights and for weights grown by 3x (hidden layers) as trained
networks' are.  Run once per library build to compare formulations (tools/ab_variants.sh swaps them)."""
import copy
import json
import sys

import torch

sys.path.insert(0, ".")
from fpv_drone_rl_agent_amd.ops_policy import FusedPolicy
from fpv_drone_rl_agent_amd.policy import ActorCritic

out = {}
for name, grow in (("initial", 1.0), ("hidden_weights_x3", 3.0)):
    torch.manual_seed(0)
    m = ActorCritic().cuda()
    with torch.no_grad():
        for lin in (m.pi[0], m.pi[2], m.vf[0], m.vf[2]):
            lin.weight.mul_(grow); lin.bias.normal_(0, 0.3)
        m.action_net.weight.mul_(30.0); m.value_net.weight.mul_(3.0)        # heads of a trained policy are far from their 0.01-gain init
    k = FusedPolicy(m)
    m64 = copy.deepcopy(m).double()
    worst_a = worst_v = 0.0
    scale_a = scale_v = 0.0
    for trial in range(8):
        obs = torch.randn(65536, 20, device="cuda", generator=torch.Generator(device="cuda").manual_seed(trial)) * 2
        with torch.no_grad():
            mu64, v64 = m64.mean_value(obs.double())
        a = torch.zeros(65536, 4, device="cuda")
        r = k.forward(obs, action=a, deterministic=True)
        worst_a = max(worst_a, float((a.double() - mu64).abs().max())); scale_a = max(scale_a, float(mu64.abs().max()))
        worst_v = max(worst_v, float((r["value"].double() - v64).abs().max())); scale_v = max(scale_v, float(v64.abs().max()))
    out[name] = {"action_mean_worst_abs_err": worst_a, "action_mean_max_abs": scale_a, "value_worst_abs_err": worst_v, "value_max_abs": scale_v}
print(json.dumps(out))
