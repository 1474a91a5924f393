"""Stress both policy GEMM modes: many launches, every row checked against an fp64 evaluation.
Building with FPV_EXTRA_HIPCC_FLAGS=-DFPV_HALFSUM_BPERMUTE brings back the ds_bpermute-after-v_pk_fma hazard that the
kernels avoid (fpv_internal.h half_sum): the bf16x6 line then reports ~300 bad rows per 10^6."""
import sys, copy, torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from fpv_drone_rl_agent_amd.policy import ActorCritic
from fpv_drone_rl_agent_amd.ops_policy import FusedPolicy
torch.manual_seed(5)
m = ActorCritic().cuda()
with torch.no_grad():
    for p in m.parameters():
        if p.dim() == 1: p.add_(torch.randn_like(p) * 0.3)
    m.action_net.weight.mul_(30.0)
m64 = copy.deepcopy(m).double()
n = 65536
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for gemm in ("bf16x6", "f32"):
    k = FusedPolicy(m, gemm=gemm)
    bad_total, worst = 0, 0.0
    for trial in range(trials):
        torch.manual_seed(1000 + trial)
        obs = torch.randn(n, 20, device="cuda") * 2
        with torch.no_grad(): mu64, v64 = m64.mean_value(obs.double())
        a = torch.zeros(n, 4, device="cuda")
        r = k.forward(obs, action=a, deterministic=True)
        e = torch.maximum((a.double() - mu64).abs().max(1).values, (r["value"].double() - v64).abs())
        bad_total += int((e > 1e-5).sum()); worst = max(worst, float(e.max()))
    print(gemm, "trials", trials, "bad rows", bad_total, "worst err", worst, flush=True)
