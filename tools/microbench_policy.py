"""Kernel-only timing of fpv_policy_forward (HIP events on the launch stream)."""
import os, sys, json
import torch
os.environ.setdefault("FPV_EXPERIMENTS", "1")     # 'f32_w16' lives in libfpv_env_experiments.so
sys.path.insert(0, ".")
from fpv_drone_rl_agent_amd.policy import ActorCritic
from fpv_drone_rl_agent_amd.ops_policy import FusedPolicy
torch.manual_seed(0)
m = ActorCritic().cuda()
for gemm, n in [(g, n) for g in ('f32', 'f32_w16', 'bf16x6') for n in (4096, 65536, 524288)]:
    fp = FusedPolicy(m, gemm=gemm)
    obs = torch.randn(n, 20, device="cuda")
    bufs = dict(obs_n=torch.empty(n, 20, device="cuda"), action=torch.empty(n, 4, device="cuda"), clipped=torch.empty(n, 4, device="cuda"),
                value=torch.empty(n, device="cuda"), logp=torch.empty(n, device="cuda"))
    rms = torch.cat([torch.zeros(20), torch.ones(20), torch.ones(1)]).double().cuda()
    for _ in range(10): fp.forward(obs, rms=rms, **bufs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): fp.forward(obs, rms=rms, **bufs)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 10
    print(json.dumps({"gemm": gemm, "n": n, "us": round(us, 2), "TFLOPs": round(n * 77e3 / us / 1e6, 1)}), flush=True)
