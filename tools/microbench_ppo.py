"""Time fpv_ppo_grad (+ fpv_adam_clip_step) on one minibatch shape; run under rocprofv3 --kernel-trace --stats for the
per-kernel split.  usage: python tools/microbench_ppo.py [rows] [total] [iters] [storage f32|bf16] [path one_kernel|two_kernels|two_kernels_bf16x6]"""
import sys, json
import torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from fpv_drone_rl_agent_amd.ops_policy import FusedPolicy
from fpv_drone_rl_agent_amd.ops import FusedAdam
from fpv_drone_rl_agent_amd.policy import ActorCritic

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
total = int(sys.argv[2]) if len(sys.argv) > 2 else 8 * rows
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 50
storage = sys.argv[4] if len(sys.argv) > 4 else "f32"
path = sys.argv[5] if len(sys.argv) > 5 else None
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = ActorCritic(log_std_init=-1.0).to(dev)
flat, fgrad = m.flatten_parameters()
fp = FusedPolicy(m)
opt = FusedAdam(flat)
obs = torch.randn(total, 20, device=dev); act = torch.randn(total, 4, device=dev) * 0.3
lp = torch.randn(total, device=dev) - 2; adv = torch.randn(total, device=dev); ret = torch.randn(total, device=dev)
idx = torch.randperm(total, device=dev)[:rows].contiguous()
ws = fp.ppo_workspace(rows); stats = torch.zeros(4, device=dev)
def step():
    fp.ppo_grad(obs, act, lp, adv, ret, index=idx, stats=stats, workspace=ws, storage=storage, path=path)
    opt.step(fgrad)
for _ in range(5): step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters): step()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / iters
flop = rows * 2 * (2 * (24 * 128 + 128 * 128) + 2 * 128 * 128 + 2 * (128 * 128 + 128 * 32 + 32 * 128))   # fwd, bwd-data, wgrad, per net pair
print(json.dumps({"rows": rows, "storage": storage, "path": path or ("one_kernel" if storage == "f32" else "two_kernels"), "us_per_minibatch": round(us, 1), "rows_per_s": round(rows / us * 1e6),
                  "mfma_TFLOPs": round(flop / us / 1e6, 1)}))
