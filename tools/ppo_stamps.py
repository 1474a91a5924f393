"""Diagnostic (variants/stamp.so only): s_memrealtime stamps of two workgroups of the one-kernel PPO gradient."""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from fpv_drone_rl_agent_amd.ops_policy import FusedPolicy
from fpv_drone_rl_agent_amd.policy import ActorCritic
rows, total = 65536, 524288
dev = torch.device("cuda:0"); torch.manual_seed(0)
m = ActorCritic(log_std_init=-1.0).to(dev); m.flatten_parameters(); fp = FusedPolicy(m)
obs = torch.randn(total, 20, device=dev); act = torch.randn(total, 4, device=dev) * 0.3
lp = torch.randn(total, device=dev) - 2; adv = torch.randn(total, device=dev); ret = torch.randn(total, device=dev)
idx = torch.randperm(total, device=dev)[:rows].contiguous()
ws = fp.ppo_workspace(rows); stats = torch.zeros(4, device=dev)
for _ in range(5):
    fp.ppo_grad(obs, act, lp, adv, ret, index=idx, stats=stats, workspace=ws, path="one_kernel")
torch.cuda.synchronize()
kTImage = 2 * 36096 + 1172
raw = ws.view(torch.uint8)[kTImage * 4: kTImage * 4 + 8 * 128 * 8].cpu().numpy().view(np.uint64).reshape(8, 128).astype(np.int64)
names = ["L1+L2", "heads,D2", "barrier", "H2 down, backward, D1", "barrier", "head gradient", "barrier", "D2 down + barrier", "A operands", "barrier+H1 down",
         "barrier", "dW2", "barrier", "D1 down", "barrier", "dW1 (to next chunk)"]
for wg in range(2):
    t0 = raw[wg * 4:(wg + 1) * 4, 0].min()
    print("workgroup", wg, "(us from the first wave's start)")
    for w in range(4):
        r = (raw[wg * 4 + w] - t0) / 100.0
        print(" wave", w, "start %.2f image %.2f" % (r[0], r[1]), "| end of chunks %.2f, partials written %.2f" % (r[2], r[3]))
        rr = raw[wg * 4 + w]
        print("   core clock over the chunks: %.3f GHz" % ((rr[122] - rr[121]) / ((rr[2] - rr[1]) * 10.0)))
        for c in range(4):
            ck = 4 + 16 * c
            prev = r[1] if c == 0 else r[ck - 16 + 14]
            print("   chunk", c, " ".join("%.2f" % (r[ck + k] - (prev if k == 0 else r[ck + k - 1])) for k in range(15)),
                  "| of the first: until layer 1 is done %.2f, layer 2 %.2f" % (r[ck + 15] - prev, r[ck] - r[ck + 15]))
print("columns:", "; ".join(names[:15]))
