import sys, torch
sys.path.insert(0, "/root/repo")
from fpv_drone_rl_agent_amd.policy import ActorCritic
from fpv_drone_rl_agent_amd.ops_policy import FusedPolicy
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
torch.manual_seed(0)
m = ActorCritic().cuda(); fp = FusedPolicy(m)
obs = torch.randn(n, 20, device="cuda")
bufs = dict(obs_n=torch.empty(n, 20, device="cuda"), action=torch.empty(n, 4, device="cuda"), clipped=torch.empty(n, 4, device="cuda"),
            value=torch.empty(n, device="cuda"), logp=torch.empty(n, device="cuda"))
rms = torch.cat([torch.zeros(20), torch.ones(20), torch.ones(1)]).double().cuda()
for _ in range(6): fp.forward(obs, rms=rms, **bufs)
torch.cuda.synchronize()
