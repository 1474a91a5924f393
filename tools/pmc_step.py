"""Launch only the env-step kernel a few times (for rocprofv3 --pmc passes)."""
import sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from fpv_drone_rl_agent_amd.engine import FpvEngine
from fpv_drone_rl_agent_amd.synthetic import synthetic_states
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 10
eng = FpvEngine(n, 0, noise_ratio=0.02)
eng.set_state(torch.from_numpy(synthetic_states(n, 0).astype(np.float32)))
act = torch.rand(n, 4, device="cuda") * 2 - 1
for _ in range(launches):
    eng.step(act)
torch.cuda.synchronize()
eng.close()
