"""fpv_gae at 65 536 drones, T = 20 / 32: HIP-event time per launch without the advantage moments, with them, with mean / std too
(the in-launch reduction costs 2.5-3.6 us of 12-13; the rest does not depend on T in this range: launch + one load round trip +
the store drain)."""
import sys, json, statistics
import torch
sys.path.insert(0, ".")
from fpv_drone_rl_agent_amd import ops
torch.manual_seed(0)
res = {}
for T in (20, 32):
    n = 65536
    r, v = torch.randn(T, n, device="cuda"), torch.randn(T, n, device="cuda")
    s = (torch.rand(T, n, device="cuda") < 0.05).to(torch.uint8)
    lv, ld = torch.randn(n, device="cuda"), torch.zeros(n, dtype=torch.uint8, device="cuda")
    adv, ret = torch.empty_like(r), torch.empty_like(r)
    stats, ms = torch.zeros(3, dtype=torch.float64, device="cuda"), torch.zeros(2, dtype=torch.float64, device="cuda")
    for name, kw in (("no_stats", {}), ("stats", {"stats": stats}), ("stats+mean_std", {"stats": stats, "mean_std": ms})):
        fn = lambda: ops.gae(r, v, s, lv, ld, 0.99, 0.95, advantages=adv, returns=ret, **kw)
        for _ in range(10): fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(100): fn()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 10)
        res[f"T{T}_{name}_us"] = round(statistics.median(ts), 2)
print(json.dumps(res))
