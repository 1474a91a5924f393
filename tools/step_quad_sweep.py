"""fpv_step / fpv_step_fused through step_quad_kernel (four tiles per workgroup, physics + helper wave per tile) against the
default dispatch (wave-pair kernel up to 32 768 drones, one wave per tile beyond), per drone count and detector: the same bits
(checked here on a few steps from the same state), and HIP-event time per launch, median of 5 repeats of L back-to-back launches.
usage: python tools/step_quad_sweep.py [out.json]"""
import json
import statistics
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from fpv_drone_rl_agent_amd import _lib as L_
from fpv_drone_rl_agent_amd.engine import FpvEngine
from fpv_drone_rl_agent_amd.synthetic import synthetic_states

lib = L_.load()
out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/step_quad_sweep.json"
rows = []
for detector in ("pixel", "analytic"):
    for n in (1024, 4096, 16384, 32768, 65536, 131072, 262144, 524288):
        L = 200
        rec = {"detector": detector, "drones": n}
        outs = {}
        for mode in ("default", "quad"):
            lib.fpv_set_quad_range(0, 1 << 40 if mode == "quad" else 0)
            eng = FpvEngine(n, 0, noise_ratio=0.02, detector=detector, seed=3, ledger_capacity=4 * n)
            st = synthetic_states(min(n, 1 << 20), 0).astype(np.float32)
            eng.set_state(torch.from_numpy(np.tile(st, (n // st.shape[0], 1))))
            g = torch.Generator(device="cuda").manual_seed(n)
            act = torch.rand(n, 4, device="cuda", generator=g) * 2 - 1
            done = torch.zeros(n, dtype=torch.uint8, device="cuda")
            ret = torch.zeros(n, device="cuda")
            part = torch.zeros((eng.partial_rows, 44), dtype=torch.float64, device="cuda")
            raw = torch.zeros(n, device="cuda")
            for _ in range(3):                        # bit-identity first: three plain and three fused steps from the same state
                eng.step(act)
            for _ in range(3):
                eng.step_fused(act, done=done, ret=ret, partials=part, gamma=0.99, reward=raw)
            torch.cuda.synchronize()
            outs[mode] = {"obs": eng.obs.clone(), "state": eng.get_state().clone(), "reward": raw.clone(), "done": done.clone(), "ret": ret.clone(),
                          "sums": part.sum(dim=0).clone(), "ledger": eng.ledger_read()["count"]}
            for name, fn in (("fpv_step", lambda: eng.step(act)),
                             ("fpv_step_fused", lambda: eng.step_fused(act, done=done, ret=ret, partials=part, gamma=0.99, reward=raw))):
                for _ in range(10):
                    fn()
                torch.cuda.synchronize()
                ts = []
                for _ in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(L):
                        fn()
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) * 1e3 / L)
                rec[f"{name}_{mode}_us"] = round(statistics.median(ts), 2)
                eng.ledger_clear()
            eng.close()
            del eng
            torch.cuda.empty_cache()
        a, b = outs["default"], outs["quad"]
        rec["same_bits"] = bool(all(torch.equal(a[k], b[k]) for k in ("obs", "state", "reward", "done", "ret")) and a["ledger"] == b["ledger"]
                                and torch.allclose(a["sums"], b["sums"], rtol=1e-12))
        rows.append(rec)
        print(json.dumps(rec), flush=True)
lib.fpv_set_quad_range(0, 0)
json.dump(rows, open(out_path, "w"), indent=1)
