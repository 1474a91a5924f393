"""Env-step kernel over drone counts (SURVEY.md §8(d) reporting item): per N in {4 096, 65 536, 524 288, 4 M, 16 M} the
plain fpv_step and the fused fpv_step_fused, timed with HIP events on the launch stream — median of 5 repeats of L
back-to-back plain launches, and of 5 replays of a HIP graph holding L launches.  Algorithmic bytes = 382 B per env-step
(hover).  usage: python tools/step_sweep.py out.json [detector]"""
import json
import statistics
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from fpv_drone_rl_agent_amd.engine import FpvEngine
from fpv_drone_rl_agent_amd.synthetic import synthetic_states

out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/step_sweep.json"
detector = sys.argv[2] if len(sys.argv) > 2 else "analytic"
ALG = 382
rows = []
for n in (4096, 65536, 524288, 4 * 1024 * 1024, 16 * 1024 * 1024):
    L = 200 if n <= 524288 else (40 if n <= 4 * 1024 * 1024 else 12)
    eng = FpvEngine(n, 0, noise_ratio=0.02, detector=detector)
    st = synthetic_states(min(n, 1 << 20), 0).astype(np.float32)
    eng.set_state(torch.from_numpy(np.tile(st, (n // st.shape[0], 1))))
    act = torch.rand(n, 4, device="cuda") * 2 - 1
    done = torch.zeros(n, dtype=torch.uint8, device="cuda")
    ret = torch.zeros(n, device="cuda")
    part = torch.zeros((eng.partial_rows, 44), dtype=torch.float64, device="cuda")
    raw = torch.zeros(n, device="cuda")
    variants = {"fpv_step": lambda: eng.step(act),
                "fpv_step_fused": lambda: eng.step_fused(act, done=done, ret=ret, partials=part, gamma=0.99, reward=raw)}
    rec = {"drones": n, "launches_per_repeat": L, "algorithmic_bytes_per_launch": ALG * n}
    for name, fn in variants.items():
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        plain = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(L):
                fn()
            e1.record()
            torch.cuda.synchronize()
            plain.append(e0.elapsed_time(e1) * 1e3 / L)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(L):
                fn()
        g.replay()
        torch.cuda.synchronize()
        graph = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            graph.append(e0.elapsed_time(e1) * 1e3 / L)
        del g
        eng.ledger_clear()
        up, ug = statistics.median(plain), statistics.median(graph)
        rec[name] = {"us_per_launch_plain_median_of_5": round(up, 3), "us_per_launch_graph_median_of_5": round(ug, 3),
                     "env_steps_per_s": round(n / min(up, ug) * 1e6), "algorithmic_GBps": round(ALG * n / min(up, ug) / 1e3, 1),
                     "frac_of_8TBps": round(ALG * n / min(up, ug) / 1e3 / 8000.0, 4)}
    rows.append(rec)
    print(json.dumps(rec), flush=True)
    eng.close()
    del eng, act, done, ret, part, raw
    torch.cuda.empty_cache()
json.dump({"detector": detector, "rows": rows}, open(out_path, "w"), indent=1)
