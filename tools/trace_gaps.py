"""Where the time of a rollout goes BETWEEN its kernels: reads a rocprofv3 kernel_trace.csv and prints, for the rollouts it finds
(a rollout = the run of kernels from one fpv::rollout_begin_kernel to the next), every kernel's start offset, duration and the idle
gap in front of it, plus the idle time between rollouts (host launch latency + synchronisation when a timed region is one rollout).
Usage: python tools/trace_gaps.py <kernel_trace.csv> [rollouts-to-print]"""
import csv
import json
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
want = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("fpv::", "")) for r in rows))
starts = [i for i, e in enumerate(ev) if e[2].startswith("rollout_begin_kernel")]
out = []
for a, b in zip(starts[:-1], starts[1:]):
    ks = ev[a:b]
    steps = [k for k in ks if k[2].startswith("rollout_step")]
    if not steps:
        continue
    busy = sum(e - s for s, e, _ in ks)
    span = ks[-1][1] - ks[0][0]
    gaps = [ks[i][0] - ks[i - 1][1] for i in range(1, len(ks))]
    step_gaps = [ks[i][0] - ks[i - 1][1] for i in range(1, len(ks)) if ks[i][2].startswith("rollout_step") and ks[i - 1][2].startswith("rollout_step")]
    out.append({"kernels": len(ks), "steps": len(steps), "span_us": span / 1e3, "busy_us": busy / 1e3, "gaps_us": sum(gaps) / 1e3,
                "median_gap_between_steps_us": sorted(step_gaps)[len(step_gaps) // 2] / 1e3 if step_gaps else None,
                "idle_before_next_rollout_us": (ev[b][0] - ks[-1][1]) / 1e3,
                "step_kernel_us": sum(e - s for s, e, _ in steps) / 1e3 / len(steps),
                "tail": [{"kernel": n[:40], "us": (e - s) / 1e3, "gap_before_us": (s - ks[i - 1][1]) / 1e3}
                         for i, (s, e, n) in enumerate(ks) if i and not n.startswith("rollout_step")]})
from collections import Counter
mode = Counter(o["steps"] for o in out).most_common(1)[0][0] if out else 0
full = [o for o in out if o["steps"] == mode and o["kernels"] < o["steps"] + 16]
for o in full[-want:]:
    print(json.dumps(o))
if full:
    med = lambda k: sorted(o[k] for o in full)[len(full) // 2]
    print(json.dumps({"rollouts": len(full), "median_span_us": med("span_us"), "median_busy_us": med("busy_us"), "median_gaps_us": med("gaps_us"),
                      "median_idle_before_next_rollout_us": med("idle_before_next_rollout_us"), "median_step_kernel_us": med("step_kernel_us")}))
