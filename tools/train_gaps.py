"""Where a training iteration's time goes: reads a rocprofv3 kernel_trace.csv of examples/train_hover.py and prints, per iteration
(from one fpv::rollout_begin_kernel to the next), the span, the device-busy time, and the idle time summed by the kernel it precedes
(host launch latency, synchronisations) — plus the busy time by kernel.  Usage: python tools/train_gaps.py <kernel_trace.csv>"""
import csv
import json
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("fpv::", "")[:48]) for r in rows)
starts = [i for i, e in enumerate(ev) if e[2].startswith("rollout_begin_kernel")]
its = []
for a, b in zip(starts[:-1], starts[1:]):
    ks = ev[a:b + 1]
    span = ks[-1][0] - ks[0][0]
    busy = defaultdict(float)
    idle = defaultdict(float)
    for i in range(len(ks) - 1):
        busy[ks[i][2]] += ks[i][1] - ks[i][0]
        g = ks[i + 1][0] - ks[i][1]
        if g > 0:
            idle[ks[i + 1][2]] += g
    its.append((span, busy, idle, len(ks) - 1))
its = [x for x in its if x[3] > 40]            # iterations with an update in them
if not its:
    sys.exit("no full iteration in the trace")
its.sort(key=lambda x: x[0])
span, busy, idle, n = its[len(its) // 2]
top = lambda d, k: {a: round(b / 1e3, 1) for a, b in sorted(d.items(), key=lambda kv: -kv[1])[:k]}
print(json.dumps({"iterations": len(its), "median_iteration": {"kernels": n, "span_us": round(span / 1e3, 1), "busy_us": round(sum(busy.values()) / 1e3, 1),
                  "idle_us": round(sum(idle.values()) / 1e3, 1), "busy_by_kernel_us": top(busy, 10), "idle_in_front_of_us": top(idle, 10)}}, indent=1))
