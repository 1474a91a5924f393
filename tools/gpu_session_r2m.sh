#!/bin/bash
set -e
mkdir -p gpurun_out/r2m
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r2m/tests.txt 2>&1 || { tail -30 gpurun_out/r2m/tests.txt; exit 1; }
tail -2 gpurun_out/r2m/tests.txt
python tools/region_overhead.py 2>/dev/null | tail -1
python bench.py --no-cpu-baseline --no-split-extra > gpurun_out/r2m/bench.json 2> gpurun_out/r2m/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-split-extra > gpurun_out/r2m/bench_s20.json 2> gpurun_out/r2m/bench_s20.err
python - <<'PY'
import json
for k in ("bench", "bench_s20"):
    d = json.load(open(f"gpurun_out/r2m/{k}.json"))
    print(k, round(d["value"]), round(d["ms_per_step"] * 1e3, 2), d["roofline"]["frac"], d["roofline"].get("frac_of_peak_within_policy_phase"))
PY
