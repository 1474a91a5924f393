#!/bin/bash
# round 2, session k: first run of the one-launch rollout step (fpv_rollstep.hip): parity tests, then both bench shapes
set -e
mkdir -p gpurun_out/r2k
timeout -k 10 600 python -m pytest tests/test_rollstep_gpu.py -x -q > gpurun_out/r2k/tests.log 2>&1 || { tail -30 gpurun_out/r2k/tests.log; exit 1; }
tail -3 gpurun_out/r2k/tests.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-split-extra > gpurun_out/r2k/bench_one.json 2> gpurun_out/r2k/bench_one.err
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-split-extra --two-launch > gpurun_out/r2k/bench_two.json 2> gpurun_out/r2k/bench_two.err
python - <<'PY'
import json
for k in ("one", "two"):
    d = json.load(open(f"gpurun_out/r2k/bench_{k}.json"))
    print(k, d["value"], d["ms_per_step"], d["repeats_ms_per_step"])
PY
