#!/bin/bash
# round 3, session c: full GPU suite + the driver's bench shape + the default shape
set -e
mkdir -p gpurun_out/r3c
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3c/pytest.log 2>&1 || { tail -60 gpurun_out/r3c/pytest.log; exit 1; }
tail -3 gpurun_out/r3c/pytest.log
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3c/bench_s20.json 2> gpurun_out/r3c/bench_s20.err || { tail -30 gpurun_out/r3c/bench_s20.err; exit 1; }
timeout -k 10 300 python bench.py > gpurun_out/r3c/bench_default.json 2> gpurun_out/r3c/bench_default.err || { tail -30 gpurun_out/r3c/bench_default.err; exit 1; }
python - <<'PY'
import json
for f in ("s20", "default"):
    d = json.loads(open(f"gpurun_out/r3c/bench_{f}.json").read().strip().splitlines()[-1])
    print(f, round(d["value"]), round(d["ms_per_step"] * 1e3, 2), d["config"]["detector"], d["roofline"]["frac"], d["roofline"]["us_per_launch"], d.get("extra", {}).get("cfg1_host_mode", {}).get("value"))
PY
