#!/bin/bash
# round 3, session a: the pixel detector as the default — full GPU suite, then the bench line with both detectors
set -e
mkdir -p gpurun_out/r3a
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3a/pytest.log 2>&1 || { tail -60 gpurun_out/r3a/pytest.log; exit 1; }
tail -3 gpurun_out/r3a/pytest.log
timeout -k 10 300 python bench.py --no-cpu-baseline --no-split-extra > gpurun_out/r3a/bench_pixel.json 2> gpurun_out/r3a/bench_pixel.err || { tail -30 gpurun_out/r3a/bench_pixel.err; exit 1; }
timeout -k 10 300 python bench.py --no-cpu-baseline --no-split-extra --detector analytic > gpurun_out/r3a/bench_analytic.json 2> gpurun_out/r3a/bench_analytic.err || { tail -30 gpurun_out/r3a/bench_analytic.err; exit 1; }
python - <<'PY'
import json
for f in ("pixel", "analytic"):
    d = json.loads(open(f"gpurun_out/r3a/bench_{f}.json").read().strip().splitlines()[-1])
    print(f, round(d["value"]), round(d["ms_per_step"] * 1e3, 2), d["config"]["detector"], d["roofline"].get("phases_us"))
PY
