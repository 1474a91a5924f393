#!/bin/bash
# round 2, session l: full GPU suite, bench lines of the one-kernel-per-step default (driver shape --steps 20 too)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2l; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
python bench.py --steps 20 --warmup 5 > $O/bench_s20.json 2> $O/bench_s20.err
python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
for k in ("s20", "default"):
    d = json.load(open(f"gpurun_out/r2l/bench_{k}.json"))
    print(k, d["value"], d["ms_per_step"], d["repeats_ms_per_step"], d["roofline"]["frac"], d["roofline"].get("phases_us"), d.get("extra"))
PY
