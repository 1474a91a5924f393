#!/bin/bash
# round 2, session o: GPU suite on the final tree, C-host figures, final bench lines (traffic field filled from profiles/kernel_traffic.json)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2o; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
tail -2 $O/tests.txt
make -C examples/c_host > /dev/null 2>&1
./examples/c_host/rollout_host 65536 32 32 > $O/c_host_one.json
./examples/c_host/rollout_host 65536 32 32 2 > $O/c_host_two.json
cat $O/c_host_one.json $O/c_host_two.json
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_s20.json 2> $O/bench_s20.err
python - <<'PY'
import json
for k in ("bench", "bench_s20"):
    d = json.load(open(f"gpurun_out/r2o/{k}.json"))
    r = d["roofline"]
    print(k, round(d["value"]), round(d["ms_per_step"] * 1e3, 2), r["frac"], r["us_per_launch"], r.get("us_per_launch_traced"), r["traffic"], r.get("phases_us"))
PY
