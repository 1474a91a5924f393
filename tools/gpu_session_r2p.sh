#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2p; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
tail -2 $O/tests.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 20 --warmup 5 --repeats 9 --no-cpu-baseline --no-split-extra --kernel-launches 5 > $O/prof.log 2>&1
find $O -name "*.db" -delete
python - <<'PY'
import csv, glob
f = sorted(glob.glob("gpurun_out/r2p/prof/**/*kernel_stats.csv", recursive=True))[-1]
for r in list(csv.DictReader(open(f)))[:14]:
    print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.2f} min {float(r["MinNs"])/1e3:8.2f}')
PY
timeout -k 10 300 python tools/region_overhead.py 2>/dev/null | tail -1
