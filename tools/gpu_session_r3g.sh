#!/bin/bash
# round 3, session g: the status words written by the ledger scan itself (no D2H copy behind the rollout), 16-step GAE load blocks — tests, the driver-shape line, the boundary between rollouts from the kernel trace
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3g; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_rollstep_gpu.py tests/test_rollout_ops_gpu.py tests/test_facades_gpu.py tests/test_training_gpu.py tests/test_c_host_gpu.py tests/test_policy_gpu.py -x -q -m gpu > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
tail -2 $O/tests.txt
for i in 1 2 3; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-mode --no-split-extra > $O/s20_$i.json 2> $O/err.txt
done
python bench.py --no-cpu-baseline --no-host-mode --no-split-extra > $O/def.json 2>> $O/err.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 20 --warmup 5 --repeats 12 --no-cpu-baseline --no-split-extra --no-host-mode > $O/prof.log 2>&1
f=$(ls $O/prof/*/*kernel_trace.csv | head -1); python tools/trace_gaps.py $f 1 > $O/boundary_s20.txt
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +6M -delete
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3g/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"] * 1e3, 2))
PY
cat $O/boundary_s20.txt | cut -c1-2500
