#!/bin/bash
# round 2, session q (final of the round): the one-kernel-per-step default measured end to end — GPU suite, bench lines (default, the driver's
# --steps 20 shape, the two-kernel shape, yaw, pixel detector), rocprofv3 kernel statistics and the PMC traffic passes
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2q; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
tail -2 $O/tests.txt
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_s20.json 2> $O/bench_s20.err
python bench.py --two-launch --no-cpu-baseline > $O/bench_two.json 2> $O/bench_two.err
python bench.py --task yaw --no-cpu-baseline > $O/bench_yaw.json 2> $O/bench_yaw.err
python bench.py --detector pixel --no-cpu-baseline --no-split-extra > $O/bench_pixel.json 2> $O/bench_pixel.err
python tools/rollout_phases.py > $O/phases_step.json 2> /dev/null
SHAPE=rollout python tools/rollout_phases.py > $O/phases_rollout.json 2> /dev/null
python tools/rollout_shapes.py > $O/shapes.txt 2> /dev/null
python tools/region_overhead.py > $O/region.json 2> /dev/null
make -C examples/c_host > /dev/null 2>&1
./examples/c_host/rollout_host 65536 32 32 > $O/c_host_one.json
./examples/c_host/rollout_host 65536 32 32 2 > $O/c_host_two.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_hover -- python3 bench.py --steps 256 --warmup 32 --repeats 2 --no-cpu-baseline --no-split-extra > $O/prof_hover.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 128 --warmup 32 --repeats 1 --no-cpu-baseline --no-split-extra --kernel-launches 20 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --steps 128 --warmup 32 --repeats 1 --no-cpu-baseline --no-split-extra --kernel-launches 20 > $O/pmc_write.log 2>&1
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +3M -delete
python - <<'PY'
import json
for k in ("bench", "bench_s20", "bench_two", "bench_yaw", "bench_pixel"):
    d = json.load(open(f"gpurun_out/r2q/{k}.json"))
    print(k, round(d["value"]), round(d["ms_per_step"] * 1e3, 2), d["roofline"]["frac"], d.get("cpu_baseline", {}).get("value"))
PY
du -sh $O
