#!/bin/bash
# round 3, session d: rollout-path tests + both bench shapes
set -e
mkdir -p gpurun_out/r3d
timeout -k 10 900 python -m pytest tests/test_facades_gpu.py tests/test_rollstep_gpu.py tests/test_rollout_ops_gpu.py tests/test_fullsize_properties_gpu.py tests/test_c_host_gpu.py tests/test_training_gpu.py -m gpu -x -q > gpurun_out/r3d/pytest.log 2>&1 || { tail -60 gpurun_out/r3d/pytest.log; exit 1; }
tail -3 gpurun_out/r3d/pytest.log
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-split-extra --no-host-mode > gpurun_out/r3d/bench_s20.json 2> gpurun_out/r3d/bench_s20.err || { tail -30 gpurun_out/r3d/bench_s20.err; exit 1; }
timeout -k 10 300 python bench.py --no-cpu-baseline --no-split-extra --no-host-mode > gpurun_out/r3d/bench_default.json 2> gpurun_out/r3d/bench_default.err || { tail -30 gpurun_out/r3d/bench_default.err; exit 1; }
python - <<'PY'
import json
for f in ("s20", "default"):
    d = json.loads(open(f"gpurun_out/r3d/bench_{f}.json").read().strip().splitlines()[-1])
    print(f, round(d["value"]), round(d["ms_per_step"] * 1e3, 2), d["config"]["detector"], d["roofline"]["frac"], d["roofline"]["us_per_launch"])
PY
