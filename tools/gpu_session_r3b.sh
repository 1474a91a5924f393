#!/bin/bash
# round 3, session b: work-item pixel walk — the parity / bit-identity tests that cover it, phase stamps, bench line
set -e
mkdir -p gpurun_out/r3b
timeout -k 10 900 python -m pytest tests/test_rollstep_gpu.py tests/test_detector_gpu.py tests/test_fullsize_properties_gpu.py tests/test_golden.py tests/test_step_parity_gpu.py -m gpu -x -q > gpurun_out/r3b/pytest.log 2>&1 || { tail -60 gpurun_out/r3b/pytest.log; exit 1; }
tail -3 gpurun_out/r3b/pytest.log
timeout -k 10 300 python tools/rollout_phases.py > gpurun_out/r3b/phases_pixel.json
cat gpurun_out/r3b/phases_pixel.json
timeout -k 10 300 python bench.py --no-cpu-baseline --no-split-extra > gpurun_out/r3b/bench_pixel.json 2> gpurun_out/r3b/bench_pixel.err || { tail -30 gpurun_out/r3b/bench_pixel.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3b/bench_pixel.json").read().strip().splitlines()[-1])
print(round(d["value"]), round(d["ms_per_step"] * 1e3, 2), d["config"]["detector"], d["roofline"].get("phases_us"))
PY
