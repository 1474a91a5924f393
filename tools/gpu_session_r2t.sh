#!/bin/bash
set -e
mkdir -p gpurun_out/r2t
timeout -k 10 600 python -m pytest tests/test_detector_gpu.py tests/test_step_parity_gpu.py tests/test_rollstep_gpu.py -x -q > gpurun_out/r2t/tests.txt 2>&1 || { tail -30 gpurun_out/r2t/tests.txt; exit 1; }
tail -2 gpurun_out/r2t/tests.txt
python bench.py --detector pixel --no-cpu-baseline --no-split-extra > gpurun_out/r2t/bench_pixel.json 2> gpurun_out/r2t/bench_pixel.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r2t/bench_pixel.json"))
print(round(d["value"]), round(d["ms_per_step"] * 1e3, 2), d["roofline"].get("phases_us"))
PY
