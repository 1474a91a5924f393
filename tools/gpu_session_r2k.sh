#!/bin/bash
set -e
mkdir -p gpurun_out/r2k
timeout -k 10 300 python tools/rollout_phases.py > gpurun_out/r2k/phases_step.txt 2>&1 || { tail -20 gpurun_out/r2k/phases_step.txt; exit 1; }
cat gpurun_out/r2k/phases_step.txt
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r2k/tests.log 2>&1 || { tail -40 gpurun_out/r2k/tests.log; exit 1; }
tail -3 gpurun_out/r2k/tests.log
timeout -k 10 300 python tools/rollout_shapes.py > gpurun_out/r2k/shapes.txt 2>&1 || { tail -20 gpurun_out/r2k/shapes.txt; exit 1; }
cat gpurun_out/r2k/shapes.txt
