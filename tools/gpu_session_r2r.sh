#!/bin/bash
mkdir -p gpurun_out/r2r
for N in 131072 262144; do DRONES=$N NSTEPS=16 timeout -k 10 300 python tools/rollout_shapes.py 2>&1 | tail -8; done | tee gpurun_out/r2r/shapes_sweep_large.txt
