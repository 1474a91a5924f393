#!/bin/bash
# two ranks on the one GPU over gloo: the multi-rank launch contract with the one-kernel default (rehearsal of the driver's N > 1 runs)
set -e
mkdir -p gpurun_out/r2s
FPV_DIST_BACKEND=gloo timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 64 --warmup 32 --no-cpu-baseline --no-split-extra --drones 32768 > gpurun_out/r2s/bench_2rank.json 2> gpurun_out/r2s/bench_2rank.err || { tail -30 gpurun_out/r2s/bench_2rank.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2s/bench_2rank.json").read().strip().splitlines()[-1])
print(d["n_gpus"], round(d["value"]), round(d["ms_per_step"] * 1e3, 2), d["config"]["kernels_per_step"], d["roofline"]["frac"])
PY
