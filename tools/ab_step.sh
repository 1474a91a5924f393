#!/bin/bash
# A/B of library builds on the stand-alone env step inside ONE gpurun call: variants/<name>.so are selected in turn
# (FPV_LIB_VARIANT; "product" = the product library) and tools/pmc_step.py-like timing is taken (HIP events, 65 536 drones, synthetic states).  Usage: bash tools/ab_step.sh A B ...
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = product ]; then unset FPV_LIB_VARIANT; else export FPV_LIB_VARIANT=$v; fi
  python - "$v" <<'PY'
import sys, numpy as np, torch
sys.path.insert(0, ".")
from fpv_drone_rl_agent_amd.engine import FpvEngine
from fpv_drone_rl_agent_amd.synthetic import synthetic_states
n = 65536
eng = FpvEngine(n, 0, noise_ratio=0.02)
st = torch.from_numpy(synthetic_states(n, 0).astype(np.float32))
act = torch.rand(n, 4, device="cuda") * 2 - 1
out = []
for rep in range(3):
    eng.set_state(st)
    for _ in range(5): eng.step(act)
    eng.set_state(st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40): eng.step(act)
    e1.record(); torch.cuda.synchronize()
    out.append(round(e0.elapsed_time(e1) * 1e3 / 40, 2))
print(sys.argv[1], out)
eng.close()
PY
done
