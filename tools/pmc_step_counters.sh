#!/bin/bash
# SQ counters of the stand-alone env-step kernel (tools/pmc_step.py <drones> 10): two rocprofv3 --pmc passes, per-launch means of the
# kernel whose name contains $2 (default step_quad_kernel) -> gpurun_out/pmc_step/summary.json
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
N=${1:-65536}; K=${2:-step_quad_kernel}
O=gpurun_out/pmc_step; mkdir -p $O
B="python3 tools/pmc_step.py $N 10"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT --output-format csv -d $O/p1 -- $B > $O/p1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM --output-format csv -d $O/p2 -- $B > $O/p2.log 2>&1
K=$K python - <<'PY'
import collections, csv, glob, json, os
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_step/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if os.environ["K"] in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {"launches": len(v), "mean": sum(v) / len(v)} for k, v in sorted(acc.items())}
json.dump(out, open("gpurun_out/pmc_step/summary.json", "w"), indent=1)
print(json.dumps({k: round(v["mean"]) for k, v in out.items()}))
PY
find $O -name "*.db" -delete
