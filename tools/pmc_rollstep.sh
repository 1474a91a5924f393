#!/bin/bash
# SQ counters of the rollout-step kernel (one kernel per step, 65 536 drones, pixel detector): two rocprofv3 --pmc passes of eight
# counters over the bench command, per-launch means of the rollout-step kernel (fpv::rollout_step_split_kernel by default) -> gpurun_out/pmc_k0/summary.json
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_k0; mkdir -p $O
B="python3 bench.py --steps 64 --warmup 32 --repeats 1 --no-cpu-baseline --no-split-extra --no-host-mode --kernel-launches 10"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $O/p1 -- $B > $O/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --output-format csv -d $O/p2 -- $B > $O/p2.log 2>&1
python - <<'PY'
import collections, csv, glob, json
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_k0/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "rollout_step_kernel" in r["Kernel_Name"] or "rollout_step_split_kernel" in r["Kernel_Name"]:     # (whichever the default runs)
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {"launches": len(v), "mean": sum(v) / len(v)} for k, v in sorted(acc.items())}
json.dump(out, open("gpurun_out/pmc_k0/summary.json", "w"), indent=1)
print(json.dumps({k: round(v["mean"]) for k, v in out.items()}))
PY
find $O -name "*.db" -delete
