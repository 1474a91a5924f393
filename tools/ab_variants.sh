#!/bin/bash
# A/B of library builds inside ONE gpurun call (box-to-box differences are larger than the effects compared): variants/*.so are
# selected in turn (FPV_LIB_VARIANT; "product" = the product library itself), twice each, and the headline bench is run.  Usage: bash tools/ab_variants.sh A C ...
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/ab; mkdir -p $O
for round in 1 2; do
for v in "$@"; do
  if [ "$v" = product ]; then unset FPV_LIB_VARIANT; else export FPV_LIB_VARIANT=$v; fi
  python bench.py --no-cpu-baseline --no-host-mode --no-split-extra > $O/${v}_$round.json 2> $O/err.txt
  python - "$v" "$round" <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/ab/{sys.argv[1]}_{sys.argv[2]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], sys.argv[2], round(d["value"]), round(d["ms_per_step"] * 1e3, 2), d["roofline"]["us_per_launch"], d["roofline"].get("phases_us"))
PY
done
done
