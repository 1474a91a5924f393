#!/bin/bash
# A/B of library builds on the PPO update inside ONE gpurun call: variants/<name>.so (built with FPV_EXTRA_HIPCC_FLAGS, e.g.
# -DFPV_PPO_STAMPS=1 for tools/ppo_stamps.py) selected in turn (FPV_LIB_VARIANT; "product" = the product library), the
# microbench run under rocprofv3 for the per-kernel split.  Usage: bash tools/ab_ppo.sh A B ...   (ARGS="rows total iters storage path")
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/ab_ppo; mkdir -p $O
for v in "$@"; do
  if [ "$v" = product ]; then unset FPV_LIB_VARIANT; else export FPV_LIB_VARIANT=$v; fi
  rm -rf $O/$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$v -- python3 tools/microbench_ppo.py ${ARGS:-65536 524288 50 f32 one_kernel} > $O/$v.log 2>&1
  echo "$v $(tail -1 $O/$v.log)"
  f=$(find $O/$v -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if r["Name"].startswith(("void fpv::ppo", "fpv::ppo", "fpv::adam")):
        print("   %-60s %8.1f us" % (r["Name"][:60], float(r["AverageNs"]) / 1e3))
PY
  find $O/$v -name "*.csv" ! -name "*kernel_stats.csv" -delete
done
