#!/bin/bash
# tools/stress_policy.py on library builds variants/<name>.so in turn, inside ONE gpurun call.  Usage: bash tools/ab_stress.sh A B ...  (TRIALS=60)
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = product ]; then unset FPV_LIB_VARIANT; else export FPV_LIB_VARIANT=$v; fi
  echo "== $v"; timeout -k 10 300 python tools/stress_policy.py ${TRIALS:-60} 2>&1 | tail -2
done
