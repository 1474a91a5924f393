#!/bin/bash
# tools/stress_policy.py on library builds variants/<name>.so in turn, inside ONE gpurun call.  Usage: bash tools/ab_stress.sh A B ...  (TRIALS=60)
cd $GRAFT_REPO_ROOT
cp fpv_drone_rl_agent_amd/libfpv_env.so /tmp/keep.so
for v in "$@"; do
  cp variants/$v.so fpv_drone_rl_agent_amd/libfpv_env.so
  echo "== $v"; timeout -k 10 300 python tools/stress_policy.py ${TRIALS:-60} 2>&1 | tail -2
done
cp /tmp/keep.so fpv_drone_rl_agent_amd/libfpv_env.so
