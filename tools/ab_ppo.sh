#!/bin/bash
# A/B of the PPO update inside one gpurun call: variants/NAME.so (FPV_LIB_VARIANT) against the default build, three runs each.
# usage: bash tools/ab_ppo.sh "NAME1 NAME2 ..." [rows] [total]
VS=${1:-ppo_v1}; R=${2:-65536}; T=${3:-524288}
for i in 1 2 3; do
  echo -n "default  "; python tools/microbench_ppo.py $R $T 50 f32 one_kernel 2>/dev/null
  for V in $VS; do echo -n "$V  "; FPV_LIB_VARIANT=$V python tools/microbench_ppo.py $R $T 50 f32 one_kernel 2>/dev/null; done
done
