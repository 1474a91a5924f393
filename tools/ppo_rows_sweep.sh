#!/bin/bash
# PPO update cost against minibatch rows / storage / weight-gradient pipe (is the update bound by the matrices' trip through HBM?)
O=gpurun_out/ppo_sweep; mkdir -p $O; rm -f $O/sweep.jsonl
for rows in 4096 8192 16384 32768 65536; do
  for st in f32 bf16; do
    python tools/microbench_ppo.py $rows $((8*65536)) 50 $st two_kernels | tee -a $O/sweep.jsonl
  done
  python tools/microbench_ppo.py $rows $((8*65536)) 50 f32 two_kernels_bf16x6 | tee -a $O/sweep.jsonl
  python tools/microbench_ppo.py $rows $((8*65536)) 50 f32 one_kernel | tee -a $O/sweep.jsonl
done
