#!/bin/bash
# The measurement set behind profiles/<TAG>_* (TAG = r05 by default): GPU suite, bench lines (default = pixel detector; the driver's
# --steps 20 shape; analytic; yaw; two-kernel; rehearsed ranks), phase stamps, C host, rocprofv3 kernel statistics, the rollout
# boundary from a kernel trace, the PMC traffic passes and the SQ counters of the rollout-step kernel, env-step sweeps over drone
# counts, the fp32-vs-fp64 trajectory record, PPO update profile.
# Run as: gpurun -- 'bash tools/gpu_session.sh'; then tools/collect_profiles.sh copies the summaries into profiles/.
# PART=a (suite, bench lines, stamps, traces, counters) / PART=b (sweeps, records, training, PPO profiles): the whole set is longer than
# one gpurun call may last; default: both.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/session; mkdir -p $O
PART=${PART:-all}
if [ "$PART" != b ]; then
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
tail -2 $O/tests.txt
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --steps 20 --warmup 5 > $O/bench_s20.json 2> $O/bench_s20.err
python bench.py --policy-gemm f32 --no-cpu-baseline --no-split-extra --no-host-mode --no-f64-rollout > $O/bench_f32.json 2> $O/bench_f32.err
python bench.py --policy-gemm f32 --steps 20 --warmup 5 --no-cpu-baseline --no-split-extra --no-host-mode > $O/bench_f32_s20.json 2> $O/bench_f32_s20.err
python bench.py --detector analytic --no-cpu-baseline --no-split-extra --no-host-mode --no-f64-rollout > $O/bench_analytic.json 2> $O/bench_analytic.err
python bench.py --two-launch --no-cpu-baseline --no-split-extra --no-host-mode --no-f64-rollout > $O/bench_two.json 2> $O/bench_two.err
python bench.py --task yaw --no-cpu-baseline --no-host-mode > $O/bench_yaw.json 2> $O/bench_yaw.err
python bench.py --task yaw --policy-gemm f32 --no-cpu-baseline --no-host-mode > $O/bench_yaw_f32.json 2> $O/bench_yaw_f32.err
FPV_REHEARSE_RANKS=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29581 python bench.py --no-cpu-baseline --no-split-extra --no-host-mode --no-f64-rollout > $O/bench_rehearsed_ranks.json 2> $O/bench_rehearsed_ranks.err
FPV_DIST_BACKEND=gloo python bench.py --gpus 2 --drones 32768 --steps 64 --warmup 32 --no-cpu-baseline --no-split-extra --no-host-mode --no-f64-rollout --repeats 3 > $O/bench_gpus2_gloo.json 2> $O/bench_gpus2_gloo.err
python tools/rollout_phases.py > $O/phases_step.json 2> /dev/null
DETECTOR=analytic python tools/rollout_phases.py > $O/phases_step_analytic.json 2> /dev/null
make -C examples/c_host > /dev/null 2>&1
./examples/c_host/rollout_host 65536 32 32 > $O/c_host_one.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_hover -- python3 bench.py --steps 256 --warmup 32 --repeats 2 --no-cpu-baseline --no-split-extra --no-host-mode --no-f64-rollout > $O/prof_hover.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_s20 -- python3 bench.py --steps 20 --warmup 5 --repeats 12 --no-cpu-baseline --no-split-extra --no-host-mode --no-f64-rollout > $O/prof_s20.log 2>&1
python tools/trace_gaps.py $(ls $O/prof_s20/*/*kernel_trace.csv | head -1) 1 > $O/boundary_s20.txt
python tools/trace_gaps.py $(ls $O/prof_hover/*/*kernel_trace.csv | head -1) 1 > $O/boundary_steady.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 128 --warmup 32 --repeats 1 --no-cpu-baseline --no-split-extra --no-host-mode --no-f64-rollout --kernel-launches 20 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --steps 128 --warmup 32 --repeats 1 --no-cpu-baseline --no-split-extra --no-host-mode --no-f64-rollout --kernel-launches 20 > $O/pmc_write.log 2>&1
bash tools/pmc_rollstep.sh > $O/pmc_k0.txt 2>&1
fi
if [ "$PART" != a ]; then
python tools/step_sweep.py $O/step_sweep_pixel.json pixel > /dev/null 2>&1
python tools/step_quad_sweep.py $O/step_quad_sweep.json > /dev/null 2>&1
python tools/trajectory_drift.py > $O/fp32_vs_fp64_trajectory.json 2> /dev/null
python tools/flip_rates.py > $O/flip_rates.json 2> /dev/null
python tools/detector_census.py $O/census.npz > /dev/null 2>&1
python tools/microbench_hostmode.py > $O/hostmode.txt 2> /dev/null || true
# PPO training end to end (learning curve + wall clock), the default update path and the two-kernel one
python examples/train_hover.py --drones 16384 --timesteps 3e8 --log-every 20 > $O/train_16384.log 2>&1
python examples/train_hover.py --drones 65536 --timesteps 6e8 --log-every 20 > $O/train_65536.log 2>&1
python examples/train_hover.py --drones 65536 --timesteps 6e8 --log-every 80 --update-path two_kernels > $O/train_65536_two_kernels.log 2>&1
for path in one_kernel two_kernels two_kernels_bf16x6; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ppo_$path -- python3 tools/microbench_ppo.py 65536 524288 50 f32 $path > $O/prof_ppo_$path.log 2>&1
  python tools/microbench_ppo.py 65536 524288 50 f32 $path >> $O/ppo_events.json 2> /dev/null
done
python tools/microbench_ppo.py 65536 524288 50 bf16 two_kernels >> $O/ppo_events.json 2> /dev/null
python tools/microbench_ppo.py 262144 2097152 20 f32 one_kernel >> $O/ppo_events.json 2> /dev/null
python tools/microbench_ppo.py 262144 2097152 20 f32 two_kernels >> $O/ppo_events.json 2> /dev/null
bash tools/ppo_rows_sweep.sh > /dev/null 2>&1; cp gpurun_out/ppo_sweep/sweep.jsonl $O/ppo_rows_sweep.jsonl
# the update against round 4's build (variants/ppo_v1.so: python -m fpv_drone_rl_agent_amd.build ppo_v1 on round 4's fpv_ppo.hip), same call
if [ -f variants/ppo_v1.so ]; then bash tools/ab_ppo.sh ppo_v1 > $O/ppo_ab.txt 2>&1; fi
# phase stamps of the one-kernel gradient: the diagnostic build (FPV_EXTRA_HIPCC_FLAGS=-DFPV_PPO_STAMPS=1, kept as variants/ppo_stamps.so)
if [ -f variants/ppo_stamps.so ]; then
  FPV_LIB_VARIANT=ppo_stamps python tools/ppo_stamps.py > $O/ppo_stamps.txt 2> /dev/null
fi
fi
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +3M -delete
python - <<'PY'
import json, os
for k in ("bench", "bench_s20", "bench_analytic", "bench_two", "bench_yaw", "bench_gpus2_gloo"):
    if not os.path.exists(f"gpurun_out/session/{k}.json"):
        continue
    d = json.loads(open(f"gpurun_out/session/{k}.json").read().strip().splitlines()[-1])
    print(k, round(d["value"]), round(d["ms_per_step"] * 1e3, 2), d["roofline"]["frac"], d["roofline"]["us_per_launch"], d.get("cpu_baseline", {}).get("value"))
PY
du -sh $O
