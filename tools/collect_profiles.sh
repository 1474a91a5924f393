#!/bin/bash
# gpurun_out/session (tools/gpu_session.sh) -> the tracked summaries under profiles/, named <TAG>_* (default r05)
set -e
cd "$(dirname "$0")/.."
TAG=${1:-r05}
O=gpurun_out/session
last() { tail -n 1 "$1"; }
last $O/bench.json > profiles/${TAG}_bench_line.json
last $O/bench_s20.json > profiles/${TAG}_bench_line_s20.json
last $O/bench_analytic.json > profiles/${TAG}_bench_line_analytic.json
last $O/bench_two.json > profiles/${TAG}_bench_line_two.json
last $O/bench_yaw.json > profiles/${TAG}_bench_line_yaw.json
[ -s $O/bench_yaw_f32.json ] && last $O/bench_yaw_f32.json > profiles/${TAG}_bench_line_yaw_exact_fp32_policy.json
[ -s $O/ppo_ab.txt ] && { echo "# PPO update, 65 536-row minibatch, this build (default) against round 4's fpv_ppo.hip (ppo_v1) inside one call: tools/ab_ppo.sh"; cat $O/ppo_ab.txt; } > profiles/${TAG}_ppo_update_ab.txt
last $O/bench_gpus2_gloo.json > profiles/${TAG}_bench_line_gpus2_gloo_one_gpu.json
cp $O/phases_step.json profiles/${TAG}_rollout_step_phases.json
cp $O/phases_step_analytic.json profiles/${TAG}_rollout_step_phases_analytic.json
cp $O/c_host_one.json profiles/${TAG}_c_host.json
cp $O/flip_rates.json profiles/${TAG}_flip_rates.json
[ -s $O/hostmode.txt ] && cp $O/hostmode.txt profiles/${TAG}_hostmode.txt
last $O/bench_f32.json > profiles/${TAG}_bench_line_exact_fp32_policy.json
last $O/bench_f32_s20.json > profiles/${TAG}_bench_line_exact_fp32_policy_s20.json
[ -f $O/census.npz ] && python tests/detector_census_report.py $O/census.npz > profiles/${TAG}_detector_census.json
cp $O/step_sweep_pixel.json profiles/${TAG}_step_sweep_pixel.json
cp $O/step_quad_sweep.json profiles/${TAG}_step_quad_sweep.json
cp $O/fp32_vs_fp64_trajectory.json profiles/${TAG}_fp32_vs_fp64_trajectory.json
python tools/summarize_prof.py ${TAG} $O/prof_hover $O/pmc_fetch $O/pmc_write > /dev/null
python tools/make_kernel_traffic.py ${TAG} > /dev/null
cp "$(ls -t $O/prof_s20/*/*kernel_stats.csv | head -1)" profiles/${TAG}_kernel_stats_s20.csv
cp gpurun_out/pmc_k0/summary.json profiles/${TAG}_rollout_step_pmc_raw.json
{ echo "# The boundary between rollouts (tools/trace_gaps.py on rocprofv3 kernel traces; durations under the profiler)"; echo;
  echo "## one 20-step rollout per timed region (python bench.py --steps 20 --warmup 5 --repeats 12): last rollout, then medians"; echo '```'; cat $O/boundary_s20.txt; echo '```'; echo;
  echo "## steady state (--steps 256 --warmup 32, 32-step rollouts back to back)"; echo '```'; cat $O/boundary_steady.txt; echo '```'; } > profiles/${TAG}_rollout_boundary.md
[ -f $O/ppo_stamps.txt ] && { echo "# tools/ppo_stamps.py on the -DFPV_PPO_STAMPS=1 build: s_memrealtime stamps of two workgroups of ppo_fused_kernel, 65 536 rows"; cat $O/ppo_stamps.txt; } > profiles/${TAG}_ppo_one_kernel_phases.txt
[ -f $O/ppo_rows_sweep.jsonl ] && cp $O/ppo_rows_sweep.jsonl profiles/${TAG}_ppo_rows_sweep.jsonl
{ echo "# PPO training end to end (pixel detector, one-kernel update): python examples/train_hover.py --drones 16384 --timesteps 3e8 --log-every 20"
  grep "^{" $O/train_16384.log; echo; echo "# --drones 65536 --timesteps 6e8"; grep "^{" $O/train_65536.log; echo
  echo "# --drones 65536 --timesteps 6e8 --update-path two_kernels (round 3's update)"; grep "^{" $O/train_65536_two_kernels.log | tail -2; } > profiles/${TAG}_train_hover_curves.txt
{ echo "# PPO update, one 65 536-row minibatch gathered out of 524 288 (tools/microbench_ppo.py): HIP events per path (gradient + clip + Adam),"
  echo "# then rocprofv3 kernel statistics of the same script, path by path"; cat $O/ppo_events.json; echo;
  for path in one_kernel two_kernels two_kernels_bf16x6; do
    echo "## path = $path"; python3 - "$(ls -t $O/prof_ppo_$path/*/*kernel_stats.csv | head -1)" <<'PY'
import csv, sys
print('"Name","Calls","AverageNs","MinNs","MaxNs"')
for r in csv.DictReader(open(sys.argv[1])):
    if "fpv::" in r["Name"]:
        print('"%s",%s,%.0f,%s,%s' % (r["Name"], r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"]))
PY
  done; } > profiles/${TAG}_ppo_update_profile.txt
tail -3 $O/tests.txt > profiles/${TAG}_gpu_tests_tail.txt
[ -f $O/bench_rehearsed_ranks.json ] && last $O/bench_rehearsed_ranks.json > profiles/${TAG}_bench_line_rehearsed_ranks.json
ls -la profiles/${TAG}_* | awk '{print $5, $9}'
