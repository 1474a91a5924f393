#!/bin/bash
# gpurun_out/r3 (tools/gpu_session_r3.sh) -> the tracked summaries under profiles/
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/r3
last() { tail -n 1 "$1"; }
last $O/bench.json > profiles/r03_bench_line.json
last $O/bench_s20.json > profiles/r03_bench_line_s20.json
last $O/bench_analytic.json > profiles/r03_bench_line_analytic.json
last $O/bench_two.json > profiles/r03_bench_line_two.json
last $O/bench_yaw.json > profiles/r03_bench_line_yaw.json
cp $O/phases_step.json profiles/r03_rollout_step_phases.json
cp $O/phases_step_analytic.json profiles/r03_rollout_step_phases_analytic.json
cp $O/c_host_one.json profiles/r03_c_host.json
cp gpurun_out/pixel_flip_rate.json profiles/r03_pixel_flip_rate.json
python tools/summarize_prof.py r03 $O/prof_hover $O/pmc_fetch $O/pmc_write > /dev/null
python tools/make_kernel_traffic.py r03 > /dev/null
cp "$(ls $O/prof_s20/*/*kernel_stats.csv | head -1)" profiles/r03_kernel_stats_s20.csv
{ echo "# The boundary between rollouts (tools/trace_gaps.py on rocprofv3 kernel traces; durations under the profiler)"; echo;
  echo "## one 20-step rollout per timed region (python bench.py --steps 20 --warmup 5 --repeats 12): last rollout, then medians"; echo '```'; cat $O/boundary_s20.txt; echo '```'; echo;
  echo "## steady state (--steps 256 --warmup 32, 32-step rollouts back to back)"; echo '```'; cat $O/boundary_steady.txt; echo '```'; } > profiles/r03_rollout_boundary.md
{ echo "# PPO update (tools/microbench_ppo.py): HIP events, then rocprofv3 kernel statistics of the same script"; cat $O/ppo_events.json; echo;
  head -12 "$(ls $O/prof_ppo/*/*kernel_stats.csv | head -1)"; } > profiles/r03_ppo_update_profile.txt
tail -3 $O/tests.txt > profiles/r03_gpu_tests_tail.txt
[ -f $O/bench_rehearsed_ranks.json ] && last $O/bench_rehearsed_ranks.json > profiles/r03_bench_line_rehearsed_ranks.json
ls -la profiles/r03_* | awk '{print $5, $9}'
