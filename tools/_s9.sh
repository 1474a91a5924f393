cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/s9
timeout -k 10 900 python -m pytest tests/test_rollstep_gpu.py tests/test_policy_gpu.py tests/test_facades_gpu.py -x -q > gpurun_out/s9/tests.txt 2>&1; tail -4 gpurun_out/s9/tests.txt
for round in 1 2 3; do for v in tilewise noslp product; do
  if [ "$v" = product ]; then unset FPV_LIB_VARIANT; else export FPV_LIB_VARIANT=$v; fi
  python bench.py --no-cpu-baseline --no-host-mode --no-f64-rollout --no-split-extra --kernel-launches 50 > gpurun_out/s9/bench_${v}_$round.json 2> gpurun_out/s9/err.txt
  python - "$v" "$round" <<'PY' | tee -a gpurun_out/s9/ab_bench.txt
import json, sys
d = json.loads(open(f"gpurun_out/s9/bench_{sys.argv[1]}_{sys.argv[2]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], sys.argv[2], round(d["value"]), round(d["ms_per_step"] * 1e3, 2), d["roofline"]["us_per_launch"], d["roofline"].get("phases_us"))
PY
done; done
unset FPV_LIB_VARIANT
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-mode --no-split-extra --repeats 12 > gpurun_out/s9/bench_s20.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/s9/prof_s20 -- python3 bench.py --steps 20 --warmup 5 --repeats 12 --no-cpu-baseline --no-split-extra --no-host-mode --no-f64-rollout > gpurun_out/s9/prof_s20.log 2>&1
python tools/trace_gaps.py $(ls gpurun_out/s9/prof_s20/*/*kernel_trace.csv | head -1) 1 > gpurun_out/s9/boundary_s20.txt; cat gpurun_out/s9/boundary_s20.txt | cut -c1-900
find gpurun_out/s9 -name "*.db" -delete; find gpurun_out/s9 -name "*kernel_trace.csv" -size +3M -delete
