cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/s11
timeout -k 10 900 python -m pytest tests/test_rollstep_gpu.py tests/test_policy_gpu.py tests/test_facades_gpu.py tests/test_fullsize_properties_gpu.py -x -q > gpurun_out/s11/tests.txt 2>&1; tail -3 gpurun_out/s11/tests.txt
python bench.py --no-cpu-baseline --no-host-mode --no-f64-rollout > gpurun_out/s11/bench.json 2> gpurun_out/s11/err.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-mode --no-split-extra > gpurun_out/s11/bench_s20.json 2> /dev/null
python - <<'PY'
import json
for k in ("bench", "bench_s20"):
    d = json.loads(open(f"gpurun_out/s11/{k}.json").read().strip().splitlines()[-1])
    print(k, round(d["value"]), round(d["ms_per_step"] * 1e3, 2), d["roofline"]["us_per_launch"], d["roofline"]["frac"], d["roofline"].get("phases_us"), {a: b for a, b in d.get("extra", {}).items() if "note" not in a})
PY
