cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/s3
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/s3/tests.txt 2>&1; tail -25 gpurun_out/s3/tests.txt
python bench.py --no-cpu-baseline > gpurun_out/s3/bench.json 2> gpurun_out/s3/bench.err; tail -3 gpurun_out/s3/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-mode --no-f64-rollout > gpurun_out/s3/bench_s20.json 2> gpurun_out/s3/bench_s20.err
python tools/flip_rates.py > gpurun_out/s3/flip_rates.json 2> gpurun_out/s3/flip_rates.err; tail -3 gpurun_out/s3/flip_rates.err
python - <<'PY'
import json
for k in ("bench", "bench_s20"):
    d = json.loads(open(f"gpurun_out/s3/{k}.json").read().strip().splitlines()[-1])
    print(k, round(d["value"]), round(d["ms_per_step"] * 1e3, 2), d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["us_per_launch"], d["roofline"].get("phases_us"), d.get("extra", {}))
PY
