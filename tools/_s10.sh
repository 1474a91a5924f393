cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/s10
for round in 1 2 3; do for v in product nofence prefetch prefetch_nofence; do
  if [ "$v" = product ]; then unset FPV_LIB_VARIANT; else export FPV_LIB_VARIANT=$v; fi
  python bench.py --no-cpu-baseline --no-host-mode --no-f64-rollout --no-split-extra --kernel-launches 50 > gpurun_out/s10/bench_${v}_$round.json 2> gpurun_out/s10/err.txt
  python - "$v" "$round" <<'PY' | tee -a gpurun_out/s10/ab_bench.txt
import json, sys
d = json.loads(open(f"gpurun_out/s10/bench_{sys.argv[1]}_{sys.argv[2]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], sys.argv[2], round(d["value"]), round(d["ms_per_step"] * 1e3, 2), d["roofline"]["us_per_launch"], d["roofline"].get("phases_us"))
PY
done; done
