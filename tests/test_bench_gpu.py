"""bench.py as the driver runs it, with real kernels: `python bench.py --gpus 2 ...` and no launcher around it — the parent starts
its two ranks itself (torch.distributed.run), both on this box's one GPU over gloo (FPV_DIST_BACKEND=gloo: the rehearsal of the
N > 1 path that one GPU allows; RCCL across GPUs is the driver's run), and hands rank 0's one JSON line on.  Exercises bench's whole
N > 1 path at HEAD: rendezvous, id offsets, the collector's all-gather + advantage statistics + rank merge behind every rollout,
barrier-bracketed timing with MAX over ranks."""
import json
import math
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_gpus_2_runs_without_a_launcher():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["FPV_DIST_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--drones", "8192", "--steps", "20", "--warmup", "5",
                          "--no-cpu-baseline", "--no-host-mode", "--no-split-extra", "--kernel-launches", "20"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 20 and rec["warmup"] == 5 and rec["scaling"] == "weak"
    assert rec["config"]["total_drones"] == 16384 and rec["config"]["detector"] == "pixel"
    assert rec["value"] is not None and math.isfinite(rec["value"]) and rec["value"] > 1e6
    assert abs(rec["value"] - 16384 * 20 / (rec["ms_per_step"] * 20 * 1e-3)) <= 1e-6 * rec["value"]       # whole-job rate over the MAX time
