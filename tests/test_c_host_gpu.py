"""The C ABI driven from plain C (examples/c_host/rollout_host.c: no Python, no torch in the process): the rollout loop of
INTEGRATION.md — policy forward with the statistics merge and the env step (as one kernel and as two), ledger scan / scatter,
GAE — must run and
report consistent figures."""
import json
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu


def test_plain_c_host_runs_the_rollout_loop():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = os.path.join(root, "examples", "c_host")
    subprocess.run(["make", "-C", d, "clean"], check=True, capture_output=True)
    subprocess.run(["make", "-C", d], check=True, capture_output=True)
    for kernels in ("1", "2"):         # fpv_rollout_step (one kernel per step, the default) and fpv_policy_forward + fpv_step_fused
        out = subprocess.run([os.path.join(d, "rollout_host"), "8192", "16", "6", kernels], check=True, capture_output=True,
                             text=True, timeout=120).stdout
        r = json.loads(out.strip().splitlines()[-1])
        assert r["ok"] and r["ledger_flags"] == 0 and r["kernels_per_step"] == int(kernels)
        assert r["obs_count"] == pytest.approx(8192 * 16 * 8 + 1e-4, rel=1e-9)       # 2 warm-up + 6 timed rollouts merged
        assert 20 < r["ep_len_mean"] < 402 and r["adv_std"] > 0
