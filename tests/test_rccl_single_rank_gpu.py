"""What one GPU can rehearse of the N > 1 job (tools/rccl_rehearsal.py, in a child process: a process group lives for the whole
process): a single-rank RCCL group — communicator, watchdog thread, a real all-gather of the collector's payload on a side stream
between two graph launches — alive while the collector captures and replays its HIP graphs, in both capture modes."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_rccl_group_and_graph_capture_coexist():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_rehearsal.py")], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["allgather_ok"] is True
    assert out["capture_global"] is True and out["capture_thread_local"] is True, out
    assert out["episodes_global"] > 0 and out["episodes_thread_local"] > 0
