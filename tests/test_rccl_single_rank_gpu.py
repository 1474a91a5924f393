"""What one GPU can rehearse of the N > 1 job (tools/rccl_rehearsal.py, in a child process: a process group lives for the whole
process): a single-rank RCCL group — communicator, watchdog thread, a real all-gather of the collector's payload on a side stream
between two graph launches — alive while the collector captures and replays its HIP graphs, in both capture modes; then the
collector's whole N > 1 arrangement run with G = 1 over that group and compared with the one-rank arrangement."""
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
    # the collector's N > 1 arrangement with G = 1 over the real collective (FPV_REHEARSE_RANKS=1) against the one-rank arrangement:
    # the same batches, and statistics that went through un-merge + rank merge equal to fp64 rounding
    assert out["rehearsed_path_is_the_multi_rank_one"] is True and out["episodes_equal"] is True, out
    assert out["advantages_max_abs_diff"] == 0.0 and out["obs_max_abs_diff"] == 0.0, out
    assert out["adv_mean_std_max_rel_diff"] < 1e-12 and out["statistics_max_rel_diff"] < 1e-12, out
