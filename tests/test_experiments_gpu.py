"""The measured alternatives that lost (the persistent whole-rollout launch, the 16-wave policy kernel, 256-thread
policy launches, the statistics merge inside the env step's launch) are compiled into libfpv_env_experiments.so only.  A process
binds ONE of the two libraries (FPV_EXPERIMENTS in the environment, read when the package is imported), so their tests — marked
`experiments`, plus the `both_libraries` ones — run here in a child process on the other library.  The child is one more process
on the card (two in all)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_experiments_library_suite():
    from fpv_drone_rl_agent_amd import _lib
    if _lib.experiments():
        pytest.skip("this process already runs on the experiments library")
    import torch
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    env = dict(os.environ, FPV_EXPERIMENTS="1")
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    log = os.path.join(out_dir, "experiments_suite.log")
    with open(log, "w") as f:
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests"), "-q", "-x", "-p", "no:cacheprovider",
                            "-m", "gpu and (experiments or both_libraries)"], cwd=ROOT, env=env, stdout=f, stderr=subprocess.STDOUT,
                           timeout=900)
    tail = "".join(open(log).readlines()[-25:])
    assert r.returncode == 0, tail
    assert " passed" in tail and "no tests ran" not in tail, tail


def test_product_library_refuses_the_experiments_without_a_gpu():
    """The guards are host-side: they fire before anything touches the device."""
    from fpv_drone_rl_agent_amd import _lib
    if _lib.experiments():
        pytest.skip("experiments library")
    with pytest.raises(RuntimeError, match="FPV_EXPERIMENTS=1"):
        _lib.require_experiments("x")
    assert not hasattr(_lib.load(), "fpv_rollout_steps")
