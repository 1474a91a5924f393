import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "experiments: needs libfpv_env_experiments.so (FPV_EXPERIMENTS=1 in the environment): the measured "
                            "alternatives that lost, DESIGN §5b; tests/test_experiments_gpu.py runs these in a child process")
    config.addinivalue_line("markers", "both_libraries: runs on the product library and again, in that child process, on the experiments library")


def pytest_collection_modifyitems(config, items):
    from fpv_drone_rl_agent_amd import _lib
    if _lib.experiments():
        return
    skip = pytest.mark.skip(reason="built into libfpv_env_experiments.so only: run with FPV_EXPERIMENTS=1 (tests/test_experiments_gpu.py does)")
    for item in items:
        if "experiments" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc
