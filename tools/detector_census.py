"""What the contour phase of the rollout-step kernel (barrier P -> barrier R, fpv_rollstep.hip rs_tail) is made of, workgroup by
workgroup — the GPU half:

  python tools/detector_census.py OUT.npz      the bench's rollout, 4 collects, then one more with the kernel's s_memrealtime stamps
      on: the stamps of the LAST launch per workgroup + the drones' poses after it (the poses its contour phase looked at, except
      for the drones that ended an episode in that step: `done` marks them).

With FPV_LIB_VARIANT=contour_probe (build.build_extension(variant="contour_probe", flags=("-DFPV_CONTOUR_PROBE=1",))) the rows
hold every wave's times and the contour phase runs twice, the second time on warm caches.

The other half, tests/detector_census_report.py IN.npz, runs the oracle's contour calls on those poses (it is test
infrastructure, so it lives under tests/) and holds the per-workgroup counts against the workgroup's P -> R time.
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def dump(path):
    import torch
    from bench import synthetic_states
    from fpv_drone_rl_agent_amd import _lib as L
    from fpv_drone_rl_agent_amd.engine import FpvEngine
    from fpv_drone_rl_agent_amd.policy import ActorCritic
    from fpv_drone_rl_agent_amd.rollout import RolloutCollector
    n, T = int(os.environ.get("DRONES", 65536)), int(os.environ.get("NSTEPS", 32))
    torch.manual_seed(0)
    eng = FpvEngine(n, 0, noise_ratio=0.02, seed=0, ledger_capacity=n * max(2, T // 8), detector="pixel")
    col = RolloutCollector(eng, ActorCritic().cuda(), n_steps=T, one_launch=True)
    col.start()
    eng.set_state(torch.from_numpy(synthetic_states(n, seed=0)))
    for _ in range(4):
        col.collect()
    grid = n // 256
    words = 64 if "probe" in os.environ.get("FPV_LIB_VARIANT", "") else 16     # (the probe build's stamp rows: fpv_rollstep.hip)
    stamps = torch.zeros((grid, words), dtype=torch.int64, device="cuda")
    L.load().fpv_debug_rollout_stamps(eng._h, C.c_void_p(stamps.data_ptr()))
    col._graph = None
    col.collect()
    col.collect()
    torch.cuda.synchronize()
    L.load().fpv_debug_rollout_stamps(eng._h, None)
    st = eng.get_state().cpu().numpy()
    np.savez_compressed(path, stamps=stamps.cpu().numpy(), pose=st[:, 0:7].astype(np.float64), done=col.done.cpu().numpy(),
                        obs=col.buf_obs[-1].cpu().numpy())
    print("wrote", path)


if __name__ == "__main__":
    dump(sys.argv[1])
