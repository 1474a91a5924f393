"""Two ranks on one GPU over gloo (rehearsal of the one-process-per-GPU layout): PPO.learn with the hand-written
update kernels, gradient all-reduce per minibatch, shared early-stop decision.  Both ranks must end with bitwise
identical parameters (same averaged gradients -> same Adam steps) although they simulate different drones, and with
bitwise identical VecNormalize statistics (fpv_rank_merge over the one all-gather's payload)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

_WORKER = r"""
import os, sys, hashlib
import torch
sys.path.insert(0, os.environ["FPV_REPO"])
from fpv_drone_rl_agent_amd import dist as D
from fpv_drone_rl_agent_amd.engine import FpvEngine
from fpv_drone_rl_agent_amd.policy import ActorCritic
from fpv_drone_rl_agent_amd.ppo import PPO
from fpv_drone_rl_agent_amd.rollout import RolloutCollector
rank, world, local = D.init_from_env("gloo")
dev = torch.device("cuda", 0)
torch.manual_seed(0)
pol = ActorCritic(log_std_init=-1.0).to(dev)
eng = FpvEngine(2048, dev, seed=0, id_offset=rank * 2048, ledger_capacity=2048 * 8)
col = RolloutCollector(eng, pol, n_steps=16, policy_seed=0)
model = PPO(col, n_epochs=3, n_minibatches=4, target_kl=0.02)
model.learn(4 * 2048 * 16 * world)
torch.cuda.synchronize()
flat = torch.cat([p.detach().reshape(-1) for p in pol.parameters()]).cpu()
assert torch.isfinite(flat).all()
col.wait_stats()
torch.cuda.synchronize()
rms = torch.cat([col.obs_norm.rms, col.rew_norm.rms]).cpu()
print("DIGEST", rank, hashlib.sha256(flat.numpy().tobytes()).hexdigest(), model.num_timesteps,
      hashlib.sha256(rms.numpy().tobytes()).hexdigest(), float(col.obs_norm.rms[40]), flush=True)
"""


def test_two_rank_training_keeps_replicas_identical(tmp_path):
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, FPV_REPO=repo, MASTER_ADDR="127.0.0.1", MASTER_PORT="29631", WORLD_SIZE="2",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    assert all(p.returncode == 0 for p in procs), outs
    dig = sorted(line.split()[1:] for out in outs for line in out.splitlines() if line.startswith("DIGEST"))
    assert len(dig) == 2 and dig[0][1] == dig[1][1], dig          # identical parameters on both ranks
    assert dig[0][2] == str(4 * 2048 * 16 * 2)                      # timesteps count every rank's drones
    # VecNormalize's statistics ride in the one all-gather: after every update both ranks hold the statistics of the union of
    # their drones' samples — bit-identical, and counting both ranks' samples (reset observations, the one warm-up step of the
    # graph capture, 4 rollouts of 16 steps)
    assert dig[0][3] == dig[1][3], dig
    assert abs(float(dig[0][4]) - (2 * 2048 * (1 + 1 + 4 * 16) + 1e-4)) < 1e-3, dig
